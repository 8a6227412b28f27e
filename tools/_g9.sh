cd $GRAFT_REPO_ROOT
python -m pytest tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -15
