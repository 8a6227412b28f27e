cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r02b_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b_gputest.log
tail -30 gpurun_out/r02b_gputest.log
