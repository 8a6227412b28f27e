cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02c_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02c_gputest.log
python tools/bench_prep.py > gpurun_out/r02c_bench_prep.json 2> gpurun_out/r02c_bench_prep.err
tail -25 gpurun_out/r02c_gputest.log; cat gpurun_out/r02c_bench_prep.json
