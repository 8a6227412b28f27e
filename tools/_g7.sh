cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02f_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02f_gputest.log
tail -12 gpurun_out/r02f_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02f_smoke.log 2>&1; tail -2 gpurun_out/r02f_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err; cat gpurun_out/r02f_bench.json
bash tools/gpu_profile.sh r02f_prof > /dev/null 2>&1; tail -3 gpurun_out/r02f_prof/summary.txt
bash tools/gpu_profile_fwbw.sh r02f_fwbw > /dev/null 2>&1
python tools/bench_config3.py > gpurun_out/r02f_bench_config3.json 2> gpurun_out/r02f_bench_config3.err; tail -1 gpurun_out/r02f_bench_config3.json
