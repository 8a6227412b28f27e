cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r02a_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a_gputest.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
python bench.py --reads 12500 --steps 2 --warmup 1 --no-cpu-baseline --no-fwbw > gpurun_out/r02a_bench_c4shard.json 2> gpurun_out/r02a_bench_c4shard.err
python bench.py --gpus 2 --steps 1 > gpurun_out/r02a_bench_gpus2.out 2>&1; echo "rc=$?" >> gpurun_out/r02a_bench_gpus2.out
nproc > gpurun_out/r02a_host.txt; lscpu | head -20 >> gpurun_out/r02a_host.txt; free -g >> gpurun_out/r02a_host.txt
bash tools/gpu_profile.sh r02a_prof > /dev/null 2>&1
tail -5 gpurun_out/r02a_gputest.log; cat gpurun_out/r02a_bench.json; cat gpurun_out/r02a_bench_c4shard.json; cat gpurun_out/r02a_bench_gpus2.out | tail -3
