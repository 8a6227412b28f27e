# Every bench line of profiles/r05_* (r04b_* in round 4): configs 2, 3, 4 (one shard), 5 (256 / 1024 reads, 12 GB budget), the host path, ragged batches.
#   bash tools/ubench/bench_all_configs.sh   (on the GPU box; writes gpurun_out/r05b/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b
mkdir -p $O
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw > $O/bench_c2_20steps.json 2>> $O/bench_c2.err
python bench.py --model r9.t --events 50000 --reads 256 --steps 3 --warmup 1 --no-fwbw > $O/bench_config5_256.json 2> $O/c5.err
python bench.py --model r9.t --events 50000 --reads 1024 --steps 3 --warmup 1 --no-fwbw > $O/bench_config5_1024.json 2>> $O/c5.err
NCHMM_WS_BUDGET_MB=12288 python bench.py --model r9.t --events 50000 --reads 256 --steps 3 --warmup 1 --no-fwbw > $O/bench_config5_256_budget12g.json 2>> $O/c5.err
python bench.py --reads 12500 --steps 5 --warmup 2 --no-fwbw --no-cpu-baseline --no-end-to-end > $O/bench_config4_shard.json 2> $O/c4.err
python tools/bench_config3.py > $O/bench_config3.json 2> $O/c3.err
python tools/bench_hostpath.py > $O/bench_hostpath.json 2> $O/hp.err
DEPTH=3 python tools/bench_ragged.py > $O/bench_ragged_1024.json 2> $O/rg.err
DEPTH=3 READS=4096 MEDIAN=5000 SIGMA=1.0 MAXLEN=50000 python tools/bench_ragged.py > $O/bench_ragged_4096.json 2>> $O/rg.err
SHAPES=1:5000,64:5000,256:5000,384:5000,512:5000,1024:5000,8:30000,96:10000,256:50000 python tools/bench_sweeps.py > $O/bench_sweeps.json 2> $O/sw.err
SWEEP=wide DEPTH=3 python tools/bench_ragged.py > $O/bench_ragged_1024_wide.json 2>> $O/rg.err
NCHMM_BENCH_SHARE_GPU0=1 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2_shared.json 2> $O/n2.err
NCHMM_BENCH_SHARE_GPU0=1 python bench.py --pool --gpus 2 --steps 3 --warmup 1 > $O/bench_pool_n2_shared.json 2>> $O/n2.err
python bench.py --pool --gpus 1 --reads 8192 --steps 3 --warmup 1 > $O/bench_pool_n1.json 2>> $O/n2.err
for t in 1 16 256 1024; do tools/bench_cpp_layer 2048 5000 $t 2>/dev/null | tail -1; done > $O/strand_combiner_threads.txt
for f in $O/*.json; do echo "== $f"; head -c 1500 $f; echo; done
tail -3 $O/*.err
