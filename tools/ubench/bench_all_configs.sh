# Every bench line of profiles/r04b_*: configs 2, 3, 4 (one shard), 5 (256 / 1024 reads, 12 GB budget), the host path, ragged batches.
#   bash tools/ubench/bench_all_configs.sh   (on the GPU box; writes gpurun_out/r04b/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b
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
for f in $O/*.json; do echo "== $f"; head -c 1500 $f; echo; done
tail -3 $O/*.err
