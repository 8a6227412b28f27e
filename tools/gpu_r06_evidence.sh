#!/bin/bash
# Round 6 evidence pass (GPU box, repo root): rocprofv3 kernel stats + PMC passes of the config-2 command and of the FB pair
# (tools/gpu_profile*.sh), the default bench line, config 5 (256 and 1024 reads of 50 000 events, r9.t) with its CPU baseline, and the
# command line with 1 / 2 / 4 worker processes on this one GPU.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
bash tools/gpu_profile.sh r06/prof_c2 > $O/prof_c2.log 2>&1
bash tools/gpu_profile_fwbw.sh r06/prof_fwbw > $O/prof_fwbw.log 2>&1
# (the traffic files first, so that the bench lines below replay them)
cp $O/prof_c2/hbm_traffic_c2.json profiles/r06_hbm_traffic_c2.json 2>/dev/null
cp $O/prof_fwbw/hbm_traffic_fwbw.json profiles/r06_hbm_traffic_fwbw.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/bench_default_20.json 2> $O/bench_default_20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --model r9.t --events 50000 --reads 256 --steps 3 --warmup 1 --no-fwbw --no-shard-leg > $O/bench_config5_256.json 2> $O/bench_config5_256.err
python bench.py --model r9.t --events 50000 --reads 1024 --steps 3 --warmup 1 --no-fwbw --no-shard-leg --no-end-to-end > $O/bench_config5_1024.json 2> $O/bench_config5_1024.err
for W in 0 1 2 4; do
  READS=8000 EVENTS=5000 THREADS=64 WORKERS=$W python tools/bench_cli.py > $O/bench_cli_w$W.json 2> $O/bench_cli_w$W.err
done
ls -la $O | tail -30
for f in bench_default_20 bench_config5_256 bench_config5_1024 bench_cli_w0 bench_cli_w1 bench_cli_w2 bench_cli_w4; do echo "== $f"; head -c 700 $O/$f.json; echo; tail -c 400 $O/$f.err; done
