#!/bin/bash
# Round 6, second evidence pass (after the FB kernels changed again): FB rocprofv3 stats + PMC traffic, then the bench lines that replay them.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
bash tools/gpu_profile_fwbw.sh r06/prof_fwbw2 > $O/prof_fwbw2.log 2>&1
cp $O/prof_fwbw2/hbm_traffic_fwbw.json profiles/r06_hbm_traffic_fwbw.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/bench2_default_20.json 2> $O/bench2_default_20.err
python bench.py > $O/bench2_default.json 2> $O/bench2_default.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_default -o b -- python3 $R/bench.py --no-cpu-baseline > $O/stats_default.log 2>&1 )
head -c 600 $O/bench2_default_20.json; echo; tail -c 300 $O/bench2_default_20.err
cat $O/prof_fwbw2/stats/fb_kernel_stats.csv | head -4
