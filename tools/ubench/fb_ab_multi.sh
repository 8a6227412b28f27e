# FB scaled kernels: the tree against several variant source files on one box in one run; per variant the bench leg (3x) and the
# per-kernel averages of a rocprofv3 --kernel-trace --stats run.   bash tools/ubench/fb_ab_multi.sh tools/ubench/_fbv/*.hip
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
FLAGS=$(make -s print-hipflags)
run() { (cd $R && for i in 1 2 3; do STEPS=20 python tools/bench_fwbw.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d['value'], 'Mevent-rounds/s kernel_ms', d['kernel_ms'], 'clock', d['shader_clock_mhz_under_load'], 'lpd', d['log_pr_data_mean'])"; done
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fbprof && STEPS=20 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/fbprof -o p -- python3 $R/tools/bench_fwbw.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/fbprof/**/*kernel_stats*.csv", recursive=True) + glob.glob("/tmp/fbprof/*kernel_stats*.csv"):
    for r in csv.DictReader(open(f)):
        if "scaled" in r["Name"]:
            print("     ", r["Name"].split("(")[0][7:], "avg %.3f ms  min %.3f" % (float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6))
PY
); }
cp fwbw_scaled_kernel.hip /tmp/fwbw_scaled_kernel.hip.tree
trap 'cp /tmp/fwbw_scaled_kernel.hip.tree fwbw_scaled_kernel.hip; make -s > /dev/null 2>&1' EXIT
echo "== tree"; run
for V in "$@"; do
  cp "$R/$V" fwbw_scaled_kernel.hip 2>/dev/null || cp "$V" fwbw_scaled_kernel.hip
  if make -s > /tmp/fb_build.log 2>&1; then echo "== variant $(basename $V)"; run; else echo "== variant $(basename $V): BUILD FAILED"; tail -5 /tmp/fb_build.log; fi
done
cp /tmp/fwbw_scaled_kernel.hip.tree fwbw_scaled_kernel.hip; make -s > /dev/null 2>&1
echo "== tree again"; run
