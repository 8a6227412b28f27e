cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NCHMM_FB_LAYOUT=8 python tools/bench_fwbw.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('layout8', d['value'], d['kernel_ms'], d['log_pr_data_mean'])"
python tools/bench_fwbw.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('layout4', d['value'], d['kernel_ms'], d['log_pr_data_mean'])"
cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02j_stats -o fb -- python3 $GRAFT_REPO_ROOT/tools/bench_fwbw.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r02j_stats/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'scaled' in row['Name']: print(row['Name'][:50], row['AverageNs'], row['MinNs'])
PY
timeout 600 python -m pytest tests/test_fwbw_gpu.py -m gpu -x -q 2>&1 | tail -3
