# Traceback: how long a run-in do the speculative segments need?  bench.py by NCHMM_TB_MARGIN (events a segment walks above its
# first owned event before its result counts), with the kernel's own tally of segments that had NOT merged with the true path
# at their boundary and were walked again (exact either way).   bash tools/ubench/tb_margin.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for m in ${MARGINS:-256 192 128 96 64 32}; do
  NCHMM_TB_MARGIN=$m NCHMM_PROFILE=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fwbw --no-end-to-end 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json, re
m = '$m'
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('margin', m, 'Mevents/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'tb_ms', d['roofline'].get('traceback_kernel_ms', 0))
    elif 'phase ticks' in l:
        r = re.search(r're-walked=(\d+) of (\d+)', l); print('margin', m, 'segments re-walked', r.group(1), 'of', r.group(2))
"
done
