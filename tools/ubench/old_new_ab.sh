# Same-box A/B of the tree against the copy of an earlier commit under _old/ (git archive <commit> | tar -x -C _old; make -C _old/nanocall_amd/csrc):
# bench.py config 2, 20 steps, interleaved ROUNDS times, then tools/bench_ragged.py on both.
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${ROUNDS:-4}
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 Mevents/s', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'clock', d['device']['shader_clock_mhz_under_load'], 'end_to_end', d.get('end_to_end', {}).get('value'), 'one_call', d.get('end_to_end', {}).get('one_call', {}).get('value'))"; }
for i in $(seq $ROUNDS); do
  (cd $R/_old && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw 2>/dev/null | line old)
  (cd $R && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw 2>/dev/null | line new)
done
(cd $R/_old && GRAFT_REPO_ROOT=$R/_old python tools/bench_ragged.py 2>/dev/null | sed 's/^/old /')
(cd $R && python tools/bench_ragged.py 2>/dev/null | sed 's/^/new /')
(cd $R/_old && GRAFT_REPO_ROOT=$R/_old READS=2048 MEDIAN=5000 SIGMA=1.0 MAXLEN=50000 python tools/bench_ragged.py 2>/dev/null | sed 's/^/old /')
(cd $R && READS=2048 MEDIAN=5000 SIGMA=1.0 MAXLEN=50000 python tools/bench_ragged.py 2>/dev/null | sed 's/^/new /')
