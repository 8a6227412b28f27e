cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mb in 0 65536 32768 16384 8192; do
  if [ $mb = 0 ]; then unset NCHMM_WS_BUDGET_MB; else export NCHMM_WS_BUDGET_MB=$mb; fi
  /usr/bin/env python bench.py --reads 12500 --steps 3 --warmup 1 --no-cpu-baseline --no-fwbw 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget_mb=$mb', 'Mev/s', d['value'], 'ms/step', d['ms_per_step'], 'launches', d['config']['forward_launches_per_step'])"
done
