# Two-operation division by sigma / eta (viterbi_kernel.hip) against the three-operation form, SAME binary, same box:
# NCHMM_DIV2=0 withholds the table at context creation, so every wave takes the three-operation columns.
#     bash tools/ubench/vit_ab_div2.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for round in 1 2; do
for v in 1 0; do
  echo "== NCHMM_DIV2=$v"
  for i in 1 2 3; do NCHMM_DIV2=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Mevents/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'tb_ms', d['roofline']['traceback_kernel_ms'], 'clock', d['device']['shader_clock_mhz_under_load'])"; done
done
done
