# Viterbi kernel A/B on one box: rebuilds viterbi_kernel.o with each -D switch in VARIANTS (BASE = none) and runs
# bench.py (no CPU baseline, no FB leg) plus the bit-exactness tests.   VARIANTS="BASE NCHMM_BP_SDWA" bash tools/ubench/vit_variant.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
for v in ${VARIANTS:-BASE}; do
  /opt/rocm/bin/hipcc $FLAGS -D$v -c viterbi_kernel.hip -o viterbi_kernel.o && make -s > /dev/null 2>&1
  echo "== $v"
  (cd $R && for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Mevents/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'clock', d['device']['shader_clock_mhz_under_load'])"; done
   [ -n "${TESTS:-}" ] && python -m pytest tests/test_viterbi_gpu.py -x -q 2>&1 | tail -2)
done
/opt/rocm/bin/hipcc $FLAGS -c viterbi_kernel.hip -o viterbi_kernel.o && make -s > /dev/null 2>&1
