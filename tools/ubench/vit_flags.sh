# Viterbi / FB kernels rebuilt with extra compiler flags (scheduler strategies etc.), one set per line of $FLAGSETS
# (separated by ';'), timed with bench.py and tools/bench_fwbw.py on one box.
#   FLAGSETS="-O3;-O3 -mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" bash tools/ubench/vit_flags.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
BASEF="--offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
IFS=';' read -ra SETS <<< "${FLAGSETS:--O3}"
for f in "${SETS[@]}"; do
  echo "== $f"
  if /opt/rocm/bin/hipcc $BASEF $f -c viterbi_kernel.hip -o viterbi_kernel.o 2>/tmp/err.txt && /opt/rocm/bin/hipcc $BASEF $f -c fwbw_scaled_kernel.hip -o fwbw_scaled_kernel.o 2>>/tmp/err.txt && make -s > /dev/null 2>&1; then
    (cd $R && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('viterbi Mevents/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])"
     python tools/bench_fwbw.py | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwbw M event-rounds/s', d['value'], 'kernel_ms', d['kernel_ms'])")
  else
    echo "build failed: $(grep -m1 error /tmp/err.txt)"
  fi
done
/opt/rocm/bin/hipcc $BASEF -O3 -c viterbi_kernel.hip -o viterbi_kernel.o && /opt/rocm/bin/hipcc $BASEF -O3 -c fwbw_scaled_kernel.hip -o fwbw_scaled_kernel.o && make -s > /dev/null 2>&1
