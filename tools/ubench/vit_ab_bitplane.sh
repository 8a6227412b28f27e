# Same box, same run: the tree's Viterbi kernels (one byte per state, 4 KiB rows) against the round-3 bit-plane experiment
# (tools/ubench/viterbi_bitplane_kernel.hip.txt: class masks of the 3-way combine stored with s_store_dwordx4, group winners one
# byte per thread, scalar traceback walk; 1.5 KiB rows).  The experiment file carries NCHMM_X_* switches (NOSTORE, FIXEDADDR,
# NOPIN, EMIS_AFTER, EMIS_FIRST, PHASES) for its own variants: VARIANTS="BASE NCHMM_X_NOSTORE" bash tools/ubench/vit_ab_bitplane.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
run() { (cd $R && for i in 1 2 3; do NCHMM_PROFILE=${PROFILE:-0} python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw --no-end-to-end 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('Mevents/s', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'tb_ms', d['roofline']['traceback_kernel_ms'], 'clock', d['device']['shader_clock_mhz_under_load'])
    elif 'phase ticks' in l:
        print(l.strip()[:330])
"; done); }
echo "== tree"; run
cp $R/tools/ubench/viterbi_bitplane_kernel.hip.txt /tmp/viterbi_bitplane_kernel.hip
for v in ${VARIANTS:-BASE}; do
  D=""; for m in $(echo $v | tr '+' ' '); do D="$D -D$m"; done
  /opt/rocm/bin/hipcc $FLAGS -DNCHMM_BP_ROW_BYTES=1536 $D -c /tmp/viterbi_bitplane_kernel.hip -o viterbi_kernel.o && /opt/rocm/bin/hipcc $FLAGS -DNCHMM_BP_ROW_BYTES=1536 -x hip -c nchmm_api.cpp -o nchmm_api.o && make -s > /dev/null 2>&1
  echo "== bit-plane experiment, $v"; run
done
rm -f viterbi_kernel.o nchmm_api.o; make -s > /dev/null 2>&1
echo "== tree again"; run
