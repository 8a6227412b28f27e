# Viterbi kernel A/B on one box over preprocessor switches of the tree's viterbi_kernel.hip (NCHMM_VIT_SKEW, NCHMM_TB_PRIO, ...):
# for every argument (a quoted list of -D flags; "" = the tree) the kernel is rebuilt with it, the library relinked, and bench.py
# run REPS times with overlapping steps and once with --serial-launches; the tree is restored at the end.
#   bash tools/ubench/vit_ab_defs.sh "" "-DNCHMM_VIT_SKEW=768" "-DNCHMM_TB_PRIO=0"
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=${REPS:-3}
cd $R/nanocall_amd/csrc
FLAGS=$(make -s print-hipflags 2>/dev/null)
[ -z "$FLAGS" ] && FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 Mevents/s', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'clock', d['device']['shader_clock_mhz_under_load'], 'cycles/step (M)', round(d['ms_per_step'] * d['device']['shader_clock_mhz_under_load'] / 1e3, 2))"; }
run() { (cd $R && for i in $(seq $REPS); do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw --no-end-to-end ${BENCH_ARGS:-} 2>/dev/null | line overlap; done
        python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw --no-end-to-end --serial-launches ${BENCH_ARGS:-} 2>/dev/null | line serial
        [ -n "${PROFILE:-}" ] && NCHMM_PROFILE=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fwbw --no-end-to-end --serial-launches ${BENCH_ARGS:-} 2>&1 >/dev/null | grep -E "phase ticks|blocks\]" ); }
for D in "$@"; do
  echo "== variant [$D]"
  /opt/rocm/bin/hipcc $FLAGS $D -c viterbi_kernel.hip -o viterbi_kernel.o && make -s > /dev/null 2>&1 && run
done
rm -f viterbi_kernel.o; make -s > /dev/null 2>&1
echo "== tree restored"
