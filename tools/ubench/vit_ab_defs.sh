# Viterbi kernel A/B on one box over the named constants of the tree's viterbi_kernel.hip (kTbSeg, kTbPrio, ...): for every
# argument (a space-separated list of NAME=VALUE; "" = the tree) a COPY of the kernel with those constants edited is compiled, the
# library relinked, and bench.py run REPS times with overlapping and once with serialised steps; the tree's object is rebuilt at the end.
#   bash tools/ubench/vit_ab_defs.sh "" "kTbSeg=40" "kTbPrio=0"         (product sources carry no switches: the edit is made here)
R=${GRAFT_REPO_ROOT:-$(pwd)}
REPS=${REPS:-3}
cd $R/nanocall_amd/csrc
# whatever ends this script (an error, an interrupt, a time limit): the tree's own object is rebuilt, never a variant's left behind
trap 'rm -f viterbi_kernel.o; make -s > /dev/null 2>&1' EXIT
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
  cp viterbi_kernel.hip /tmp/viterbi_variant.hip
  for kv in $D; do
    name=${kv%%=*}; val=${kv#*=}
    grep -q "constexpr [a-z ]* $name = " /tmp/viterbi_variant.hip || { echo "no constant $name in viterbi_kernel.hip"; continue 2; }
    sed -i "s/\(constexpr [a-z ]* $name = \)[0-9]*;/\1$val;/" /tmp/viterbi_variant.hip
  done
  /opt/rocm/bin/hipcc $FLAGS -I$R/nanocall_amd/csrc -c /tmp/viterbi_variant.hip -o viterbi_kernel.o && make -s > /dev/null 2>&1 && run
done
rm -f viterbi_kernel.o; make -s > /dev/null 2>&1
echo "== tree restored"
