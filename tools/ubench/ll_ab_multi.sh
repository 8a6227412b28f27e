# Low-latency sweep A/B on one box: the tree's viterbi_ll_kernel.hip against variant source files (experiments kept as text),
# same library otherwise; tools/bench_sweeps.py with the form forced, tree measured before and after.
#   bash tools/ubench/ll_ab_multi.sh variant1.hip.txt [variant2 ...]      SHAPES / REPS as in tools/bench_sweeps.py; TESTS=1: parity tests per variant
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
# whatever ends this script (an error, an interrupt, a time limit): the tree's own object is rebuilt, never a variant's left behind
trap 'rm -f viterbi_ll_kernel.o; make -s > /dev/null 2>&1' EXIT
FLAGS=$(make -s print-hipflags)
export SHAPES=${SHAPES:-1:5000,256:5000,1024:5000}
run() { (cd $R && python tools/bench_sweeps.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'reads' in d: print('  ', d['reads'], 'x', d['events_per_read'], ' ll kernel_ms', d['ll']['kernel_ms'], 'us/event', d['ll']['us_per_event_of_a_read'], ' wide kernel_ms', d['wide']['kernel_ms'], 'same_bits', d['same_bits'])
    else: print('  ', d)"); }
echo "== tree"; run
for V in "$@"; do
  cp "$(realpath $R/$V 2>/dev/null || realpath $V)" /tmp/viterbi_ll_variant.hip
  /opt/rocm/bin/hipcc $FLAGS -I$R/nanocall_amd/csrc -c /tmp/viterbi_ll_variant.hip -o viterbi_ll_kernel.o && make -s > /dev/null 2>&1
  echo "== variant $(basename $V)"; run
  [ -n "${TESTS:-}" ] && (cd $R && NCHMM_VIT_SWEEP=ll python -m pytest tests/test_viterbi_gpu.py -x -q 2>&1 | tail -2)
done
rm -f viterbi_ll_kernel.o; make -s > /dev/null 2>&1
echo "== tree again"; run
