# Viterbi kernel A/B on one box: the tree's viterbi_kernel.hip against another source file (an experiment kept as text), same
# library otherwise, three bench runs each, tree measured before and after.   bash tools/ubench/vit_ab_file.sh <variant.hip[.txt]> [TESTS=1]
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$(realpath "$1")
cd $R/nanocall_amd/csrc
# whatever ends this script (an error, an interrupt, a time limit): the tree's own object is rebuilt, never a variant's left behind
trap 'rm -f viterbi_kernel.o; make -s > /dev/null 2>&1' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
run() { (cd $R && for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Mevents/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'tb_ms', d['roofline'].get('traceback_kernel_ms', 0), 'clock', d['device']['shader_clock_mhz_under_load'])"; done); }
echo "== tree"; run
cp "$V" /tmp/viterbi_variant.hip
/opt/rocm/bin/hipcc $FLAGS -c /tmp/viterbi_variant.hip -o viterbi_kernel.o && make -s > /dev/null 2>&1
echo "== variant $(basename $V)"; run
[ -n "${TESTS:-}" ] && (cd $R && python -m pytest tests/test_viterbi_gpu.py -x -q 2>&1 | tail -2)
rm -f viterbi_kernel.o; make -s > /dev/null 2>&1
echo "== tree again"; run
