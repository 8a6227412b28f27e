# Several Viterbi kernel variants against the tree on one box in one run: tree, each variant (2 bench runs, optional parity
# tests with TESTS=1), tree again.   bash tools/ubench/vit_ab_multi.sh <variant1.hip.txt> <variant2.hip.txt> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
# whatever ends this script (an error, an interrupt, a time limit): the tree's own object is rebuilt, never a variant's left behind
trap 'rm -f viterbi_kernel.o; make -s > /dev/null 2>&1' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
run() { (cd $R && for i in $(seq 1 ${RUNS:-2}); do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Mevents/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'tb_ms', d['roofline'].get('traceback_kernel_ms', 0), 'clock', d['device']['shader_clock_mhz_under_load'])"; done); }
echo "== tree"; run
for v in "$@"; do
    cp "$(realpath $R/$v 2>/dev/null || realpath $v)" /tmp/viterbi_variant.hip
    if /opt/rocm/bin/hipcc $FLAGS -c /tmp/viterbi_variant.hip -o viterbi_kernel.o 2>/tmp/variant_build.log && make -s > /dev/null 2>&1; then
        echo "== variant $(basename $v)"; run
        [ -n "${TESTS:-}" ] && (cd $R && python -m pytest tests/test_viterbi_gpu.py -x -q 2>&1 | tail -1)
    else
        echo "== variant $(basename $v): BUILD FAILED"; tail -5 /tmp/variant_build.log
    fi
done
rm -f viterbi_kernel.o; make -s > /dev/null 2>&1
echo "== tree again"; run
