# FB scaled kernels A/B on one box: the tree's fwbw_scaled_kernel.hip against another source file, same library otherwise.
#     bash tools/ubench/fb_ab_file.sh <variant.hip[.txt]>   [TESTS=1 runs the FB + EM parity tests on the variant]
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$(realpath "$1")
cd $R/nanocall_amd/csrc
# whatever ends this script (an error, an interrupt, a time limit): the tree's own object is rebuilt, never a variant's left behind
trap 'rm -f fwbw_scaled_kernel.o; make -s > /dev/null 2>&1' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
run() { (cd $R && for i in 1 2 3; do STEPS=20 python tools/bench_fwbw.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Mevent-rounds/s kernel_ms', d['kernel_ms'], 'frac', d['roofline']['frac'], 'clock', d['shader_clock_mhz_under_load'], 'lpd', d['log_pr_data_mean'])"; done); }
echo "== tree"; run
cp "$V" /tmp/fwbw_variant.hip
/opt/rocm/bin/hipcc $FLAGS -c /tmp/fwbw_variant.hip -o fwbw_scaled_kernel.o && make -s > /dev/null 2>&1
echo "== variant $(basename $V)"; run
[ -n "${TESTS:-}" ] && (cd $R && python -m pytest tests/test_fwbw_gpu.py tests/test_train_reads_gpu.py -x -q 2>&1 | tail -3)
rm -f fwbw_scaled_kernel.o; make -s > /dev/null 2>&1
echo "== tree again"; run
