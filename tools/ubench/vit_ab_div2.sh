R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/nanocall_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize"
run() { (cd $R && for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwbw --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Mevents/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'clock', d['device']['shader_clock_mhz_under_load'])"; done); }
echo "== tree"; run
cp $R/tools/ubench/viterbi_div2_proto.hip.txt /tmp/viterbi_div2.hip
/opt/rocm/bin/hipcc $FLAGS -c /tmp/viterbi_div2.hip -o viterbi_kernel.o && make -s > /dev/null 2>&1
echo "== 2-op division prototype"; run
(cd $R && python -m pytest tests/test_viterbi_gpu.py -x -q 2>&1 | tail -3)
rm -f viterbi_kernel.o; make -s > /dev/null 2>&1
echo "== tree again"; run
