cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_fwbw_gpu.py tests/test_train_reads_gpu.py -m gpu -x -q > gpurun_out/r02d_fbtest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02d_fbtest.log
tail -5 gpurun_out/r02d_fbtest.log
python tools/bench_fwbw.py > gpurun_out/r02d_bench_fwbw.json 2>&1; cat gpurun_out/r02d_bench_fwbw.json
bash tools/gpu_profile_fwbw.sh r02d_fwbw > /dev/null 2>&1; grep -E "scaled_kernel.*(AverageNs|SQ_INSTS_VALU |SQ_WAIT_ANY|SQ_WAVE_CYCLES)" gpurun_out/r02d_fwbw/summary.txt | head
# Viterbi variant: max3-based group scans
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwbw > gpurun_out/r02d_vit_base.json 2>&1
make -C nanocall_amd/csrc clean > /dev/null; make -C nanocall_amd/csrc -j16 HIPFLAGS='--offload-arch=gfx950 $(CXXFLAGS) -fno-slp-vectorize -DNCHMM_SCAN_MAX3' > gpurun_out/r02d_build.log 2>&1
python -m pytest tests/test_viterbi_gpu.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwbw > gpurun_out/r02d_vit_max3.json 2>&1
python -c "
import json
for f in ('gpurun_out/r02d_vit_base.json','gpurun_out/r02d_vit_max3.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['kernel_ms'])
    except Exception as e: print(f, 'ERR', e, open(f).read()[-500:])
"
