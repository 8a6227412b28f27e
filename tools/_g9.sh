cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwbw 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['value'], d['roofline']['kernel_ms'])"
make -C nanocall_amd/csrc clean > /dev/null; make -C nanocall_amd/csrc -j16 HIPFLAGS='--offload-arch=gfx950 $(CXXFLAGS) -fno-slp-vectorize -DNCHMM_PK_EMISSION' > gpurun_out/r02i_build.log 2>&1
python -m pytest tests/test_viterbi_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-fwbw 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pk', d['value'], d['roofline']['kernel_ms'], d['cpu_baseline']['parity_checked_reads'])"
