cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {  # tag, extra flags
  make -C nanocall_amd/csrc clean > /dev/null; make -C nanocall_amd/csrc -j16 HIPFLAGS="--offload-arch=gfx950 \$(CXXFLAGS) -fno-slp-vectorize $2" > gpurun_out/r02e_build_$1.log 2>&1
  python tools/bench_fwbw.py > gpurun_out/r02e_fb_$1.json 2>/dev/null
  timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fwbw > gpurun_out/r02e_vit_$1.json 2>/dev/null
  python - "$1" <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/r02e_fb_{t}.json').read().strip().splitlines()[-1]); fb=(d['value'], d['kernel_ms'])
except Exception as e: fb=('ERR',str(e)[:80])
try:
    d=json.loads(open(f'gpurun_out/r02e_vit_{t}.json').read().strip().splitlines()[-1]); v=(d['value'], d['roofline']['kernel_ms'])
except Exception as e: v=('ERR',str(e)[:80])
print(t, 'FB Mev-rounds/s, kernel ms', fb, ' VIT Mev/s, kernel ms', v)
PY
}
run base ""
run noload "-DNCHMM_EXP_NOLOAD"
run nobarrier "-DNCHMM_EXP_NOBARRIER"
run nostore "-DNCHMM_EXP_NOSTORE"
run noload_nobarrier "-DNCHMM_EXP_NOLOAD -DNCHMM_EXP_NOBARRIER"
run nostore_nobarrier "-DNCHMM_EXP_NOSTORE -DNCHMM_EXP_NOBARRIER"
make -C nanocall_amd/csrc clean > /dev/null; make -C nanocall_amd/csrc -j16 > /dev/null 2>&1
