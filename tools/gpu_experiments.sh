#!/bin/bash
# tools/gpu_experiments.sh <which> -- the round-2 measurement recipes, run on the GPU box from the repo root
# (gpurun -- 'bash tools/gpu_experiments.sh budgets').  Results go to gpurun_out/; what was kept is under profiles/.
#
#   budgets   config-4 shard throughput against the back-pointer workspace budget (tail of a launch vs launch count)
#   rates     tools/ubench/valu_rate.hip: ns per wave-instruction per SIMD by instruction class and occupancy
#   hbm       torch fill / sum / copy rates of the device (calibration of "achievable" for the roofline fractions)
#   hostpath  nchmm_viterbi / nchmm_viterbi_raw from pageable host memory (wall vs kernels), and the box's PCIe / host-copy rates
#   cli       tools/bench_cli.py: FAST5 files -> nanocall -> FASTA, wall time by stage
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
case "${1:-}" in
budgets)
  for mb in 0 65536 32768 16384 8192; do
    if [ $mb = 0 ]; then unset NCHMM_WS_BUDGET_MB; else export NCHMM_WS_BUDGET_MB=$mb; fi
    python bench.py --reads 12500 --steps 3 --warmup 1 --no-cpu-baseline --no-fwbw 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget_mb=$mb', 'Mevents/s', d['value'], 'ms/step', d['ms_per_step'], 'launches', d['config']['forward_launches_per_step'])"
  done ;;
rates)
  (cd tools/ubench && hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate) | tee gpurun_out/valu_rate.txt ;;
hbm)
  python tools/ubench/hbm_rate.py | tee gpurun_out/hbm_rate.txt ;;
hostpath)
  python tools/bench_hostpath.py | tee gpurun_out/bench_hostpath.json
  python tools/ubench/pcie_rate.py | tee gpurun_out/pcie_rate.txt ;;
cli)
  READS=${READS:-8000} EVENTS=${EVENTS:-5000} THREADS=${THREADS:-32} python tools/bench_cli.py | tee gpurun_out/bench_cli.json ;;
*)
  echo "usage: bash tools/gpu_experiments.sh budgets|rates|hbm|hostpath|cli" ; exit 2 ;;
esac
