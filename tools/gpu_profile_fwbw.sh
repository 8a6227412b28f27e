#!/bin/bash
# tools/gpu_profile_fwbw.sh <tag> -- PMC passes for the FB kernel (tools/bench_fwbw.py), run on the GPU box.
TAG=${1:-fwbw}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STEPS=${STEPS:-20}     # the --stats average runs over every launch: enough warm ones to outweigh the first two
BENCH="python3 $ROOT/tools/bench_fwbw.py"
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o fb -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq1 -o fb -- $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $OUT/pmc_sq2 -o fb -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fb -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o fb -- $BENCH > $OUT/pmc_write.log 2>&1
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
