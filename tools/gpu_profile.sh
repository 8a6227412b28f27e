#!/bin/bash
# tools/gpu_profile.sh <tag> -- run on the GPU box (via gpurun) from the repo root.
# Produces gpurun_out/<tag>/{bench.json, stats/, pmc_*/}: the bench line, the rocprofv3 kernel-trace
# stats of the same command, and PMC passes (separate runs, no tracing flags mixed in).
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --serial-launches: every step behind the previous one, so that each kernel in the trace ran alone and its duration is its own (by default
# consecutive steps roll into each other on the lanes of the context and a launch spans more than its share of the wall time)
# --no-end-to-end: that leg streams batches through the lanes, i.e. overlaps its launches again
BENCH="python3 $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-fwbw --no-end-to-end --no-shard-leg --serial-launches"
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o vit -- $BENCH > $OUT/stats.log 2>&1
# the default command (what the driver runs): steps overlap at their edges
BENCH_DEFAULT="python3 $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-fwbw --no-end-to-end --no-shard-leg"
$BENCH_DEFAULT > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_overlap -o vit -- $BENCH_DEFAULT > $OUT/stats_overlap.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq1 -o vit -- $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $OUT/pmc_sq2 -o vit -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o vit -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o vit -- $BENCH > $OUT/pmc_write.log 2>&1
# the low-latency form (viterbi_ll_kernel) and the emission kernel: a launch of 256 reads (one per CU) and a strand on its own
BENCH_LL="python3 $ROOT/tools/bench_sweeps.py"
SHAPES=256:5000,1:5000 REPS=4 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats_ll -o ll -- $BENCH_LL > $OUT/stats_ll.log 2>&1
find $OUT -name '*.csv' | head -50 > $OUT/files.txt
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
