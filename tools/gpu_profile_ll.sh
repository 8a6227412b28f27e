#!/bin/bash
# tools/gpu_profile_ll.sh <tag> -- hardware counters of the low-latency sweep (viterbi_ll_kernel) beside the wide one, on the GPU box
# from the repo root (via gpurun).  The workload: 256 reads x 5000 events (one read per CU; the wide form puts them on 256 of its
# 512 block slots), each form forced, through tools/bench_sweeps.py.  PMC passes are separate runs with no tracing flags.
# Output: gpurun_out/<tag>/ll_counters.txt (per-dispatch means and per wave-column figures).
TAG=${1:-llprof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SHAPES=256:5000 REPS=4 MODES=wide,ll
CMD="python3 $ROOT/tools/bench_sweeps.py"
$CMD > $OUT/sweeps.json 2> $OUT/sweeps.err
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq1 -o ll -- $CMD > $OUT/pmc_sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $OUT/pmc_sq2 -o ll -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ll -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o ll -- $CMD > $OUT/pmc_write.log 2>&1
python3 - $OUT <<'PY' > $OUT/ll_counters.txt
import collections, csv, glob, os, sys
out = sys.argv[1]
acc = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "viterbi" not in kn: continue
            acc[(kn.split("(")[0].split("::")[-1], row["Counter_Name"])].append(float(row["Counter_Value"]))
reads, events = 256, 5000
print(f"workload: {reads} reads x {events} events, each form forced (tools/bench_sweeps.py); per-dispatch means of rocprofv3 --pmc passes")
print(open(os.path.join(out, "sweeps.json")).read().strip()[:1500])
mean = {k: sum(v) / len(v) for k, v in acc.items()}
for kern, waves_per_read, label in (("viterbi_kernel", 8, "wide: 8 waves per read"), ("viterbi_ll_kernel", 16, "low-latency: 16 waves per read")):
    ks = {c: mean[(k, c)] for (k, c) in mean if k == kern}
    if not ks: continue
    wc = reads * events * waves_per_read          # wave-columns per launch
    print(f"== {kern} ({label}; {wc:.3g} wave-columns per launch) ==")
    for c in sorted(ks): print(f"  {c:24s} {ks[c]:.6g}   per wave-column {ks[c] / wc:.2f}   per read-column {ks[c] / (reads * events):.1f}")
    if "SQ_ACTIVE_INST_ANY" in ks and "SQ_WAVE_CYCLES" in ks:
        print(f"  issue share: SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = {ks['SQ_ACTIVE_INST_ANY'] / ks['SQ_WAVE_CYCLES']:.3f}; "
              f"SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES = {ks.get('SQ_WAIT_INST_LDS', 0) / ks['SQ_WAVE_CYCLES']:.3f}; "
              f"SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = {ks.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, ks.get('SQ_ACTIVE_INST_LDS', 1)):.3f}")
    if "FETCH_SIZE" in ks and "WRITE_SIZE" in ks:
        print(f"  HBM traffic per launch (KiB counters as reported): fetch {ks['FETCH_SIZE'] * 1024 / 1e9:.3f} GB, write {ks['WRITE_SIZE'] * 1024 / 1e9:.3f} GB; "
              f"algorithmic 4113 B x {reads * events} events = {4113 * reads * events / 1e9:.3f} GB")
PY
cat $OUT/ll_counters.txt
