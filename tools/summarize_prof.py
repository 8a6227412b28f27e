#!/usr/bin/env python3
"""Summarise a tools/gpu_profile.sh output directory: kernel stats + per-dispatch PMC means."""
import csv, glob, os, sys, collections
out = sys.argv[1]
print("== bench line ==")
try:
    print(open(os.path.join(out, "bench.json")).read().strip()[:1500])
except Exception as e:
    print("no bench.json", e)
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    print(f"== {os.path.basename(d)} (per-dispatch mean over viterbi_kernel dispatches) ==")
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "viterbi" not in row.get("Kernel_Name", ""):
                continue
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"  {k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
