#!/usr/bin/env python3
"""Summarise a tools/gpu_profile.sh output directory: bench line, rocprofv3 kernel stats, and
per-dispatch PMC means per kernel (separate --pmc passes)."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
print("== bench line ==")
try:
    print(open(os.path.join(out, "bench.json")).read().strip()[:2500])
except Exception as e:
    print("no bench.json", e)
print("== kernel stats (rocprofv3 --kernel-trace --stats, same command) ==")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    print(f"== {os.path.basename(d)} (per-dispatch mean, nchmm kernels) ==")
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "nchmm" not in kn:
                continue
            acc[(kn.split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kn, cn), v in sorted(acc.items()):
        print(f"  {kn:28s} {cn:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
