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
# the --stats average runs over EVERY launch of the command: warm-up steps (cold clocks) and the host-pointer leg's calls
# (the kernel after a PCIe gap) as well as the timed steps; the per-call trace separates them
# bench.py's launches, in order: settling launches (as many as it takes), W warm-up steps, K timed steps, then max(3, min(K, 8))
# launches one behind the other for roofline.kernel_ms (the profiled commands carry --no-end-to-end): the timed region is counted
# from the END of the trace
steps = int(os.environ.get("PROF_STEPS", 12))
tail = max(3, min(steps, 8))
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
    d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
         for r in csv.DictReader(open(f)) if "viterbi_kernel" in r["Kernel_Name"]]
    d = [x[1] for x in sorted(d)]
    if len(d) >= steps + tail:
        timed, rest = d[-(steps + tail):-tail], d[:-(steps + tail)] + d[-tail:]
        print(f"viterbi_kernel ({len(d)} launches, every one behind the previous): the {steps} of bench.py's timed region mean "
              f"{sum(timed) / len(timed):.3f} ms [{min(timed):.3f}, {max(timed):.3f}]; the other {len(rest)} (settling, warm-up, the "
              f"kernel_ms leg) mean {sum(rest) / max(1, len(rest)):.3f} ms")
# the default bench command (consecutive steps roll into each other on the context's lanes): how the launches overlap
for f in glob.glob(os.path.join(out, "stats_overlap", "**", "*kernel_trace.csv"), recursive=True):
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "viterbi_kernel" in r["Kernel_Name"])
    if len(d) >= steps + tail:
        t = d[-(steps + tail):-tail]
        span = (t[-1][1] - t[0][0]) / 1e6
        dur = [(b - a) / 1e6 for a, b in t]
        lap = [(t[i][1] - t[i + 1][0]) / 1e6 for i in range(len(t) - 1)]
        print(f"== overlapping steps (default bench command), the {steps} launches of the timed region: first start to last end {span:.3f} ms = "
              f"{span / len(t):.3f} ms per launch; each launch lasts {sum(dur) / len(dur):.3f} ms [{min(dur):.3f}, {max(dur):.3f}] from the "
              f"dispatch of its first block to the exit of its last; consecutive launches overlap by {sum(lap) / len(lap):.3f} ms "
              f"[{min(lap):.3f}, {max(lap):.3f}]")
traffic = collections.defaultdict(dict)
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
        short = kn.split("::")[-1]
        if cn in ("FETCH_SIZE", "WRITE_SIZE"):
            traffic[short][cn + "_KiB"] = sum(v) / len(v)
        elif cn == "SQ_INSTS_VALU":
            traffic[short][cn] = sum(v) / len(v)
# the figures bench.py replays for roofline.traffic / valu_floor_ms, keyed on the kernel source so they cannot go stale
if "viterbi_kernel" in traffic and "FETCH_SIZE_KiB" in traffic["viterbi_kernel"] and "WRITE_SIZE_KiB" in traffic["viterbi_kernel"]:
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("viterbi_kernel.hip", "viterbi_ll_kernel.hip", "emission_kernel.hip", "viterbi_common.hpp", "nchmm_device.h"):      # (= bench.py kernel_source_hash)
        h.update(open(os.path.join(root, "nanocall_amd", "csrc", f), "rb").read())
    reads, events = int(os.environ.get("PROF_READS", 1024)), int(os.environ.get("PROF_EVENTS", 5000))
    doc = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ_INSTS_VALU (separate passes, tools/gpu_profile.sh), per-dispatch means. "
                       "FETCH_SIZE as reported: the sweep's reads are scalar/uniform loads and 12 B/event, the in-block traceback's are scattered 16-byte "
                       "loads that cost a 64-byte sector each (95 % of the figure) -- neither is the wide coalesced streaming pattern the "
                       "microarch guide's 2x under-count applies to; writes are 8-byte-per-lane coalesced stores.",
           "workload": {"reads": reads, "events": events}, "kernel_source_sha256_16": h.hexdigest()[:16]}
    doc.update(traffic)
    json.dump(doc, open(os.path.join(out, "hbm_traffic_c2.json"), "w"), indent=1)
    print("== wrote", os.path.join(out, "hbm_traffic_c2.json"))
fb = {k: v for k, v in traffic.items() if k.startswith("fwbw_") and "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v}
if fb:
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("fwbw_scaled_kernel.hip", "fwbw_common.hpp", "nchmm_device.h"):
        h.update(open(os.path.join(root, "nanocall_amd", "csrc", f), "rb").read())
    doc = {"_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_profile_fwbw.sh), per-dispatch means, AS REPORTED. "
                       "The backward sweep streams its alpha rows with 16-byte-per-lane loads, which gfx950's FETCH_SIZE counts at half their "
                       "bytes (MI355X_MICROARCH.md, HBM section): bench.py doubles FETCH_SIZE when it fills fwbw.roofline.traffic.",
           "workload": {"windows": int(os.environ.get("PROF_WINDOWS", 4096)), "events": int(os.environ.get("PROF_EVENTS_PER_WINDOW", 100))},
           "kernel_source_sha256_16": h.hexdigest()[:16]}
    doc.update(fb)
    json.dump(doc, open(os.path.join(out, "hbm_traffic_fwbw.json"), "w"), indent=1)
    print("== wrote", os.path.join(out, "hbm_traffic_fwbw.json"))
