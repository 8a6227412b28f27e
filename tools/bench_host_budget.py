#!/usr/bin/env python3
"""Rehearsal of ONE rank of the 8-GPU run on a 1-GPU box: does 1/8 of the host keep one GPU fed?

The reference gives every read to one of `-t` pfor threads (nanocall.cpp:611-621) and writes FASTA as chunks complete
(:859-866).  Here a rank owns one GPU and, on an 8-GPU node, 1/8 of the host cores.  This tool runs the rank's work twice --
with the whole host, and confined (sched_setaffinity, BEFORE anything creates a thread or a context) to 1/8 of the physical
cores with their SMT siblings -- and reports medians of >= 5 repetitions:

  generation     bench.generate_shard + events_prepare of the rank's 12 500 x 5 000-event shard (what bench.py --gpus 8 does)
  decode         nchmm_viterbi on the whole shard: host arrays in and out, pipelined over read ranges (nchmm_pipeline.cpp)
  epilogue       nchmm_base_seq + nchmm_write_fasta per read on the rank's host threads (tools/bench_epilogue, C++ threads)
  rank_wall_s    `python bench.py --gpus 1 --reads 12500 --steps 20 --warmup 5` (legs that only N = 1 runs skipped): the wall time
                 of one rank of the driver's 8-GPU bench, generation included, against its 1 800 s limit

  python tools/bench_host_budget.py            -> one JSON object (both modes)
  python tools/bench_host_budget.py --mode confined|whole   (internal: one mode, one JSON line)"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_topology():
    """[(physical id, core id) -> sorted logical cpus] for the CPUs this process may use"""
    usable = set(os.sched_getaffinity(0))
    cores, cpu, phys, core = {}, None, 0, None
    for line in open("/proc/cpuinfo"):
        if line.startswith("processor"):
            cpu = int(line.split(":")[1])
        elif line.startswith("physical id"):
            phys = int(line.split(":")[1])
        elif line.startswith("core id"):
            core = int(line.split(":")[1])
        elif not line.strip():
            if cpu in usable and core is not None:
                cores.setdefault((phys, core), []).append(cpu)
            cpu, phys, core = None, 0, None
    return cores


def one_mode(mode, reads, events, reps):
    cores = cpu_topology()
    keys = sorted(cores)
    if mode == "confined":
        keep = keys[: max(1, len(keys) // 8)]
        cpus = sorted(c for k in keep for c in cores[k])
        os.sched_setaffinity(0, cpus)            # before numpy / torch / the library create any thread
    n_cpus = len(os.sched_getaffinity(0))
    n_phys = len({k for k in keys if set(cores[k]) & os.sched_getaffinity(0)})
    import numpy as np
    import bench
    import nanocall_amd as na
    from nanocall_amd import shard
    out = {"mode": mode, "logical_cpus": n_cpus, "physical_cores": n_phys, "reads": reads, "events_per_read": events}
    table = na.builtin_model("r73.t")
    mine = shard.lpt_partition(np.full(8 * reads, events, np.int64), 8)[0]
    host_threads = max(1, min(8, n_cpus))        # what bench.py gives a rank of 8
    gen = []
    for _ in range(3):
        t0 = time.perf_counter()
        off, mean, stdv, start = bench.generate_shard(table, mine, events, host_threads)
        cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
        gen.append(time.perf_counter() - t0)
    del mean, stdv, start
    out["generation_and_prepare_s"] = {"median": round(float(np.median(gen)), 3), "all": [round(x, 3) for x in gen], "threads": host_threads}
    total = reads * events
    ctx = na.Context(0)
    ctx.put_model(0, na.scaled_model_table(table))
    ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    ctx.viterbi(off[:1025], cm[:1024 * events], sd[:1024 * events], ls[:1024 * events])     # sizes nothing big; loads the kernels
    dec = []
    launches0 = int(ctx.counters()[3])
    for i in range(reps + 1):
        t0 = time.perf_counter()
        states, logp, status = ctx.viterbi(off, cm, sd, ls)
        dec.append(time.perf_counter() - t0)
    dec = dec[1:]                                # the first call sizes workspace + staging
    assert (status == 0).all()
    out["decode_host_pointers"] = {"median_s": round(float(np.median(dec)), 4), "Mevents_per_s": round(total / float(np.median(dec)) / 1e6, 1),
                                   "all_s": [round(x, 4) for x in dec], "launch_pairs_per_call": (int(ctx.counters()[3]) - launches0) // 2 // (reps + 1),
                                   "shader_clock_mhz_under_load": round(ctx.shader_clock_mhz())}
    gpu_rate = out["decode_host_pointers"]["Mevents_per_s"]
    ctx.close()
    # epilogue on the rank's threads (C++ tool; inherits this process's affinity)
    tool = os.path.join(ROOT, "tools", "bench_epilogue")
    if not os.path.exists(tool):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tools"), "bench_epilogue"], check=True)
    tlist = sorted({t for t in (1, 4, 8, 16, 32, 64, 128, n_cpus) if t <= n_cpus})
    by_t = {}
    for _ in range(reps):
        p = subprocess.run([tool, str(reads)] + [str(t) for t in tlist], capture_output=True, text=True, check=True,
                           env=dict(os.environ, GPU_MEVENTS_PER_S=str(gpu_rate)))
        for t, v in json.loads(p.stdout.strip().splitlines()[-1])["by_threads"].items():
            by_t.setdefault(int(t), []).append(v["Mevents_per_s"])
    out["epilogue_Mevents_per_s_by_threads"] = {str(t): {"median": round(float(np.median(v)), 1), "min": round(min(v), 1), "max": round(max(v), 1), "reps": len(v)}
                                                for t, v in sorted(by_t.items())}
    best = max(float(np.median(v)) for v in by_t.values())
    out["epilogue_best_median_over_this_rank_gpu_rate"] = round(best / gpu_rate, 2)
    # one rank of the driver's 8-GPU bench (weak scaling: 12 500 reads per GPU), generation included
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--reads", str(reads), "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-fwbw", "--no-end-to-end"], capture_output=True, text=True)
    wall = time.perf_counter() - t0
    line = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {}
    out["rank_of_8_bench"] = {"wall_s": round(wall, 1), "rc": p.returncode, "value_Mevents_per_s": line.get("value"),
                              "host_generation_s": line.get("config", {}).get("host_generation_s"), "driver_limit_s": 1800}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("whole", "confined"))
    ap.add_argument("--reads", type=int, default=12500)
    ap.add_argument("--events", type=int, default=5000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    if a.mode:
        return one_mode(a.mode, a.reads, a.events, a.reps)
    res = {}
    for mode in ("whole", "confined"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", mode, "--reads", str(a.reads), "--events", str(a.events),
                            "--reps", str(a.reps)], capture_output=True, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stderr[-3000:])
            sys.exit(1)
        res[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    w, c = res["whole"], res["confined"]
    res["summary"] = {
        "decode_confined_over_whole": round(c["decode_host_pointers"]["Mevents_per_s"] / w["decode_host_pointers"]["Mevents_per_s"], 3),
        "generation_confined_over_whole_time": round(c["generation_and_prepare_s"]["median"] / w["generation_and_prepare_s"]["median"], 2),
        "epilogue_keeps_up_when_confined": c["epilogue_best_median_over_this_rank_gpu_rate"] >= 1.0,
        "rank_wall_s_confined": c["rank_of_8_bench"]["wall_s"], "driver_limit_s": 1800,
        "what": "one rank of the 8-GPU run rehearsed on one GPU with 1/8 of the host's physical cores (and their SMT siblings)"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
