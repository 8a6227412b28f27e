#!/usr/bin/env python3
"""bench.py -- Mevents/s of the HIP Viterbi decode on the BASELINE.json config-2 workload
(1024 synthetic reads x 5 000 events, template-only, builtin r73.t model, 4096-state HMM).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; reads shard with no collective on the
   data path -- each rank decodes its own 1024-read shard ("weak" scaling); one all-reduce gathers
   the counters and the max-over-ranks time.)

A "step" is one full pass of the hot path over the batch: forward sweep + back-pointer streaming +
traceback for every read, inputs already resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_EVENT = 4113       # 4096 B back-pointers + 12 B inputs + 1 B traceback read + 4 B state out
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(table, n_events, threads):
    """Time the CPU oracle (port of the reference's Viterbi, reference memory layout) on a bounded
    sample of the same workload: 4 reads per thread, read-parallel like the reference's pfor."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nc_oracle as oracle
    import nanocall_amd as na
    from nanocall_amd import synth

    n_reads = 4 * threads
    ev = synth.generate(table, n_reads, n_events)
    om = oracle.Model(table, (1.0, 0.0, 0.0, 1.0, 1.0, 1.0))
    ot = oracle.Transitions(0.3, 0.1)
    prepped = [na.events_prepare(ev["mean"][r], ev["stdv"][r], ev["start"][r], 0.0) for r in range(n_reads)]
    results = [None] * n_reads

    def work(tid):
        for r in range(tid, n_reads, threads):
            results[r] = oracle.viterbi(om, ot, *prepped[r])

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    # T = 1 as well: the first two reads again, one thread
    t1 = time.perf_counter()
    for r in range(min(2, n_reads)):
        oracle.viterbi(om, ot, *prepped[r])
    dt1 = time.perf_counter() - t1
    return dict(value=n_reads * n_events / dt / 1e6, unit="Mevents/s", cores=threads, kind="port",
                sample=f"{n_reads} reads x {n_events} events of the same synthetic workload, "
                       f"{threads} read-parallel threads, oracle/nc_oracle.c (reference matrix layout), {dt:.1f} s",
                single_thread_value=round(min(2, n_reads) * n_events / dt1 / 1e6, 5)), results, prepped


def measured_traffic(n_reads, n_events):
    """HBM bytes per launch of viterbi_kernel from the PMC passes committed under profiles/ (rocprofv3
    cannot run inside this process); null when no profile of this exact workload exists."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_c2.json")))
        if t["workload"] == {"reads": n_reads, "events": n_events}:
            k = t["viterbi_kernel"]
            return int((k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024)
    except Exception:
        pass
    return None


def measured_valu(n_reads, n_events):
    """VALU wave-instructions per thread and event of viterbi_kernel from the same committed PMC pass (SQ_INSTS_VALU /
    (events x 8 waves per block)): the kernel is bound by VALU issue, not by HBM -- reported next to the contract's HBM roofline."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_c2.json")))
        if t["workload"] == {"reads": n_reads, "events": n_events}:
            return round(t["viterbi_kernel"]["SQ_INSTS_VALU"] / (n_reads * n_events * 8.0), 1)
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1024, help="reads per GPU")
    ap.add_argument("--events", type=int, default=5000, help="events per read")
    ap.add_argument("--model", default="r73.t")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import nanocall_amd as na
    from nanocall_amd import synth, shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    table = na.builtin_model(args.model)
    n_reads, n_events = args.reads, args.events
    # each rank owns reads [rank*n_reads, (rank+1)*n_reads) of the global synthetic set
    ev = synth.generate(table, n_reads, n_events, first_read=rank * n_reads)
    off, mean, stdv, start = synth.flat_batch(ev)
    cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
    total = n_reads * n_events

    ctx = na.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.put_model(0, na.scaled_model_table(table))
    ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))

    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_cm, d_sd, d_ls = (torch.from_numpy(x).to(dev) for x in (cm, sd, ls))
    d_state = torch.empty(total, dtype=torch.int16, device=dev)
    d_logp = torch.empty(n_reads, dtype=torch.float32, device=dev)
    d_status = torch.zeros(n_reads, dtype=torch.int32, device=dev)

    def step():
        ctx.viterbi_dev(n_reads, n_events, total, d_off, d_cm, d_sd, d_ls, d_state, d_logp, d_status)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events recorded by the library around the kernel on the launch stream; reading them
        # waits for that launch only (steps are serialised on one stream anyway)
        kernel_ms.append(ctx.last_kernel_ms()[:2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, dev if world > 1 else None)
    counters = shard.gather_counters(ctx.counters(), dev if world > 1 else None)

    if os.environ.get("NCHMM_PROFILE") == "1" and rank == 0:
        tk = ctx.profile_ticks()
        sys.stderr.write(f"[phase ticks, 100 MHz, summed over blocks] forward={tk[0]} traceback={tk[1]} block={tk[2]} "
                         f"blocks={tk[3]} traceback segments re-walked={tk[4]} of {tk[5]} speculative\n")
        raw = ctx.profile_blocks()
        ng = ctx.grid_slots()
        pb = raw[:4096].reshape(2048, 2)[:ng].astype(np.int64)
        ids = raw[4096:4096 + ng]
        hwid = (ids & np.uint64(0xFFFFFFFF)).astype(np.int64)
        xcc = (ids >> np.uint64(32)).astype(np.int64) & 15
        dur = (pb[:, 1] - pb[:, 0]) / 1e5
        cu = (hwid >> 8) & 15
        sh = (hwid >> 12) & 1
        se = (hwid >> 13) & 7
        for x in range(8):
            m = xcc == x
            if m.sum():
                sys.stderr.write(f"[xcc {x}] n={m.sum()} dur mean={dur[m].mean():.2f} min={dur[m].min():.2f} max={dur[m].max():.2f}\n")
        key = xcc * 4096 + se * 512 + sh * 256 + cu
        import collections
        cnt = collections.Counter(key.tolist())
        sys.stderr.write(f"[placement] distinct CUs={len(cnt)} blocks/CU histogram={collections.Counter(cnt.values())}\n")
        for nb in sorted(set(cnt.values())):
            ks = [k for k, v in cnt.items() if v == nb]
            m = np.isin(key, ks)
            sys.stderr.write(f"[placement] CUs with {nb} blocks: mean dur {dur[m].mean():.2f} ms\n")
        t0b = pb[:, 0].min()
        st, en = (pb[:, 0] - t0b) / 1e5, (pb[:, 1] - t0b) / 1e5
        sys.stderr.write(f"[blocks] start ms min/med/max = {st.min():.2f}/{np.median(st):.2f}/{st.max():.2f}  end ms min/med/max = "
                         f"{en.min():.2f}/{np.median(en):.2f}/{en.max():.2f}  dur ms min/med/max = {(en-st).min():.2f}/{np.median(en-st):.2f}/{(en-st).max():.2f}\n")
        q = np.percentile(en - st, [5, 25, 75, 95])
        sys.stderr.write(f"[blocks] dur percentiles 5/25/75/95 = {q}\n")
    status = d_status.cpu().numpy()
    assert (status == 0).all(), "a read failed to decode"

    result = None
    if rank == 0:
        value = world * total * args.steps / dt / 1e6
        k_ms = float(np.mean([k[0] for k in kernel_ms]))
        tb_ms = float(np.mean([k[1] for k in kernel_ms]))
        achieved = BYTES_PER_EVENT * total / (k_ms * 1e-3) / 1e9
        result = {
            "metric": "Mevents/s Viterbi (4096-state HMM)", "value": round(value, 3), "unit": "Mevents/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n_reads} reads x {n_events} events per GPU, template-only Viterbi, "
                                   f"builtin {args.model} 6-mer model, identity scaling, transitions p_skip=.3 p_stay=.1",
                       "reads_per_gpu": n_reads, "events_per_read": n_events, "parallelism": f"read-sharded x{world}",
                       "grid_slots": ctx.grid_slots()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(n_reads, n_events),
                         "kernel": "nchmm::viterbi_kernel", "kernel_ms": round(k_ms, 3), "traceback_kernel_ms": round(tb_ms, 3),
                         "bytes_per_event": BYTES_PER_EVENT, "events_per_launch": total,
                         "valu_instructions_per_thread_event": measured_valu(n_reads, n_events)},
            "counters": {"reads": int(counters[0]), "events": int(counters[1]), "bp_bytes": int(counters[2])},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            base, oracle_results, prepped = cpu_baseline(table, n_events, threads)
            # parity in the same run: every read the CPU timed must match the GPU output bit for bit
            states = d_state.cpu().numpy().view(np.uint16)
            logp = d_logp.cpu().numpy()
            for r, (s, mv, lp) in enumerate(oracle_results):
                assert np.array_equal(states[r * n_events:(r + 1) * n_events], s), f"read {r}: path differs from oracle"
                assert np.float32(lp).tobytes() == np.float32(logp[r]).tobytes(), f"read {r}: path log-prob differs"
            base["parity_checked_reads"] = len(oracle_results)
            base["gpu_over_cpu"] = round(value / base["value"], 1)
            base["value"] = round(base["value"], 5)
            result["cpu_baseline"] = base
        print(json.dumps(result), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
