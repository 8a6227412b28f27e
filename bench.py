#!/usr/bin/env python3
"""bench.py -- Mevents/s of the HIP Viterbi decode (4096-state pore HMM) on the BASELINE.json workloads.

  python bench.py --gpus N --steps K --warmup W

  N = 1 : BASELINE config 2 -- 1024 synthetic reads x 5 000 events, template-only, builtin r73.t model.
  N > 1 : BASELINE config 4, sharded -- 12 500 reads x 5 000 events PER GPU (100 000 reads over 8 GPUs), one
          rank per GPU, reads assigned by nanocall_amd.shard.lpt_partition, no collective on the data path;
          one all-reduce (RCCL) gathers the counters and the max-over-ranks time.  A shard is ONE launch: the
          back-pointer workspace is one region per resident thread block (12 GB), not one row per event (256 GB).
          When the script is started directly (no RANK in the environment) it launches its N ranks itself
          (python -m torch.distributed.run) BEFORE touching the GPU, and fails if fewer than N GPUs are visible.
  --scaling strong : BASELINE config 4 AS WRITTEN at every N -- the same 100 000 reads x 5 000 events split N ways by
          lpt_partition (N = 1 decodes all 100 000 on one GPU), so that
          value(N) / value(1) is the strong-scaling curve north_star asks for ("scaling": "strong").  --reads then
          means the GLOBAL read count.  The default stays weak (per-GPU work fixed), which is what the driver's
          N = 1 / 2 / 4 / 8 runs without extra flags measure.

A "step" is one full pass of the hot path over the batch: forward sweep + back-pointer streaming + traceback
for every read, inputs already resident in HBM.  Consecutive steps are queued on the context's three lanes
(nchmm_viterbi_dev_enqueue) and joined once: the thread blocks of step k+1 start where those of step k run out of
reads (--serial-launches: every step behind the previous one, for profiling).  Rank 0 prints ONE JSON line; at N = 1 it also carries
`cpu_baseline` (the oracle timed on the host cores, GPU output checked bit for bit against it) and `fwbw`
(the forward-backward / EM-statistics kernels on the config-3 window shape).
"""
import argparse
import concurrent.futures
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_EVENT = 4113       # 4096 B back-pointers + 12 B inputs + 1 B traceback read + 4 B state out
FB_BYTES_PER_EVENT_ROUND = 32780   # 16 384 B alpha written + 16 384 B alpha read + 12 B inputs
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
SCLK_GHZ = 2.4               # same guide: peak engine clock
C4_READS_PER_GPU = 12500     # BASELINE config 4: 100 000 reads over 8 GPUs
C4_TOTAL_READS = 100000


def kernel_source_hash():
    """sha256 (first 16 hex) of the Viterbi kernel source: profiles are keyed on it so that replayed PMC figures
    cannot outlive the kernel they were measured on."""
    h = hashlib.sha256()
    for f in ("viterbi_kernel.hip", "viterbi_ll_kernel.hip", "emission_kernel.hip", "viterbi_common.hpp", "nchmm_device.h"):
        h.update(open(os.path.join(ROOT, "nanocall_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def fb_kernel_source_hash():
    """the same for the forward-backward kernels the `fwbw` leg times"""
    h = hashlib.sha256()
    for f in ("fwbw_scaled_kernel.hip", "fwbw_common.hpp", "nchmm_device.h"):
        h.update(open(os.path.join(ROOT, "nanocall_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def cpu_quota_cores():
    """CPUs' worth of time this process's cgroup allows (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited.  A container may
    show 256 CPUs and be allowed the time of 16: threads beyond that are only throttled (measured on the boxes this runs on: 128
    spinning processes do less work per second than 32)."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else -(-int(q) // int(p))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else -(-q // p)
    except (OSError, ValueError):
        return None


def baseline_threads(cpu_threads, cap=None):
    """threads of a CPU baseline: --cpu-threads if given, else one per physical core -- but no more than the cgroup's CPU quota, which
    is what the host can actually run at once ("T = number of physical host cores used")"""
    if cpu_threads:
        return cpu_threads
    logical, physical = physical_cores()
    t = max(1, min(physical, logical))
    q = cpu_quota_cores()
    if q:
        t = min(t, q)
    return min(t, cap) if cap else t


def physical_cores():
    """(logical CPUs usable by this process, physical cores behind them) from /proc/cpuinfo."""
    try:
        usable = sorted(os.sched_getaffinity(0))
    except AttributeError:
        usable = list(range(os.cpu_count() or 1))
    cores = set()
    try:
        cpu, phys, core = None, 0, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip():
                if cpu in usable and core is not None:
                    cores.add((phys, core))
                cpu, phys, core = None, 0, None
    except OSError:
        pass
    return len(usable), (len(cores) or len(usable))


def cpu_baseline(table, n_events, threads, n_reads):
    """Time the CPU oracle (port of the reference's Viterbi, reference memory layout) on a bounded sample of the
    same workload, read-parallel like the reference's pfor (one read per worker at a time): `threads` workers
    (default: one per PHYSICAL core of the host) share `n_reads` reads; T = 1 beside it.
    Long reads (config 5, 50 000 events: a 1.6 GB matrix per read, Viterbi.hpp:50): 8 reads on 8 threads, as BASELINE.md
    section 3 sizes it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nc_oracle as oracle
    import nanocall_amd as na
    from nanocall_amd import synth

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
    n_single = min(2 if n_events <= 10000 else 1, n_reads)
    for r in range(n_single):
        oracle.viterbi(om, ot, *prepped[r])
    dt1 = time.perf_counter() - t1
    logical, physical = physical_cores()
    quota = cpu_quota_cores()
    return dict(value=n_reads * n_events / dt / 1e6, unit="Mevents/s", cores=threads, kind="port",
                host_logical_cpus=logical, host_physical_cores=physical, host_cpu_quota_cores=quota,
                sample=f"{n_reads} reads x {n_events} events of the same synthetic workload on {threads} read-parallel threads "
                       f"(host: {logical} usable logical CPUs, {physical} physical cores, cgroup CPU quota {quota if quota else 'none'}; {-(-n_reads // threads)} read(s) per thread), "
                       f"oracle/nc_oracle.c (reference matrix layout, 8 B per cell), {dt:.1f} s wall = {dt * threads:.0f} CPU-seconds; "
                       f"single_thread_value = {n_single} of those reads on one thread, {dt1:.1f} s",
                single_thread_value=round(n_single * n_events / dt1 / 1e6, 5)), results, prepped


def committed_pmc(n_reads, n_events):
    """PMC figures of viterbi_kernel from the newest profiles/*hbm_traffic*.json whose workload AND kernel source
    hash match this tree (rocprofv3 cannot run inside this process).  None when the kernel changed since."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_c2*.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("workload") == {"reads": n_reads, "events": n_events} and t.get("kernel_source_sha256_16") == kernel_source_hash():
            best = dict(t, _file=os.path.relpath(f, ROOT))
    return best


def committed_pmc_fb(n_win, n_events):
    """the FB kernels' PMC figures (profiles/*hbm_traffic_fwbw*.json), same keying"""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_fwbw*.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("workload") == {"windows": n_win, "events": n_events} and t.get("kernel_source_sha256_16") == fb_kernel_source_hash():
            best = dict(t, _file=os.path.relpath(f, ROOT))
    return best


def generate_shard(table, read_ids, n_events, threads):
    """Synthetic events of the given (ascending) global read ids, generated in chunks on a few host threads
    (numpy releases the GIL) -> flat (off, mean, stdv, start)."""
    from nanocall_amd import synth
    read_ids = np.asarray(read_ids, np.int64)
    n = len(read_ids)
    mean = np.empty((n, n_events), np.float32)
    stdv = np.empty((n, n_events), np.float32)
    start = np.empty((n, n_events), np.float32)
    # runs of consecutive ids, cut into chunks of <= 512 reads
    chunks = []
    i = 0
    while i < n:
        j = i + 1
        while j < n and read_ids[j] == read_ids[j - 1] + 1 and j - i < 512:
            j += 1
        chunks.append((i, j))
        i = j

    def work(c):
        a, b = c
        ev = synth.generate(table, b - a, n_events, first_read=int(read_ids[a]))
        mean[a:b], stdv[a:b], start[a:b] = ev["mean"], ev["stdv"], ev["start"]

    with concurrent.futures.ThreadPoolExecutor(max(1, threads)) as ex:
        list(ex.map(work, chunks))
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(n_events)
    return off, mean.reshape(-1), stdv.reshape(-1), start.reshape(-1)


def end_to_end_leg(ctx, off, cm, sd, ls, d_state, d_logp, reps):
    """SURVEY 8d's second figure: the same batch from pageable HOST arrays to host arrays (12 B/event up, 2 B/event + 8 B/read
    down over PCIe).  Never the headline `value`; every output compared with the device-resident run of the timed region.
      value     a caller that streams batches (nchmm_viterbi_begin / _end, two in flight: batch k+1 is copied in and queued
                while batch k computes, batch k is handed over under the kernels of batch k+1) -- what the library's own
                chunk loops do.  Wall per batch over `reps` batches after 6 warm-up batches (the shader clock needs ~50 ms of
                uninterrupted load to come back up after the gaps of the one-call loop before it).
      one_call  nchmm_viterbi, one synchronous call per batch: the copies are exposed AND every launch starts after an idle
                gap, which costs the sweep itself 6-9 % (profiles/r04_pipeline_timeline.md)."""
    import torch
    n_reads = off.shape[0] - 1
    total = int(off[-1])
    want_states = d_state.cpu().numpy().view(np.uint16)
    want_logp = d_logp.cpu().numpy()
    ctx.use_own_stream()
    try:
        ctx.viterbi(off, cm, sd, ls)                 # first call sizes the library's staging buffers
        wall = []
        for _ in range(reps):
            t0 = time.perf_counter()
            states, logp, status = ctx.viterbi(off, cm, sd, ls)
            wall.append(time.perf_counter() - t0)
        assert (status == 0).all()
        assert np.array_equal(states, want_states), "host-pointer path decodes differently"
        assert logp.tobytes() == want_logp.tobytes()
        one_ms = float(np.median(wall)) * 1e3
        outs = [(np.empty(total, np.uint16), np.empty(n_reads, np.float32), np.zeros(n_reads, np.int32)) for _ in range(2)]
        warm = 6
        tk = ctx.viterbi_begin(off, cm, sd, ls, out=outs[0])
        per, same = [], True
        for i in range(1, warm + reps + 1):
            t0 = time.perf_counter()
            nxt = ctx.viterbi_begin(off, cm, sd, ls, out=outs[i & 1])
            st, lp, status = ctx.viterbi_end(tk)
            tk = nxt
            per.append(time.perf_counter() - t0)
            if i in (1, warm + reps):            # (the comparison is host work between batches: keep it out of most iterations)
                same = same and np.array_equal(st[:total], want_states) and lp[:n_reads].tobytes() == want_logp.tobytes() and (status == 0).all()
        st, lp, status = ctx.viterbi_end(tk)
        same = same and np.array_equal(st[:total], want_states) and lp[:n_reads].tobytes() == want_logp.tobytes()
        assert same, "streamed batches decode differently"
        ms = float(np.mean(per[warm:warm + reps - 1])) * 1e3     # (the last timed iteration carries the comparison)
    finally:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    return {"metric": "Mevents/s Viterbi, host pointers in and out (PCIe inclusive)", "value": round(total / ms / 1e3, 3), "unit": "Mevents/s",
            "ms_per_batch": round(ms, 3), "batches": reps - 1, "warm_up_batches": warm,
            "path": "nchmm_viterbi_begin / nchmm_viterbi_end, two batches in flight: pageable host SoA events -> H2D on the copy-in stream "
                    "under the previous batch's kernel -> viterbi_kernel (sweep + in-block traceback) writing states / log-probs into pinned "
                    "host memory -> memcpy into the caller's arrays under the next batch's kernel",
            "one_call": {"value": round(total / one_ms / 1e3, 3), "unit": "Mevents/s", "ms_per_call": round(one_ms, 3), "calls": reps,
                         "path": "nchmm_viterbi: H2D -> viterbi_kernel -> D2H, one synchronous call per batch"},
            "pcie_bytes_per_batch": int(12 * total + 8 * (n_reads + 1) + 2 * total + 8 * n_reads),
            "identical_to_device_resident_run": True}


def ragged_leg(ctx, table, reps=3):
    """The headline's reads are all 5000 events long; real reads are log-normally long, and a read is sequential.  This leg
    decodes 1024 reads with lengths lognormal(median 4000, sigma 0.7) clipped to [200, 30 000] (tools/bench_ragged.py's batch,
    seeded) from host arrays: one synchronous call per batch (it lasts as long as its longest read: the plan takes the
    one-read-per-CU form of the sweep, nchmm_plan.hpp) and a caller that streams batches, three in flight (the tail of one is
    covered by the next).  Both results compared with each other; never the headline `value`."""
    import torch
    import nanocall_amd as na
    from nanocall_amd import synth
    R = 1024
    rng = np.random.default_rng(20261002)
    lens = np.clip(np.round(np.exp(rng.normal(np.log(4000.0), 0.7, R))), 200, 30000).astype(np.int64)
    ev = synth.generate(table, R, int(lens.max()))
    keep = np.arange(int(lens.max()))[None, :] < lens[:, None]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    cm, sd, ls = na.events_prepare(ev["mean"][keep], ev["stdv"][keep], ev["start"][keep], 0.0)
    del ev, keep
    total = int(off[-1])
    ctx.use_own_stream()
    try:
        sw0 = ctx.sweep_stats()
        st, lp, status = ctx.viterbi(off, cm, sd, ls)              # sizes the staging buffers
        one = []
        for _ in range(reps):
            t0 = time.perf_counter()
            st, lp, status = ctx.viterbi(off, cm, sd, ls)
            one.append(time.perf_counter() - t0)
        k_ms = ctx.last_kernel_ms()[0]
        sw1 = ctx.sweep_stats()
        assert (status == 0).all()
        depth, nb = 3, 9
        best = 1e9
        for _ in range(2):
            tk = []
            t0 = time.perf_counter()
            for _b in range(nb):
                if len(tk) == depth:
                    ctx.viterbi_end(tk.pop(0))
                tk.append(ctx.viterbi_begin(off, cm, sd, ls))
            while len(tk) > 1:
                ctx.viterbi_end(tk.pop(0))
            st2, lp2, status2 = ctx.viterbi_end(tk.pop(0))
            best = min(best, time.perf_counter() - t0)
        assert np.array_equal(st[:total], st2[:total]) and lp[:R].tobytes() == lp2[:R].tobytes(), "streamed ragged batches decode differently"
    finally:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    return {"workload": "1024 reads, lengths lognormal(median 4000, sigma 0.7) clipped to [200, 30000] events, host arrays in and out",
            "events": total, "longest_read_events": int(lens.max()),
            "one_call": {"value": round(total / min(one) / 1e6, 1), "unit": "Mevents/s", "ms_per_call": round(min(one) * 1e3, 2),
                         "kernel_ms": round(k_ms, 2), "launches_wide_ll": [int(sw1[0] - sw0[0]) // (reps + 1), int(sw1[1] - sw0[1]) // (reps + 1)],
                         "note": "bounded below by the longest read: one block, one event after the other"},
            "streaming": {"value": round(nb * total / best / 1e6, 1), "unit": "Mevents/s", "batches": nb, "in_flight": depth},
            "output_sha256_16": hashlib.sha256(np.ascontiguousarray(st[:total]).tobytes() + np.ascontiguousarray(lp[:R]).tobytes()).hexdigest()[:16]}


def fwbw_cpu_baseline(tables, off, cm, sd, ls, strand, gpu_lpd, threads, n_cpu_win):
    """The oracle's Forward_Backward::fill (Forward_Backward.hpp:72-125, as Parameter_Trainer::fill_train_data calls it per window,
    Parameter_Trainer.hpp:141-155) timed on `n_cpu_win` of the SAME windows the GPU leg timed, window-parallel on `threads` host
    threads, and every one of those windows' log Pr(data) compared with the GPU's: within 1e-4 relative (north_star), in this run."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nc_oracle as oracle
    n_win = len(off) - 1
    pick = np.unique(np.linspace(0, n_win - 1, n_cpu_win).astype(np.int64))          # spread over reads and over both strands
    oms = [oracle.Model(t, (1.0, 0.0, 0.0, 1.0, 1.0, 1.0)) for t in tables]
    ot = oracle.Transitions(0.3, 0.1)
    lpd = np.zeros(len(pick), np.float64)

    def work(tid):
        for k in range(tid, len(pick), threads):
            w = int(pick[k])
            a, b = int(off[w]), int(off[w + 1])
            lpd[k] = float(oracle.fwbw(oms[int(strand[w])], ot, cm[a:b], sd[a:b], ls[a:b], want_matrices=False)[0])

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    rel = np.abs(gpu_lpd[pick].astype(np.float64) - lpd) / np.abs(lpd)
    assert np.isfinite(lpd).all() and (rel <= 1e-4).all(), f"forward log-likelihoods differ from the oracle: max relative {rel.max():.3g} (window {int(pick[rel.argmax()])})"
    events = int(sum(int(off[w + 1] - off[w]) for w in pick))
    logical, physical = physical_cores()
    return dict(value=round(events / dt / 1e6, 6), unit="Mevent-rounds/s", cores=threads, kind="port",
                host_logical_cpus=logical, host_physical_cores=physical,
                sample=f"{len(pick)} of the {n_win} timed windows ({events} event-rounds), window-parallel on {threads} threads, oracle/nc_oracle.c "
                       f"nco_fwbw_fill (reference layout: alpha + beta matrices, log-space logsumset per cell), {dt:.1f} s wall = {dt * min(threads, len(pick)):.0f} CPU-seconds",
                parity_checked_windows=int(len(pick)), parity_tolerance_rel=1e-4, parity_max_rel=float(f"{rel.max():.3g}"))


def fwbw_leg(ctx, dev, steps, cpu_threads=0, with_cpu=True):
    """The forward-backward + EM-statistics kernels on the BASELINE config-3 window shape: 1024 2D reads x
    (2 strands x 2 windows x 100 events) = 4096 windows, one pass ("event-round") per window, inputs resident; beside it the
    oracle's forward-backward on a sample of the same windows across the host cores, with the log-likelihoods compared."""
    import torch
    import nanocall_amd as na
    from nanocall_amd import synth
    n_reads, n_ev = 1024, 100
    t0, t1 = na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")
    e0 = synth.generate(t0, n_reads, 2 * n_ev)
    e1 = synth.generate(t1, n_reads, 2 * n_ev, first_read=10 ** 6)
    pick = lambda k: np.stack([e0[k][:, :n_ev], e0[k][:, n_ev:], e1[k][:, :n_ev], e1[k][:, n_ev:]], 1).reshape(-1)
    cm, sd, ls = na.events_prepare(pick("mean"), pick("stdv"), None, 0.0)
    n_win = n_reads * 4
    total = n_win * n_ev
    off = (np.arange(n_win + 1) * n_ev).astype(np.int64)
    strand = np.tile(np.array([0, 0, 1, 1], np.int32), n_reads)
    for s, t in enumerate((t0, t1)):
        ctx.put_model(2 + s, na.scaled_model_table(t))
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_off, d_cm, d_sd, d_ls, d_slot = d(off), d(cm), d(sd), d(ls), d(strand + 2)
    d_tr = torch.zeros(n_win, dtype=torch.int32, device=dev)
    d_sp = torch.tensor([0.1, 0.3], dtype=torch.float32, device=dev).repeat(n_win, 1).contiguous()
    d_lpd = torch.empty(n_win, dtype=torch.float32, device=dev)
    d_pm = torch.empty(total * 6, dtype=torch.float32, device=dev)
    d_st = torch.empty(n_win * 3, dtype=torch.float32, device=dev)

    def step():
        ctx.fwbw_dev(n_win, n_ev, total, d_off, d_cm, d_sd, d_ls, d_lpd, d_pm, d_st, d_scaled_slot=d_slot,
                     d_trans_slot=d_tr, d_st_params=d_sp)

    # untimed, until the pair's duration has settled (at least 6 launches, then four in a row within 1 %, at most 40): this leg starts
    # behind the CPU baseline's seconds of idle GPU, and a device that comes out of idle takes 50 ms and more of load to reach the
    # clock it then holds (the headline leg settles the same way)
    settle = []
    for _ in range(40):
        step()
        settle.append(ctx.last_kernel_ms()[2])
        if len(settle) >= 6 and max(settle[-4:]) <= 1.01 * min(settle[-4:]):
            break
    torch.cuda.synchronize()
    ks = []
    t_0 = time.perf_counter()
    for _ in range(steps):
        step()
        ks.append(ctx.last_kernel_ms()[2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_0
    k_ms = float(np.mean(ks))
    achieved = FB_BYTES_PER_EVENT_ROUND * total / (k_ms * 1e-3) / 1e9
    lpd = d_lpd.cpu().numpy()
    assert np.isfinite(lpd).all()
    traffic, traffic_src = None, None
    pmc = committed_pmc_fb(n_win, n_ev)
    if pmc:
        # FETCH_SIZE counts the backward sweep's 16-byte-per-lane streaming row reads at half their bytes on gfx950
        # (MI355X_MICROARCH.md, HBM section): doubled here, as that guide prescribes; writes as reported
        traffic = int(sum((2.0 * k.get("FETCH_SIZE_KiB", 0.0) + k.get("WRITE_SIZE_KiB", 0.0)) * 1024
                          for name, k in pmc.items() if name.startswith("fwbw_") and isinstance(k, dict)))
        traffic_src = pmc["_file"]
    base = None
    if with_cpu:
        # bounded sample: 128 windows over the threads the host may run at once (at most 64)
        threads = baseline_threads(cpu_threads, cap=64)
        base = fwbw_cpu_baseline((t0, t1), off, cm, sd, ls, strand, lpd, threads, 128)
        base["gpu_over_cpu"] = round(total * steps / dt / 1e6 / base["value"], 1)
    return {"metric": "FB + EM-statistics event-rounds/s", "value": round(total * steps / dt / 1e6, 3), "unit": "Mevent-rounds/s",
            "workload": "4096 windows x 100 events (config-3 shape: 1024 2D reads x 2 strands x 2 windows), r73.t / r73.c.p1",
            "steps": steps, "settling_launches": len(settle), "ms_per_step": round(dt / steps * 1e3, 3), "cpu_baseline": base,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_source_sha256_16": fb_kernel_source_hash(),
                         "kernel": "nchmm::fwbw_forward_scaled_kernel + nchmm::fwbw_backward_scaled_kernel",
                         "kernel_ms": round(k_ms, 3), "bytes_per_event_round": FB_BYTES_PER_EVENT_ROUND,
                         "event_rounds_per_launch": total},
            "log_pr_data_mean": float(lpd.mean())}


def oracle_train_job(oracle, opts, t0, t1, windows, wst):
    """One iteration of the reference's 2D model loop (train_reads, nanocall.cpp:360-426) on the oracle's train_one_round
    (Parameter_Trainer.hpp:541-579): round loop, stop on `done`, roll-back when the fit falls, round limit 2 x scaling_max_rounds,
    minimum progress.  -> (pm[6], st[4], fit, rounds, event-rounds of forward-backward it cost)"""
    off = np.concatenate([[0], np.cumsum([len(w[0]) for w in windows])]).astype(np.uint64)
    mean, stdv, start = (np.concatenate([w[k] for w in windows]) for k in range(3))
    crt_pm, crt_st = np.float32([1, 0, 0, 1, 1, 1]), np.float32([opts.default_p_stay, opts.default_p_skip] * 2)
    crt_fit, rnd, ev_rounds = np.float32(-np.inf), 0, 0
    while True:
        old_pm, old_st, old_fit = crt_pm.copy(), crt_st.copy(), crt_fit
        r = oracle.train_one_round(off, np.asarray(wst, np.uint32), mean, stdv, start, t0, t1, old_pm, old_st, opts.default_p_stay,
                                   opts.default_p_skip, opts.train_drift, bool(opts.train_scaling), bool(opts.train_transitions))
        ev_rounds += int(off[-1])
        crt_pm, crt_st, crt_fit = r["pm"], r["st"].copy(), r["fit"]
        if r["done"]:
            break
        if crt_fit < old_fit:
            crt_pm, crt_st, crt_fit = old_pm, old_st, old_fit
            break
        rnd += 1
        if rnd >= 2 * opts.scaling_max_rounds or (rnd > 1 and crt_fit < old_fit + opts.scaling_min_progress):
            break
    return crt_pm, crt_st, crt_fit, rnd, ev_rounds


def config3_leg(ctx, host_threads, cpu_threads, with_cpu, n_reads=1024, n_ev=5000):
    """BASELINE config 3 end to end: 1024 2D reads (template 5000 events from r73.t + complement 5000 from r73.c.p1), candidate
    pairs {t} x {c.p1, c.p2} = 2048 jobs, Parameter_Trainer EM on 2 x 100-event windows per strand, 4 rounds per pair
    (--scaling-max-rounds 2 in 2D semantics, minimum progress 0: nanocall.cpp:420), then Viterbi of both strands of every pair with
    its trained parameters and the choice of the better pair (nanocall.cpp:692-782).  Host arrays in and out, every host stage inside
    the clock (nchmm_train_reads, nchmm_basecall_reads).  Beside it the reference's loop on the oracle for a sample of the jobs, on
    the host cores, with round counts and fits compared (fit = the sum of the windows' forward log-likelihoods: 1e-4 relative)."""
    import torch
    import nanocall_amd as na
    from nanocall_amd import api
    names, strands = ["r73.c.p1", "r73.c.p2", "r73.t"], [1, 1, 0]           # sorted by name, like the reference's std::map
    tables = [na.builtin_model(n) for n in names]
    states = np.stack([na.model_load(t) for t in tables])
    t_gen = time.perf_counter()
    _, m0, s0, b0 = generate_shard(tables[2], np.arange(n_reads), n_ev, host_threads)
    _, m1, s1, b1 = generate_shard(tables[0], 10 ** 6 + np.arange(n_reads), n_ev, host_threads)
    il = lambda a, b: np.stack([a.reshape(n_reads, n_ev), b.reshape(n_reads, n_ev)], 1).reshape(-1)      # read-major, template then complement
    mean, stdv, start = il(m0, m1), il(s0, s1), il(b0, b1)
    del m0, m1, s0, s1, b0, b1
    _, stdv, _ = na.events_prepare(mean, stdv, None, 0.0)                    # Event::update_logs: stdv 0 -> .01
    t_gen = time.perf_counter() - t_gen
    so = (np.arange(2 * n_reads + 1) * n_ev).astype(np.uint64)
    opts = api.train_opts(scaling_max_rounds=2, scaling_min_progress=0.0)
    jr, j0, j1 = api.train_enumerate(opts, strands, so, np.ones(n_reads, np.uint8))
    nj = len(jr)
    ctx.use_own_stream()
    try:
        ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)     # sizes the workspaces (a chunk loop's first chunk)
        em = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
            em.append(time.perf_counter() - t0)
        bc, dec = None, []
        for _ in range(4):
            t0 = time.perf_counter()
            bc = ctx.basecall_reads(opts, states, so, mean, stdv, start, jr, j0, j1, out["pm"], out["st"], out=bc)
            dec.append(time.perf_counter() - t0)
        try:
            mhz = round(ctx.shader_clock_mhz())
        except Exception:
            mhz = None
    finally:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    assert (bc["best_job"] >= 0).all(), "a strand was left without a decoded candidate"
    rounds = out["rounds"].astype(np.int64)
    half = opts.scaling_num_events // 2
    ev_rounds = int((rounds * 4 * half).sum())             # 2 strands x 2 windows of `half` events per round
    t_em, t_dec = float(np.median(em)), float(np.median(dec[1:]))
    res = {"workload": f"BASELINE config 3: {n_reads} 2D reads x ({n_ev} template + {n_ev} complement events), pairs {{r73.t}} x {{r73.c.p1, r73.c.p2}} = {nj} jobs, "
                       f"EM on 2 x {half}-event windows per strand, then Viterbi of both strands of every pair; host arrays in and out",
           "jobs": int(nj), "em_rounds_per_job": {"mean": float(rounds.mean()), "min": int(rounds.min()), "max": int(rounds.max())},
           "em": {"value": round(ev_rounds / t_em / 1e6, 3), "unit": "Mevent-rounds/s", "wall_s": round(t_em, 4), "event_rounds": ev_rounds,
                  "calls": len(em), "path": "nchmm_train_reads: window gather + drift correction, forward-backward + statistics kernels, fp64 outer sums, 3x3 solve, round loop"},
           "decode": {"value": round(2 * nj * n_ev / t_dec / 1e6, 3), "unit": "Mevents/s", "wall_s": round(t_dec, 4), "events": int(2 * nj * n_ev),
                      "calls": len(dec) - 1, "path": "nchmm_basecall_reads: candidate tables on the device, raw copy-in, event prep, Viterbi of every candidate strand, winner choice"},
           "end_to_end_s": round(t_em + t_dec, 4), "shader_clock_mhz": mhz, "host_generation_s": round(t_gen, 1),
           "fit_mean": float(out["fit"].mean()),
           "output_sha256_16": hashlib.sha256(bc["states"].tobytes() + bc["best_logp"].tobytes() + out["pm"].tobytes()).hexdigest()[:16]}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import nc_oracle as oracle
        # bounded sample: 16 jobs on 16 threads -- a job is 4 rounds x 4 windows of the oracle's forward-backward (~0.5 s per window on
        # an idle host, 6 s with 128 of them running at once on a 16-CPU quota): ~15 s of wall
        threads = baseline_threads(cpu_threads, cap=16)
        pick = np.unique(np.linspace(0, nj - 1, max(16, threads)).astype(np.int64))
        got = [None] * len(pick)

        def work(tid):
            for k in range(tid, len(pick), threads):
                j = int(pick[k])
                windows, wst = [], []
                for s_ in (0, 1):
                    lo, hi = int(so[2 * jr[j] + s_]), int(so[2 * jr[j] + s_ + 1])
                    for sl in (slice(lo, lo + half), slice(hi - half, hi)):
                        windows.append((mean[sl], stdv[sl], start[sl])); wst.append(s_)
                got[k] = oracle_train_job(oracle, opts, tables[j0[j]], tables[j1[j]], windows, wst)

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(min(threads, len(pick)))]
        threads = len(th)
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        same_rounds = [int(out["rounds"][j]) == g[3] for j, g in zip(pick, got)]
        rel = np.array([abs(float(out["fit"][j]) - float(g[2])) / abs(float(g[2])) for j, g, ok in zip(pick, got, same_rounds) if ok])
        assert len(rel) and (rel <= 1e-4).all(), f"config 3: a job's fit differs from the reference loop on the oracle: max relative {rel.max():.3g}"
        cpu_rounds = int(sum(g[4] for g in got))
        # the trained parameters themselves, reported (not asserted: var / var_sd carry the reference's own fp32 noise, which compounds over
        # free-running rounds -- tests/test_fwbw_gpu.py holds them round by round against a float64 evaluation): largest relative distance
        # to the oracle's loop over the sampled jobs; shift and drift on the 60 pA level scale
        t_span = float(n_ev) * 0.02
        den = lambda q, v: {1: 60.0, 2: 60.0 / t_span}.get(q, abs(float(v)))
        pm_rel = {nme: float(f"{max(abs(float(out['pm'][j][q]) - float(g[0][q])) / den(q, g[0][q]) for j, g, ok in zip(pick, got, same_rounds) if ok):.3g}")
                  for q, nme in enumerate(("scale", "shift", "drift", "var", "scale_sd", "var_sd"))}
        st_rel = float(f"{max(float(np.max(np.abs(out['st'][j] - g[1]) / np.abs(g[1]))) for j, g, ok in zip(pick, got, same_rounds) if ok):.3g}")
        res["cpu_baseline"] = dict(value=round(cpu_rounds / dt / 1e6, 6), unit="Mevent-rounds/s", cores=threads, kind="port",
                                   sample=f"{len(pick)} of the {nj} jobs through the reference's round loop on oracle/nc_oracle.c (nco_train_one_round: forward-backward of 4 "
                                          f"windows + train_pm_params + train_st_params per round), job-parallel on {threads} threads, {dt:.1f} s wall; EM stage only",
                                   parity_checked_jobs=int(len(pick)), jobs_with_equal_round_count=int(sum(same_rounds)),
                                   parity_tolerance_rel=1e-4, parity_fit_max_rel=float(f"{rel.max():.3g}"),
                                   trained_params_max_rel_vs_oracle=pm_rel, trained_transitions_max_rel_vs_oracle=st_rel,
                                   gpu_over_cpu=round(res["em"]["value"] / (cpu_rounds / dt / 1e6), 1))
    return res


def same_shard_leg(ctx, table, dev, n_events, host_threads, steps, warmup):
    """BASELINE config-4's per-GPU shard (12 500 reads x n_events) on this one GPU, timed exactly like the headline: W warm-up
    steps, K steps queued on the lanes and joined once, inputs resident."""
    import torch
    import nanocall_amd as na
    n = C4_READS_PER_GPU
    off, mean, stdv, start = generate_shard(table, np.arange(n), n_events, host_threads)
    cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
    del mean, stdv, start
    total = n * n_events
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_cm, d_sd, d_ls = (torch.from_numpy(x).to(dev) for x in (cm, sd, ls))
    outs = [(torch.empty(total, dtype=torch.int16, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
             torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(3)]
    k = [0]

    def step():
        ctx.viterbi_dev_enqueue(n, n_events, total, d_off, d_cm, d_sd, d_ls, *outs[k[0] % 3])
        k[0] += 1

    for _ in range(max(1, warmup)):
        step()
    ctx.viterbi_dev_join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.viterbi_dev_join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    try:
        mhz = round(ctx.shader_clock_mhz())
    except Exception:
        mhz = None
    assert (outs[0][2].cpu().numpy() == 0).all()
    return {"value": round(total * steps / dt / 1e6, 3), "unit": "Mevents/s", "reads": n, "events_per_read": n_events, "steps": steps,
            "ms_per_step": round(dt / steps * 1e3, 3), "shader_clock_mhz": mhz,
            "note": "the shard one rank of the N > 1 (weak-scaling) run decodes, on one GPU: divide the N-GPU value by THIS for the scaling ratio"}


def pool_leg(args):
    """--pool: one process, one nchmm_pool over --gpus devices, the decode stage of the command line from host arrays
    (nchmm_pool_basecall_reads: LPT partition by events, per-device raw copy-in + device-side event prep + tables + Viterbi +
    winner choice, results back in input order), every host stage inside the clock.  NCHMM_BENCH_SHARE_GPU0=1 (test hook): all
    pool members on GPU 0.  Prints one JSON line; the pool's counters come back through its one RCCL all-reduce when the
    devices are distinct."""
    import nanocall_amd as na
    from nanocall_amd import api
    share = os.environ.get("NCHMM_BENCH_SHARE_GPU0") == "1"
    have = na.device_count()
    if have < (1 if share else args.gpus):
        sys.stderr.write(f"bench.py --pool: --gpus {args.gpus} requested but only {have} GPU(s) are visible\n")
        return 2
    ids = [0] * args.gpus if share else list(range(args.gpus))
    n_events = args.events
    n_reads = args.reads or (1024 if args.gpus == 1 else C4_READS_PER_GPU) * (1 if args.scaling == "strong" else args.gpus)
    table = na.builtin_model(args.model)
    states = np.stack([na.model_load(table)])
    t_gen = time.perf_counter()
    off, mean, stdv, start = generate_shard(table, np.arange(n_reads), n_events, max(1, min(32, os.cpu_count() or 1)))
    t_gen = time.perf_counter() - t_gen
    # template-only reads: strand 0 = the events, strand 1 empty (Fast5_Summary's strand bounds)
    so = np.zeros(2 * n_reads + 1, np.uint64)
    so[1::2] = off[1:]
    so[2::2] = off[1:]
    opts = api.train_opts()
    jr, j0, j1 = api.train_enumerate(opts, [0], so, np.zeros(n_reads, np.uint8))
    nj = len(jr)
    pm = np.tile(np.float32([1, 0, 0, 1, 1, 1]), (nj, 1))
    st = np.tile(np.float32([opts.default_p_stay, opts.default_p_skip] * 2), (nj, 1))
    total = n_reads * n_events
    with api.Pool(ids) as pool:
        walls = []
        out = None
        for _ in range(args.warmup + args.steps):
            t0 = time.perf_counter()
            out = pool.basecall_reads(opts, states, so, mean, stdv, start, jr, j0, j1, pm, st)
            walls.append(time.perf_counter() - t0)
        counters, used_rccl = pool.counters()
    timed = walls[args.warmup:]
    assert (out["best_job"][:, 0] >= 0).all(), "a read was not decoded"
    st_chk = out["states"][: min(n_reads, 64) * n_events].reshape(-1, n_events).astype(np.int64)
    a, b = st_chk[:, :-1], st_chk[:, 1:]
    assert ((a == b) | ((a & 1023) == (b >> 2)) | ((a & 255) == (b >> 4))).all(), "decoded path leaves the stay/step/skip graph"
    print(json.dumps({
        "metric": "Mevents/s Viterbi, device pool from host arrays (PCIe and host stages inclusive)", "value": round(total / float(np.mean(timed)) / 1e6, 3),
        "unit": "Mevents/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
        "ms_per_step": round(float(np.mean(timed)) * 1e3, 3),
        "step_ms": {"min": round(min(timed) * 1e3, 3), "median": round(float(np.median(timed)) * 1e3, 3), "max": round(max(timed) * 1e3, 3)},
        "data": "synthetic", "dtype": "f32",
        "config": {"workload": f"{n_reads} template-only reads x {n_events} events, builtin {args.model}, one candidate per read, through "
                               f"nchmm_pool_basecall_reads over devices {ids}", "reads_total": n_reads, "events_per_read": n_events,
                   "pool_devices": ids, "shared_gpu0_test_hook": share, "host_generation_s": round(t_gen, 1)},
        "counters": {"reads": int(counters[0]), "events": int(counters[1]), "launches": int(counters[3])},
        "counters_through_rccl": bool(used_rccl),
        "output_sha256_16": hashlib.sha256(out["states"].tobytes() + out["best_logp"].tobytes()).hexdigest()[:16]}), flush=True)
    return 0


def launch_ranks(args):
    """--gpus N without a launcher: start N ranks as a child process group.  Nothing here initialises the GPU
    (torch.cuda.device_count() only counts), and the child is spawned, never exec'ed over this process."""
    import torch
    have = torch.cuda.device_count()
    if os.environ.get("NCHMM_BENCH_SHARE_GPU0") == "1" and have >= 1:
        have = args.gpus          # test hook: all ranks share GPU 0 (see main())
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; refusing to report a "
                         f"{args.gpus}-GPU number from fewer devices\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: 1024 at N=1 = config 2, 12500 at N>1 = config-4 shard); "
                                                          "with --scaling strong: reads in total (default 100000 = config 4)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: per-GPU work fixed (default).  strong: BASELINE config 4 as written, the same 100 000 reads split over N")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg")
    ap.add_argument("--serial-launches", action="store_true",
                    help="queue every step behind the previous one (nchmm_viterbi_dev) instead of letting consecutive steps roll into "
                         "each other on the context's three lanes (nchmm_viterbi_dev_enqueue / _join); profiling runs use it so that "
                         "every kernel's duration in the trace is its own")
    ap.add_argument("--events", type=int, default=5000, help="events per read")
    ap.add_argument("--model", default="r73.t")
    ap.add_argument("--no-ragged", action="store_true", help="skip the leg on log-normally long reads (one call / streaming, host arrays; --no-end-to-end skips it too)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fwbw", action="store_true", help="skip the forward-backward leg (and config 3 with it)")
    ap.add_argument("--no-config3", action="store_true", help="skip the config-3 leg (2D reads, 4-round EM + decode, end to end)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-shard-leg", action="store_true",
                    help="N = 1 only: skip `n1_same_shard` (the 12 500-read shard every rank of an N > 1 run decodes, timed on this GPU, so "
                         "that the 1 -> N ratio of a weak-scaling series compares like with like)")
    ap.add_argument("--pool", action="store_true",
                    help="instead of the rank-per-GPU run: ONE process drives nchmm_pool_basecall_reads over --gpus devices from host arrays "
                         "(what the nanocall command line does) and reports Mevents/s including every host stage")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if args.pool:
        sys.exit(pool_leg(args))
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus}\n")
        sys.exit(2)

    import torch
    import torch.distributed as dist
    import nanocall_amd as na
    from nanocall_amd import shard

    # TEST HOOK (tests/test_bench_gpu.py): NCHMM_BENCH_SHARE_GPU0=1 puts every rank on GPU 0 with gloo for the two small
    # exchanges, so that the N > 1 code path (LPT shard per rank, per-rank generation, counter all-reduce, max over ranks,
    # JSON assembly) runs on a 1-GPU box.  RCCL cannot form a communicator from two ranks on one device; the line says so.
    share_gpu0 = os.environ.get("NCHMM_BENCH_SHARE_GPU0") == "1"
    if share_gpu0:
        local_rank_dev = 0
    else:
        local_rank_dev = local_rank
    if torch.cuda.device_count() <= local_rank_dev:
        sys.stderr.write(f"bench.py: rank {rank} has no GPU {local_rank_dev} ({torch.cuda.device_count()} visible)\n")
        sys.exit(2)
    dev = torch.device("cuda", local_rank_dev)
    torch.cuda.set_device(dev)
    # A process group whenever a launcher started this rank (torch.distributed.run exports RANK / MASTER_ADDR) -- also for ONE
    # rank: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` then takes the very calls an 8-rank run takes
    # (RCCL communicator, barrier, the counters' all-reduce, the per-rank all-gather) on the one GPU a test box has.
    group = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if group:
        if share_gpu0:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    table = na.builtin_model(args.model)
    n_events = args.events
    strong = args.scaling == "strong"
    if strong:
        global_reads = args.reads or C4_TOTAL_READS
        reads_per_gpu = -(-global_reads // world)          # nominal (LPT gives every rank floor or ceil of it)
    else:
        reads_per_gpu = args.reads or (1024 if world == 1 else C4_READS_PER_GPU)
        global_reads = reads_per_gpu * world
    # the global read set and its partition: every rank computes the same LPT assignment and takes its own shard
    global_lengths = np.full(global_reads, n_events, np.int64)
    mine = shard.lpt_partition(global_lengths, world)[rank]
    n_reads = len(mine)
    host_threads = max(1, min(32 if strong else 8, (os.cpu_count() or 1) // max(1, world)))
    t_gen = time.perf_counter()
    off, mean, stdv, start = generate_shard(table, mine, n_events, host_threads)
    cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
    del mean, stdv, start
    t_gen = time.perf_counter() - t_gen
    total = n_reads * n_events

    mem_free_before, mem_total = na.device_mem_info(local_rank_dev)
    ctx = na.Context(local_rank_dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # launches are ordered with torch's work on this stream
    ctx.put_model(0, na.scaled_model_table(table))
    ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))

    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_cm, d_sd, d_ls = (torch.from_numpy(x).to(dev) for x in (cm, sd, ls))
    # one set of outputs per compute lane of the library (kVitLanes = 3): steps k and k + 3 share a lane, which orders them;
    # steps on different lanes may be resident together and must not be handed the same arrays (include/nanocall_hip.h)
    N_LANES = 3
    outs = [(torch.empty(total, dtype=torch.int16, device=dev), torch.empty(n_reads, dtype=torch.float32, device=dev),
             torch.zeros(n_reads, dtype=torch.int32, device=dev)) for _ in range(1 if args.serial_launches else N_LANES)]
    d_state, d_logp, d_status = outs[0]

    # A step = one batch through the hot path: viterbi_kernel sweeps every read and each block walks its read back as soon as
    # the last column is done.  The steps are queued with nchmm_viterbi_dev_enqueue on the context's three lanes and joined once
    # at the end: the blocks of step k+1 start where the blocks of step k run out of reads, so no CU waits for the slowest
    # block of a step (what a caller with more than one batch does; --serial-launches queues each step behind the previous one).
    n_step = [0]

    def step():
        o = outs[n_step[0] % len(outs)]
        n_step[0] += 1
        if args.serial_launches:
            ctx.viterbi_dev(n_reads, n_events, total, d_off, d_cm, d_sd, d_ls, *o)
        else:
            ctx.viterbi_dev_enqueue(n_reads, n_events, total, d_off, d_cm, d_sd, d_ls, *o)

    # Untimed, before the W warm-up steps: launches until the kernel time has settled (at least 8, then four in a row within 1 %,
    # at most 40).  A GPU that comes out of idle -- a fresh lease, or a profiler session that has just ended -- takes anything from
    # 50 to several hundred ms of load to reach the clock it then holds (profiles/r04_hostpath_gap.json); W = 5 steps are 75 ms.
    settle = []
    for _ in range(40):
        ctx.viterbi_dev(n_reads, n_events, total, d_off, d_cm, d_sd, d_ls, *outs[0])
        settle.append(ctx.last_kernel_ms()[0])
        if len(settle) >= 8 and max(settle[-4:]) <= 1.01 * min(settle[-4:]):
            break
        if len(settle) >= 3 and sum(settle) > 2500.0:      # (launches of a whole shard: 2.5 s of load is settled enough)
            break
    c_settle = [int(x) for x in ctx.counters()]      # (the reported counters cover warm-up + timed steps, as before)
    for _ in range(args.warmup):
        step()
    ctx.viterbi_dev_join()
    torch.cuda.synchronize()
    launches0 = int(ctx.counters()[3]) - c_settle[3]
    if group:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.viterbi_dev_join()
    torch.cuda.synchronize()
    if group:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.synchronize()            # (reports a block that found no back-pointer region; cannot happen)
    local_counters = np.array([int(a) - b for a, b in zip(ctx.counters(), c_settle)] + [1], dtype=np.int64)   # last slot: "a rank was here"

    # the hot kernels are VALU-issue bound, so their duration follows the shader clock -- and the boxes of one pool do not
    # all sustain the same clock (the same binary: 15.8 ms and 23.2 ms per launch on two boxes), nor does one box hold one clock
    # through a run.  One ~3 ms probe per leg, while that leg's load is still on the chip.
    def clock():
        try:
            return ctx.shader_clock_mhz()
        except Exception as e:          # a diagnostic must not cost the run its line
            sys.stderr.write(f"bench.py: shader clock probe failed: {e}\n")
            return float("nan")

    sclk_mhz = clock()           # right behind the timed region
    # every output set of the timed region: the same bits (the steps decode the same batch)
    ref_state, ref_logp = outs[0][0].cpu().numpy().tobytes(), outs[0][1].cpu().numpy().tobytes()
    for o in outs[1:min(len(outs), n_step[0])]:
        assert o[0].cpu().numpy().tobytes() == ref_state and o[1].cpu().numpy().tobytes() == ref_logp, "overlapping steps returned different bits"
    # the kernel's own duration: launches one behind the other (nothing beside them), HIP events recorded by the library around
    # each on the stream it runs on; reading them waits for that launch only.  Wall-clocked as a whole too: the serial figure.
    kernel_ms = []
    n_serial = max(3, min(args.steps, 8))
    # (one untimed launch first: the first launch behind the join, the synchronisation and the clock probe starts on a GPU that
    # has been idle for a few hundred microseconds and has been seen to take 15 % longer -- it is the change of regime, not the kernel)
    ctx.viterbi_dev(n_reads, n_events, total, d_off, d_cm, d_sd, d_ls, *outs[0])
    torch.cuda.synchronize()
    t_serial = time.perf_counter()
    for _ in range(n_serial):
        ctx.viterbi_dev(n_reads, n_events, total, d_off, d_cm, d_sd, d_ls, *outs[0])
        kernel_ms.append(ctx.last_kernel_ms()[:2])
    torch.cuda.synchronize()
    t_serial = time.perf_counter() - t_serial
    sclk_serial_mhz = clock()    # right behind the serial launches
    red_dev = dev if (group and not share_gpu0) else None      # gloo reduces host tensors
    # per rank: wall of the timed region, shader clock behind it, the kernel alone -- one all-gather; the step time of the line is
    # the maximum over ranks, as the contract says
    per_rank = shard.gather_per_rank([dt, sclk_mhz, float(np.mean([k[0] for k in kernel_ms])), float(n_reads)], red_dev)
    dt = float(per_rank[:, 0].max())
    launches_per_step = (int(local_counters[3]) - launches0) // max(1, args.steps)
    counters = shard.gather_counters(local_counters, red_dev)

    # device memory at the high-water mark (the workspace and staging buffers are kept between calls, so "now" is the peak)
    lib_now, lib_peak = ctx.mem_stats()
    mem_free_after, _ = na.device_mem_info(local_rank_dev)
    if os.environ.get("NCHMM_PROFILE") == "1" and rank == 0:
        profile_report(ctx)
    status = d_status.cpu().numpy()
    assert (status == 0).all(), "a read failed to decode"
    # size-independent sanity on the full output of this rank: every consecutive state pair is an arc of the graph
    n_chk_reads = min(n_reads, 256)
    st = d_state[: n_chk_reads * n_events].cpu().numpy().view(np.uint16).reshape(n_chk_reads, n_events).astype(np.int64)
    a, b = st[:, :-1], st[:, 1:]
    arc_ok = (a == b) | ((a & 1023) == (b >> 2)) | ((a & 255) == (b >> 4))
    assert arc_ok.all(), "decoded path leaves the stay/step/skip graph"

    sw = ctx.sweep_stats()        # launches wide, launches ll (since the context was created: settling + warm-up + timed + serial)
    # N = 1, config 2: the shard every rank of an N > 1 run decodes (12 500 reads), on this GPU -- the like-for-like N = 1 point
    # of a weak-scaling series (a 12 500-read launch is 3-5 % faster per GPU than a 1024-read one: longer queue, same tail)
    same_shard = None
    if rank == 0 and world == 1 and not strong and not args.no_shard_leg and reads_per_gpu == 1024 and n_events == 5000:
        try:
            same_shard = same_shard_leg(ctx, table, dev, n_events, host_threads, args.steps, args.warmup)
        except Exception as e:      # a secondary leg must not cost the run its headline line
            sys.stderr.write(f"bench.py: same-shard leg failed: {e}\n")
            same_shard = {"error": str(e)}

    if rank == 0:
        value = global_reads * n_events * args.steps / dt / 1e6
        k_ms = float(np.mean([k[0] for k in kernel_ms]))
        events_per_launch = total if launches_per_step <= 1 else None
        events_per_block = total / max(1, min(ctx.grid_slots(), n_reads))
        cyc = lambda mhz, ms: (round(mhz * ms * 1e3 / events_per_block, 1) if mhz == mhz else None)
        if strong:
            which = "config 4 as written" if (global_reads == C4_TOTAL_READS and n_events == 5000) else "custom (strong scaling)"
            workload = (f"BASELINE {which}: {global_reads} reads x {n_events} events in total, split over {world} GPU(s) by lpt_partition "
                        f"({n_reads} reads on rank 0), template-only Viterbi, builtin {args.model} 6-mer model, identity scaling, "
                        f"transitions p_skip=.3 p_stay=.1")
        else:
            which = "config 2" if (world == 1 and reads_per_gpu == 1024 and n_events == 5000) else (
                "config-4 shard" if reads_per_gpu == C4_READS_PER_GPU and n_events == 5000 else "custom")
            workload = (f"BASELINE {which}: {reads_per_gpu} reads x {n_events} events per GPU ({reads_per_gpu * world} reads "
                        f"over {world} GPU(s)), template-only Viterbi, builtin {args.model} 6-mer model, identity scaling, "
                        f"transitions p_skip=.3 p_stay=.1")
        if not group:
            collective = "none (one rank)"
        elif share_gpu0:
            collective = f"gloo, communicator of {dist.get_world_size()} ranks all on GPU 0 (test hook)"
        else:
            collective = (f"rccl: one all-reduce of 9 counters + one all-gather of 4 numbers per rank (step time, clock, kernel time, "
                          f"reads) over a communicator of {dist.get_world_size()} ranks (one per GPU); nothing on the data path")
        result = {
            "metric": "Mevents/s Viterbi (4096-state HMM)", "value": round(value, 3), "unit": "Mevents/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            # vs_baseline: the reference publishes no number for this metric, so the contract's field stays
            # null; the ratio to the CPU path timed in this run is vs_cpu_baseline (= cpu_baseline.gpu_over_cpu, filled in below).
            # Boxes of the pool differ by 20 % in the clock they sustain and the kernels follow the clock: the headline's
            # clock-normalised form -- shader cycles a thread block spends per event of its reads -- travels right behind it.
            "vs_baseline": None, "vs_cpu_baseline": None,
            "cycles_per_block_event": cyc(sclk_mhz, dt / args.steps * 1e3), "shader_clock_mhz": (round(sclk_mhz) if sclk_mhz == sclk_mhz else None),
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "clock_normalised": {"cycles_per_block_event": cyc(sclk_mhz, dt / args.steps * 1e3),
                                            "shader_clock_mhz": (round(sclk_mhz) if sclk_mhz == sclk_mhz else None),
                                            "note": "value x this = blocks x clock; compare THIS across boxes and rounds (the pool's boxes sustain 1.9-2.35 GHz)"},
                       "reads_per_gpu": reads_per_gpu, "reads_total": global_reads, "events_per_read": n_events,
                       "parallelism": f"read-sharded x{world} (LPT, no data-path collective)",
                       "collective": collective,
                       "grid_slots": ctx.grid_slots(), "forward_launches_per_step": launches_per_step,
                       "settling_launches_before_warmup": len(settle),
                       "host_generation_s": round(t_gen, 1)},
            "counters": {"reads": int(counters[0]), "events": int(counters[1]), "bp_bytes": int(counters[2])},
            # taken from the all-reduced counters (every rank adds 1), not from the communicator's size
            "ranks_in_collective": int(counters[8]),
            # the same steps queued one behind the other (nchmm_viterbi_dev), wall-clocked right after the timed region: the
            # figure rounds 1-3 reported as `value`, kept beside the overlapping one for comparison across rounds
            "serial_launches": {"value": round(total * world * n_serial / t_serial / 1e6, 3) if world == 1 else None,
                                "ms_per_step": round(t_serial / n_serial * 1e3, 3), "steps": n_serial,
                                "launch_ms": {"min": round(float(np.min([k[0] for k in kernel_ms])), 3),
                                              "median": round(float(np.median([k[0] for k in kernel_ms])), 3),
                                              "max": round(float(np.max([k[0] for k in kernel_ms])), 3)},
                                "note": "rank 0; launch_ms = hipEvents around each launch on its stream"},
            # shader cycles a thread block spends per event of its reads (clock of the leg x time of the leg / events per block):
            # comparable across boxes and legs where Mevents/s and ms are not
            "cycles_per_event": {"timed_region": cyc(sclk_mhz, dt / args.steps * 1e3), "serial_launches": cyc(sclk_serial_mhz, k_ms),
                                 "events_per_block": round(events_per_block, 1),
                                 "note": "timed_region: overlapping steps (ms_per_step); serial_launches: a launch alone (kernel_ms); "
                                         "each with the clock probed right behind that leg"},
            "output_sha256_16": hashlib.sha256(d_state.cpu().numpy().tobytes() + d_logp.cpu().numpy().tobytes()).hexdigest()[:16],
            "device": {"peak_mem_bytes": int(mem_free_before - mem_free_after), "library_peak_bytes": int(lib_peak),
                       "workspace_budget_mb": (int(os.environ["NCHMM_WS_BUDGET_MB"]) if os.environ.get("NCHMM_WS_BUDGET_MB") else None),
                       "total_mem_bytes": int(mem_total),
                       "mem_note": "peak_mem_bytes = device memory taken by this process between context creation and the end of the timed "
                                   "region (inputs + outputs of the batch, the library's tables, staging and back-pointer workspace); "
                                   "library_peak_bytes = the library's own share at its high-water mark (nchmm_mem_stats)",
                       "shader_clock_mhz_under_load": (round(sclk_mhz, 0) if sclk_mhz == sclk_mhz else None), "peak_shader_clock_mhz": round(SCLK_GHZ * 1e3, 0),
                       "shader_clock_mhz_by_leg": {"timed_region": (round(sclk_mhz, 0) if sclk_mhz == sclk_mhz else None),
                                                   "serial_launches": (round(sclk_serial_mhz, 0) if sclk_serial_mhz == sclk_serial_mhz else None)},
                       "note": "rank 0, ~3 ms full-chip VALU probe right behind each leg (nchmm_shader_clock_mhz)"},
        }
        if group:
            stat = lambda v: {"min": round(float(np.min(v)), 3), "median": round(float(np.median(v)), 3), "max": round(float(np.max(v)), 3)}
            result["ranks"] = {"ms_per_step": stat(per_rank[:, 0] / args.steps * 1e3), "kernel_ms": stat(per_rank[:, 2]),
                               "shader_clock_mhz": [round(float(x)) if x == x else None for x in per_rank[:, 1]],
                               "reads": [int(x) for x in per_rank[:, 3]],
                               "note": "one all-gather of four numbers per rank; `ms_per_step` of the line is the maximum"}
        if same_shard is not None:
            result["n1_same_shard"] = same_shard
        kernel_name = "nchmm::viterbi_kernel" if sw[1] == 0 else ("nchmm::viterbi_ll_kernel" if sw[0] == 0 else "nchmm::viterbi_kernel + nchmm::viterbi_ll_kernel")
        result["config"]["sweep_launches_wide_ll"] = [int(sw[0]), int(sw[1])]
        if events_per_launch:
            achieved = BYTES_PER_EVENT * events_per_launch / (k_ms * 1e-3) / 1e9
            pmc = committed_pmc(n_reads, n_events)
            roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "kernel": kernel_name, "kernel_ms": round(k_ms, 3),
                    "kernel_ms_note": f"mean of {len(kernel_ms)} launches queued one behind the other right after the timed region (hipEvents "
                                      "on the launch stream): sweep + the traceback every block does when its read ends, nothing running "
                                      "beside it; in the timed region consecutive launches overlap at their edges, so a step costs "
                                      "ms_per_step, less than a launch lasts.  (Rounds 1-3 had a separate traceback kernel, ~0.45 ms per "
                                      "launch, which their kernel_ms / frac did not include.)",
                    "steps_overlap": not args.serial_launches,
                    # the same bytes against the rate the timed region sustained (launches overlapping at their edges): what a
                    # caller with more than one batch gets per launch; `frac` above stays the conservative kernel-alone figure
                    "frac_at_step_rate": round(BYTES_PER_EVENT * events_per_launch / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                    "bytes_per_event": BYTES_PER_EVENT, "events_per_launch": events_per_launch,
                    "kernel_source_sha256_16": kernel_source_hash()}
            if pmc:
                k = pmc["viterbi_kernel"]
                roof["traffic"] = int((k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024)
                roof["traffic_source"] = pmc.get("_file")
                valu = float(k["SQ_INSTS_VALU"])
                roof["valu_instructions_per_thread_event"] = round(valu / (n_reads * n_events * 8.0), 1)
                # the kernel is VALU-issue bound: floor = wave-instructions / SIMDs x 2 cycles (wave64 on a 32-lane-per-clock
                # fp32 pipe) at the peak engine clock; half-rate ops (compare / select / max / integer) make the real floor higher
                n_simd = 4 * (ctx.grid_slots() // 2)
                roof["valu_floor_ms"] = round(valu / n_simd * 2.0 / (SCLK_GHZ * 1e9) * 1e3, 3)
                roof["valu_floor_frac"] = round(roof["valu_floor_ms"] / k_ms, 4)
            result["roofline"] = roof
        else:
            # a step of several launches: report the whole step against the roofline
            achieved = BYTES_PER_EVENT * total / (dt / args.steps) / 1e9
            result["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                                  "kernel": f"{kernel_name} x{launches_per_step} per step (wall)",
                                  "bytes_per_event": BYTES_PER_EVENT, "events_per_step": total,
                                  "last_launch_kernel_ms": round(k_ms, 3)}
        if world == 1 and not args.no_end_to_end and total <= 64 * 1024 * 1024:
            try:
                result["end_to_end"] = end_to_end_leg(ctx, off, cm, sd, ls, d_state, d_logp, max(6, min(args.steps, 16)))
                c_e2e = clock()
                result["device"]["shader_clock_mhz_by_leg"]["end_to_end"] = round(c_e2e, 0) if c_e2e == c_e2e else None
            except Exception as e:      # a secondary leg must not cost the run its headline line
                sys.stderr.write(f"bench.py: end-to-end leg failed: {e}\n")
                result["end_to_end"] = {"error": str(e)}
        if world == 1 and not args.no_ragged and not args.no_end_to_end and not args.serial_launches:     # (both host-pointer legs go together)
            try:
                result["ragged"] = ragged_leg(ctx, table)
            except Exception as e:      # a secondary leg must not cost the run its headline line
                sys.stderr.write(f"bench.py: ragged leg failed: {e}\n")
                result["ragged"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            threads = baseline_threads(args.cpu_threads)      # T = physical cores the host may actually run at once
            if n_events <= 10000:
                # bounded sample: 256 reads (2 per thread on a 128-core host, 16 on a 16-CPU quota) -- ~10-20 s of wall, 164 MB of matrix per thread
                n_cpu_reads = max(256, threads)
            else:
                # long reads (config 5): 8 reads on 8 threads -- 1.6 GB of matrix per 50 000-event read
                threads = args.cpu_threads or min(8, threads)
                n_cpu_reads = min(n_reads, max(8, threads))
            base, oracle_results, prepped = cpu_baseline(table, n_events, threads, n_cpu_reads)
            # parity in the same run: every read the CPU timed must match the GPU output bit for bit
            states = d_state.cpu().numpy().view(np.uint16)
            logp = d_logp.cpu().numpy()
            n_chk = min(len(oracle_results), n_reads)
            for r, (s, mv, lp) in enumerate(oracle_results[:n_chk]):
                assert np.array_equal(states[r * n_events:(r + 1) * n_events], s), f"read {r}: path differs from oracle"
                assert np.float32(lp).tobytes() == np.float32(logp[r]).tobytes(), f"read {r}: path log-prob differs"
            base["parity_checked_reads"] = n_chk
            base["gpu_over_cpu"] = round(value / base["value"], 1)
            result["vs_cpu_baseline"] = base["gpu_over_cpu"]
            base["value"] = round(base["value"], 5)
            result["cpu_baseline"] = base
        if world == 1 and not args.no_fwbw:
            try:
                result["fwbw"] = fwbw_leg(ctx, dev, max(3, args.steps), args.cpu_threads, not args.no_cpu_baseline)
            except AssertionError:      # (a parity failure is not a secondary matter)
                raise
            except Exception as e:      # the secondary leg must not cost the run its headline line
                sys.stderr.write(f"bench.py: forward-backward leg failed: {e}\n")
                result["fwbw"] = {"error": str(e)}
        if world == 1 and not args.no_fwbw and not args.no_config3 and n_events == 5000:
            try:
                result["config3"] = config3_leg(ctx, host_threads, args.cpu_threads, not args.no_cpu_baseline)
            except AssertionError:
                raise
            except Exception as e:
                sys.stderr.write(f"bench.py: config-3 leg failed: {e}\n")
                result["config3"] = {"error": str(e)}
        print(json.dumps(result), flush=True)
    ctx.close()
    if group:
        dist.barrier()
        dist.destroy_process_group()


def profile_report(ctx):
    import collections
    tk = ctx.profile_ticks()
    sys.stderr.write(f"[phase ticks, 100 MHz, summed over blocks] forward={tk[0]} traceback={tk[1]} block={tk[2]} "
                     f"blocks={tk[3]} traceback segments re-walked={tk[4]} of {tk[5]} speculative; "
                     f"exact rescans={tk[6]} exact tie combines={tk[7]}\n")
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
    cnt = collections.Counter(key.tolist())
    sys.stderr.write(f"[placement] distinct CUs={len(cnt)} blocks/CU histogram={collections.Counter(cnt.values())}\n")
    t0b = pb[:, 0].min()
    st, en = (pb[:, 0] - t0b) / 1e5, (pb[:, 1] - t0b) / 1e5
    sys.stderr.write(f"[blocks] start ms min/med/max = {st.min():.2f}/{np.median(st):.2f}/{st.max():.2f}  end ms min/med/max = "
                     f"{en.min():.2f}/{np.median(en):.2f}/{en.max():.2f}  dur ms min/med/max = {(en-st).min():.2f}/{np.median(en-st):.2f}/{(en-st).max():.2f}\n")


if __name__ == "__main__":
    main()
