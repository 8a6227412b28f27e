#!/usr/bin/env python3
"""Randomised bit-parity sweep of the Viterbi path against the CPU oracle (run on the GPU box).

Per configuration: a random builtin model, random (valid) scaling parameters and transition probabilities, READS ragged reads.
Round 6: the events are no longer only draws from the model they are decoded with -- each read takes one of the kinds of
tests/adversarial.py (model-matched, another builtin model's stream, uniform levels, constant runs of 50-500 events, 1 % spikes at
+-20 sigma, stdv log-uniform over [0.01, 50], stdv == 0, abasic stretches) and a length log-uniform over [1, LONGEST] -- and every
configuration is decoded in EVERY form of the sweep named in FORMS (one context per form, the form forced with nchmm_set_sweep;
"auto" = the plan's choice), each compared with the oracle's k-mer path and path log-probability bit for bit.  Per form the sweep
also reports how often the two exactness branches of the kernels fired (nchmm_profile_ticks()[6..7]: group rescans, cells decided
by the tie rule) -- the contract (Viterbi.hpp:79-89,125-132) rests on them.

The inputs and the oracle's decode of a configuration are made by worker PROCESSES (they never touch the GPU); the parent decodes.

  CONFIGS=500 READS=6 LONGEST=30000 FORMS=wide,ll,ahead WORKERS=48 OUT=gpurun_out/parity_sweep.json python tools/parity_sweep.py
  KINDS=matched  LONGEST=2500 reproduces the round-5 sweep's distribution.
  APIS=host,raw,dev,strand: which entry points decode every configuration (default host = nchmm_viterbi):
    raw     nchmm_viterbi_raw: raw events in, drift correction + log(stdv) on the device (the glibc logf port)
    dev     nchmm_viterbi_dev: device pointers, the plan made on the device from the offsets (longest-first order, outliers)
    strand  nchmm_viterbi_strand: one strand per call from a thread per read, combined into launches by the library"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_READS = int(os.environ.get("READS", 6))
LONGEST = int(os.environ.get("LONGEST", 30000))
SEED = int(os.environ.get("SEED", 20260606))
KINDS = tuple(k for k in os.environ.get("KINDS", "").split(",") if k)


def default_workers():
    """oracle worker processes: the CPUs the box allows (its cgroup quota, when it has one), at most 48"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, min(48, -(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(48, (os.cpu_count() or 2) // 2))


def make_config(c):
    """(worker process) inputs of configuration c and the oracle's decode of every read"""
    import nanocall_amd as na
    from nanocall_amd import models
    import nc_oracle as oracle
    import adversarial
    kinds = KINDS or adversarial.KINDS
    rng = np.random.default_rng([SEED, c])
    meta, tables = models._load()
    m = int(rng.integers(len(tables)))
    table = tables[m]
    params = (float(rng.uniform(0.8, 1.2)), float(rng.uniform(-6, 6)), float(rng.uniform(-0.01, 0.01)),
              float(rng.uniform(0.7, 1.5)), float(rng.uniform(0.8, 1.25)), float(rng.uniform(0.5, 2.0)))
    p_skip, p_stay = float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.05, 0.4))
    lens = adversarial.log_uniform_lengths(rng, N_READS, LONGEST)
    read_kinds = [kinds[int(rng.integers(len(kinds)))] for _ in lens]
    other = tables[(m + 1 + int(rng.integers(len(tables) - 1))) % len(tables)]
    cms, sds, lss, raws = [], [], [], []
    for r, (n, kind) in enumerate(zip(lens, read_kinds)):
        mean, stdv, start = adversarial.events(kind, table, params, n, seed=1000 * c + r, other_table=other)
        cm, sd, ls = na.events_prepare(mean, stdv, start, params[2])
        cms.append(cm); sds.append(sd); lss.append(ls); raws.append((mean, stdv, start))
    om, ot = oracle.Model(table, params), oracle.Transitions(p_skip, p_stay)
    want = []
    for cm, sd, ls in zip(cms, sds, lss):
        s, mv, lp = oracle.viterbi(om, ot, cm, sd, ls)
        want.append((s, np.float32(lp)))
    return dict(c=c, model=m, name=meta["names"][m], params=params, trans=(p_skip, p_stay), lens=lens, kinds=read_kinds,
                off=np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64), cm=np.concatenate(cms), sd=np.concatenate(sds),
                ls=np.concatenate(lss), want=want, raw_mean=np.concatenate([x[0] for x in raws]), raw_stdv=np.concatenate([x[1] for x in raws]),
                raw_start=np.concatenate([x[2] for x in raws]))


def main():
    n_cfg = int(os.environ.get("CONFIGS", 40))
    forms = [f for f in os.environ.get("FORMS", os.environ.get("SWEEP", "auto")).split(",") if f]
    workers = int(os.environ.get("WORKERS", default_workers()))
    t0 = time.time()
    pool = mp.get_context("spawn").Pool(workers)          # before anything here touches the GPU; spawn: no forked HIP state either way
    todo = pool.imap(make_config, range(n_cfg), chunksize=1)
    import nanocall_amd as na
    from nanocall_amd import models
    meta, tables = models._load()
    os.environ["NCHMM_PROFILE"] = "1"
    ctxs = {}
    for f in forms:
        ctxs[f] = na.Context(0)
        ctxs[f].set_sweep(f)
    del os.environ["NCHMM_PROFILE"]
    apis = [a for a in os.environ.get("APIS", "host").split(",") if a]
    if "dev" in apis:
        import torch
        dev = torch.device("cuda", 0)
    if "strand" in apis:
        from concurrent.futures import ThreadPoolExecutor
        strand_pool = ThreadPoolExecutor(max(8, N_READS))

    def decode(ctx, api, cfg, table):
        """-> (states, logp, status) of every read of the configuration through one entry point"""
        off = cfg["off"]
        n = len(off) - 1
        lens = np.diff(off.astype(np.int64))
        if api == "host":
            return ctx.viterbi(off, cfg["cm"], cfg["sd"], cfg["ls"])
        if api == "raw":
            return ctx.viterbi_raw(cfg["raw_mean"], cfg["raw_stdv"], cfg["raw_start"], off[:-1].astype(np.uint64), lens.astype(np.uint32), np.full(n, cfg["params"][2], np.float32))
        if api == "dev":
            d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (off.astype(np.int64), cfg["cm"], cfg["sd"], cfg["ls"])]
            total = int(off[-1])
            d_state = torch.empty(max(total, 1), dtype=torch.int16, device=dev)
            d_logp = torch.empty(n, dtype=torch.float32, device=dev)
            d_status = torch.zeros(n, dtype=torch.int32, device=dev)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.viterbi_dev(n, int(lens.max()), total, *d, d_state, d_logp, d_status)
            torch.cuda.synchronize()
            ctx.use_own_stream()
            return d_state.cpu().numpy().view(np.uint16)[:total], d_logp.cpu().numpy(), d_status.cpu().numpy()
        if api == "strand":
            t6 = na.scaled_model_table(table, cfg["params"])
            futs = [strand_pool.submit(ctx.viterbi_strand, t6, cfg["trans"][0], cfg["trans"][1], cfg["cm"][int(off[r]):int(off[r + 1])],
                                       cfg["sd"][int(off[r]):int(off[r + 1])], cfg["ls"][int(off[r]):int(off[r + 1])]) for r in range(n)]
            got = [f.result() for f in futs]
            return (np.concatenate([g[0] for g in got]), np.array([g[1] for g in got], np.float32), np.array([0 if g[2] == 0 else 1 for g in got], np.int32))
        raise ValueError(api)

    stats = {f: dict(mismatches=0, rescans=0, tie_rule_cells=0) for f in forms}
    api_reads = {a: 0 for a in apis}
    by_kind = {}
    events = reads = longest_seen = 0
    for cfg in todo:
        table = tables[cfg["model"]]
        for f in forms:
            ctx = ctxs[f]
            for api in apis:
                ctx.put_model(0, na.scaled_model_table(table, cfg["params"]))
                ctx.put_transitions(0, *na.transitions_fast(*cfg["trans"]))
                states, logp, status = decode(ctx, api, cfg, table)
                tk = ctx.profile_ticks()
                stats[f]["rescans"] += int(tk[6]); stats[f]["tie_rule_cells"] += int(tk[7])
                api_reads[api] += len(cfg["lens"])
                for r, (n, (s, lp)) in enumerate(zip(cfg["lens"], cfg["want"])):
                    a, b = int(cfg["off"][r]), int(cfg["off"][r + 1])
                    ok = status[r] == 0 and np.array_equal(s, states[a:b]) and np.float32(lp).tobytes() == np.float32(logp[r]).tobytes()
                    if not ok:
                        stats[f]["mismatches"] += 1
                        print(f"MISMATCH form {f} api {api} config {cfg['c']} model {cfg['name']} params {cfg['params']} trans {cfg['trans']} read {r} "
                              f"kind {cfg['kinds'][r]} len {n}", flush=True)
        for n, k in zip(cfg["lens"], cfg["kinds"]):
            by_kind[k] = by_kind.get(k, 0) + 1
            events += n; reads += 1; longest_seen = max(longest_seen, n)
    pool.close(); pool.join()
    out = {"configs": n_cfg, "reads_per_config": N_READS, "reads_checked_per_form": reads, "events_per_form": events, "longest_read": longest_seen,
           "length_distribution": f"log-uniform over [1, {LONGEST}]", "reads_by_kind": by_kind, "forms": stats, "read_decodes_by_entry_point": api_reads,
           "launches_wide_ll_reads_wide_ll": {f: list(ctxs[f].sweep_stats()) for f in forms},
           "ahead_launches_reads_events": {f: list(ctxs[f].ahead_stats()) for f in forms},
           "mismatches": sum(s["mismatches"] for s in stats.values()), "seed": SEED, "oracle_workers": workers, "seconds": round(time.time() - t0, 1)}
    for ctx in ctxs.values():
        ctx.close()
    line = json.dumps(out)
    print(line)
    if os.environ.get("OUT"):
        os.makedirs(os.path.dirname(os.path.abspath(os.environ["OUT"])), exist_ok=True)
        with open(os.environ["OUT"], "w") as fh:
            fh.write(line + "\n")
    return 1 if out["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
