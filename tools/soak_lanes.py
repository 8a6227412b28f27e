#!/usr/bin/env python3
"""Soak of what lets launches overlap: the back-pointer region pool (regions taken and returned by the blocks themselves, per
XCD), the three compute lanes, up to three batches in flight.  A handful of ragged batches (1 .. MAXREADS reads of 50 .. MAXLEN
events, log-normal) is decoded once per batch by the one-call form, then ITER times in random order

  * through nchmm_viterbi_begin / _end with 1, 2 or 3 batches in flight (chosen at random as it goes), and
  * through nchmm_viterbi_dev_enqueue on device copies of the batches with one join at the end of each iteration,

and every result must equal the first one bit for bit (a stale back-pointer row, a region handed to two blocks, a launch
reading another launch's queue: all show up as a different path).  ORACLE short reads of every batch are also compared with
the CPU oracle, so that "equal to the first result" means "right".   ITER=30 python tools/soak_lanes.py   (on the GPU box)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nanocall_amd as na                 # noqa: E402
from nanocall_amd import synth            # noqa: E402

ITER = int(os.environ.get("ITER", 30))
NB = int(os.environ.get("BATCHES", 7))
MAXREADS = int(os.environ.get("MAXREADS", 2500))
MAXLEN = int(os.environ.get("MAXLEN", 20000))
ORACLE = int(os.environ.get("ORACLE", 6))
rng = np.random.default_rng(int(os.environ.get("SEED", 777)))
table = na.builtin_model("r73.t")

batches = []
for b in range(NB):
    n = int(rng.integers(1, MAXREADS + 1)) if b else MAXREADS          # (one batch at the maximum: several ranges)
    lens = np.clip(np.round(np.exp(rng.normal(np.log(1500), 1.0, n))), 50, MAXLEN).astype(np.int64)
    if b == 1:
        lens[: min(n, 3)] = 0                                           # empty reads ride along
    ev = synth.generate(table, n, int(lens.max()), first_read=100000 * b)
    keep = np.arange(int(lens.max()))[None, :] < lens[:, None]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    cm, sd, ls = na.events_prepare(ev["mean"][keep], ev["stdv"][keep], ev["start"][keep], 0.0)
    batches.append(dict(off=off, cm=cm, sd=sd, ls=ls, lens=lens))
    del ev

ctx = na.Context(0)
ctx.put_model(0, na.scaled_model_table(table))
ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
for B in batches:
    B["ref"] = ctx.viterbi(B["off"], B["cm"], B["sd"], B["ls"])
    assert (B["ref"][2] == 0).all()

# the oracle on the shortest non-empty reads of every batch
import nc_oracle as oracle                # noqa: E402
om, ot = oracle.Model(table, (1.0, 0.0, 0.0, 1.0, 1.0, 1.0)), oracle.Transitions(0.3, 0.1)
n_oracle = 0
for B in batches:
    nz = np.flatnonzero(B["lens"] > 0)
    for r in nz[np.argsort(B["lens"][nz])][:ORACLE]:
        a, e = int(B["off"][r]), int(B["off"][r + 1])
        s, mv, lp = oracle.viterbi(om, ot, B["cm"][a:e], B["sd"][a:e], B["ls"][a:e])
        assert np.array_equal(s, B["ref"][0][a:e]) and np.float32(lp).tobytes() == np.float32(B["ref"][1][r]).tobytes()
        n_oracle += 1


def same(B, got):
    st, lp, status = got
    nz = B["lens"] > 0
    return np.array_equal(st, B["ref"][0]) and lp[nz].tobytes() == B["ref"][1][nz].tobytes() and (status == 0).all()


bad = 0
events = 0
t0 = time.time()
# ---- host-pointer streaming, 1 .. 3 in flight ----
for it in range(ITER):
    order = rng.permutation(NB)
    fly = []
    for b in order:
        depth = int(rng.integers(1, 4))
        while len(fly) >= depth:
            bb, tk = fly.pop(0)
            bad += not same(batches[bb], ctx.viterbi_end(tk))
        B = batches[b]
        fly.append((b, ctx.viterbi_begin(B["off"], B["cm"], B["sd"], B["ls"])))
        events += int(B["off"][-1])
    while fly:
        bb, tk = fly.pop(0)
        bad += not same(batches[bb], ctx.viterbi_end(tk))
t_stream = time.time() - t0

# ---- device-pointer enqueue / join ----
import torch                               # noqa: E402
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for B in batches:
    B["d"] = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (B["off"].astype(np.int64), B["cm"], B["sd"], B["ls"])]
    n, tot = len(B["lens"]), int(B["off"][-1])
    B["o"] = (torch.empty(max(tot, 1), dtype=torch.int16, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
              torch.empty(n, dtype=torch.int32, device=dev))
torch.cuda.synchronize()
t0 = time.time()
for it in range(ITER):
    for B in batches:
        for o in B["o"]:
            o.fill_(-1)
    for b in rng.permutation(NB):
        B = batches[b]
        ctx.viterbi_dev_enqueue(len(B["lens"]), int(B["lens"].max()), int(B["off"][-1]), *B["d"], *B["o"])
        events += int(B["off"][-1])
    ctx.viterbi_dev_join()
    ctx.synchronize()
    for B in batches:
        tot = int(B["off"][-1])
        bad += not same(B, (B["o"][0][:tot].cpu().numpy().view(np.uint16), B["o"][1].cpu().numpy(), B["o"][2].cpu().numpy()))
t_dev = time.time() - t0
peak = ctx.mem_stats()[1]
ctx.close()
print(json.dumps({"batches": NB, "reads": [int(len(B["lens"])) for B in batches], "events_per_batch": [int(B["off"][-1]) for B in batches],
                  "longest_read": int(max(B["lens"].max() for B in batches)), "iterations": ITER, "batch_decodes": 2 * ITER * NB,
                  "events_decoded": events, "mismatching_batches": int(bad), "oracle_checked_reads": n_oracle,
                  "streaming_s": round(t_stream, 1), "device_enqueue_s": round(t_dev, 1), "library_peak_bytes": int(peak)}))
sys.exit(1 if bad else 0)
