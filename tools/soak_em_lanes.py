#!/usr/bin/env python3
"""Soak of the EM driver's parts and lanes (nchmm_train.cpp) on one context that also decodes.

The second EM lane computes on the stream of Viterbi lane 1 and takes its alpha rows from the context's one workspace; parts change
lanes from round to round; the lanes' staging and arenas are reused from call to call.  ITER times: a random subset of a pool of 2D
reads (matched and adversarial windows, tests/adversarial.py) is trained with the jobs in one part (NCHMM_EM_LANES=1), then in parts
on two lanes -- under a forward-backward budget drawn at random per context, so that the number of parts runs from 2 to dozens --
and the two results must be equal bit for bit (parameters, fits, round counts, preferences).  Around the training calls the same
context decodes: a ragged batch through the one-call form, and (every third iteration) a device-resident batch queued with
nchmm_viterbi_dev_enqueue BEFORE the training call and joined after it, each compared with the first decode of that batch.

  ITER=40 python tools/soak_em_lanes.py      (GPU box)  -> one JSON line, exit code 1 on any difference"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import nanocall_amd as na                 # noqa: E402
from nanocall_amd import api, synth       # noqa: E402
import adversarial                        # noqa: E402

ITER = int(os.environ.get("ITER", 40))
POOL = int(os.environ.get("POOL", 400))
rng = np.random.default_rng(int(os.environ.get("SEED", 99)))
names = ["r73.c.p1", "r73.c.p2", "r73.t"]
strands = [1, 1, 0]
tables = [na.builtin_model(n) for n in names]
states = np.stack([na.model_load(t) for t in tables])
opts = api.train_opts(scaling_max_rounds=3, scaling_num_events=200, scaling_select_threshold=5.0)

pool = []
for r in range(POOL):
    two_d = r % 4 != 3
    rd = []
    for s in range(2):
        n = int(rng.integers(120, 1200)) if (s == 0 or two_d) else 0
        if not n:
            rd.append(None)
            continue
        kind = adversarial.KINDS[int(rng.integers(len(adversarial.KINDS)))] if rng.random() < 0.3 else "matched"
        params = (float(rng.uniform(0.9, 1.1)), float(rng.uniform(-4, 4)), float(rng.uniform(-0.002, 0.002)), 1.0, 1.0, 1.0)
        m, sd, t = adversarial.events(kind, tables[2] if s == 0 else tables[r % 2], params, n, seed=70000 + 2 * r + s, other_table=tables[(r + 1) % 3])
        _, sd, _ = na.events_prepare(m, sd, t, 0.0)
        rd.append((m, sd, t))
    pool.append((rd, 1 if two_d else 0))


def subset_arrays(idx):
    mean, stdv, start, so, tog = [], [], [], [0], []
    for i in idx:
        rd, t = pool[i]
        for e in rd:
            if e is not None:
                mean.append(e[0]); stdv.append(e[1]); start.append(e[2])
            so.append(so[-1] + (0 if e is None else len(e[0])))
        tog.append(t)
    return np.concatenate(mean), np.concatenate(stdv), np.concatenate(start), np.array(so, np.uint64), tog


# the decode batch (ragged, r73.t, the default transition weights: training rewrites transition slot 0 with those very values)
lens = np.clip(np.round(np.exp(rng.normal(np.log(1500), 0.9, 700))), 50, 15000).astype(np.int64)
ev = synth.generate(tables[2], len(lens), int(lens.max()), first_read=5_000_000)
keep = np.arange(int(lens.max()))[None, :] < lens[:, None]
v_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
v_cm, v_sd, v_ls = na.events_prepare(ev["mean"][keep], ev["stdv"][keep], ev["start"][keep], 0.0)
del ev

import torch                              # noqa: E402
dev = torch.device("cuda", 0)
d_in = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (v_off.astype(np.int64), v_cm, v_sd, v_ls)]
total = int(v_off[-1])

t0 = time.time()
stats = dict(iterations=0, jobs=0, parts_budgets_mb=[], decodes=0, async_decodes=0, differences=0)
ctx = None
ref = None
for it in range(ITER):
    if it % 8 == 0:          # a fresh context now and then, with another budget (read when the context is made)
        if ctx is not None:
            ctx.close()
        mb = int(rng.choice([48, 96, 256, 1024, 16384]))
        os.environ["NCHMM_FB_BUDGET_MB"] = str(mb)
        ctx = na.Context(0)
        del os.environ["NCHMM_FB_BUDGET_MB"]
        stats["parts_budgets_mb"].append(mb)
        ctx.put_model(0, na.scaled_model_table(tables[2]))
        ctx.put_transitions(0, *na.transitions_fast(opts.default_p_skip, opts.default_p_stay))
        first = ctx.viterbi(v_off, v_cm, v_sd, v_ls)
        assert (first[2] == 0).all()
        if ref is None:
            ref = first
        elif not (np.array_equal(ref[0], first[0]) and ref[1].tobytes() == first[1].tobytes()):
            stats["differences"] += 1; print("decode differs on a fresh context", it, flush=True)
    idx = rng.choice(POOL, int(rng.integers(40, POOL + 1)), replace=False)
    mean, stdv, start, so, tog = subset_arrays(idx)
    jr, j0, j1 = api.train_enumerate(opts, strands, so, tog)
    pending = None
    if it % 3 == 2:
        d_state = torch.empty(total, dtype=torch.int16, device=dev)
        d_logp = torch.empty(len(lens), dtype=torch.float32, device=dev)
        d_status = torch.zeros(len(lens), dtype=torch.int32, device=dev)
        ctx.viterbi_dev_enqueue(len(lens), int(lens.max()), total, *d_in, d_state, d_logp, d_status)
        pending = (d_state, d_logp, d_status)
    two = ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    if pending is not None:
        ctx.viterbi_dev_join()
        torch.cuda.synchronize()
        ok = (np.array_equal(pending[0].cpu().numpy().view(np.uint16), ref[0]) and pending[1].cpu().numpy().tobytes() == ref[1].tobytes()
              and (pending[2].cpu().numpy() == 0).all())
        stats["async_decodes"] += 1
        if not ok:
            stats["differences"] += 1; print("queued decode differs", it, flush=True)
    os.environ["NCHMM_EM_LANES"] = "1"
    one = ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    del os.environ["NCHMM_EM_LANES"]
    for k in ("pm", "st", "fit", "rounds", "preferred"):
        if one[k].tobytes() != two[k].tobytes():
            stats["differences"] += 1; print("training differs", it, k, len(jr), flush=True)
    if it % 2 == 1:
        got = ctx.viterbi(v_off, v_cm, v_sd, v_ls)
        stats["decodes"] += 1
        if not (np.array_equal(ref[0], got[0]) and ref[1].tobytes() == got[1].tobytes() and (got[2] == 0).all()):
            stats["differences"] += 1; print("decode differs", it, flush=True)
    stats["iterations"] += 1; stats["jobs"] += int(len(jr))
ctx.close()
stats["seconds"] = round(time.time() - t0, 1)
print(json.dumps(stats))
sys.exit(1 if stats["differences"] else 0)
