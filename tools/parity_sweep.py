#!/usr/bin/env python3
"""Randomised bit-parity sweep of the Viterbi path against the CPU oracle: random builtin model, random (valid) scaling
parameters and transition probabilities per configuration, ragged reads.  Every read must match the oracle's k-mer path
and path log-probability bit for bit.   CONFIGS=40 READS=6 SWEEP=auto|wide|ll python tools/parity_sweep.py   (run on the GPU box;
READS > the number of CUs makes the plan pick the wide form for some configurations)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nanocall_amd as na                 # noqa: E402
from nanocall_amd import models, synth    # noqa: E402
import nc_oracle as oracle                # noqa: E402

n_cfg, n_reads = int(os.environ.get("CONFIGS", 40)), int(os.environ.get("READS", 6))
n_check = int(os.environ.get("CHECK_READS", n_reads))     # reads per configuration compared with the oracle (the first ones; all are decoded)
rng = np.random.default_rng(int(os.environ.get("SEED", 20260101)))
meta, tables = models._load()
ctx = na.Context(0)
ctx.set_sweep(os.environ.get("SWEEP", "auto"))     # auto | wide | ll: which form of the sweep is checked (nchmm_set_sweep)
t0 = time.time()
events = mismatches = 0
for c in range(n_cfg):
    m = int(rng.integers(len(tables)))
    table = tables[m]
    params = (float(rng.uniform(0.8, 1.2)), float(rng.uniform(-6, 6)), float(rng.uniform(-0.01, 0.01)),
              float(rng.uniform(0.7, 1.5)), float(rng.uniform(0.8, 1.25)), float(rng.uniform(0.5, 2.0)))
    p_skip, p_stay = float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.05, 0.4))
    lens = [int(x) for x in rng.integers(1, 2500, n_reads)]
    ev = synth.generate(table, n_reads, max(lens), first_read=1000 * c)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    cat = lambda k: np.concatenate([ev[k][r, :n] for r, n in enumerate(lens)])
    mean = cat("mean") * np.float32(params[0]) + np.float32(params[1])          # events that fit the scaled model
    cm, sd, ls = na.events_prepare(mean, cat("stdv"), cat("start"), params[2])
    ctx.put_model(0, na.scaled_model_table(table, params))
    ctx.put_transitions(0, *na.transitions_fast(p_skip, p_stay))
    states, logp, status = ctx.viterbi(off, cm, sd, ls)
    om, ot = oracle.Model(table, params), oracle.Transitions(p_skip, p_stay)
    for r, n in enumerate(lens[:n_check]):
        a, b = int(off[r]), int(off[r + 1])
        s, mv, lp = oracle.viterbi(om, ot, cm[a:b], sd[a:b], ls[a:b])
        ok = status[r] == 0 and np.array_equal(s, states[a:b]) and np.float32(lp).tobytes() == np.float32(logp[r]).tobytes()
        if not ok:
            mismatches += 1
            print(f"MISMATCH config {c} model {meta['names'][m]} params {params} trans {(p_skip, p_stay)} read {r} len {n}", flush=True)
        events += n
print(json.dumps({"configs": n_cfg, "reads_decoded": n_cfg * n_reads, "reads_checked": n_cfg * min(n_check, n_reads), "events": events, "mismatches": mismatches,
                  "seconds": round(time.time() - t0, 1), "sweep": os.environ.get("SWEEP", "auto"),
                  "launches_wide_ll_reads_wide_ll": list(ctx.sweep_stats())}))
sys.exit(1 if mismatches else 0)
