#!/usr/bin/env python3
"""Randomised agreement sweep of the two forward-backward implementations (rescaled linear space vs base-2 log space)
and of both against the CPU oracle's log-likelihood: random builtin model, scaling parameters, transition
probabilities per configuration.   CONFIGS=60 python tools/fb_sweep.py   (run on the GPU box)"""
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

n_cfg, n_win, n_ev = int(os.environ.get("CONFIGS", 60)), 8, 100
rng = np.random.default_rng(int(os.environ.get("SEED", 424242)))
meta, tables = models._load()
fast = na.Context(0)
os.environ["NCHMM_FB_FORCE_LOG"] = "1"
slow = na.Context(0)
del os.environ["NCHMM_FB_FORCE_LOG"]
rel = lambda a, b, floor: np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.maximum(np.abs(np.asarray(b, np.float64)), floor)
worst = dict(lpd_fast_vs_log=0.0, lpd_fast_vs_oracle=0.0, pm_fast_vs_log=0.0, st_fast_vs_log=0.0)
redone = 0
t0 = time.time()
for c in range(n_cfg):
    table = tables[int(rng.integers(len(tables)))]
    params = (float(rng.uniform(0.8, 1.2)), float(rng.uniform(-6, 6)), 0.0, float(rng.uniform(0.6, 2.0)),
              float(rng.uniform(0.7, 1.4)), float(rng.uniform(0.3, 3.0)))
    p_skip, p_stay = float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.05, 0.4))
    ev = synth.generate(table, n_win, n_ev, first_read=5000 + 100 * c)
    mean = ev["mean"].reshape(-1) * np.float32(params[0]) + np.float32(params[1])
    cm, sd, ls = na.events_prepare(mean, ev["stdv"].reshape(-1), None, 0.0)
    off = (np.arange(n_win + 1) * n_ev).astype(np.uint64)
    outs = []
    for ctx in (fast, slow):
        ctx.put_model(0, na.scaled_model_table(table, params))
        ctx.put_transitions(0, *na.transitions_fast(p_skip, p_stay))
        before = int(ctx.counters()[7])
        outs.append(ctx.fwbw(off, cm, sd, ls, pm_params=params, st_params=np.tile(np.float32([p_stay, p_skip]), (n_win, 1))))
        if ctx is fast:
            redone += int(ctx.counters()[7]) - before
    f, s = outs
    om, ot = oracle.Model(table, params), oracle.Transitions(p_skip, p_stay)
    lpd = np.array([oracle.fwbw(om, ot, cm[w * n_ev:(w + 1) * n_ev], sd[w * n_ev:(w + 1) * n_ev], ls[w * n_ev:(w + 1) * n_ev], want_matrices=False)[0]
                    for w in range(n_win)])
    worst["lpd_fast_vs_log"] = max(worst["lpd_fast_vs_log"], float(rel(f["log_pr_data"], s["log_pr_data"], 1.0).max()))
    worst["lpd_fast_vs_oracle"] = max(worst["lpd_fast_vs_oracle"], float(rel(f["log_pr_data"], lpd, 1.0).max()))
    worst["pm_fast_vs_log"] = max(worst["pm_fast_vs_log"], float(rel(f["pm_sums"], s["pm_sums"], 1e-3).max()))
    worst["st_fast_vs_log"] = max(worst["st_fast_vs_log"], float(rel(np.exp(f["st_sums"]), np.exp(s["st_sums"]), 1e-6).max()))
print(json.dumps({"configs": n_cfg, "windows": n_cfg * n_win, "windows_redone_in_log_space": redone, "worst_relative_differences": worst,
                  "seconds": round(time.time() - t0, 1)}))
ok = worst["lpd_fast_vs_log"] <= 1e-5 and worst["lpd_fast_vs_oracle"] <= 1e-4 and worst["pm_fast_vs_log"] <= 2e-3 and worst["st_fast_vs_log"] <= 2e-3
sys.exit(0 if ok else 1)
