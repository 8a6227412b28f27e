import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nanocall_amd as na
from nanocall_amd import synth
rng = np.random.default_rng(5)
t = na.builtin_model("r73.t")
n_win, n_ev = 2048, 100
ev = synth.generate(t, n_win, n_ev)
mean = ev["mean"].reshape(-1).copy(); stdv = ev["stdv"].reshape(-1).copy()
ctx = na.Context(0)
ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
off = (np.arange(n_win + 1) * n_ev).astype(np.uint64)
for name, p_sd, p_lv, lv_sigma in (("clean", 0, 0, 0), ("1% stdv x3-15, 0.5% level N(0,10pA)", 0.01, 0.005, 10.0), ("3% stdv x3-15, 2% level N(0,15pA)", 0.03, 0.02, 15.0)):
    m, s = mean.copy(), stdv.copy()
    k = rng.random(m.shape[0]) < p_sd
    s[k] *= rng.uniform(3, 15, k.sum()).astype(np.float32)
    k2 = rng.random(m.shape[0]) < p_lv
    m[k2] += rng.normal(0, lv_sigma, k2.sum()).astype(np.float32) if lv_sigma else 0
    cm, sd, ls = na.events_prepare(m, s, None, 0.0)
    before = int(ctx.counters()[7])
    out = ctx.fwbw(off, cm, sd, ls, st_params=np.tile(np.float32([0.1, 0.3]), (n_win, 1)))
    flagged = int(ctx.counters()[7]) - before
    print(f"{name}: {flagged} of {n_win} windows redone in log space ({100.0 * flagged / n_win:.1f} %), finite lpd: {np.isfinite(out['log_pr_data']).all()}")
