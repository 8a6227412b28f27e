#!/usr/bin/env python3
"""Which windows of tools/fb_sweep.py's adversarial sweep disagree, and how (debug aid): prints the windows whose fast-path log Pr(data)
is more than 5e-5 (relative) from the log-space path's, and those whose pm / st sums differ by more than 2e-3.  CONFIGS=300 python tools/ubench/fb_sweep_debug.py"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import multiprocessing as mp
import fb_sweep

def main():
    n_cfg = int(os.environ.get("CONFIGS", 300))
    pool = mp.get_context("spawn").Pool(int(os.environ.get("WORKERS", 64)))
    todo = pool.imap(fb_sweep.make_config, range(n_cfg), chunksize=1)
    import nanocall_amd as na
    from nanocall_amd import models
    meta, tables = models._load()
    fast = na.Context(0)
    os.environ["NCHMM_FB_FORCE_LOG"] = "1"
    slow = na.Context(0)
    del os.environ["NCHMM_FB_FORCE_LOG"]
    shown = 0
    for cfg in todo:
        table, params = tables[cfg["model"]], cfg["params"]
        outs = []
        for ctx in (fast, slow):
            ctx.put_model(0, na.scaled_model_table(table, params))
            ctx.put_transitions(0, *na.transitions_fast(*cfg["trans"]))
            outs.append(ctx.fwbw(cfg["off"], cfg["cm"], cfg["sd"], cfg["ls"], pm_params=params, st_params=np.tile(np.float32([cfg["trans"][1], cfg["trans"][0]]), (fb_sweep.N_WIN, 1))))
        f, s = outs
        off = cfg["off"].astype(np.int64)
        for w in range(fb_sweep.N_WIN):
            a, b = int(off[w]), int(off[w + 1])
            lf, ls_, lo = float(f["log_pr_data"][w]), float(s["log_pr_data"][w]), float(cfg["lpd"][w])
            pmf, pms = f["pm_sums"].reshape(-1, 6)[a:b].astype(np.float64), s["pm_sums"].reshape(-1, 6)[a:b].astype(np.float64)
            with np.errstate(all="ignore"):
                pm_rel = np.abs(pmf - pms) / np.maximum(np.abs(pms), 1e-3)
                stf, sts = np.exp(f["st_sums"].reshape(-1, 3)[w].astype(np.float64)), np.exp(s["st_sums"].reshape(-1, 3)[w].astype(np.float64))
                st_rel = np.abs(stf - sts) / np.maximum(np.abs(sts), 1e-6)
            bad_l = abs(lf - ls_) > 5e-5 * max(abs(ls_), 1.0)
            bad_pm = not np.isfinite(pm_rel).all() or pm_rel.max() > 2e-3
            bad_st = not np.isfinite(st_rel).all() or st_rel.max() > 2e-3
            if (bad_l or bad_pm or bad_st) and shown < 40:
                shown += 1
                ev = np.unravel_index(np.nanargmax(np.where(np.isfinite(pm_rel), pm_rel, 1e30)), pm_rel.shape)
                print(json.dumps({"config": cfg["c"], "window": w, "kind": cfg["kinds"][w], "len": b - a, "lpd_fast": lf, "lpd_log": ls_, "lpd_oracle": lo,
                                  "bad": [bool(bad_l), bool(bad_pm), bool(bad_st)], "pm_worst_at_event": int(ev[0]), "pm_fast_row": pmf[ev[0]].tolist(), "pm_log_row": pms[ev[0]].tolist(),
                                  "st_fast": f["st_sums"].reshape(-1, 3)[w].tolist(), "st_log": s["st_sums"].reshape(-1, 3)[w].tolist(),
                                  "event": [float(cfg["cm"][a + ev[0]]), float(cfg["sd"][a + ev[0]])], "params": params, "trans": cfg["trans"], "model": cfg["model"]}), flush=True)
    pool.close(); pool.join()

if __name__ == "__main__":
    main()
