#!/usr/bin/env python3
"""Where does the log-space forward-backward lose digits on a window no state explains?  One training window with an abasic stretch
(job 51 of tools/em_sweep.py SEED=4242: log Pr(data) ~ -2e4): alpha / beta matrices and the per-event pm sums of this library's
log-space kernels, of the oracle (the reference's fp32 arithmetic) and in float64.   python tools/ubench/fb_log_noise.py (GPU box)"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
os.environ["SEED"] = "4242"
import em_sweep, fb_truth
import nanocall_amd as na
import nc_oracle as oracle

def main():
    strands, kinds = em_sweep.make_read(25)
    tabs = [na.builtin_model(n) for n in em_sweep.NAMES]
    ident = np.float32([1, 0, 0, 1, 1, 1])
    os.environ["NCHMM_FB_FORCE_LOG"] = "1"
    ctx = na.Context(0)
    del os.environ["NCHMM_FB_FORCE_LOG"]
    rel = lambda a, b, floor: np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.maximum(np.abs(np.asarray(b, np.float64)), floor)
    for s_, m_ in ((0, 2), (1, 1)):
        mean, stdv, start = strands[s_]
        for sl in (slice(0, 100), slice(300, 400)):
            cm, sd, ls = na.events_prepare(mean[sl], stdv[sl], None, 0.0)
            t6 = na.scaled_model_table(tabs[m_], ident)
            tr = na.transitions_fast(0.3, 0.1)
            lpd64, al64, be64 = fb_truth.fwbw64(t6, *tr, cm, sd)
            ctx.put_model(0, t6); ctx.put_transitions(0, *tr)
            off = np.array([0, len(cm)], np.uint64)
            g = ctx.fwbw(off, cm, sd, ls, pm_params=ident, st_params=np.float32([[0.1, 0.3]]), want_matrices=True)
            gn = ctx.fwbw(off, cm, sd, ls, pm_params=ident, st_params=np.float32([[0.1, 0.3]]))      # no matrices: columns relative to integer offsets (NORM)
            om, ot = oracle.Model(tabs[m_], ident), oracle.Transitions(0.3, 0.1)
            olpd, oal, obe = oracle.fwbw(om, ot, cm, sd, ls, want_matrices=True)
            u = na.model_load(tabs[m_]).astype(np.float64); u0 = 1.0 / (u[:, 1] ** 2)
            def sums(al, be, lpd):
                p = np.exp(np.asarray(al, np.float64) + np.asarray(be, np.float64) - float(lpd))
                return np.stack([p @ u0, p @ (u0 * u[:, 0]), p @ (u0 * u[:, 0] ** 2), p @ u[:, 4], p @ (u[:, 4] / u[:, 2]), p @ (u[:, 4] / u[:, 2] ** 2)], 1), p.sum(1)
            t, tp = sums(al64, be64, lpd64)
            so, sop = sums(oal, obe, olpd)                       # the oracle's fp32 matrices, posterior formed in float64
            sg, sgp = sums(g["alpha"], g["beta"], g["log_pr_data"][0])   # this library's matrices, the same
            # only states that carry mass matter: compare alpha + beta - lpd where the float64 posterior is above 1e-6
            mask = (al64 + be64 - lpd64) > np.log(1e-6)
            d_o = np.abs((oal.astype(np.float64) + obe - float(olpd)) - (al64 + be64 - lpd64))[mask]
            d_g = np.abs((g["alpha"].astype(np.float64) + g["beta"] - float(g["log_pr_data"][0])) - (al64 + be64 - lpd64))[mask]
            print(json.dumps({"strand": s_, "window": [sl.start, sl.stop], "kind": kinds[s_], "log_pr_data": {"float64": float(lpd64), "oracle": float(olpd), "gpu": float(g["log_pr_data"][0])},
                              "log_posterior_abs_error_where_p>1e-6": {"oracle_mean": float(d_o.mean()), "oracle_max": float(d_o.max()), "gpu_mean": float(d_g.mean()), "gpu_max": float(d_g.max())},
                              "posterior_row_sum_worst": {"oracle": float(np.abs(sop - 1).max()), "gpu_matrices": float(np.abs(sgp - 1).max())},
                              "pm_sums_max_rel_to_float64": {"oracle_matrices": float(rel(so, t, 1e-3).max()), "gpu_matrices": float(rel(sg, t, 1e-3).max()),
                                                             "gpu_kernel_sums": float(rel(g["pm_sums"].reshape(-1, 6), t, 1e-3).max()),
                                                             "gpu_kernel_sums_relative_columns": float(rel(gn["pm_sums"].reshape(-1, 6), t, 1e-3).max())},
                              "log_pr_data_relative_columns": float(gn["log_pr_data"][0]),
                              "alpha_abs_error_max": {"oracle": float(np.abs(oal - al64)[mask].max()), "gpu": float(np.abs(g["alpha"] - al64)[mask].max())},
                              "beta_abs_error_max": {"oracle": float(np.abs(obe - be64)[mask].max()), "gpu": float(np.abs(g["beta"] - be64)[mask].max())}}), flush=True)

if __name__ == "__main__":
    main()
