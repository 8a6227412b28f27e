#!/usr/bin/env python3
"""tools/fb_truth.py -- where does the fp32 noise of one EM round come from?

Runs round 0 of the 2D EM fixture (tests/golden/em_2d_drift1.npz: 4 windows x 100 events, identity
scaling, p_stay 0.1 / p_skip 0.3) three ways and prints the six trained scaling parameters:

  truth   forward-backward and the per-event inner sums in float64 (numpy; same fp32 inputs: model
          tables, transition weights, events), finished by nchmm_train_pm_finish
  oracle  the CPU restatement of the reference (fp32 throughout, as the reference is) -- the fixture
  gpu     the HIP kernels

The relative distance of `oracle` and `gpu` to `truth` is what the parameter tolerances in
tests/test_fwbw_gpu.py are set from: the reference's own fp32 arithmetic is no closer to the real-number
answer than the GPU's.  Needs a GPU (run through gpurun); the float64 part is plain numpy.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanocall_amd as na  # noqa: E402

S = 4096
LOG_2PI = np.log(2.0 * np.pi)


def lse(a, axis):
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    return (m + np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True))).squeeze(axis)


def emission64(t6, x, y):
    mu, sg, eta, lam = (t6[:, k].astype(np.float64) for k in (0, 1, 3, 4))
    a = (x - mu) / sg
    n = -np.log(sg) - (LOG_2PI + a * a) / 2.0
    ig = (np.log(lam) - LOG_2PI - 3.0 * np.log(y) - lam * (y - eta) ** 2 / (eta * eta * y)) / 2.0
    return n + ig


def fwbw64(t6, rp, pred, logw, cm, sd):
    n = cm.shape[0]
    deg = np.diff(rp.astype(np.int64))
    dmax = int(deg.max())
    P = np.zeros((S, dmax), np.int64)
    W = np.full((S, dmax), -np.inf)
    SP = [[] for _ in range(S)]
    for j in range(S):
        a, b = int(rp[j]), int(rp[j + 1])
        P[j, : b - a] = pred[a:b]
        W[j, : b - a] = logw[a:b]
        for k in range(a, b):
            SP[int(pred[k])].append((j, float(logw[k])))
    smax = max(len(v) for v in SP)
    Q = np.zeros((S, smax), np.int64)
    V = np.full((S, smax), -np.inf)
    for p, v in enumerate(SP):
        for k, (j, w) in enumerate(v):
            Q[p, k] = j
            V[p, k] = w
    E = np.stack([emission64(t6, float(cm[i]), float(sd[i])) for i in range(n)])
    al = np.empty((n, S))
    be = np.zeros((n, S))
    al[0] = E[0] - np.log(float(S))
    for i in range(1, n):
        al[i] = E[i] + lse(W + al[i - 1][P], 1)
    for i in range(n - 2, -1, -1):
        g = E[i + 1] + be[i + 1]
        be[i] = lse(V + g[Q], 1)
    lpd = lse(al[n - 1], 0)
    return lpd, al, be


def setup():
    z = np.load(os.path.join(ROOT, "tests", "golden", "em_2d_drift1.npz"))
    tabs = [na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")]
    pm = np.array([1, 0, 0, 1, 1, 1], np.float32)
    d = dict(z=z, pm=pm, strand=z["strand"].astype(np.int64), off=z["off"], mean=z["mean"], start=z["start"])
    d["rp"], d["pred"], d["logw"] = na.transitions_fast(0.3, 0.1)
    d["cm"], d["sd"], d["ls"] = na.events_prepare(z["mean"], z["stdv"], z["start"], 0.0)
    d["t6"] = [na.scaled_model_table(t, pm) for t in tabs]
    return d


def truth_round0(d):
    """float64 forward-backward + inner sums of fixture round 0 -> (fit, pm_sums[n,6] f64, params[6] f32)."""
    cm, sd, off, strand = d["cm"], d["sd"], d["off"], d["strand"]
    sums = np.zeros((cm.shape[0], 6))
    fit64 = 0.0
    for w in range(len(strand)):
        a, b = int(off[w]), int(off[w + 1])
        m = d["t6"][strand[w]].astype(np.float64)
        lpd, al, be = fwbw64(d["t6"][strand[w]], d["rp"], d["pred"], d["logw"], cm[a:b], sd[a:b])
        fit64 += lpd
        p = np.exp(al + be - lpd)
        u0 = 1.0 / (m[:, 1] ** 2)
        sums[a:b, 0] = p @ u0
        sums[a:b, 1] = p @ (u0 * m[:, 0])
        sums[a:b, 2] = p @ (u0 * m[:, 0] ** 2)
        sums[a:b, 3] = p @ m[:, 4]
        sums[a:b, 4] = p @ (m[:, 4] / m[:, 3])
        sums[a:b, 5] = p @ (m[:, 4] / m[:, 3] ** 2)
    truth, _ = na.train_pm_finish(sums.astype(np.float32), d["mean"], sd, d["start"], d["pm"], train_drift=True)
    return fit64, sums, truth


def finish64(sums, x, y, t, crt_pm, train_drift):
    """Parameter_Trainer::train_pm_params' outer sums and solve (Parameter_Trainer.hpp:297-427) entirely in float64:
    products, accumulation, the 3x3 solve and the closed forms.  -> params[6] float64 (scale, shift, drift, var, scale_sd, var_sd)"""
    s0, s1, s2, l0, l1, l2 = (sums[:, k] for k in range(6))
    x, y, t = x.astype(np.float64), y.astype(np.float64), t.astype(np.float64)
    A = np.zeros((3, 3))
    B = np.zeros(3)
    A[0, 0], A[0, 1], A[1, 1] = s0.sum(), s1.sum(), s2.sum()
    B[0], B[1] = (s0 * x).sum(), (s1 * x).sum()
    if train_drift:
        A[0, 2], A[1, 2], A[2, 2], B[2] = (s0 * t).sum(), (s1 * t).sum(), (s0 * t * t).sum(), (s0 * x * t).sum()
    else:
        A[2, 2] = 1.0
    A[1, 0], A[2, 0], A[2, 1] = A[0, 1], A[0, 2], A[1, 2]
    D, Vn, Vd, Up = (s0 * x * x).sum(), (l2 * y).sum(), l1.sum(), (l0 / y).sum()
    a, b, c = np.linalg.solve(A, B)
    n = float(len(x))
    d_numer = (D + a * a * A[0, 0] + b * b * A[1, 1] + c * c * A[2, 2] + 2 * a * b * A[0, 1] + 2 * a * c * A[0, 2] + 2 * b * c * A[1, 2]
               - 2 * (a * B[0] + b * B[1] + c * B[2]))
    v = Vn / Vd
    return np.array([b, a, c if train_drift else float(crt_pm[2]), np.sqrt(d_numer / n), v, n / (Up - Vd / v)])


GOLDEN = os.path.join(ROOT, "tests", "golden", "em_2d_drift1_truth64.json")
GOLDEN_ALL = os.path.join(ROOT, "tests", "golden", "em_2d_truth64_all_rounds.json")


def truth_all_rounds():
    """Every round of both EM fixtures (drift trained / not trained), teacher-forced from the fixture's previous-round
    parameters exactly as tests/test_fwbw_gpu.py::test_em_rounds_against_golden runs them: forward-backward in float64 on
    the SCALED model with the round's transitions and drift-corrected events, inner sums in float64 over the UNSCALED
    model (Parameter_Trainer.hpp:273-296), finished by nchmm_train_pm_finish.  -> {drift: [ {fit, params[6]} per round ]}"""
    out = {}
    tabs = [na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")]
    unscaled = [na.model_load(t).astype(np.float64) for t in tabs]      # S x 10: level_mean, level_stdv, sd_mean, sd_stdv, sd_lambda, ...
    for drift in (1, 0):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"em_2d_drift{drift}.npz"))
        mean, stdv, start, strand, off = z["mean"], z["stdv"], z["start"], z["strand"].astype(np.int64), z["off"]
        pm = np.array([1, 0, 0, 1, 1, 1], np.float32)
        stp = np.array([[0.1, 0.3], [0.1, 0.3]], np.float32)
        rounds = []
        for rnd, exp in enumerate(z["rounds"]):
            t6 = [na.scaled_model_table(t, pm) for t in tabs]
            tr = [na.transitions_fast(float(stp[st, 1]), float(stp[st, 0])) for st in range(2)]
            cm, sd, ls = na.events_prepare(mean, stdv, start, float(pm[2]))
            sums = np.zeros((cm.shape[0], 6))
            fit64 = 0.0
            for w in range(len(strand)):
                a, b = int(off[w]), int(off[w + 1])
                st = strand[w]
                lpd, al, be = fwbw64(t6[st], *tr[st], cm[a:b], sd[a:b])
                fit64 += lpd
                p = np.exp(al + be - lpd)
                u = unscaled[st]
                u0 = 1.0 / (u[:, 1] ** 2)
                sums[a:b, 0] = p @ u0
                sums[a:b, 1] = p @ (u0 * u[:, 0])
                sums[a:b, 2] = p @ (u0 * u[:, 0] ** 2)
                sums[a:b, 3] = p @ u[:, 4]
                sums[a:b, 4] = p @ (u[:, 4] / u[:, 2])
                sums[a:b, 5] = p @ (u[:, 4] / u[:, 2] ** 2)
            # two finishes: the reference's (float products of float32 sums, nchmm_train_pm_finish) fed the float64 sums
            # rounded once, and the same closed forms evaluated entirely in float64 -- the real-number answer
            mixed, done = na.train_pm_finish(sums.astype(np.float32), mean, sd, start, pm, train_drift=bool(drift))
            truth = finish64(sums, mean, sd, start, pm, bool(drift))
            rounds.append({"fit": fit64, "params": [float(v) for v in truth], "params_f64_sums_f32_finish": [float(v) for v in mixed],
                           "done": bool(done),
                           "oracle_params": [float(v) for v in exp[1:7]], "oracle_fit": float(exp[0])})
            pm, stp = exp[1:7].astype(np.float32), exp[7:11].astype(np.float32).reshape(2, 2)
        out[str(drift)] = rounds
    return out


def main():
    if "--golden-all" in sys.argv:
        with open(GOLDEN_ALL, "w") as f:
            json.dump({"made_by": "tools/fb_truth.py --golden-all", "rounds": truth_all_rounds()}, f, indent=1)
        print("wrote", GOLDEN_ALL)
        return
    d = setup()
    z, pm, strand, off, mean, start = d["z"], d["pm"], d["strand"], d["off"], d["mean"], d["start"]
    cm, sd, ls, t6 = d["cm"], d["sd"], d["ls"], d["t6"]
    fit64, sums, truth = truth_round0(d)
    if "--golden" in sys.argv:      # CPU only: the fixture tests/test_fwbw_gpu.py holds the GPU to
        with open(GOLDEN, "w") as f:
            json.dump({"made_by": "tools/fb_truth.py --golden", "fit": fit64, "params": [float(v) for v in truth],
                       "pm_sums_event_totals": [float(v) for v in sums.sum(axis=0)]}, f, indent=1)
        print("wrote", GOLDEN)
        return
    rp, pred, logw = d["rp"], d["pred"], d["logw"]

    # ---- gpu ----
    ctx = na.Context(0)
    for s in range(2):
        ctx.put_model(12 + s, t6[s])
        ctx.put_transitions(12 + s, rp, pred, logw)
    out = ctx.fwbw(off, cm, sd, ls, scaled_slot=12 + strand, pm_params=pm, trans_slot=12 + strand,
                   st_params=np.tile(np.float32([0.1, 0.3]), (len(strand), 1)))
    gpu, _ = na.train_pm_finish(out["pm_sums"], mean, sd, start, pm, train_drift=True)
    sum_err = np.abs(out["pm_sums"] - sums).max(axis=0) / np.abs(sums).max(axis=0)

    oracle = z["rounds"][0][1:7]
    names = ["scale", "shift", "drift", "var", "scale_sd", "var_sd"]
    res = {"fit": {"truth": fit64, "oracle": float(z["rounds"][0][0]), "gpu": float(np.sum(out["log_pr_data"], dtype=np.float64))},
           "pm_sums_max_rel_err_gpu_vs_truth": [float(v) for v in sum_err], "params": {}}
    for k, nme in enumerate(names):
        # shift is an offset on a ~60 pA level scale, drift on ~60 pA per read time span
        den = {"shift": 60.0, "drift": 60.0 / float(start.max())}.get(nme, abs(float(truth[k])))
        res["params"][nme] = {"truth": float(truth[k]), "oracle": float(oracle[k]), "gpu": float(gpu[k]),
                              "oracle_rel_err": abs(float(oracle[k]) - float(truth[k])) / den,
                              "gpu_rel_err": abs(float(gpu[k]) - float(truth[k])) / den}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
