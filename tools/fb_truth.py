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


# ------------------------------------------------------------------------------------------------
# BASELINE config 3, one read's pair, FREE-RUNNING: four EM rounds in float64, each from the float64 result of the one before --
# scaling AND transitions (Parameter_Trainer.hpp:230-427, :434-532) -- beside the same loop on the fp32 oracle.  What
# tests/test_fullsize_gpu.py holds the GPU's var / var_sd of that pair to: the reference's own fp32 arithmetic compounds its
# per-round noise over the rounds, and the distance of the oracle to this real-number answer is the yardstick.
# ------------------------------------------------------------------------------------------------
GOLDEN_C3 = os.path.join(ROOT, "tests", "golden", "config3_read511_truth64.json")


def scaled_t6_f64(unscaled, pm):
    """Pore_Model::scale (Pore_Model.hpp:190-201) in float64 -> the columns emission64 reads (0 mu, 1 sigma, 3 eta, 4 lambda)"""
    scale, shift, drift, var, scale_sd, var_sd = (float(v) for v in pm)
    t6 = np.zeros((S, 6))
    t6[:, 0] = unscaled[:, 0] * scale + shift
    t6[:, 1] = unscaled[:, 1] * var
    t6[:, 3] = unscaled[:, 2] * scale_sd
    t6[:, 4] = unscaled[:, 4] * var_sd
    return t6


def emission64_tab(t6, x, y):
    mu, sg, eta, lam = t6[:, 0], t6[:, 1], t6[:, 3], t6[:, 4]
    a = (x - mu) / sg
    return (-np.log(sg) - (LOG_2PI + a * a) / 2.0) + (np.log(lam) - LOG_2PI - 3.0 * np.log(y) - lam * (y - eta) ** 2 / (eta * eta * y)) / 2.0


def train_st64(al, be, lpd, E, kmers, p_stay, p_skip):
    """the three log-sums of train_st_params (Parameter_Trainer.hpp:451-515) over one window, float64 -> (denom, stay, skip) as lists of log terms"""
    lp_stay, lp_step4 = np.log(p_stay), np.log((1.0 - p_stay - p_skip) / 4.0)
    j1 = kmers
    n = al.shape[0]
    lp = al[:-1][:, j1] + be[:-1][:, j1] - lpd                                   # log P(S_i = j1)
    tail = E[1:] + be[1:]                                                          # emis(j2, e_{i+1}) + beta_{i+1}(j2)
    stay = np.minimum(al[:-1][:, j1] + lp_stay + tail[:, j1] - lpd, lp)            # :470-488
    terms = [stay]
    for b in range(4):
        j2 = ((j1 << 2) | b) & 4095
        terms.append(al[:-1][:, j1] + lp_step4 + tail[:, j2] - lpd)
    d01 = np.minimum(lse(np.stack(terms, 0), 0), lp)                               # :489-510
    with np.errstate(divide="ignore"):
        skip = np.log(np.exp(lp) - np.exp(d01))                                    # :511-512
    return lp.ravel(), stay.ravel(), skip.ravel()


def em_free_running64(tabs_unscaled, windows, strands, rounds, train_drift):
    """windows: [(mean, stdv, start)] raw fp32 events; strands: their strand.  -> per round {fit, pm[6], st[4]} in float64"""
    kmers = np.asarray(na.st_train_kmers(), np.int64)
    pm = np.array([1, 0, 0, 1, 1, 1], np.float64)
    st = np.array([[0.1, 0.3], [0.1, 0.3]], np.float64)                             # (p_stay, p_skip) per strand
    out = []
    for _ in range(rounds):
        t6 = [scaled_t6_f64(u, pm) for u in tabs_unscaled]
        tr = [na.transitions_fast(float(st[s, 1]), float(st[s, 0])) for s in range(2)]
        fit = 0.0
        sums, xs, ys, ts = [], [], [], []
        logs = {0: [[], [], []], 1: [[], [], []]}
        for (mean, stdv, start), s in zip(windows, strands):
            x, y, t = mean.astype(np.float64), stdv.astype(np.float64), start.astype(np.float64)
            cm = x - pm[2] * t
            n = len(x)
            rp, pred, logw = tr[s]
            E = np.stack([emission64_tab(t6[s], cm[i], y[i]) for i in range(n)])
            # forward-backward on these emissions (fwbw64's recursions)
            deg = np.diff(rp.astype(np.int64)); dmax = int(deg.max())
            if "P" not in em_free_running64.__dict__:
                P = np.zeros((S, dmax), np.int64); SP = [[] for _ in range(S)]
                for j in range(S):
                    a, b = int(rp[j]), int(rp[j + 1])
                    P[j, : b - a] = pred[a:b]
                    for k in range(a, b): SP[int(pred[k])].append((j, k))
                smax = max(len(v) for v in SP)
                Q = np.zeros((S, smax), np.int64); QK = np.full((S, smax), -1, np.int64)
                for p_, v in enumerate(SP):
                    for k, (j, kk) in enumerate(v): Q[p_, k] = j; QK[p_, k] = kk
                em_free_running64.P, em_free_running64.Q, em_free_running64.QK = P, Q, QK           # (the topology does not depend on the weights)
            P, Q, QK = em_free_running64.P, em_free_running64.Q, em_free_running64.QK
            W = np.full(P.shape, -np.inf)
            for j in range(S):
                a, b = int(rp[j]), int(rp[j + 1]); W[j, : b - a] = logw[a:b]
            V = np.where(QK >= 0, logw.astype(np.float64)[np.maximum(QK, 0)], -np.inf)
            al = np.empty((n, S)); be = np.zeros((n, S))
            al[0] = E[0] - np.log(float(S))
            for i in range(1, n): al[i] = E[i] + lse(W + al[i - 1][P], 1)
            for i in range(n - 2, -1, -1): be[i] = lse(V + (E[i + 1] + be[i + 1])[Q], 1)
            lpd = lse(al[n - 1], 0)
            fit += lpd
            p = np.exp(al + be - lpd)
            u = tabs_unscaled[s]
            u0 = 1.0 / (u[:, 1] ** 2)
            sm = np.stack([p @ u0, p @ (u0 * u[:, 0]), p @ (u0 * u[:, 0] ** 2), p @ u[:, 4], p @ (u[:, 4] / u[:, 2]), p @ (u[:, 4] / u[:, 2] ** 2)], 1)
            sums.append(sm); xs.append(x); ys.append(y); ts.append(t)
            for acc, v in zip(logs[s], train_st64(al, be, lpd, E, kmers, st[s, 0], st[s, 1])): acc.append(v)
        new_pm = finish64(np.concatenate(sums), np.concatenate(xs), np.concatenate(ys), np.concatenate(ts), pm, train_drift)
        new_st = st.copy()
        for s in (0, 1):
            if not logs[s][0]: continue
            denom, stay, skip = (lse(np.concatenate(v), 0) for v in logs[s])
            new_st[s] = np.clip([np.exp(stay - denom), np.exp(skip - denom)], 0.05, 0.4)      # :516-530
        out.append({"fit": float(fit), "pm": [float(v) for v in new_pm], "st": [float(v) for v in new_st.ravel()]})
        pm, st = new_pm, new_st
    return out


def config3_read_truth(read=511, n_ev=5000, half=100):
    """the pair (r73.t, r73.c.p1) of read `read` of BASELINE config 3 as tests/test_fullsize_gpu.py and bench.py make it"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nc_oracle as oracle
    from nanocall_amd import synth
    tabs = [na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")]
    unscaled = [na.model_load(t).astype(np.float64) for t in tabs]
    e = [synth.generate(tabs[0], 1, n_ev, first_read=read), synth.generate(tabs[1], 1, n_ev, first_read=10 ** 6 + read)]
    windows, strands = [], []
    for s in (0, 1):
        m, sd, t = e[s]["mean"][0], e[s]["stdv"][0], e[s]["start"][0]
        _, sd, _ = na.events_prepare(m, sd, None, 0.0)
        for sl in (slice(0, half), slice(n_ev - half, n_ev)):
            windows.append((m[sl], sd[sl], t[sl])); strands.append(s)
    truth = em_free_running64(unscaled, windows, strands, 4, True)
    # the same loop on the fp32 oracle (the reference's arithmetic), free-running from its own results
    off = np.concatenate([[0], np.cumsum([len(w[0]) for w in windows])]).astype(np.uint64)
    cat = lambda k: np.concatenate([w[k] for w in windows])
    pm, st = np.float32([1, 0, 0, 1, 1, 1]), np.float32([0.1, 0.3, 0.1, 0.3])
    orc = []
    for _ in range(4):
        r = oracle.train_one_round(off, np.asarray(strands, np.uint32), cat(0), cat(1), cat(2), tabs[0], tabs[1], pm, st, 0.1, 0.3, 1, True, True)
        orc.append({"fit": float(r["fit"]), "pm": [float(v) for v in r["pm"]], "st": [float(v) for v in r["st"]]})
        pm, st = r["pm"], r["st"]
    return {"made_by": "tools/fb_truth.py --config3-read", "read": read, "events_per_strand": n_ev, "window_events": half,
            "pair": ["r73.t", "r73.c.p1"], "truth64_rounds": truth, "oracle_fp32_rounds": orc}


GOLDEN_CPP = os.path.join(ROOT, "tests", "golden", "cpp_layer_read900_truth64.json")


def cpp_layer_truth():
    """the 2D read of tests/test_cpp_layer_gpu.py (template 420 events from r73.t, complement 380 from r73.c.p2, windows of 60 events),
    four free-running rounds in float64, with and without drift training"""
    from nanocall_amd import synth
    names = ["r73.t.006.ont.model", "r73.c.p2.006.ont.model"]
    tabs = [na.builtin_model(n) for n in names]
    unscaled = [na.model_load(t).astype(np.float64) for t in tabs]
    evs = [synth.generate(tabs[0], 1, 420, first_read=900), synth.generate(tabs[1], 1, 380, first_read=901)]
    windows, strands = [], []
    for s_, e in enumerate(evs):
        m, sd, t = e["mean"][0], na.events_prepare(e["mean"][0], e["stdv"][0], None, 0.0)[1], e["start"][0]
        n = len(m)
        half = min(120, n) // 2
        for sl in (slice(0, half), slice(n - half, n)):
            windows.append((m[sl], sd[sl], t[sl])); strands.append(s_)
    return {"made_by": "tools/fb_truth.py --cpp-layer-read", "models": names, "scaling_num_events": 120,
            "rounds_by_train_drift": {str(d): em_free_running64(unscaled, windows, strands, 4, bool(d)) for d in (1, 0)}}


def main():
    if "--cpp-layer-read" in sys.argv:
        with open(GOLDEN_CPP, "w") as f:
            json.dump(cpp_layer_truth(), f, indent=1)
        print("wrote", GOLDEN_CPP)
        return
    if "--config3-read" in sys.argv:
        doc = config3_read_truth()
        names = ["scale", "shift", "drift", "var", "scale_sd", "var_sd"]
        t, o = doc["truth64_rounds"][-1], doc["oracle_fp32_rounds"][-1]
        doc["oracle_rel_distance_after_4_rounds"] = {n: abs(o["pm"][k] - t["pm"][k]) / {1: 60.0, 2: 60.0 / 100.0}.get(k, abs(t["pm"][k])) for k, n in enumerate(names)}
        doc["oracle_rel_distance_after_4_rounds"]["st"] = [abs(a - b) / abs(b) for a, b in zip(o["st"], t["st"])]
        with open(GOLDEN_C3, "w") as f:
            json.dump(doc, f, indent=1)
        print(json.dumps(doc["oracle_rel_distance_after_4_rounds"]))
        print("wrote", GOLDEN_C3)
        return
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
