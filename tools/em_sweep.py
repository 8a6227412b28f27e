#!/usr/bin/env python3
"""Randomised sweep of the EM driver (nchmm_train_reads = train_reads, nanocall.cpp:292-574) against the reference's round loop on the
CPU oracle (Parameter_Trainer::train_one_round, Parameter_Trainer.hpp:541-579), on 2D reads whose training windows are NOT draws
from the models they are trained with (the event kinds of tests/adversarial.py): what happens to the control flow -- round counts,
roll-backs, the singular-matrix stop -- when windows contain constant runs, spikes, abasic stretches, another model's levels.

Per job: the number of rounds, the final fit (1e-4 relative wherever the round counts agree) and the distance of the trained
parameters.  Where the two disagree on the number of rounds the job is listed with both fits per round, so that one can see whether
a decision was taken at a margin inside the fp32 noise of the oracle's log-space arithmetic (tools/fb_sweep.py measures that noise
against float64).   JOBS=96 WORKERS=16 OUT=gpurun_out/em_sweep.json python tools/em_sweep.py   (GPU box)"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED = int(os.environ.get("SEED", 31337))
N_EV = 400            # events per strand; scaling_num_events 200 -> two windows of 100 at either end
NAMES = ["r73.c.p1", "r73.c.p2", "r73.t"]


def make_read(r):
    """raw events of read r: (mean, stdv, start) per strand, kinds per strand"""
    import nanocall_amd as na
    import adversarial
    rng = np.random.default_rng([SEED, r])
    tabs = [na.builtin_model(n) for n in NAMES]
    kinds = [adversarial.KINDS[int(rng.integers(len(adversarial.KINDS)))] if rng.random() < 0.7 else "matched" for _ in range(2)]
    params = (float(rng.uniform(0.9, 1.1)), float(rng.uniform(-4, 4)), float(rng.uniform(-0.003, 0.003)), 1.0, 1.0, 1.0)
    out = []
    for s, kind in enumerate(kinds):
        t = tabs[2] if s == 0 else tabs[int(rng.integers(2))]
        mean, stdv, start = adversarial.events(kind, t, params, N_EV, seed=50000 + 10 * r + s, other_table=tabs[(s + 1) % 3])
        _, stdv, _ = na.events_prepare(mean, stdv, None, 0.0)
        out.append((mean, stdv, start))
    return out, kinds


def oracle_job(args):
    """(worker process) the reference loop for one job"""
    import nanocall_amd as na
    from nanocall_amd import api
    import nc_oracle as oracle
    import bench
    r, m0, m1 = args
    tabs = [na.builtin_model(n) for n in NAMES]
    opts = api.train_opts(scaling_max_rounds=2)
    strands, _ = make_read(r)
    half = opts.scaling_num_events // 2
    windows, wst = [], []
    for s in (0, 1):
        mean, stdv, start = strands[s]
        for sl in (slice(0, half), slice(N_EV - half, N_EV)):
            windows.append((mean[sl], stdv[sl], start[sl])); wst.append(s)
    pm, st, fit, rnd, _ = bench.oracle_train_job(oracle, opts, tabs[m0], tabs[m1], windows, wst)
    return [float(x) for x in pm], [float(x) for x in st], float(fit), int(rnd)


def main():
    n_reads = int(os.environ.get("READS", 48))
    workers = int(os.environ.get("WORKERS", 16))
    t0 = time.time()
    import nanocall_amd as na
    from nanocall_amd import api
    tabs = [na.builtin_model(n) for n in NAMES]
    states = np.stack([na.model_load(t) for t in tabs])
    reads = [make_read(r) for r in range(n_reads)]
    mean = np.concatenate([np.concatenate([s[0] for s in rd[0]]) for rd in reads])
    stdv = np.concatenate([np.concatenate([s[1] for s in rd[0]]) for rd in reads])
    start = np.concatenate([np.concatenate([s[2] for s in rd[0]]) for rd in reads])
    so = (np.arange(2 * n_reads + 1) * N_EV).astype(np.uint64)
    opts = api.train_opts(scaling_max_rounds=2)
    jr, j0, j1 = api.train_enumerate(opts, [1, 1, 0], so, np.ones(n_reads, np.uint8))
    pool = mp.get_context("spawn").Pool(workers)
    fut = pool.map_async(oracle_job, [(int(r), int(a), int(b)) for r, a, b in zip(jr, j0, j1)], chunksize=1)
    ctx = na.Context(0)
    os.environ["NCHMM_FB_FORCE_LOG"] = "1"
    ctx_log = na.Context(0)
    del os.environ["NCHMM_FB_FORCE_LOG"]
    import nc_oracle as oracle
    out = ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    want = fut.get()
    pool.close(); pool.join()
    same = [int(out["rounds"][k]) == w[3] for k, w in enumerate(want)]
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)
    fit_rel = [rel(out["fit"][k], w[2]) for k, w in enumerate(want) if same[k] and np.isfinite(w[2])]
    names = ("scale", "shift", "drift", "var", "scale_sd", "var_sd")
    den = lambda q, v: {1: 60.0, 2: 60.0 / (N_EV * 0.02)}.get(q, abs(float(v)))
    pm_rel = {nme: sorted(abs(float(out["pm"][k][q]) - w[0][q]) / den(q, w[0][q]) for k, w in enumerate(want) if same[k]) for q, nme in enumerate(names)}
    pct = lambda v, p: float(v[min(len(v) - 1, int(p * len(v)))]) if v else None
    differ = [dict(job=k, read=int(jr[k]), kinds=reads[int(jr[k])][1], gpu_rounds=int(out["rounds"][k]), oracle_rounds=w[3], gpu_fit=float(out["fit"][k]), oracle_fit=w[2])
              for k, w in enumerate(want) if not same[k]]
    # the jobs whose final fit is furthest from the oracle's: evaluated in float64 as well (four free-running rounds, scaling and
    # transitions: tools/fb_truth.py em_free_running64) -- which of the two fp32 answers is the noisy one?
    import fb_truth
    worst = sorted(((rel(out["fit"][k], w[2]), k) for k, w in enumerate(want) if same[k] and np.isfinite(w[2])), reverse=True)[: int(os.environ.get("TRUTH_JOBS", 3))]
    unscaled = [na.model_load(t).astype(np.float64) for t in tabs]
    half = opts.scaling_num_events // 2
    truth_rows = []
    for d, k in worst:
        r, m0, m1 = int(jr[k]), int(j0[k]), int(j1[k])
        windows, wst = [], []
        for s_ in (0, 1):
            mean_, stdv_, start_ = reads[r][0][s_]
            for sl in (slice(0, half), slice(N_EV - half, N_EV)):
                windows.append((mean_[sl], stdv_[sl], start_[sl])); wst.append(s_)
        rounds = int(out["rounds"][k])
        t = fb_truth.em_free_running64([unscaled[m0], unscaled[m1]], windows, wst, max(rounds, 1), bool(opts.train_drift))
        tf, tpm = t[-1]["fit"], np.array(t[-1]["pm"])
        # round 1 on its own (no amplification by later rounds): the scaling parameters after ONE round from the identity, by
        # this library's two forward-backward paths (rescaled with its log-space redo; everything in log space), by the oracle, in float64
        off_w = np.concatenate([[0], np.cumsum([len(w_[0]) for w_ in windows])]).astype(np.uint64)
        cat = lambda i: np.concatenate([w_[i] for w_ in windows])
        cm_, sd_, ls_ = na.events_prepare(cat(0), cat(1), cat(2), 0.0)
        ident = np.float32([1, 0, 0, 1, 1, 1])
        r1 = {}
        for tag, c_ in (("rescaled_with_redo", ctx), ("log_space", ctx_log)):
            for s_, m_ in enumerate((m0, m1)):
                c_.put_model(40 + s_, na.scaled_model_table(tabs[m_], ident))
            c_.put_transitions(40, *na.transitions_fast(opts.default_p_skip, opts.default_p_stay))
            fb = c_.fwbw(off_w, cm_, sd_, ls_, scaled_slot=40 + np.asarray(wst, np.int32), pm_params=ident, trans_slot=np.full(len(wst), 40, np.int32),
                         st_params=np.tile(np.float32([opts.default_p_stay, opts.default_p_skip]), (len(wst), 1)))
            # (the pm sums of each strand are over ITS unscaled model; train_pm_finish takes them as they come)
            r1[tag] = [float(x) for x in na.train_pm_finish(fb["pm_sums"], cat(0), cat(1), cat(2), ident, train_drift=bool(opts.train_drift))[0]]
        o1 = oracle.train_one_round(off_w, np.asarray(wst, np.uint32), cat(0), cat(1), cat(2), tabs[m0], tabs[m1], ident, np.float32([opts.default_p_stay, opts.default_p_skip] * 2),
                                    opts.default_p_stay, opts.default_p_skip, opts.train_drift, True, True)
        t1 = np.array(t[0]["pm"])
        dist = lambda v: [abs(float(a) - float(b)) / {1: 60.0, 2: 60.0 / (N_EV * 0.02)}.get(q, abs(float(b))) for q, (a, b) in enumerate(zip(v, t1))]
        round1 = {"float64": [float(x) for x in t1], "distance_rescaled_with_redo": dist(r1["rescaled_with_redo"]), "distance_log_space": dist(r1["log_space"]),
                  "distance_oracle": dist(o1["pm"])}
        truth_rows.append({"job": k, "read": r, "kinds": reads[r][1], "round_1_pm_distance_to_float64": round1, "models": [NAMES[m0], NAMES[m1]], "rounds": rounds, "gpu_fit": float(out["fit"][k]), "oracle_fit": want[k][2], "float64_fit": tf,
                           "float64_fit_by_round": [x["fit"] for x in t],
                           "gpu_fit_rel_to_float64": rel(out["fit"][k], tf), "oracle_fit_rel_to_float64": rel(want[k][2], tf),
                           "gpu_pm": [float(x) for x in out["pm"][k]], "oracle_pm": want[k][0], "float64_pm": [float(x) for x in tpm]})
    res = {"jobs": len(want), "reads": n_reads, "worst_fit_jobs_against_float64": truth_rows, "events_per_strand": N_EV, "jobs_with_equal_round_count": int(sum(same)),
           "round_count_histogram_gpu": {str(v): int((out["rounds"] == v).sum()) for v in sorted(set(int(x) for x in out["rounds"]))},
           "fit_max_rel_where_rounds_agree": max(fit_rel) if fit_rel else None,
           "trained_parameter_distance_to_oracle": {n: {"median": pct(v, 0.5), "p90": pct(v, 0.9), "max": pct(v, 1.0)} for n, v in pm_rel.items()},
           "jobs_that_differ_in_round_count": differ[:20], "seed": SEED, "seconds": round(time.time() - t0, 1)}
    fr = sorted(fit_rel)
    res["fit_rel_percentiles"] = {"p50": pct(fr, 0.5), "p90": pct(fr, 0.9), "p99": pct(fr, 0.99), "max": pct(fr, 1.0)}
    res["jobs_with_fit_beyond_1e-4"] = int(sum(x > 1e-4 for x in fr))
    line = json.dumps(res)
    print(line)
    if os.environ.get("OUT"):
        os.makedirs(os.path.dirname(os.path.abspath(os.environ["OUT"])), exist_ok=True)
        open(os.environ["OUT"], "w").write(line + "\n")
    fr = sorted(fit_rel)
    # per call the log-likelihoods agree to 1e-5 (tools/fb_sweep.py); over free-running rounds a window that no state explains (an
    # abasic stretch inside it: log Pr ~ -2e4) feeds both fp32 log-space implementations ~1 % of noise, which the rounds amplify.
    # What must hold: the control flow (every round count), and the fit of the jobs without such windows.
    ok = all(same) and (not fr or pct(fr, 0.9) <= 1e-4)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
