"""The batched EM driver (nchmm_train_reads = train_reads of nanocall.cpp:292-574) against a straight
transcription of that loop driven by the CPU oracle's train_one_round: same control flow (round counts,
roll-backs, selected models), parameters within the EM tolerances of test_fwbw_gpu.py."""
import numpy as np
import pytest

import nanocall_amd as na
from nanocall_amd import api, synth
import nc_oracle as oracle

pytestmark = pytest.mark.gpu


def reference_train_job(opts, tables, windows, strands, m, pm, st):
    """One iteration of the reference's model loop (nanocall.cpp:360-426 / :476-542) on the oracle."""
    off = np.concatenate([[0], np.cumsum([len(w[0]) for w in windows])]).astype(np.uint64)
    mean = np.concatenate([w[0] for w in windows]); stdv = np.concatenate([w[1] for w in windows])
    start = np.concatenate([w[2] for w in windows])
    two_d = m[0] >= 0 and m[1] >= 0
    t0 = tables[m[0] if m[0] >= 0 else m[1]]
    t1 = tables[m[1] if m[1] >= 0 else m[0]]
    crt_pm, crt_st, crt_fit, rnd = np.float32(pm), np.float32(st), np.float32(-np.inf), 0
    while True:
        old_pm, old_st, old_fit = crt_pm.copy(), crt_st.copy(), crt_fit
        r = oracle.train_one_round(off, np.asarray(strands, np.uint32), mean, stdv, start, t0, t1, old_pm, old_st,
                                   opts.default_p_stay, opts.default_p_skip, opts.train_drift, bool(opts.train_scaling),
                                   bool(opts.train_transitions))
        crt_pm, crt_fit = r["pm"], r["fit"]
        new_st = r["st"].copy()
        for s in range(2):                      # the oracle leaves NaN for an absent strand, like the reference
            if s not in strands:
                new_st[2 * s:2 * s + 2] = old_st[2 * s:2 * s + 2]
        crt_st = new_st
        if r["done"]:
            break
        if crt_fit < old_fit:
            crt_pm, crt_st, crt_fit = old_pm, old_st, old_fit
            break
        rnd += 1
        limit = 2 * opts.scaling_max_rounds if two_d else opts.scaling_max_rounds
        if rnd >= limit or (rnd > 1 and crt_fit < old_fit + opts.scaling_min_progress):
            break
    return crt_pm, crt_st, crt_fit, rnd


def test_train_reads_matches_reference_loop(gpu_ctx):
    opts = api.train_opts(scaling_max_rounds=2, scaling_num_events=120, scaling_select_threshold=5.0)
    names = ["r73.c.p1", "r73.c.p2", "r73.t"]           # sorted by name, like the reference's std::map
    strands = [1, 1, 0]
    tables = [na.builtin_model(n) for n in names]
    states = np.stack([na.model_load(t) for t in tables])
    # read 0: 2D (template from r73.t, complement from r73.c.p1); read 1: template only; read 2: complement too short
    ev = [(synth.generate(tables[2], 1, 700, first_read=500), synth.generate(tables[0], 1, 650, first_read=501)),
          (synth.generate(tables[2], 1, 90, first_read=502), None),
          (synth.generate(tables[2], 1, 300, first_read=503), synth.generate(tables[1], 1, 4, first_read=504))]
    together = [1, 0, 0]
    mean, stdv, start, so = [], [], [], [0]
    for e0, e1 in ev:
        for e in (e0, e1):
            if e is not None:
                m, s, t = e["mean"][0], e["stdv"][0], e["start"][0]
                _, s, _ = na.events_prepare(m, s, t, 0.0)      # Event::update_logs: stdv 0 -> .01
                mean.append(m); stdv.append(s); start.append(t)
            so.append(so[-1] + (0 if e is None else len(e["mean"][0])))
    mean, stdv, start = np.concatenate(mean), np.concatenate(stdv), np.concatenate(start)
    so = np.array(so, np.uint64)
    jr, j0, j1 = api.train_enumerate(opts, strands, so, together)
    # read 0 -> (t, c.p1), (t, c.p2); read 1 -> (t,-1); read 2 -> (t,-1) only (4 complement events < min_ed_events)
    assert list(zip(jr, j0, j1)) == [(0, 2, 0), (0, 2, 1), (1, 2, -1), (2, 2, -1)]
    out = gpu_ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    for k, (r, a, b) in enumerate(zip(jr, j0, j1)):
        windows, wst = [], []
        for s, mm in ((0, a), (1, b)):
            if mm < 0:
                continue
            lo, hi = int(so[2 * r + s]), int(so[2 * r + s + 1])
            half = min(opts.scaling_num_events, hi - lo) // 2
            for sl in (slice(lo, lo + half), slice(hi - half, hi)):
                windows.append((mean[sl], stdv[sl], start[sl])); wst.append(s)
        pm, st, fit, rnd = reference_train_job(opts, tables, windows, wst, (a, b), [1, 0, 0, 1, 1, 1], [0.1, 0.3, 0.1, 0.3])
        assert out["rounds"][k] == rnd, (k, out["rounds"][k], rnd)
        assert abs(out["fit"][k] - fit) <= 1e-4 * abs(fit), (k, out["fit"][k], fit)
        got = out["pm"][k]
        for q in (0, 4):
            assert abs(got[q] - pm[q]) <= 2e-4 * abs(pm[q]), (k, q, got, pm)
        # var, var_sd: ill-conditioned (fp32 reference noise ~1e-4 per round, test_fwbw_gpu.py) and this loop
        # runs free, each round starting from its own previous parameters
        for q in (3, 5):
            assert abs(got[q] - pm[q]) <= 1e-3 * abs(pm[q]), (k, q, got, pm)  # (free-running rounds against the fp32 ORACLE: 1e-3 -- 48 full-size jobs measured at most 6.9e-4 / 2.1e-4, profiles/r06_notes.md section 4; against a float64 evaluation the bound is 5e-4, tests/test_fullsize_gpu.py)
        assert abs(got[1] - pm[1]) <= 2e-4 * 60 and abs(got[2] - pm[2]) <= 2e-4 * 60 / max(float(start.max()), 1.0)
        assert np.allclose(out["st"][k], st, rtol=5e-4, atol=0), (k, out["st"][k], st)
    # selection: read 0's pair trained on the matching complement model must win by > threshold or not at all,
    # exactly as the oracle-side fits say
    fits = out["fit"][:2]
    exp = int(np.argmax(fits)) if abs(fits[0] - fits[1]) > opts.scaling_select_threshold else -1
    assert out["preferred"][0, 2] == exp
    assert out["preferred"][1, 0] == 2 and out["preferred"][2, 0] == 3       # single candidates are always preferred
    assert out["preferred"][2, 1] == -1 and out["preferred"][0, 0] == -1


@pytest.fixture
def own_ctx():
    """(a context of its own: a hundred jobs fill model / transition slots that tests of the shared context expect unset)"""
    ctx = na.Context(0)
    yield ctx
    ctx.close()


def test_jobs_trained_in_two_parts_on_two_lanes_equal_the_one_part_run(own_ctx, monkeypatch):
    gpu_ctx = own_ctx
    """A call with 64 jobs or more trains them in two parts, each on its own EM lane of the context, so that the host's share of
    one part's round hides behind the other part's kernels (nchmm_train.cpp).  A job's rounds do not depend on what it is batched
    with: 60 reads -- 2D with two complement candidates, template only, of adversarial kinds (some of their jobs stop early, some
    roll back, so the parts shrink unevenly) -- give bit-identical parameters, fits, round counts and preferences whether they
    train in one part (NCHMM_EM_LANES=1), two, or twenty (a small forward-backward budget), and on a second call of the same context
    (the lanes' buffers reused)."""
    import adversarial
    opts = api.train_opts(scaling_max_rounds=3, scaling_num_events=160, scaling_select_threshold=5.0)
    names = ["r73.c.p1", "r73.c.p2", "r73.t"]
    strands = [1, 1, 0]
    tables = [na.builtin_model(n) for n in names]
    states = np.stack([na.model_load(t) for t in tables])
    rng = np.random.default_rng(77)
    mean, stdv, start, so, together = [], [], [], [0], []
    params = (1.05, -2.0, 0.0004, 1.2, 0.95, 1.3)
    for r in range(60):
        two_d = r % 3 != 2
        for s in range(2):
            n = int(rng.integers(100, 900)) if (s == 0 or two_d) else 0
            if n:
                kind = adversarial.KINDS[(r + s) % len(adversarial.KINDS)] if r % 4 == 0 else "matched"
                m, sd, t = adversarial.events(kind, tables[2] if s == 0 else tables[r % 2], params, n, seed=4000 + 2 * r + s, other_table=tables[1])
                _, sd, _ = na.events_prepare(m, sd, t, 0.0)
                mean.append(m); stdv.append(sd); start.append(t)
            so.append(so[-1] + n)
        together.append(1 if two_d else 0)
    mean, stdv, start = np.concatenate(mean), np.concatenate(stdv), np.concatenate(start)
    so = np.array(so, np.uint64)
    jr, j0, j1 = api.train_enumerate(opts, strands, so, together)
    assert len(jr) >= 64
    monkeypatch.setenv("NCHMM_EM_LANES", "1")
    one = gpu_ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    monkeypatch.delenv("NCHMM_EM_LANES")
    two = gpu_ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    again = gpu_ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    # ... and in many parts taking turns on the two lanes: a budget of 64 MiB is 4096 events of alpha rows, 2048 per lane -- some
    # twenty parts of five jobs, which finish at different rounds
    monkeypatch.setenv("NCHMM_FB_BUDGET_MB", "64")
    small = na.Context(0)
    try:
        many = small.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
        assert int(small.mem_stats()[1]) < (1 << 30)                 # (its workspaces stayed small: 16 KiB x 2 x 2048 events and change)
    finally:
        small.close()
    monkeypatch.delenv("NCHMM_FB_BUDGET_MB")
    assert len(set(one["rounds"].tolist())) > 1                      # the jobs do not all stop together
    for other in (two, again, many):
        for k in ("pm", "st", "fit", "rounds", "preferred"):
            assert one[k].tobytes() == other[k].tobytes(), k
    # ... and a Viterbi launch behind it finds its lanes as they were (the second EM lane computes on Viterbi lane 1's stream)
    cm, sd, ls = na.events_prepare(mean[:700], stdv[:700], start[:700], 0.0)
    gpu_ctx.put_model(0, na.scaled_model_table(tables[2]))
    gpu_ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    s1, lp1, st1 = gpu_ctx.viterbi(np.array([0, 700], np.uint64), cm, sd, ls)
    s2, lp2, st2 = gpu_ctx.viterbi(np.array([0, 700], np.uint64), cm, sd, ls)
    assert st1[0] == 0 and np.array_equal(s1, s2) and lp1[0] == lp2[0]


def test_basecall_reads_picks_best_model_and_matches_oracle(gpu_ctx):
    """basecall_reads: candidates decoded with their own parameters, winner by (summed) path log-prob."""
    opts = api.train_opts()
    names = ["r73.c.p1", "r73.c.p2", "r73.t"]
    strands = [1, 1, 0]
    tables = [na.builtin_model(n) for n in names]
    states10 = np.stack([na.model_load(t) for t in tables])
    e0 = synth.generate(tables[2], 1, 600, first_read=700)
    e1 = synth.generate(tables[1], 1, 500, first_read=701)      # complement really comes from c.p2
    e2 = synth.generate(tables[2], 1, 350, first_read=702)
    mean = np.concatenate([e0["mean"][0], e1["mean"][0], e2["mean"][0]])
    stdv = np.concatenate([e0["stdv"][0], e1["stdv"][0], e2["stdv"][0]])
    start = np.concatenate([e0["start"][0], e1["start"][0], e2["start"][0]])
    so = np.array([0, 600, 1100, 1450, 1450], np.uint64)
    jr, j0, j1 = api.train_enumerate(opts, strands, so, [1, 0])
    assert list(zip(jr, j0, j1)) == [(0, 2, 0), (0, 2, 1), (1, 2, -1)]
    pm = np.float32([[1.01, 0.3, 0.001, 1.05, 0.98, 1.1], [0.99, -0.2, 0.0, 1.0, 1.02, 0.9], [1, 0, 0, 1, 1, 1]])
    st = np.float32([[0.11, 0.27, 0.1, 0.3], [0.1, 0.3, 0.12, 0.25], [0.1, 0.3, 0.1, 0.3]])
    out = gpu_ctx.basecall_reads(opts, states10, so, mean, stdv, start, jr, j0, j1, pm, st)
    # oracle: every candidate
    exp = {}
    for k, (r, a, b) in enumerate(zip(jr, j0, j1)):
        for s, m in ((0, a), (1, b)):
            if m < 0:
                continue
            lo, hi = int(so[2 * r + s]), int(so[2 * r + s + 1])
            om = oracle.Model(tables[m], pm[k])
            ot = oracle.Transitions(float(st[k, 2 * s + 1]), float(st[k, 2 * s]))
            cm, sd, ls = oracle.events_prepare(mean[lo:hi], stdv[lo:hi], start[lo:hi], float(pm[k, 2]))
            exp[(k, s)] = oracle.viterbi(om, ot, cm, sd, ls)
    tot = [np.float32(exp[(k, 0)][2] + exp[(k, 1)][2]) for k in (0, 1)]
    win = 1 if tot[1] >= tot[0] else 0
    assert win == 1                                           # the matching complement model wins
    assert out["best_job"].tolist() == [[win, win], [2, -1]]
    for s in range(2):
        lo, hi = int(so[s]), int(so[s + 1])
        assert np.array_equal(out["states"][lo:hi], exp[(win, s)][0])
        assert out["best_logp"][0, s].tobytes() == np.float32(exp[(win, s)][2]).tobytes()
    assert np.array_equal(out["states"][1100:1450], exp[(2, 0)][0])
    assert out["best_logp"][1, 0].tobytes() == np.float32(exp[(2, 0)][2]).tobytes() and np.isnan(out["best_logp"][1, 1])
    # with a preferred pair only that one is decoded
    pref = np.full((2, 3), -1, np.int32); pref[0, 2] = 0
    out2 = gpu_ctx.basecall_reads(opts, states10, so, mean, stdv, start, jr, j0, j1, pm, st, preferred=pref)
    assert out2["best_job"].tolist() == [[0, 0], [2, -1]]
    assert np.array_equal(out2["states"][0:600], exp[(0, 0)][0])


def test_pool_shard_with_reads_but_no_jobs(gpu_ctx):
    """Two contexts on device 0 (the sharding logic on a one-GPU box): the second read is too short to have a candidate, so
    its shard has a read and NO job.  The pool must neither fail (null job arrays) nor leave that read's results unset: it
    reports what the single-context call reports -- best_job -1, best_logp NaN -- and decodes the other shard as before."""
    opts = api.train_opts()
    names = ["r73.c.p1", "r73.c.p2", "r73.t"]
    tables = [na.builtin_model(n) for n in names]
    states10 = np.stack([na.model_load(t) for t in tables])
    e0 = synth.generate(tables[2], 1, 500, first_read=810)
    e1 = synth.generate(tables[2], 1, 30, first_read=811)
    mean = np.concatenate([e0["mean"][0], e1["mean"][0]])
    stdv = np.concatenate([e0["stdv"][0], e1["stdv"][0]])
    start = np.concatenate([e0["start"][0], e1["start"][0]])
    so = np.array([0, 500, 500, 530, 530], np.uint64)
    jr, j0, j1 = np.int32([0]), np.int32([2]), np.int32([-1])        # read 1 has no candidate at all
    pm = np.float32([[1, 0, 0, 1, 1, 1]])
    st = np.float32([[0.1, 0.3, 0.1, 0.3]])
    one = gpu_ctx.basecall_reads(opts, states10, so, mean, stdv, start, jr, j0, j1, pm, st)
    with api.Pool([0, 0]) as pool:
        assert len(pool) == 2
        two = pool.basecall_reads(opts, states10, so, mean, stdv, start, jr, j0, j1, pm, st)
        tr = pool.train_reads(opts, states10, so, mean, stdv, start, jr, j0, j1)
    assert two["best_job"].tolist() == one["best_job"].tolist() == [[0, -1], [-1, -1]]
    assert np.array_equal(two["states"][:500], one["states"][:500])
    assert two["best_logp"][0, 0].tobytes() == one["best_logp"][0, 0].tobytes()
    assert np.isnan(two["best_logp"][1]).all() and np.isnan(two["best_logp"][0, 1])
    assert tr["rounds"].shape == (1,) and tr["rounds"][0] >= 1


def test_pool_over_two_distinct_devices_uses_rccl_for_the_counters(gpu_ctx):
    """nchmm_pool_* on two DIFFERENT devices: the read-parallel contract of nanocall.cpp:611-621 (every read decoded once,
    results in input order whatever device took it) and the one collective of the design -- the RCCL all-reduce of the
    counters (used_rccl == 1).  Same reads through the single context give the same bytes."""
    if api.device_count() < 2:       # (asked here, not in a decorator: collection must not load the library or touch HIP)
        pytest.skip("needs two distinct GPUs (the driver's multi-GPU node)")
    opts = api.train_opts()
    names = ["r73.c.p1", "r73.c.p2", "r73.t"]
    tables = [na.builtin_model(n) for n in names]
    states10 = np.stack([na.model_load(t) for t in tables])
    lens = [700, 300, 520, 640, 410, 333]
    evs = [synth.generate(tables[2], 1, n, first_read=900 + i) for i, n in enumerate(lens)]
    mean = np.concatenate([e["mean"][0] for e in evs])
    stdv = np.concatenate([e["stdv"][0] for e in evs])
    start = np.concatenate([e["start"][0] for e in evs])
    so = np.zeros(2 * len(lens) + 1, np.uint64)
    so[1::2] = np.cumsum(lens)          # strand 0 = the read, strand 1 empty
    so[2::2] = np.cumsum(lens)
    jr, j0, j1 = api.train_enumerate(opts, [1, 1, 0], so, [0] * len(lens))
    pm = np.tile(np.float32([1, 0, 0, 1, 1, 1]), (len(jr), 1))
    st = np.tile(np.float32([0.1, 0.3, 0.1, 0.3]), (len(jr), 1))
    before = gpu_ctx.counters().astype(np.int64)
    one = gpu_ctx.basecall_reads(opts, states10, so, mean, stdv, start, jr, j0, j1, pm, st)
    delta = gpu_ctx.counters().astype(np.int64) - before
    with api.Pool([0, 1]) as pool:
        two = pool.basecall_reads(opts, states10, so, mean, stdv, start, jr, j0, j1, pm, st)
        counters, used_rccl = pool.counters()
    assert used_rccl
    # reads and events decoded: the two fresh contexts together did exactly what the single context did
    assert counters[0] == delta[0] and counters[1] == delta[1] == sum(lens)
    assert np.array_equal(two["states"], one["states"])
    assert two["best_job"].tolist() == one["best_job"].tolist()
    assert two["best_logp"].tobytes() == one["best_logp"].tobytes()


def test_short_soak_of_parts_lanes_and_decodes_on_one_context():
    """tools/soak_em_lanes.py, a short run: training calls in one part and in parts on the two lanes under budgets from 48 MiB to
    16 GiB, decodes of a ragged batch between them and device-resident decodes queued across them -- no difference anywhere
    (profiles/r06_soak_em_lanes.json is the long run)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_em_lanes.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ITER="17", POOL="300"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["iterations"] == 17 and d["differences"] == 0 and d["async_decodes"] >= 5 and len(d["parts_budgets_mb"]) == 3
