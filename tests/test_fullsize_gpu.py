"""Full-size cases (BASELINE.json configs 2 and 5) through size-independent properties, plus one
50 000-event read against the oracle."""
import os

import numpy as np
import pytest

import nanocall_amd as na
from nanocall_amd import synth
from helpers import IDENT, rescore_path, oracle_viterbi_batch, assert_bits_equal

pytestmark = pytest.mark.gpu


def test_config2_full_batch_properties(gpu_ctx, r73t):
    """1024 reads x 5000 events: (1) deterministic across two runs, (2) every decoded transition is an arc of
    the HMM, (3) the reported path log-probability equals the score recomputed along the decoded path with the
    reference's float operations (bit for bit) on a sample of reads, (4) a checksum of the state array is
    stable under splitting the batch in two calls."""
    n_reads, n_events = 1024, 5000
    ev = synth.generate(r73t, n_reads, n_events)
    off, mean, stdv, start = synth.flat_batch(ev)
    cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
    gpu_ctx.put_model(0, na.scaled_model_table(r73t, IDENT))
    gpu_ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    s1, lp1, st1 = gpu_ctx.viterbi(off, cm, sd, ls)
    s2, lp2, st2 = gpu_ctx.viterbi(off, cm, sd, ls)
    assert np.array_equal(s1, s2) and np.array_equal(lp1.view(np.uint32), lp2.view(np.uint32))
    assert (st1 == 0).all() and np.isfinite(lp1).all()
    # (2) arcs: prev -> cur is stay, step or skip-1
    S = s1.reshape(n_reads, n_events).astype(np.int64)
    prev, cur = S[:, :-1], S[:, 1:]
    valid = (prev == cur) | ((prev & 1023) == (cur >> 2)) | ((prev & 255) == (cur >> 4))
    assert valid.all()
    # (3) rescoring
    for r in (0, 1, 511, 1023):
        a, b = r * n_events, (r + 1) * n_events
        score, ok = rescore_path(r73t, IDENT, 0.3, 0.1, cm[a:b], sd[a:b], ls[a:b], s1[a:b])
        assert ok and np.float32(score).tobytes() == lp1[r].tobytes(), (r, score, lp1[r])
    # (4) split batch
    h = n_reads // 2
    oa, ob = off[: h + 1], off[h:] - off[h]
    sa, lpa, _ = gpu_ctx.viterbi(oa, cm[: h * n_events], sd[: h * n_events], ls[: h * n_events])
    sb, lpb, _ = gpu_ctx.viterbi(ob, cm[h * n_events:], sd[h * n_events:], ls[h * n_events:])
    assert np.array_equal(np.concatenate([sa, sb]), s1)
    assert np.array_equal(np.concatenate([lpa, lpb]).view(np.uint32), lp1.view(np.uint32))


def test_config5_r9_50k_event_read_against_oracle(gpu_ctx, r9t):
    """One 50 000-event R9 read (the reference allocates a 1.6 GB matrix for it) + short neighbours."""
    lens = [50000, 17, 3000]
    ev = synth.generate(r9t, 3, max(lens), first_read=300)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    cat = lambda k: np.concatenate([ev[k][r, :n] for r, n in enumerate(lens)])
    cm, sd, ls = na.events_prepare(cat("mean"), cat("stdv"), cat("start"), 0.0)
    gpu_ctx.put_model(0, na.scaled_model_table(r9t, IDENT))
    gpu_ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    states, logp, status = gpu_ctx.viterbi(off, cm, sd, ls)
    ostates, ologp = oracle_viterbi_batch(r9t, IDENT, 0.3, 0.1, off, cm, sd, ls)
    assert np.array_equal(states, ostates)
    assert_bits_equal(logp, ologp, "path probability")
    mv, seq = na.base_seq(states[:50000])
    assert 45000 < len(seq) < 60000


def test_config5_long_reads_decode_identically_under_a_small_workspace_budget(r9t):
    """BASELINE config 5 with bounded memory (SURVEY 8d hard part 7; the reference needs 1.57 GB per 50 k-event read,
    Viterbi.hpp:50): a 50 000-event R9 read needs a 205 MB back-pointer region; under NCHMM_WS_BUDGET_MB=256 there is room
    for exactly one, so one thread block sweeps and walks back the four reads one after the other through it, and they
    decode to the same bits as with a region per resident block -- through the host-pointer form and through the
    device-pointer form.  One read is re-scored along its decoded path with the reference's float operations."""
    import os
    import torch
    n_reads, n_events = 4, 50000
    ev = synth.generate(r9t, n_reads, n_events, first_read=500)
    off, mean, stdv, start = synth.flat_batch(ev)
    cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)

    def run(budget_mb):
        if budget_mb:
            os.environ["NCHMM_WS_BUDGET_MB"] = str(budget_mb)
        try:
            ctx = na.Context(0)
            ctx.put_model(0, na.scaled_model_table(r9t, IDENT))
            ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
            states, logp, status = ctx.viterbi(off, cm, sd, ls)                     # host pointers (pipeline ranges)
            launches_host = int(ctx.counters()[3])
            dev = torch.device("cuda", 0)
            d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (off.astype(np.int64), cm, sd, ls)]
            d_state = torch.empty(n_reads * n_events, dtype=torch.int16, device=dev)
            d_logp = torch.empty(n_reads, dtype=torch.float32, device=dev)
            d_status = torch.empty(n_reads, dtype=torch.int32, device=dev)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.viterbi_dev(n_reads, n_events, n_reads * n_events, *d, d_state, d_logp, d_status)  # device pointers
            torch.cuda.synchronize()
            launches_dev = int(ctx.counters()[3]) - launches_host
            peak = ctx.mem_stats()[1]
            assert np.array_equal(d_state.cpu().numpy().view(np.uint16), states) and d_logp.cpu().numpy().tobytes() == logp.tobytes()
            ctx.close()
        finally:
            os.environ.pop("NCHMM_WS_BUDGET_MB", None)
        return states, logp, status, launches_host, launches_dev, peak

    s0, lp0, st0, lh0, ld0, peak0 = run(0)
    s1, lp1, st1, lh1, ld1, peak1 = run(256)
    assert lh0 == 1 and ld0 == 1 and lh1 == 1 and ld1 == 1, (lh0, ld0, lh1, ld1)
    assert np.array_equal(s0, s1) and lp0.tobytes() == lp1.tobytes() and (st0 == 0).all() and (st1 == 0).all()
    assert peak1 < peak0 / 2 and peak1 < (400 << 20), (peak0, peak1)      # one region against one per block that may be resident
    score, ok = rescore_path(r9t, IDENT, 0.3, 0.1, cm[:n_events], sd[:n_events], ls[:n_events], s1[:n_events])
    assert ok and np.float32(score).tobytes() == lp1[0].tobytes()


def test_config4_shard_12500_reads_sub_batched(r73t):
    """The per-GPU shard of BASELINE config 4 (100 000 reads x 5 000 events over 8 GPUs = 12 500 reads, 62.5 M events):
    the reference layout would need 2 TB for it, one 4 KiB row per event 256 GB; here a block walks a read back as soon as
    it has swept it, so the workspace is one 20 MB region per resident block (12 GB) however many reads there are, and
    the call is a few launches over contiguous read ranges only to overlap the copy-in.  Properties: every read decodes, every decoded transition is an arc of the HMM, the
    reported log-probabilities equal the score recomputed along the decoded path (bit for bit, sampled across the
    sub-batches), and reads decode to the same bits alone as inside the shard (batch independence)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from nanocall_amd import shard
    n_reads, n_events = 12500, 5000
    mine = shard.lpt_partition(np.full(100000, n_events), 8)[3]            # rank 3's reads: 37 500 .. 49 999
    assert len(mine) == n_reads and mine[0] == 37500
    off, mean, stdv, start = bench.generate_shard(r73t, mine, n_events, threads=min(16, os.cpu_count() or 1))
    cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
    del mean, stdv, start
    ctx = na.Context(0)
    try:
        ctx.put_model(0, na.scaled_model_table(r73t, IDENT))
        ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        launches0 = int(ctx.counters()[3])
        states, logp, status = ctx.viterbi(off, cm, sd, ls)
        n_sub = int(ctx.counters()[3]) - launches0
        assert n_sub >= 2, "the shard was expected to go up range by range"
        assert ctx.mem_stats()[1] < (16 << 30), "the back-pointer workspace must not grow with the number of reads"
        assert (status == 0).all() and np.isfinite(logp).all()
        S = states.reshape(n_reads, n_events)
        for lo in range(0, n_reads, 2500):                                   # arcs, 2500 reads at a time
            blk = S[lo:lo + 2500].astype(np.int32)
            prev, cur = blk[:, :-1], blk[:, 1:]
            assert ((prev == cur) | ((prev & 1023) == (cur >> 2)) | ((prev & 255) == (cur >> 4))).all()
        for r in (0, 6249, 6250, 12499):                                     # rescoring across the sub-batch boundary region
            a, b = r * n_events, (r + 1) * n_events
            score, ok = rescore_path(r73t, IDENT, 0.3, 0.1, cm[a:b], sd[a:b], ls[a:b], states[a:b])
            assert ok and np.float32(score).tobytes() == logp[r].tobytes(), (r, score, logp[r])
        for lo in (0, n_reads - 300):                                        # batch independence
            a, b = lo * n_events, (lo + 300) * n_events
            s2, lp2, _ = ctx.viterbi(off[: 301], cm[a:b], sd[a:b], ls[a:b])
            assert np.array_equal(s2, states[a:b]) and np.array_equal(lp2.view(np.uint32), logp[lo:lo + 300].view(np.uint32))
        # the first read of the shard against the oracle itself
        os_, ol = oracle_viterbi_batch(r73t, IDENT, 0.3, 0.1, np.array([0, n_events], np.uint64), cm[:n_events], sd[:n_events], ls[:n_events])
        assert np.array_equal(states[:n_events], os_) and ol[0].tobytes() == logp[0].tobytes()
    finally:
        ctx.close()


def test_config3_2d_reads_four_round_em_at_full_size():
    """BASELINE config 3 at the size tools/bench_config3.py times it: 1024 template+complement reads of 5000 events per strand,
    two candidate model pairs each, exactly four Parameter_Trainer rounds per pair (nanocall.cpp:360-426 with
    scaling_max_rounds 2, :420), then both strands of every pair decoded with its trained parameters and the better pair kept
    (nanocall.cpp:692-712).  Size-independent checks: (1) a pair's training result does not depend on the batch it is in -- six
    pairs trained on their own return the same bits; (2) one of them equals the reference's loop driven by the CPU oracle, within
    the EM tolerances of test_train_reads_gpu.py; (3) every round counter says four and every fit is finite; (4) every read
    gets a winner, whose decode equals that read decoded in a batch of its own, bit for bit, and stays on the stay / step / skip
    graph; (5) the second call returns the same bits as the first."""
    from nanocall_amd import api
    from test_train_reads_gpu import reference_train_job
    with na.Context(0) as ctx:        # (its own context: 2048 candidate tables go through the model slots)
        _config3(ctx, api, reference_train_job)


def _config3(gpu_ctx, api, reference_train_job):
    n_reads, n_ev = 1024, 5000
    names = ["r73.c.p1", "r73.c.p2", "r73.t"]           # sorted by name, like the reference's std::map
    strands = [1, 1, 0]
    tables = [na.builtin_model(n) for n in names]
    states = np.stack([na.model_load(t) for t in tables])
    e0 = synth.generate(tables[2], n_reads, n_ev)
    e1 = synth.generate(tables[0], n_reads, n_ev, first_read=10**6)
    mean = np.stack([e0["mean"], e1["mean"]], 1).reshape(-1)
    stdv = np.stack([e0["stdv"], e1["stdv"]], 1).reshape(-1)
    start = np.stack([e0["start"], e1["start"]], 1).reshape(-1)
    del e0, e1
    _, stdv, _ = na.events_prepare(mean, stdv, None, 0.0)          # Event::update_logs: stdv 0 -> .01
    so = (np.arange(2 * n_reads + 1) * n_ev).astype(np.uint64)
    opts = api.train_opts(scaling_max_rounds=2, scaling_min_progress=0.0)
    jr, j0, j1 = api.train_enumerate(opts, strands, so, np.ones(n_reads, np.uint8))
    assert len(jr) == 2 * n_reads and list(zip(jr[:2], j0[:2], j1[:2])) == [(0, 2, 0), (0, 2, 1)]
    out = gpu_ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    # (3)
    assert (out["rounds"] == 4).all() and np.isfinite(out["fit"]).all() and np.isfinite(out["pm"]).all() and np.isfinite(out["st"]).all()
    assert (out["pm"][:, 0] > 0.5).all() and (out["pm"][:, 0] < 2.0).all() and (out["st"] > 0).all() and (out["st"] < 1).all()
    # (5)
    again = gpu_ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
    for k in ("pm", "st", "fit", "rounds", "preferred"):
        assert out[k].tobytes() == again[k].tobytes(), k
    # (1) six pairs on their own
    for r in (0, 511, 1023):
        a, b = int(so[2 * r]), int(so[2 * r + 2])
        so1 = (so[2 * r:2 * r + 3] - so[2 * r]).astype(np.uint64)
        jr1, j01, j11 = api.train_enumerate(opts, strands, so1, np.ones(1, np.uint8))
        one = gpu_ctx.train_reads(opts, states, so1, mean[a:b], stdv[a:b], start[a:b], jr1, j01, j11)
        for k in ("pm", "st", "fit", "rounds"):
            assert one[k].tobytes() == out[k][2 * r:2 * r + 2].tobytes(), (r, k, one[k], out[k][2 * r:2 * r + 2])
    # (2) the reference's loop on the oracle, pair (t, c.p1) of read 511
    r, k = 511, 2 * 511
    windows, wst = [], []
    for s in (0, 1):
        lo, hi = int(so[2 * r + s]), int(so[2 * r + s + 1])
        half = min(opts.scaling_num_events, hi - lo) // 2
        for sl in (slice(lo, lo + half), slice(hi - half, hi)):
            windows.append((mean[sl], stdv[sl], start[sl])); wst.append(s)
    pm, st, fit, rnd = reference_train_job(opts, tables, windows, wst, (2, 0), [1, 0, 0, 1, 1, 1], [0.1, 0.3, 0.1, 0.3])
    assert rnd == 4 and abs(out["fit"][k] - fit) <= 1e-4 * abs(fit), (rnd, out["fit"][k], fit)
    got = out["pm"][k]
    for q in (0, 4):
        assert abs(got[q] - pm[q]) <= 2e-4 * abs(pm[q]), (q, got, pm)
    assert np.allclose(out["st"][k], st, rtol=5e-4, atol=0), (out["st"][k], st)
    # var and var_sd (Parameter_Trainer.hpp:406-426: differences of large sums) carry fp32 noise of ~1e-4 per round in the
    # reference's own arithmetic, and four free-running rounds compound it.  They are held to the same 5e-4 as in
    # tests/test_fwbw_gpu.py -- against the REAL-NUMBER answer: the same four rounds evaluated in float64 from the same fp32
    # inputs (tools/fb_truth.py --config3-read -> tests/golden/config3_read511_truth64.json; the fp32 oracle sits 1.5e-4 / 6e-5
    # from it for this pair).  Everything else against that answer too, at the tolerances above.
    import json
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_read511_truth64.json")))
    truth, orc = doc["truth64_rounds"][-1], doc["oracle_fp32_rounds"][-1]
    assert doc["read"] == r and np.allclose(orc["pm"], pm, rtol=1e-6, atol=1e-7), (orc["pm"], pm)       # the fixture is this pair's oracle loop
    t_pm, t_st = np.array(truth["pm"]), np.array(truth["st"])
    for q, tol in ((0, 2e-4), (4, 2e-4), (3, 5e-4), (5, 5e-4)):
        assert abs(got[q] - t_pm[q]) <= tol * abs(t_pm[q]), (q, got, t_pm)
        assert abs(pm[q] - t_pm[q]) <= tol * abs(t_pm[q]), ("oracle", q, pm, t_pm)
    assert abs(got[1] - t_pm[1]) <= 2e-4 * 60 and abs(got[2] - t_pm[2]) <= 2e-4 * 60 / float(start[int(so[2 * r + 1]) - 1])
    assert np.allclose(out["st"][k], t_st, rtol=5e-4, atol=0), (out["st"][k], t_st)
    assert abs(out["fit"][k] - truth["fit"]) <= 1e-4 * abs(truth["fit"])
    # (4) decode with the trained parameters
    bc = gpu_ctx.basecall_reads(opts, states, so, mean, stdv, start, jr, j0, j1, out["pm"], out["st"])
    assert (bc["best_job"] >= 0).all() and (bc["best_job"][:, 0] == bc["best_job"][:, 1]).all()
    assert (bc["best_job"][:, 0] // 2 == np.arange(n_reads)).all() and np.isfinite(bc["best_logp"]).all()
    # (the complement strands were generated from r73.c.p1: the pair trained on it should win nearly everywhere)
    assert (bc["best_job"][:, 0] % 2 == 0).mean() > 0.95
    S = bc["states"].reshape(2 * n_reads, n_ev).astype(np.int64)
    prev, cur = S[:, :-1], S[:, 1:]
    assert ((prev == cur) | ((prev & 1023) == (cur >> 2)) | ((prev & 255) == (cur >> 4))).all()
    for r in (0, 511, 1023):
        a, b = int(so[2 * r]), int(so[2 * r + 2])
        so1 = (so[2 * r:2 * r + 3] - so[2 * r]).astype(np.uint64)
        jr1, j01, j11 = api.train_enumerate(opts, strands, so1, np.ones(1, np.uint8))
        one = gpu_ctx.basecall_reads(opts, states, so1, mean[a:b], stdv[a:b], start[a:b], jr1, j01, j11, out["pm"][2 * r:2 * r + 2], out["st"][2 * r:2 * r + 2])
        assert np.array_equal(one["states"], bc["states"][a:b]) and one["best_logp"].tobytes() == bc["best_logp"][r:r + 1].tobytes()
        assert one["best_job"][0, 0] == bc["best_job"][r, 0] - 2 * r


def test_contexts_that_find_the_device_memory_taken_shrink_their_workspace(r9t):
    """Four contexts in one process, each decoding twelve 60 000-event reads: the first three take 70 GB of back-pointer regions each
    (288 regions of 246 MB), so what the fourth sized its budget from is gone when it allocates.  It retries with what is free -- fewer
    regions, the same decode -- instead of failing the batch with NCHMM_E_NOMEM (round 6: four worker processes on one GPU, or a test
    harness with a context per sweep form, ran into exactly that)."""
    n_reads, n_events = 12, 60000
    ev = synth.generate(r9t, n_reads, n_events, first_read=4100)
    off, mean, stdv, start = synth.flat_batch(ev)
    cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
    ctxs, outs = [], []
    try:
        for k in range(4):
            ctx = na.Context(0)
            ctxs.append(ctx)
            ctx.put_model(0, na.scaled_model_table(r9t, IDENT))
            ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
            outs.append(ctx.viterbi(off, cm, sd, ls))
        peaks = [c.mem_stats()[1] >> 30 for c in ctxs]
    finally:
        for c in ctxs:
            c.close()
    for s, lp, st in outs[1:]:
        assert np.array_equal(s, outs[0][0]) and lp.tobytes() == outs[0][1].tobytes() and (st == 0).all()
    assert peaks[0] >= 60 and min(peaks) < peaks[0], peaks          # (somebody did have to make do with less)
    score, ok = rescore_path(r9t, IDENT, 0.3, 0.1, cm[:n_events], sd[:n_events], ls[:n_events], outs[3][0][:n_events])
    assert ok and np.float32(score).tobytes() == outs[3][1][0].tobytes()
