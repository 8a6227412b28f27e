"""GPU parity of the two forms of the Viterbi sweep (viterbi_kernel.hip `wide`: 8 waves per read; viterbi_ll_kernel.hip `ll`:
16 waves per read, one read per CU): each FORCED through the C ABI (nchmm_set_sweep) on the cases that exercise the exactness
machinery -- ties to the lowest predecessor, the sum-by-sum rescan, the traceback's re-walk, reads around the traceback's
geometry (128 and 256 segments of 80 events per round), unreachable states, ragged batches with empty reads -- bit for bit
against the CPU oracle; and the plan's choice (NCHMM_SWEEP_AUTO, nchmm_plan.hpp choose_sweep) observed through
nchmm_sweep_stats.  Reference: Viterbi.hpp:44-99,120-142 (arithmetic), nanocall.cpp:687-689,93 (one strand per call)."""
import os

import numpy as np
import pytest

import nanocall_amd as na
from nanocall_amd import synth
from helpers import IDENT, ragged_batch, oracle_viterbi_batch, assert_bits_equal

pytestmark = pytest.mark.gpu

FORMS = ("wide", "ll", "ahead")      # "ahead": ll with the emissions of the longest reads computed in front of the sweep (emission_kernel.hip)


def _ctx(form, **env):
    for k, v in env.items():
        os.environ[k] = v
    try:
        ctx = na.Context(0)
    finally:
        for k in env:
            del os.environ[k]
    ctx.set_sweep(form)
    return ctx


def _check(ctx, table, params, p_skip, p_stay, off, cm, sd, ls, form):
    ctx.put_model(0, na.scaled_model_table(table, params))
    ctx.put_transitions(0, *na.transitions_fast(p_skip, p_stay))
    before, ahead_before = ctx.sweep_stats(), ctx.ahead_stats()
    states, logp, status = ctx.viterbi(off, cm, sd, ls)
    after, ahead_after = ctx.sweep_stats(), ctx.ahead_stats()
    launches = (after[0] - before[0], after[1] - before[1])
    low_latency = form in ("ll", "ahead")
    assert launches[0 if low_latency else 1] == 0 and launches[1 if low_latency else 0] > 0, (form, launches)
    lens = np.diff(off.astype(np.int64))
    if form == "ahead" and 0 < lens.max() <= 16384 and len(lens) <= 2 * ctx.grid_slots():
        # (one-call batches of up to two grid-fulls go up as one launch: its longest reads, as far as the 256 MiB buffer holds)
        assert ahead_after[0] > ahead_before[0] and ahead_after[1] > ahead_before[1], (ahead_before, ahead_after)
    if form != "ahead":
        assert ahead_after == ahead_before
    ostates, ologp = oracle_viterbi_batch(table, params, p_skip, p_stay, off, cm, sd, ls)
    nz = np.diff(off.astype(np.int64)) > 0
    assert np.array_equal(states, ostates), form
    assert_bits_equal(logp[nz], ologp[nz], f"path probability ({form})")
    assert np.isnan(logp[~nz]).all() and (status == 0).all()
    return states, logp


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("lens", [[1], [2], [3, 0, 4], [64, 257, 400], [1000, 1, 37, 0, 512, 1025, 2049]])
def test_ragged_reads_bit_exact_in_each_form(form, lens, r73t):
    ctx = _ctx(form)
    try:
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, lens, first_read=21)
        _check(ctx, r73t, IDENT, 0.3, 0.1, off, cm, sd, ls, form)
        params = (1.05, 2.5, 0.002, 1.1, 0.9, 1.2)
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, lens, first_read=5, drift=params[2])
        _check(ctx, r73t, params, 0.17, 0.12, off, cm, sd, ls, form)
    finally:
        ctx.close()


@pytest.mark.parametrize("form", FORMS)
def test_exact_ties_and_rescans_in_each_form(form, r73t, r9t):
    """Identical states make every comparison a tie (lowest predecessor wins, Viterbi.hpp:84 strict >); pairs of identical
    states tie inside groups only; the profile counters show the tie and rescan branches ran."""
    ctx = _ctx(form, NCHMM_PROFILE="1")
    try:
        t = r73t.copy()
        t[:, :] = t[0, :]
        n = 50
        mean = np.full(n, t[0, 0], np.float32)
        stdv = np.full(n, t[0, 2], np.float32)
        cm, sd, ls = na.events_prepare(mean, stdv, None, 0.0)
        _check(ctx, t, IDENT, 0.3, 0.1, np.array([0, n], np.uint64), cm, sd, ls, form)
        t2 = r73t.copy()
        t2[1::2] = t2[0::2]
        off, mean, stdv, start, cm, sd, ls = ragged_batch(t2, [300, 411], first_read=3)
        _check(ctx, t2, IDENT, 0.3, 0.1, off, cm, sd, ls, form)
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r9t, [2000], first_read=5)
        _check(ctx, r9t, IDENT, 0.3, 0.1, off, cm, sd, ls, form)
        tk = ctx.profile_ticks()
        assert tk[6] > 0 and tk[7] > 0, ("rescan / tie branches not exercised", tk)
    finally:
        ctx.close()


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("margin", ["0", "2"])
def test_traceback_rewalk_is_exact_in_each_form(form, margin, r73t):
    ctx = _ctx(form, NCHMM_TB_MARGIN=margin, NCHMM_PROFILE="1")
    try:
        lens = [5000, 1023, 1024, 2100, 4097]
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, lens, first_read=60)
        _check(ctx, r73t, IDENT, 0.3, 0.1, off, cm, sd, ls, form)
        tk = ctx.profile_ticks()
        assert tk[5] > 0 and tk[4] > 0, "the re-walk path was not exercised"
    finally:
        ctx.close()


@pytest.mark.parametrize("form", FORMS)
def test_read_lengths_around_the_traceback_geometry_of_each_form(form, r73t):
    """128 (wide) / 256 (ll) segments of up to 80 events per round: 10 240 / 20 480 events -- one event either side of a segment,
    of a round and of two rounds of either form."""
    ctx = _ctx(form)
    try:
        for lens in ([79, 80, 81, 82, 160, 161, 10239, 10240, 10241, 10242], [20479, 20480, 20481, 20482, 40961]):
            off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, lens, first_read=7)
            _check(ctx, r73t, IDENT, 0.3, 0.1, off, cm, sd, ls, form)
    finally:
        ctx.close()


@pytest.mark.parametrize("form", FORMS)
def test_zero_stdv_and_out_of_range_events_in_each_form(form, r73t):
    """stdv = 0 becomes 0.01 (Event.hpp:39-42); an event outside the range the reciprocal division is validated for sends its
    chunk through true division -- same bits either way."""
    ctx = _ctx(form)
    try:
        off, mean, stdv, start, _, _, _ = ragged_batch(r73t, [200, 1300], first_read=9)
        stdv[::7] = 0.0
        mean[1100] = np.float32(3.0e6)          # |x| > 2^20: leaves the fast range (viterbi_common.hpp: event_in_fast_range)
        stdv[777] = np.float32(2000.0)          # y > 1024
        cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
        _check(ctx, r73t, IDENT, 0.3, 0.1, off, cm, sd, ls, form)
    finally:
        ctx.close()


def test_both_forms_return_the_same_bits_on_a_full_grid(r73t):
    """more reads than either form has block slots, two model / transition slots picked per read: the low-latency form's queue,
    region pool and tickets work like the wide form's"""
    ctx = na.Context(0)
    try:
        n_reads = ctx.grid_slots() + 37
        rng = np.random.default_rng(11)
        lens = rng.integers(3, 60, size=n_reads).tolist()
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, lens, first_read=100)
        ctx.put_model(0, na.scaled_model_table(r73t, IDENT))
        ctx.put_model(1, na.scaled_model_table(r73t, (0.97, 1.0, 0.0, 1.1, 1.0, 0.9)))
        ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        ctx.put_transitions(1, *na.transitions_fast(0.28, 0.09))
        slot = (np.arange(n_reads) % 2).astype(np.int32)
        res = {}
        for form in FORMS:
            ctx.set_sweep(form)
            res[form] = ctx.viterbi(off, cm, sd, ls, model_slot=slot, trans_slot=slot)
        for form in ("ll", "ahead"):
            assert np.array_equal(res["wide"][0], res[form][0]) and res["wide"][1].tobytes() == res[form][1].tobytes(), form
            assert (res[form][2] == 0).all()
    finally:
        ctx.close()


def test_the_plan_picks_the_form_from_the_read_lengths(r73t):
    """NCHMM_SWEEP_AUTO: one read per call (the reference's call shape) and any batch of at most one read per CU take the
    low-latency form; a batch of equal reads that fills the wide form's slots twice takes the wide form; a batch whose
    duration one long read would set takes the low-latency form -- and the bits never depend on the choice."""
    ctx = na.Context(0)
    try:
        ctx.put_model(0, na.scaled_model_table(r73t, IDENT))
        ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        slots = ctx.grid_slots()
        n_cu = slots // 2

        def run(lens):
            ev = synth.generate(r73t, len(lens), int(max(lens)))
            keep = np.arange(int(max(lens)))[None, :] < np.asarray(lens)[:, None]
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            cm, sd, ls = na.events_prepare(ev["mean"][keep], ev["stdv"][keep], ev["start"][keep], 0.0)
            out = {}
            for form in ("auto", "wide"):
                ctx.set_sweep(form)
                b = ctx.sweep_stats()
                out[form] = ctx.viterbi(off, cm, sd, ls)
                a = ctx.sweep_stats()
                out[form + "_launches"] = (a[0] - b[0], a[1] - b[1])
            assert np.array_equal(out["auto"][0], out["wide"][0]) and out["auto"][1].tobytes() == out["wide"][1].tobytes()
            return out["auto_launches"]

        a0 = ctx.ahead_stats()
        assert run([700]) == (0, 1)                                   # one strand ...
        assert ctx.ahead_stats()[0] == a0[0] + 1                      # ... with its emissions computed ahead by the idle CUs
        assert run([300] * n_cu) == (0, 1)                            # one read per CU
        w, l = run([200] * (2 * slots))                               # equal reads, slots filled twice
        assert l == 0 and w >= 1
        w, l = run([150] * 600 + [20000])                             # one read longer than the rest of the batch's share
        assert w == 0 and l >= 1
    finally:
        ctx.close()


def test_device_pointer_form_picks_the_form_from_the_stated_shape(r73t):
    """nchmm_viterbi_dev has no lengths on the host: the form follows from (reads, longest, total) as the caller states them."""
    torch = pytest.importorskip("torch")
    ctx = na.Context(0)
    try:
        ctx.put_model(0, na.scaled_model_table(r73t, IDENT))
        ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        dev = torch.device("cuda:0")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        res = {}
        for n_reads in (8, 3 * ctx.grid_slots()):
            ev = synth.generate(r73t, n_reads, 120)
            off, mean, stdv, start = synth.flat_batch(ev)
            cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
            d_off, d_cm, d_sd, d_ls = t(off.astype(np.int64)), t(cm), t(sd), t(ls)
            for form in ("auto", "wide"):
                ctx.set_sweep(form)
                d_state = torch.zeros(int(off[-1]), dtype=torch.int16, device=dev)
                d_logp = torch.zeros(n_reads, dtype=torch.float32, device=dev)
                d_status = torch.zeros(n_reads, dtype=torch.int32, device=dev)
                b = ctx.sweep_stats()
                ctx.viterbi_dev(n_reads, 120, int(off[-1]), d_off, d_cm, d_sd, d_ls, d_state, d_logp, d_status)
                torch.cuda.synchronize()
                a = ctx.sweep_stats()
                res[(n_reads, form)] = (d_state.cpu().numpy().tobytes(), d_logp.cpu().numpy().tobytes(), (a[0] - b[0], a[1] - b[1]))
            assert res[(n_reads, "auto")][:2] == res[(n_reads, "wide")][:2]
        assert res[(8, "auto")][2] == (0, 1) and res[(3 * ctx.grid_slots(), "auto")][2] == (1, 0)
    finally:
        ctx.use_own_stream()
        ctx.close()


def test_device_pointer_form_plans_on_the_device(r73t):
    """nchmm_viterbi_dev with the offsets only on the device (plan_kernel.hip): (a) a ragged batch is handed out longest first
    without anything being waited for; (b) one 200 000-event read among 2 000 short ones -- too long for a full pool of
    back-pointer regions within the budget -- gets a region of its own beside the pooled launch instead of putting the whole
    batch on as many blocks as the budget has regions of that length for (DESIGN.md section 10 of round 4: 216 of 512).
    Same bits as the host-pointer form, which plans on the host; (b) within 10 % of its time."""
    import time
    torch = pytest.importorskip("torch")
    ctx = na.Context(0)
    try:
        ctx.put_model(0, na.scaled_model_table(r73t, IDENT))
        ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        dev = torch.device("cuda:0")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        rng = np.random.default_rng(5)
        cases = {"ragged": np.clip(np.round(np.exp(rng.normal(np.log(900), 0.8, 700))), 50, 9000).astype(np.int64),
                 "one long read": np.concatenate([rng.integers(200, 400, 1000), [200000], rng.integers(200, 400, 1000)]).astype(np.int64)}
        for name, lens in cases.items():
            n_reads, longest = len(lens), int(lens.max())
            ev = synth.generate(r73t, 1, longest)            # one long stream, cut into the reads (any events do: both forms see the same)
            pick = np.concatenate([np.arange(n) + (int(i) * 7919) % max(1, longest - n) for i, n in enumerate(lens)])
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            cm, sd, ls = na.events_prepare(ev["mean"][0][pick], ev["stdv"][0][pick], ev["start"][0][pick], 0.0)
            ctx.use_own_stream()
            ctx.viterbi(off, cm, sd, ls)                         # (sizes the host form's staging)
            t0 = time.perf_counter()
            h_state, h_logp, h_status = ctx.viterbi(off, cm, sd, ls)
            t_host = time.perf_counter() - t0
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            d_off, d_cm, d_sd, d_ls = t(off.astype(np.int64)), t(cm), t(sd), t(ls)
            d_state = torch.zeros(int(off[-1]), dtype=torch.int16, device=dev)
            d_logp = torch.zeros(n_reads, dtype=torch.float32, device=dev)
            d_status = torch.ones(n_reads, dtype=torch.int32, device=dev)
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.viterbi_dev(n_reads, longest, int(off[-1]), d_off, d_cm, d_sd, d_ls, d_state, d_logp, d_status)
                torch.cuda.synchronize()
                t_dev = time.perf_counter() - t0
            assert np.array_equal(d_state.cpu().numpy().view(np.uint16), h_state), name
            assert d_logp.cpu().numpy().tobytes() == h_logp.tobytes() and (d_status.cpu().numpy() == 0).all(), name
            if name == "one long read":
                assert t_dev <= 1.10 * t_host + 0.005, (t_dev, t_host)
    finally:
        ctx.use_own_stream()
        ctx.close()
