"""Bit parity of the Viterbi path where exact float ties are densest: LONG and OUT-OF-MODEL reads, every read against the oracle, in
every form of the sweep.

The bit-exact contract rests on two rare branches of the kernels -- the exact rescan of a predecessor group when a smaller alpha
could round to the same sum as the group's maximum, and the tie rule across the stay / step / skip classes (lowest predecessor
index: Viterbi.hpp:79-89 is an ascending scan with strict >; :125-132 the final arg-max).  Their firing rate rises with |alpha|,
i.e. with the length of a read (alpha falls ~3 per event; at 50 000 events one fp32 ulp is 2^-6) and with how badly the events fit
the model.  The reference admits reads of up to 100 000 events (--max-ed-events, nanocall.cpp:65).  Here: 16 reads of 50 000 and 4
of 100 000 events over r9.t and r73.t with trained-looking scaling parameters and non-default transition probabilities, events of
eight kinds (tests/adversarial.py: model-matched, another model's, uniform levels, constant runs, +-20 sigma spikes, heavy-tailed
stdv, stdv == 0, abasic stretches), plus 200 short reads of the same kinds -- decoded once per form (wide / ll / ahead / the plan's
choice) and once under a workspace budget that sends the long reads through regions of their own (the outlier launch), and EVERY
read compared bit for bit (k-mer path and path log-probability) with oracle.viterbi run on the host's cores.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import nanocall_amd as na
import nc_oracle as oracle
import adversarial

pytestmark = pytest.mark.gpu

MODELS = {"r9.t": (1.07, -3.2, 0.0008, 1.3, 0.92, 1.4), "r73.t": (0.94, 4.1, -0.0005, 0.85, 1.1, 0.8)}     # name -> pm params
TRANS = [(0.22, 0.14), (0.3, 0.1)]         # (p_skip, p_stay): slot 0 non-default, slot 1 the default (nanocall.cpp:80-81)
N_LONG_50K, N_LONG_100K, N_SHORT = 16, 4, 200


def _build_batch():
    names = list(MODELS)
    tables = {nm: na.builtin_model(nm) for nm in names}
    rng = np.random.default_rng(606)
    lens = [50000] * N_LONG_50K + [100000] * N_LONG_100K + [int(x) for x in np.exp(rng.uniform(np.log(1), np.log(4000), N_SHORT))]
    reads = []
    for r, n in enumerate(lens):
        nm = names[r % 2]
        kind = adversarial.KINDS[(r // 2) % len(adversarial.KINDS)]
        other = tables[names[(r + 1) % 2]]
        mean, stdv, start = adversarial.events(kind, tables[nm], MODELS[nm], n, seed=9000 + r, other_table=other)
        cm, sd, ls = na.events_prepare(mean, stdv, start, MODELS[nm][2])
        reads.append(dict(model=r % 2, trans=(r // 3) % 2, kind=kind, n=n, cm=cm, sd=sd, ls=ls))
    order = rng.permutation(len(reads))                 # the long reads anywhere in the batch, not in front
    reads = [reads[i] for i in order]
    off = np.concatenate([[0], np.cumsum([d["n"] for d in reads])]).astype(np.uint64)
    cat = lambda k: np.concatenate([d[k] for d in reads])
    return dict(reads=reads, off=off, cm=cat("cm"), sd=cat("sd"), ls=cat("ls"),
                model_slot=np.array([d["model"] for d in reads], np.int32), trans_slot=np.array([d["trans"] for d in reads], np.int32),
                tables=[tables[nm] for nm in names], params=[MODELS[nm] for nm in names])


@pytest.fixture(scope="module")
def batch():
    return _build_batch()


@pytest.fixture(scope="module")
def oracle_decode(batch):
    """oracle.viterbi of every read, 16 at a time on the host's cores (the reference layout: n x 4096 x 8 B, 3.3 GB for a 100 000-event
    read -- Viterbi.hpp:50)."""
    oms = [oracle.Model(t, p) for t, p in zip(batch["tables"], batch["params"])]
    ots = [oracle.Transitions(*t) for t in TRANS]

    def one(d):
        s, mv, lp = oracle.viterbi(oms[d["model"]], ots[d["trans"]], d["cm"], d["sd"], d["ls"])
        return s, np.float32(lp)

    # longest first, so that the 100 000-event reads do not end the pool's run on their own
    idx = sorted(range(len(batch["reads"])), key=lambda i: -batch["reads"][i]["n"])
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        got = list(ex.map(lambda i: one(batch["reads"][i]), idx))
    out = [None] * len(idx)
    for i, g in zip(idx, got):
        out[i] = g
    return out


def _decode(batch, form, **env):
    for k, v in env.items():           # (the budgets are read when a workspace is first sized, not when the context is created)
        os.environ[k] = v
    ctx = None
    try:
        ctx = na.Context(0)
        ctx.set_sweep(form)
        for s, (t, p) in enumerate(zip(batch["tables"], batch["params"])):
            ctx.put_model(s, na.scaled_model_table(t, p))
        for s, (p_skip, p_stay) in enumerate(TRANS):
            ctx.put_transitions(s, *na.transitions_fast(p_skip, p_stay))
        ctx.profile_ticks()            # (reset)
        launches0 = int(ctx.counters()[3])
        states, logp, status = ctx.viterbi(batch["off"], batch["cm"], batch["sd"], batch["ls"], batch["model_slot"], batch["trans_slot"])
        info = dict(launches=int(ctx.counters()[3]) - launches0, ticks=[int(x) for x in ctx.profile_ticks()[:8]], sweep=list(ctx.sweep_stats()),
                    ahead=list(ctx.ahead_stats()), peak_mb=int(ctx.mem_stats()[1]) >> 20)
    finally:
        for k in env:
            del os.environ[k]
        if ctx is not None:
            ctx.close()
    return states, logp, status, info


def _compare(batch, oracle_decode, states, logp, status, what):
    bad = []
    off = batch["off"]
    for r, (d, (os_, olp)) in enumerate(zip(batch["reads"], oracle_decode)):
        a, b = int(off[r]), int(off[r + 1])
        if status[r] != 0 or not np.array_equal(states[a:b], os_) or np.float32(logp[r]).tobytes() != np.float32(olp).tobytes():
            first = int(np.argmax(states[a:b] != os_)) if not np.array_equal(states[a:b], os_) else -1
            bad.append((r, d["kind"], d["n"], int(status[r]), float(logp[r]), float(olp), first))
    assert not bad, f"{what}: {len(bad)} of {len(batch['reads'])} reads differ from the oracle (read, kind, events, status, logp, oracle logp, first differing event): {bad[:6]}"


def test_the_batch_is_what_the_docstring_says(batch, oracle_decode):
    lens = np.diff(batch["off"].astype(np.int64))
    assert (lens == 50000).sum() == N_LONG_50K and (lens == 100000).sum() == N_LONG_100K and len(lens) == N_LONG_50K + N_LONG_100K + N_SHORT
    kinds_long = {d["kind"] for d in batch["reads"] if d["n"] >= 50000}
    assert kinds_long == set(adversarial.KINDS)
    # the oracle itself is in the regime the test is about: |alpha| in the hundreds of thousands at the end of the long reads
    lp_long = [abs(float(lp)) for d, (_, lp) in zip(batch["reads"], oracle_decode) if d["n"] >= 50000]
    assert min(lp_long) > 1e5 and all(np.isfinite(lp_long)), lp_long


@pytest.mark.parametrize("form", ["wide", "ll", "ahead", "auto"])
def test_long_and_out_of_model_reads_bit_exact_in_every_form(form, batch, oracle_decode):
    # "ahead": room in the emission buffer for every long read (16 KiB per event; the default 256 MiB holds 16 384 events)
    states, logp, status, info = _decode(batch, form, NCHMM_PROFILE="1", NCHMM_EM_BUDGET_MB="24000")
    print(f"long reads, form {form}: {info}")
    _compare(batch, oracle_decode, states, logp, status, f"form {form}")
    # the exactness branches ran: group rescans (ticks[6]) and cells decided by the tie rule (ticks[7])
    assert info["ticks"][6] > 0 and info["ticks"][7] > 0, info
    if form in ("wide", "ll", "ahead"):
        low = form != "wide"
        assert info["sweep"][1 if low else 0] > 0 and info["sweep"][0 if low else 1] == 0, info
    if form == "ahead":
        assert info["ahead"][0] > 0 and info["ahead"][2] >= 100000, info       # at least the longest read went ahead


def test_long_reads_through_regions_of_their_own_under_a_small_workspace_budget(batch, oracle_decode):
    """NCHMM_WS_BUDGET_MB = 32 GiB: a pooled region holds ~9 700 events, the twenty long reads (one in eleven) are outliers and run as
    one more launch on 23 regions of 410 MB beside the pooled launch of the short ones (nchmm_plan.hpp: plan_outliers)."""
    states, logp, status, info = _decode(batch, "auto", NCHMM_PROFILE="1", NCHMM_WS_BUDGET_MB="32768")
    print(f"long reads, 32 GiB workspace budget: {info}")
    _compare(batch, oracle_decode, states, logp, status, "outlier launch")
    assert info["launches"] == 2, info
    assert info["peak_mb"] < 40000, info
