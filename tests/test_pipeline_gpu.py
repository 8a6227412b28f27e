"""The host-pointer pipeline (nchmm_pipeline.cpp): a batch cut into read ranges over a copy-in stream and two compute lanes
(consecutive ranges overlap: the blocks of one start where the blocks of the other run out of reads) must decode exactly
what one launch decodes -- i.e. what the oracle decodes (Viterbi.hpp:44-142)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import nanocall_amd as na
from helpers import IDENT, ragged_batch, oracle_viterbi_batch, assert_bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["NC_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NC_ROOT"], "tests")); sys.path.insert(0, os.path.join(os.environ["NC_ROOT"], "oracle"))
import nanocall_amd as na
from helpers import IDENT, ragged_batch, oracle_viterbi_batch
t = na.builtin_model("r73.t")
lens = [300, 17, 0, 256, 1, 511, 90, 700, 33, 257, 64, 5, 400, 128]
off, mean, stdv, start, cm, sd, ls = ragged_batch(t, lens, first_read=21)
with na.Context(0) as ctx:
    ctx.put_model(0, na.scaled_model_table(t, IDENT)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    st, lp, status = ctx.viterbi(off, cm, sd, ls)
    launches = int(ctx.counters()[3])
    src = off[:-1].astype(np.uint64); ln = np.diff(off.astype(np.int64)).astype(np.uint32)
    st2, lp2, status2 = ctx.viterbi_raw(mean, stdv, start, src, ln, np.zeros(len(lens), np.float32))
ost, olp = oracle_viterbi_batch(t, IDENT, 0.3, 0.1, off, cm, sd, ls)
nz = np.diff(off.astype(np.int64)) > 0
print(json.dumps({"launches": launches, "states_equal": bool(np.array_equal(st, ost)), "logp_equal": bool(lp[nz].tobytes() == olp[nz].tobytes()),
                  "status_ok": bool((status == 0).all()), "raw_equal": bool(np.array_equal(st2, ost) and lp2[nz].tobytes() == olp[nz].tobytes()),
                  "empty_nan": bool(np.isnan(lp[~nz]).all())}))
"""


@pytest.mark.parametrize("reads_per_range", [1, 3, 5])
def test_forced_ranges_match_the_oracle(reads_per_range):
    """NCHMM_PIPE_READS (test hook) cuts 14 ragged reads -- one of them empty -- into 14 / 5 / 3 ranges: every range
    boundary, both lanes, the per-range LPT order, the prepared AND the raw (device gather) form, against the oracle."""
    env = dict(os.environ, NCHMM_PIPE_READS=str(reads_per_range), NC_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-c", _CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["launches"] >= 3, out
    assert out["states_equal"] and out["logp_equal"] and out["status_ok"] and out["raw_equal"] and out["empty_nan"], out


def test_three_batches_in_flight_equal_one_call_each(gpu_ctx, r73t):
    """begin(0); begin(1); begin(2); end(0); begin(3); end(1); end(2); end(3) -- a streaming caller that keeps every lane busy --
    returns for every batch what the one-call form returns, which is what the oracle returns.  The batches are ragged (a launch
    lasts as long as its longest read: the short batches overtake the long one on the other lanes)."""
    gpu_ctx.put_model(0, na.scaled_model_table(r73t, IDENT))
    gpu_ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    batches = [ragged_batch(r73t, lens, first_read=fr) for lens, fr in (([200, 31, 1400], 1), ([64, 0, 129, 500, 7], 9), ([350], 30), ([90, 600, 2], 44))]
    t0 = gpu_ctx.viterbi_begin(batches[0][0], *batches[0][4:])
    t1 = gpu_ctx.viterbi_begin(batches[1][0], *batches[1][4:])
    t2 = gpu_ctx.viterbi_begin(batches[2][0], *batches[2][4:])
    assert gpu_ctx.viterbi_in_flight() == 3
    with pytest.raises(na.NchmmError):       # a fourth one has no lane
        gpu_ctx.viterbi_begin(batches[3][0], *batches[3][4:])
    with pytest.raises(na.NchmmError):       # and the one-call form would have to jump the queue
        gpu_ctx.viterbi(batches[3][0], *batches[3][4:])
    # the device-pointer forms share the lanes with the batches in flight: refused too (the pointers are never looked at)
    from nanocall_amd._lib import lib
    fake = 4096
    assert lib().nchmm_viterbi_dev_enqueue(gpu_ctx._h, 1, 10, 10, fake, fake, fake, fake, None, None, None, fake, fake, None) == -1
    r0 = gpu_ctx.viterbi_end(t0)
    t3 = gpu_ctx.viterbi_begin(batches[3][0], *batches[3][4:])
    r1 = gpu_ctx.viterbi_end(t1)
    r2 = gpu_ctx.viterbi_end(t2)
    r3 = gpu_ctx.viterbi_end(t3)
    assert gpu_ctx.viterbi_in_flight() == 0
    for (off, mean, stdv, start, cm, sd, ls), (st, lp, status) in zip(batches, (r0, r1, r2, r3)):
        ost, olp = oracle_viterbi_batch(r73t, IDENT, 0.3, 0.1, off, cm, sd, ls)
        nz = np.diff(off.astype(np.int64)) > 0
        assert np.array_equal(st, ost)
        assert_bits_equal(lp[nz], olp[nz], "path probability")
        assert (status == 0).all()
    with pytest.raises(na.NchmmError):       # nothing left to end
        gpu_ctx.viterbi_end(t3)
    # and the context still serves the one-call form
    st, lp, status = gpu_ctx.viterbi(batches[0][0], *batches[0][4:])
    assert np.array_equal(st, r0[0])


def test_overlapping_device_launches_match_the_oracle(r73t):
    """nchmm_viterbi_dev_enqueue x 5 on the three lanes, then one join: the launches roll into each other (blocks of launch k+1
    start where blocks of launch k run out of reads, taking over their back-pointer regions), each into its own outputs.
    Every batch must decode what the oracle decodes; the regions are sized by the first (longest) batch and reused."""
    import torch
    dev = torch.device("cuda", 0)
    ctx = na.Context(0)
    try:
        ctx.put_model(0, na.scaled_model_table(r73t, IDENT))
        ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        specs = [([700, 40, 300, 129, 511], 3), ([64, 0, 257, 90], 11), ([400] * 6, 17), ([1, 2, 3, 600], 29), ([513, 128, 33], 41)]
        jobs = []
        for lens, fr in specs:
            off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, lens, first_read=fr)
            d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (off.astype(np.int64), cm, sd, ls)]
            outs = (torch.empty(int(off[-1]), dtype=torch.int16, device=dev), torch.full((len(lens),), 7.0, dtype=torch.float32, device=dev),
                    torch.full((len(lens),), 99, dtype=torch.int32, device=dev))
            jobs.append((lens, off, cm, sd, ls, d, outs))
        torch.cuda.synchronize()
        for lens, off, cm, sd, ls, d, outs in jobs:
            ctx.viterbi_dev_enqueue(len(lens), max(lens), int(off[-1]), *d, *outs)
        ctx.viterbi_dev_join()
        got = [(o[0].cpu().numpy().view(np.uint16), o[1].cpu().numpy(), o[2].cpu().numpy()) for *_, o in jobs]   # (torch's stream: after the join)
        ctx.synchronize()
        for (lens, off, cm, sd, ls, d, outs), (st, lp, status) in zip(jobs, got):
            ost, olp = oracle_viterbi_batch(r73t, IDENT, 0.3, 0.1, off, cm, sd, ls)
            nz = np.asarray(lens) > 0
            assert np.array_equal(st, ost) and lp[nz].tobytes() == olp[nz].tobytes() and np.isnan(lp[~nz]).all() and (status == 0).all()
    finally:
        ctx.close()


def test_lane_and_region_soak_small():
    """tools/soak_lanes.py at a size that takes seconds: ragged batches decoded in random order with 1-3 batches in flight and
    through nchmm_viterbi_dev_enqueue, every result bit-identical to the one-call result (and, on short reads, to the oracle)."""
    env = dict(os.environ, ITER="4", BATCHES="5", MAXREADS="900", MAXLEN="4000", ORACLE="3")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_lanes.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    import json
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["mismatching_batches"] == 0 and out["batch_decodes"] == 40 and out["oracle_checked_reads"] >= 12, out


_POISON_CHILD = r"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ["NC_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NC_ROOT"], "tests")); sys.path.insert(0, os.path.join(os.environ["NC_ROOT"], "oracle"))
import nanocall_amd as na
from helpers import IDENT, ragged_batch
t = na.builtin_model("r73.t")
off, mean, stdv, start, cm, sd, ls = ragged_batch(t, [300, 200, 100], first_read=5)
out = {}
with na.Context(0) as ctx:
    ctx.put_model(0, na.scaled_model_table(t, IDENT)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    t0 = time.time()
    out["codes"] = []
    for attempt in range(2):          # the second launch must find the lane's ticket count where the host expects it
        try:
            ctx.viterbi(off, cm, sd, ls)
            out["codes"].append(0)
        except na.NchmmError as e:
            out["codes"].append(e.code)
    out["seconds"] = round(time.time() - t0, 2)
print(json.dumps(out))
"""


def test_a_pool_without_free_regions_fails_loudly_and_in_bounded_time():
    """A block that finds no back-pointer region (cannot happen with a sane pool: the test hook marks every region taken, as a
    kernel that died holding them would) gives up after its bounded wait and reports through pinned host memory: the call
    returns an error within seconds -- it neither hangs the device nor hands back whatever the output arrays held."""
    env = dict(os.environ, NCHMM_TEST_POISON_POOL="1", NC_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-c", _POISON_CHILD], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["codes"] == [-3, -3] and out["seconds"] < 30, out      # NCHMM_E_HIP, both times


def test_one_strand_per_call_from_many_threads_is_combined_and_exact(r73t, r9t):
    """nchmm_viterbi_strand: the reference's call shape (basecall_strand: one strand per call with its own scaled model and
    transitions, from every pfor worker at once, nanocall.cpp:611-621,645-690).  40 threads decode 120 strands of 5 different
    (model, scaling, transition) kinds through ONE context; every strand must equal the oracle's decode, and the launches must
    have been shared (fewer launches than strands)."""
    from concurrent.futures import ThreadPoolExecutor
    import nc_oracle as oracle
    kinds = [(r73t, IDENT, 0.3, 0.1), (r73t, (1.04, 1.5, 0.0, 1.1, 0.95, 1.2), 0.25, 0.12), (r9t, IDENT, 0.3, 0.1),
             (r9t, (0.97, -2.0, 0.0, 0.9, 1.05, 0.8), 0.33, 0.08), (r73t, (1.0, 0.5, 0.0, 1.3, 1.0, 1.0), 0.17, 0.2)]
    tables = [na.scaled_model_table(t, p) for t, p, _, _ in kinds]
    rng = np.random.default_rng(5)
    jobs = []
    for r in range(120):
        k = r % len(kinds)
        n = int(rng.integers(1, 420))
        off, mean, stdv, start, cm, sd, ls = ragged_batch(kinds[k][0], [n], first_read=1000 + r)
        jobs.append((k, cm, sd, ls))
    unscaled = [na.model_load(t) for t, _, _, _ in kinds]

    def call(ctx, i, j):
        # every other strand hands its model over as (unscaled table, parameters) -- scaled on the device -- instead of as the scaled
        # table: both kinds sit side by side in the batches
        k = j[0]
        if i % 2:
            return ctx.viterbi_strand_scaled(unscaled[k], kinds[k][1], kinds[k][2], kinds[k][3], j[1], j[2], j[3])
        return ctx.viterbi_strand(tables[k], kinds[k][2], kinds[k][3], j[1], j[2], j[3])

    with na.Context(0) as ctx:
        launches0 = int(ctx.counters()[3])
        with ThreadPoolExecutor(40) as ex:
            got = list(ex.map(lambda ij: call(ctx, ij[0], ij[1]), enumerate(jobs)))
        launches = int(ctx.counters()[3]) - launches0
        # a lone caller gets a launch to itself
        st1, lp1, rc1 = ctx.viterbi_strand(tables[0], 0.3, 0.1, *jobs[0][1:])
    assert launches < len(jobs) / 2, launches
    oms = [oracle.Model(t, p) for t, p, _, _ in kinds]
    ots = [oracle.Transitions(ps, pt) for _, _, ps, pt in kinds]
    for (k, cm, sd, ls), (st, lp, rc) in zip(jobs, got):
        s, mv, olp = oracle.viterbi(oms[k], ots[k], cm, sd, ls)
        assert rc == 0 and np.array_equal(st, s) and np.float32(lp).tobytes() == np.float32(olp).tobytes()
    assert rc1 == 0 and np.array_equal(st1, got[0][0]) and lp1 == got[0][1]


def test_training_windows_per_call_from_many_threads_equal_one_batched_call(r73t):
    """nchmm_fwbw_windows: train_one_round's shape (one read per call: its two models scaled by the read's current parameters,
    its four windows, from every pfor worker, Parameter_Trainer.hpp:541-579 inside nanocall.cpp:282-579).  32 threads run 96
    reads through ONE context; every read's log-likelihoods and EM sums must be bit-identical to what nchmm_fwbw gives for the
    same windows in a batch of their own over models scaled on the HOST, and the calls must have shared launches."""
    from concurrent.futures import ThreadPoolExecutor
    c_tab = na.builtin_model("r73.c.p1")
    unscaled = [na.model_load(r73t), na.model_load(c_tab)]
    rng = np.random.default_rng(11)
    reads = []
    for r in range(96):
        pm = (float(rng.uniform(0.95, 1.05)), float(rng.uniform(-2, 2)), 0.0, float(rng.uniform(0.9, 1.2)), float(rng.uniform(0.9, 1.1)), float(rng.uniform(0.8, 1.3)))
        st = [(float(rng.uniform(0.06, 0.15)), float(rng.uniform(0.2, 0.35))) for _ in range(2)]        # (p_stay, p_skip) per strand
        lens = [int(rng.integers(60, 101)) for _ in range(4)]
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, lens, first_read=3000 + r)
        wm = np.array([0, 0, 1, 1], np.int32)
        stp = np.array([st[m] for m in wm], np.float32)
        reads.append(dict(p_skip=[st[0][1], st[1][1]], p_stay=[st[0][0], st[1][0]], off=off, cm=cm, sd=sd, ls=ls, wm=wm, pm=pm, stp=stp))
    with na.Context(0) as ctx:
        launches0 = int(ctx.counters()[3])
        with ThreadPoolExecutor(32) as ex:
            got = list(ex.map(lambda R: ctx.fwbw_windows(unscaled, R["pm"], R["p_skip"], R["p_stay"], R["off"], R["cm"], R["sd"], R["ls"], R["wm"], R["stp"]), reads))
        launches = int(ctx.counters()[3]) - launches0
    assert launches < len(reads) / 2, launches
    with na.Context(0) as ref:
        for R, G in zip(reads, got):
            for m, tab in enumerate((r73t, c_tab)):
                ref.put_model(m, na.scaled_model_table(tab, R["pm"]))
                ref.put_transitions(m, *na.transitions_fast(R["p_skip"][m], R["p_stay"][m]))
            want = ref.fwbw(R["off"], R["cm"], R["sd"], R["ls"], scaled_slot=R["wm"], pm_params=R["pm"], trans_slot=R["wm"], st_params=R["stp"])
            for k in ("log_pr_data", "pm_sums", "st_sums"):
                assert G[k].tobytes() == want[k].tobytes(), k


def test_combined_calls_edge_shapes(r73t):
    """nchmm_viterbi_strand on an empty and a one-event strand, nchmm_fwbw_windows on a one-event window and on a call with an
    empty window between two others: the same answers as the batched entry points."""
    tab = na.scaled_model_table(r73t, IDENT)
    with na.Context(0) as ctx:
        ctx.put_model(0, tab)
        ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        z = np.zeros(0, np.float32)
        s, lp, rc = ctx.viterbi_strand(tab, 0.3, 0.1, z, z, z)
        assert rc == 0 and np.isnan(lp) and len(s) == 0
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, [1], first_read=3)
        s, lp, rc = ctx.viterbi_strand(tab, 0.3, 0.1, cm, sd, ls)
        ost, olp = oracle_viterbi_batch(r73t, IDENT, 0.3, 0.1, off, cm, sd, ls)
        assert rc == 0 and np.array_equal(s, ost) and np.float32(lp).tobytes() == olp[0].tobytes()
        un = [na.model_load(r73t)]
        for lens in ([1], [5, 0, 7]):
            off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, lens, first_read=9)
            zero = np.zeros(len(lens), np.int32)
            g = ctx.fwbw_windows(un, IDENT, [0.3], [0.1], off, cm, sd, ls, zero)
            w = ctx.fwbw(off, cm, sd, ls, scaled_slot=zero, pm_params=IDENT, trans_slot=zero, st_params=np.tile(np.float32([0.1, 0.3]), (len(lens), 1)))
            for k in ("log_pr_data", "pm_sums", "st_sums"):
                same = (g[k].view(np.uint32) == w[k].view(np.uint32)) | (np.isnan(g[k]) & np.isnan(w[k]))
                assert same.all(), (lens, k)


_OUTLIER_CHILD = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["NC_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NC_ROOT"], "tests")); sys.path.insert(0, os.path.join(os.environ["NC_ROOT"], "oracle"))
import nanocall_amd as na
from helpers import IDENT, ragged_batch, oracle_viterbi_batch
t = na.builtin_model("r73.t")
rng = np.random.default_rng(3)
lens = [int(x) for x in rng.integers(40, 400, 60)]
lens[7] = 5200; lens[41] = 3900; lens[59] = 0          # two outliers (and an empty read) among short reads
off, mean, stdv, start, cm, sd, ls = ragged_batch(t, lens, first_read=77)
with na.Context(0) as ctx:
    ctx.put_model(0, na.scaled_model_table(t, IDENT)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    st, lp, status = ctx.viterbi(off, cm, sd, ls)
    launches = int(ctx.counters()[3])
    src = off[:-1].astype(np.uint64); ln = np.diff(off.astype(np.int64)).astype(np.uint32)
    st2, lp2, status2 = ctx.viterbi_raw(mean, stdv, start, src, ln, np.zeros(len(lens), np.float32))
    t1 = ctx.viterbi_begin(off, cm, sd, ls); t2 = ctx.viterbi_begin(off, cm, sd, ls)      # two batches with outliers in flight
    r1 = ctx.viterbi_end(t1); r2 = ctx.viterbi_end(t2)
    peak = ctx.mem_stats()[1]
ost, olp = oracle_viterbi_batch(t, IDENT, 0.3, 0.1, off, cm, sd, ls)
nz = np.diff(off.astype(np.int64)) > 0
same = lambda s, l: bool(np.array_equal(s, ost) and l[nz].tobytes() == olp[nz].tobytes())
print(json.dumps({"launches": launches, "one_call": same(st, lp), "raw": same(st2, lp2), "streamed": same(*r1[:2]) and same(*r2[:2]),
                  "status_ok": bool((status == 0).all() and (status2 == 0).all()), "peak_mb": peak >> 20}))
"""


def test_a_few_very_long_reads_get_regions_of_their_own():
    """Under a workspace budget in which a full pool of regions cannot hold the longest read (2 GB: 576 regions of ~590 events),
    the two reads that are longer go through regions of their own as one more launch beside the pooled one, instead of putting the
    whole batch on a handful of blocks: same decode as the oracle through the one-call, the raw and the streaming forms, and the
    workspace stays inside the budget."""
    env = dict(os.environ, NCHMM_WS_BUDGET_MB="2000", NC_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-c", _OUTLIER_CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["launches"] == 2 and out["one_call"] and out["raw"] and out["streamed"] and out["status_ok"], out
    assert out["peak_mb"] < 2300, out


def test_strands_and_training_windows_at_once_on_one_context(r73t):
    """nchmm_viterbi_strand and nchmm_fwbw_windows in progress on ONE context at the same time (a host that trains some reads while
    it decodes others): their batches take turns on the device; every result as if it had been alone."""
    from concurrent.futures import ThreadPoolExecutor
    import nc_oracle as oracle
    tab = na.scaled_model_table(r73t, IDENT)
    un = [na.model_load(r73t)]
    strands, wins = [], []
    for r in range(40):
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, [150 + 7 * r], first_read=500 + r)
        strands.append((cm, sd, ls))
        off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, [80, 100], first_read=900 + r)
        wins.append((off, cm, sd, ls))
    with na.Context(0) as ctx:
        with ThreadPoolExecutor(24) as ex:
            fs = [ex.submit(ctx.viterbi_strand, tab, 0.3, 0.1, *s) for s in strands]
            fw = [ex.submit(ctx.fwbw_windows, un, IDENT, [0.3], [0.1], w[0], w[1], w[2], w[3], np.zeros(2, np.int32)) for w in wins]
            got_s = [f.result() for f in fs]
            got_w = [f.result() for f in fw]
        ctx.put_model(0, tab)
        ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        for w, g in zip(wins, got_w):
            want = ctx.fwbw(w[0], w[1], w[2], w[3], scaled_slot=np.zeros(2, np.int32), pm_params=IDENT, trans_slot=np.zeros(2, np.int32),
                            st_params=np.tile(np.float32([0.1, 0.3]), (2, 1)))
            for k in ("log_pr_data", "pm_sums", "st_sums"):
                assert g[k].tobytes() == want[k].tobytes(), k
    om, ot = oracle.Model(r73t, IDENT), oracle.Transitions(0.3, 0.1)
    for (cm, sd, ls), (st, lp, rc) in zip(strands, got_s):
        s, mv, olp = oracle.viterbi(om, ot, cm, sd, ls)
        assert rc == 0 and np.array_equal(st, s) and np.float32(lp).tobytes() == np.float32(olp).tobytes()


def test_cross_process_counter_reduction_on_a_communicator_of_one():
    """nchmm_rccl_unique_id + nchmm_counters_allreduce (what the worker processes of `nanocall --gpus N` call, one rank per process):
    ncclGetUniqueId, ncclCommInitRank and one ncclAllReduce(sum) of eight uint64 -- with one rank the sums are the inputs.  Invalid
    arguments are refused before anything is initialised."""
    from nanocall_amd import api
    from nanocall_amd._lib import lib
    uid = api.rccl_unique_id()
    assert uid.shape == (128,) and uid.any()
    mine = np.array([3, 1 << 40, 0, 7, 11, 13, 17, (1 << 63) + 5], np.uint64)
    assert np.array_equal(api.counters_allreduce(0, 1, 0, uid, mine), mine)
    assert lib().nchmm_counters_allreduce(0, 2, 2, uid.ctypes.data, mine.ctypes.data) == -1          # rank outside [0, n_ranks)
    assert lib().nchmm_counters_allreduce(-1, 1, 0, uid.ctypes.data, mine.ctypes.data) == -1
    assert lib().nchmm_rccl_unique_id(None) == -1
