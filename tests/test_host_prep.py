"""The product's host-side prep (C++ in libnanocall_hip.so, no GPU) against the CPU oracle: bit-exact."""
import os

import numpy as np
import pytest

import nanocall_amd as na
from nanocall_amd import synth
import nc_oracle as oracle

PARAM_SETS = [(1.0, 0.0, 0.0, 1.0, 1.0, 1.0), (1.05, 2.5, 0.002, 1.1, 0.9, 1.2), (0.93, -4.0, -0.001, 0.8, 1.3, 0.7)]


@pytest.mark.parametrize("model", range(6))
def test_model_load_and_scale_bit_exact(model):
    t = na.builtin_model(model)
    st = na.model_load(t)
    assert np.array_equal(st.view(np.uint32), oracle.Model(t).states().view(np.uint32))
    for p in PARAM_SETS:
        a = na.model_scale(st, p)
        b = oracle.Model(t, p)
        assert np.array_equal(a.view(np.uint32), b.states().view(np.uint32))
        assert np.array_equal(na.model_pack6(a).view(np.uint32), b.table6().view(np.uint32))
    # scaling twice composes like the reference (logs are added, not recomputed)
    a2 = na.model_scale(na.model_scale(st, PARAM_SETS[1]), PARAM_SETS[2])
    b2 = oracle.Model(t, PARAM_SETS[1]); b2.scale(PARAM_SETS[2])
    assert np.array_equal(a2.view(np.uint32), b2.states().view(np.uint32))


@pytest.mark.parametrize("pp", [(0.3, 0.1), (0.28, 0.09), (0.17, 0.12), (0.05, 0.4), (0.4, 0.05)])
def test_transitions_fast_bit_exact(pp):
    rp, pred, w = na.transitions_fast(*pp)
    rp2, idx2, w2 = oracle.Transitions(*pp).from_csr()
    assert np.array_equal(rp, rp2) and np.array_equal(pred, idx2.astype(np.uint16))
    assert np.array_equal(w.view(np.uint32), w2.view(np.uint32))


def test_events_prepare_bit_exact(r73t):
    ev = synth.generate(r73t, 1, 500, first_read=3)
    mean, stdv, start = ev["mean"][0], ev["stdv"][0].copy(), ev["start"][0]
    stdv[::9] = 0.0
    for drift in (0.0, 0.002, -0.01):
        a = na.events_prepare(mean, stdv, start, drift)
        b = oracle.events_prepare(mean, stdv, start, drift)
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    cm, sd, ls = na.events_prepare(mean, stdv, None, 0.0)
    assert np.array_equal(cm, mean) and (sd[::9] == np.float32(0.01)).all()


def test_empty_and_single_event_inputs():
    cm, sd, ls = na.events_prepare(np.zeros(0, np.float32), np.zeros(0, np.float32))
    assert cm.size == sd.size == ls.size == 0
    mv, seq = na.base_seq(np.zeros(0, np.uint16))
    assert mv.size == 0 and seq == ""
    mv, seq = na.base_seq(np.array([0b000110110001], np.uint16))
    assert mv.tolist() == [0] and seq == "ACGTAC"


def test_base_seq_and_fasta_match_oracle():
    rng = np.random.default_rng(5)
    # a plausible path: stay / step / skip moves plus a few arbitrary jumps (move 6)
    k = int(rng.integers(4096)); path = [k]
    for _ in range(3000):
        u = rng.random()
        if u < 0.1: pass
        elif u < 0.7: k = ((k << 2) | int(rng.integers(4))) & 4095
        elif u < 0.97: k = ((k << 4) | int(rng.integers(16))) & 4095
        else: k = int(rng.integers(4096))
        path.append(k)
    path = np.array(path, np.uint16)
    mv, seq = na.base_seq(path)
    omv = np.array([0] + [oracle.lib().nco_kmer_min_skip(int(a), int(b)) for a, b in zip(path[:-1], path[1:])], np.int32)
    assert np.array_equal(mv, omv)
    assert seq == oracle.base_seq(path, omv)
    for width in (80, 60, 1, 10000):
        assert na.write_fasta("read:file:0", seq, width) == oracle.write_fasta("read:file:0", seq, width)
    assert na.write_fasta("x", "", 80) == ">x\n"


def test_st_train_kmers_match_oracle():
    a, b = na.st_train_kmers(), oracle.st_train_kmers()
    assert np.array_equal(a.astype(np.uint32), b)
    assert 0 < len(a) < 4096


def test_error_codes_not_exceptions():
    from nanocall_amd._lib import lib
    L = lib()
    assert L.nchmm_model_load(None, None) == -1
    assert L.nchmm_transitions_fast(0.3, 0.1, None, None, None, None) == -1
    assert b"invalid" in L.nchmm_strerror(-1)
    assert L.nchmm_abi_version() >= 1


def test_host_abi_under_sanitizers():
    """tools/asan_host.cpp: the batch plan of the host-pointer pipeline (ranges, longest-first order, outliers: nchmm_plan.hpp, 400
    random shapes x forms), the choice between the two forms of the sweep (schedule bounds and the decisions that must hold, 300 shapes), the call combiner with a host stand-in for the device (24 threads, nchmm_combine.hpp), and
    every device-free ABI function (model load/scale/pack, transitions, event prep incl. the
    threaded path, base sequence, FASTA, train finishes) compiled with -fsanitize=address,undefined and run on edge sizes."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "tools"), "asan-host"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -lasan" in (r.stderr + r.stdout):
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "host ABI under ASan/UBSan: ok" in r.stdout and "batch plans: ok" in r.stdout and "combiner:" in r.stdout
    assert "sweep choice: ok" in r.stdout


def test_sweep_plan_holds_on_machines_that_run_at_other_rates(tmp_path):
    """tools/plan_props.cpp: the choice between the forms of the Viterbi sweep (nchmm_plan.hpp: choose_sweep / plan_ahead /
    choose_sweep_bounds) prices them with event rates measured at one clock (2.1 GHz); the boxes of the pool sustain 1.9-2.35 GHz.
    Over 600 random batch shapes x four scale factors x eight groups of rates: a common factor changes nothing; on a machine whose
    compute rates are 0.9-1.1 x the built-in ones the decision costs at most 5 % of the best form's duration (10 % at 0.8 / 1.25),
    and changes only near the break-even; with the clock known to 5 % (rates_at_clock: a context that has measured it, or
    NCHMM_PLAN_CLOCK_MHZ) at most 5 % everywhere; the three-number form of device-pointer callers is within 5 % of the best form it
    can be given on equal-length batches.  Built with UBSan (the same functions run under ASan in test_host_abi_under_sanitizers)."""
    import json
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "plan_props"
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(root, "nanocall_amd", "csrc"),
                        os.path.join(root, "tools", "plan_props.cpp"), "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), "600"], capture_output=True, text=True, timeout=600)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and d["ok"], d
    g = d["groups"]
    assert g["uniform"]["form_changes"] == 0 and g["clock"]["worst_regret_within_10_percent"] <= 0.05 and g["clock_known"]["worst_regret"] <= 0.05
    assert all(v["changes_away_from_break_even"] == 0 for v in g.values())


def test_randomised_scalings_and_transitions_bit_exact():
    """300 random scaling-parameter sets (scale 0.5-2, shift +-30, var / scale_sd / var_sd over 0.1-10: wider than any trained read) on
    every builtin table, applied once and twice, and 200 random (p_skip, p_stay) pairs down to 1e-4 and up to a sum of 0.999: the
    scaled model images and the transition CSRs bit-identical to the oracle's (Pore_Model.hpp:126-138,190-201; State_Transitions.hpp:125-224)."""
    rng = np.random.default_rng(99)
    tabs = [na.builtin_model(m) for m in range(6)]
    loaded = [na.model_load(t) for t in tabs]
    lu = lambda lo, hi: float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
    for k in range(300):
        m = int(rng.integers(6))
        p = (rng.uniform(0.5, 2.0), rng.uniform(-30, 30), rng.uniform(-0.05, 0.05), lu(0.1, 10), lu(0.1, 10), lu(0.1, 10))
        p = tuple(float(np.float32(x)) for x in p)
        a, b = na.model_scale(loaded[m], p), oracle.Model(tabs[m], p)
        assert np.array_equal(a.view(np.uint32), b.states().view(np.uint32)), (k, p)
        assert np.array_equal(na.model_pack6(a).view(np.uint32), b.table6().view(np.uint32)), (k, p)
        assert np.array_equal(na.scaled_model_table(tabs[m], p).view(np.uint32), b.table6().view(np.uint32))
        if k % 10 == 0:
            q = tuple(float(np.float32(x)) for x in (rng.uniform(0.8, 1.2), rng.uniform(-5, 5), 0.0, lu(0.5, 2), lu(0.5, 2), lu(0.5, 2)))
            b.scale(q)
            assert np.array_equal(na.model_scale(a, q).view(np.uint32), b.states().view(np.uint32)), (k, p, q)
    for k in range(200):
        p_stay = lu(1e-4, 0.6)
        p_skip = min(lu(1e-4, 0.6), 0.999 - p_stay)
        rp, pred, w = na.transitions_fast(p_skip, p_stay)
        rp2, idx2, w2 = oracle.Transitions(p_skip, p_stay).from_csr()
        assert np.array_equal(rp, rp2) and np.array_equal(pred, idx2.astype(np.uint16)), (k, p_skip, p_stay)
        assert np.array_equal(w.view(np.uint32), w2.view(np.uint32)), (k, p_skip, p_stay)
