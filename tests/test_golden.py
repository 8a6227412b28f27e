"""Committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the
oracle; "parity unpinned" w.r.t. the reference -- see that script's header) against (a) the oracle
as it is now and (b) the product's host prep.  No GPU."""
import glob
import os

import numpy as np
import pytest

import nanocall_amd as na
import nc_oracle as oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VIT = sorted(glob.glob(os.path.join(G, "viterbi_*.npz")))


@pytest.mark.parametrize("path", VIT, ids=[os.path.basename(p)[8:-4] for p in VIT])
def test_viterbi_fixture_reproduced_by_oracle(path):
    z = np.load(path)
    n = len(z["mean"])
    if n > 2100:
        pytest.skip("kept for the GPU suite (CPU oracle needs >0.5 s and 100 MB here)")
    table = na.builtin_model(str(z["model"]))
    params = z["params"]
    om = oracle.Model(table, params)
    ot = oracle.Transitions(float(z["p_skip"]), float(z["p_stay"]))
    cm, sd, ls = na.events_prepare(z["mean"], z["stdv"], z["start"], float(params[2]))   # product host prep
    st, mv, lp = oracle.viterbi(om, ot, cm, sd, ls)
    assert np.array_equal(st, z["states"]) and np.array_equal(mv, z["moves"])
    assert np.float32(lp).view(np.uint32) == z["path_logp_bits"]
    mv2, seq = na.base_seq(st)
    assert np.array_equal(mv2, z["moves"]) and seq == str(z["seq"])
    assert na.write_fasta(os.path.basename(path)[8:-4] + ":synthetic:0", seq, 80) == str(z["fasta"])


@pytest.mark.parametrize("idx", [0, 3])
def test_scaled_model_fixture(idx):
    z = np.load(os.path.join(G, f"scaled_model_{idx}.npz"))
    st = na.model_scale(na.model_load(na.builtin_model(str(z["model"]))), z["params"])
    assert np.array_equal(st[:16].view(np.uint32), z["head"].view(np.uint32))
    assert np.array_equal(st[-16:].view(np.uint32), z["tail"].view(np.uint32))
    assert int(st.view(np.uint32).astype(np.uint64).sum()) == int(z["sum_bits"])


def test_fwbw_fixture_reproduced_by_oracle():
    z = np.load(os.path.join(G, "fwbw_r73t_2x100.npz"))
    table = na.builtin_model("r73.t")
    om, ot = oracle.Model(table, (1, 0, 0, 1, 1, 1)), oracle.Transitions(0.3, 0.1)
    w = 0   # one window keeps the CPU suite short; the GPU suite checks both
    cm, sd, ls = na.events_prepare(z[f"w{w}_mean"], z[f"w{w}_stdv"], z[f"w{w}_start"], 0.0)
    lpd, al, be = oracle.fwbw(om, ot, cm, sd, ls)
    assert abs(lpd - z[f"w{w}_log_pr_data"]) <= 1e-4 * abs(z[f"w{w}_log_pr_data"])
    for i, j, a, b in z[f"w{w}_probe_cells"]:
        assert abs(al[int(i), int(j)] - a) <= 1e-4 * abs(a) and abs(be[int(i), int(j)] - b) <= 1e-4 * max(abs(b), 1)
    post = al[50] + be[50] - lpd
    assert np.array_equal(np.argsort(-post)[:5].astype(np.int32), z[f"w{w}_top5_states"])


def test_logsumset_restatement_is_a_log_sum_exp():
    rng = np.random.default_rng(0)
    for n in (1, 2, 21, 4096):
        v = (rng.standard_normal(n) * 30 - 500).astype(np.float32)
        exact = np.logaddexp.reduce(v.astype(np.float64))
        assert abs(float(oracle.logsumset(v)) - exact) <= 1e-5 * abs(exact)
    assert oracle.logsumset(np.zeros(0, np.float32)) == -np.inf
    assert oracle.logsumset(np.array([-np.inf, -3.0], np.float32)) == np.float32(-3.0)
