"""Shared helpers for the parity tests (oracle side + input construction)."""
import numpy as np

import nanocall_amd as na
from nanocall_amd import synth
import nc_oracle as oracle

IDENT = (1.0, 0.0, 0.0, 1.0, 1.0, 1.0)


def ragged_batch(table, lens, first_read=0, drift=0.0):
    """Synthetic ragged batch -> (off, mean, stdv, start, cmean, stdv', log_stdv)."""
    lens = list(lens)
    ev = synth.generate(table, len(lens), max(max(lens), 1), first_read=first_read)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    cat = lambda k: np.concatenate([ev[k][r, :n] for r, n in enumerate(lens)]) if sum(lens) else np.zeros(0, np.float32)
    mean, stdv, start = cat("mean"), cat("stdv"), cat("start")
    cm, sd, ls = na.events_prepare(mean, stdv, start, drift)
    return off, mean, stdv, start, cm, sd, ls


def oracle_viterbi_batch(table, params, p_skip, p_stay, off, cm, sd, ls):
    om = oracle.Model(table, params)
    ot = oracle.Transitions(p_skip, p_stay)
    states = np.empty(int(off[-1]), np.uint16)
    logp = np.empty(len(off) - 1, np.float32)
    for r in range(len(off) - 1):
        a, b = int(off[r]), int(off[r + 1])
        if b == a:
            logp[r] = np.nan
            continue
        s, mv, lp = oracle.viterbi(om, ot, cm[a:b], sd[a:b], ls[a:b])
        states[a:b] = s
        logp[r] = lp
    return states, logp


def assert_bits_equal(a, b, what=""):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    same = a.view(np.uint32) == b.view(np.uint32)
    assert same.all(), f"{what}: {np.count_nonzero(~same)} of {a.size} floats differ bitwise; first {a[~same][:3]} vs {b[~same][:3]}"


def rescore_path(table, params, p_skip, p_stay, cm, sd, ls, states):
    """Recompute the Viterbi score ALONG a given state path with the reference's float operations
    (alpha_0 = e_0 - log 4096; alpha_i = (w + alpha_{i-1}) + e_i, Viterbi.hpp:59,82,90).  O(n): a
    full-size check that needs no CPU DP.  Returns (float32 score, all_arcs_valid)."""
    om = oracle.Model(table, params)
    rp, idx, w = oracle.Transitions(p_skip, p_stay).from_csr()
    a = np.float32(om.emission(int(states[0]), cm[0], sd[0], ls[0])) - np.float32(np.log(np.float32(4096.0)))
    a = np.float32(a)
    ok = True
    for i in range(1, len(states)):
        j, p = int(states[i]), int(states[i - 1])
        row = idx[rp[j]:rp[j + 1]]
        k = np.searchsorted(row, p)
        if k >= len(row) or row[k] != p:
            ok = False
            break
        a = np.float32(np.float32(w[rp[j] + k]) + a)
        a = np.float32(a + np.float32(om.emission(j, cm[i], sd[i], ls[i])))
    return a, ok
