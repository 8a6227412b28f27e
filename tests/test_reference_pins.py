"""Pins against the REAL reference (no GPU).

* oracle/_ref = the reference's own Kmer.hpp and Builtin_Model.cpp compiled unmodified (built by
  `make -C oracle ref` in the container that has /root/reference; the prebuilt .so travels).
  When it is absent these tests skip -- they never fall back to comparing the oracle with itself.
* SURVEY.md section 8a-4 records 18 mask->weight values, the arc count and the degree histogram that
  a run of the reference's compute_transitions_fast(.3, .1) produced.
"""
import collections

import numpy as np
import pytest

import nanocall_amd as na
import nc_oracle as oracle

ref = oracle.ref()
needs_ref = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
def test_kmer_algebra_matches_reference_exhaustively_per_state():
    L = oracle.lib()
    import ctypes as C
    buf_o, buf_r = C.create_string_buffer(8), C.create_string_buffer(8)
    nl_o, nl_r = np.zeros(16, np.uint32), np.zeros(16, np.uint32)
    for i in range(4096):
        assert L.nco_kmer_max_self_overlap(i) == ref.ref_kmer_max_self_overlap(i)
        for k in range(1, 7):
            assert L.nco_kmer_prefix(i, k) == ref.ref_kmer_prefix(i, k)
            assert L.nco_kmer_suffix(i, k) == ref.ref_kmer_suffix(i, k)
        L.nco_kmer_to_string(i, buf_o)
        ref.ref_kmer_to_string(i, buf_r)
        assert buf_o.value == buf_r.value
        assert L.nco_kmer_to_int(buf_o.value) == ref.ref_kmer_to_int(buf_r.value) == i
        for d, n in ((1, 4), (2, 16)):
            L.nco_kmer_neighbour_list(i, d, nl_o.ctypes.data)
            assert ref.ref_kmer_neighbour_list(i, d, nl_r.ctypes.data) == n
            assert np.array_equal(nl_o[:n], nl_r[:n])


@needs_ref
def test_min_skip_matches_reference_on_structured_and_random_pairs():
    L = oracle.lib()
    rng = np.random.default_rng(1)
    pairs = [(int(a), int(b)) for a, b in rng.integers(0, 4096, size=(20000, 2))]
    for a in rng.integers(0, 4096, size=300):     # every successor class of a few hundred states
        a = int(a)
        pairs += [(a, a)] + [(a, ((a << (2 * d)) | int(x)) & 4095) for d in range(1, 7) for x in rng.integers(0, 4 ** min(d, 6), 3)]
    for a, b in pairs:
        assert L.nco_kmer_min_skip(a, b) == ref.ref_kmer_min_skip(a, b)
    # and the product's host code agrees on a state path
    path = rng.integers(0, 4096, size=2000).astype(np.uint16)
    mv, _ = na.base_seq(path)
    exp = [0] + [ref.ref_kmer_min_skip(int(x), int(y)) for x, y in zip(path[:-1], path[1:])]
    assert np.array_equal(mv, np.array(exp, np.int32))


@needs_ref
def test_builtin_model_tables_are_bit_identical_to_reference():
    assert ref.ref_builtin_num() == 6
    names, strands = na.builtin_names(), na.builtin_strands()
    for i in range(6):
        assert ref.ref_builtin_name(i).decode() == names[i]
        assert ref.ref_builtin_strand(i) == strands[i]
        assert ref.ref_builtin_size(i) == 4096 * 4
        r = np.ctypeslib.as_array(ref.ref_builtin_table(i), shape=(4096 * 4,)).reshape(4096, 4)
        assert np.array_equal(r.view(np.uint32), na.builtin_model(i).view(np.uint32)), names[i]


# SURVEY.md section 8a-4 [probe]: overlap mask -> log weight for (p_skip, p_stay) = (.3, .1)
SURVEY_WEIGHTS = {0x01: -2.30258298, 0x02: -1.89711869, 0x04: -4.23891115, 0x09: -2.2942965, 0x11: -2.30210304,
                  0x12: -1.89679861, 0x14: -4.23558855, 0x15: -2.16743112, 0x21: -2.30255532, 0x22: -1.89710021,
                  0x24: -4.23871946, 0x29: -2.29426885, 0x2a: -1.8915683, 0x31: -2.30207539, 0x32: -1.89678013,
                  0x3c: -4.1795001, 0x3e: -1.79995608, 0x3f: -1.3268708}


def _mask(i, j):
    m = 1 if i == j else 0
    for l in range(1, 6):
        if (i & ((1 << (2 * (6 - l))) - 1)) == (j >> (2 * l)):
            m |= 1 << l
    return m


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_transitions_match_survey_probe_of_the_reference(who):
    if who == "oracle":
        rp, idx, w = oracle.Transitions(0.3, 0.1).from_csr()
    else:
        rp, idx, w = na.transitions_fast(0.3, 0.1)
    assert len(idx) == 85936
    assert collections.Counter(np.diff(rp.astype(np.int64)).tolist()) == {21: 4068, 20: 12, 17: 12, 16: 4}
    seen = {}
    for j in range(4096):
        row = idx[rp[j]:rp[j + 1]].astype(np.int64)
        assert (np.diff(row) > 0).all()          # ascending predecessors: the tie rule depends on it
        for a in range(rp[j], rp[j + 1]):
            seen.setdefault(_mask(int(idx[a]), j), set()).add(np.float32(w[a]).tobytes())
    assert set(seen) == set(SURVEY_WEIGHTS)
    for m, vals in seen.items():
        assert len(vals) == 1, hex(m)
        v = np.frombuffer(next(iter(vals)), np.float32)[0]
        # SURVEY prints 9 significant digits: that identifies a float32 uniquely
        assert np.float32(SURVEY_WEIGHTS[m]) == v, (hex(m), v)
