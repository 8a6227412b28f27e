"""Device logf (nchmm_logf: the port of glibc 2.35 logf in nanocall_amd/csrc/nchmm_device.h) against the HOST libm's
logf -- what Event::update_logs calls (Event.hpp:43) -- over EVERY binary32 input, bit for bit.  This is the licence
for computing log_stdv on the device inside the bit-exact Viterbi path (nchmm_viterbi_raw)."""
import concurrent.futures
import os

import numpy as np
import pytest

import nanocall_amd as na
import nc_oracle as oracle
from helpers import IDENT, ragged_batch

pytestmark = pytest.mark.gpu


def test_logf_all_binary32_inputs_bit_identical_to_host_libm(gpu_ctx):
    chunk = 1 << 26
    threads = max(1, min(32, (os.cpu_count() or 1)))
    total_bad = 0

    def check(lo):
        bits = np.arange(lo, lo + chunk, dtype=np.uint64).astype(np.uint32)
        x = bits.view(np.float32)
        y = gpu_ctx.logf(x)
        # host side in slices on several threads (ctypes releases the GIL)
        sl = np.array_split(np.arange(chunk), threads)
        with concurrent.futures.ThreadPoolExecutor(threads) as ex:
            res = list(ex.map(lambda s: oracle.logf_mismatches(x[s[0]:s[-1] + 1], y[s[0]:s[-1] + 1]), sl))
        bad = sum(r[0] for r in res)
        first = [int(s[0]) + r[1] for s, r in zip(sl, res) if r[0]]
        return bad, (first[0] if first else -1)

    # every non-negative float: +0, subnormals, normals, +inf, NaNs (0x00000000 .. 0x7fffffff): 32 chunks of 2^26
    for lo in range(0, 1 << 31, chunk):
        bad, first = check(lo)
        assert bad == 0, f"{bad} mismatches in [{lo:#x}, {lo + chunk:#x}); first at bits {lo + first:#x}"
        total_bad += bad
    # negative inputs (x < 0 -> NaN, -0 -> -inf, -NaN): one chunk from each end of the negative range and a random sample
    for lo in (0x80000000, 0xfc000000):
        bad, first = check(lo)
        assert bad == 0, (hex(lo), bad, first)
    rng = np.random.default_rng(0)
    x = rng.integers(0x80000000, 1 << 32, size=1 << 22, dtype=np.uint64).astype(np.uint32).view(np.float32)
    assert oracle.logf_mismatches(x, gpu_ctx.logf(x))[0] == 0


def test_viterbi_raw_equals_host_prepared_viterbi(gpu_ctx, r73t):
    """nchmm_viterbi_raw (drift correction, 0 -> .01, log on the device; candidates sharing raw events) gives the bits of
    nchmm_viterbi on host-prepared events (nchmm_events_prepare), which are the oracle's."""
    lens = [700, 1, 0, 4500, 64]
    off, mean, stdv, start, _, _, _ = ragged_batch(r73t, lens, first_read=31)
    stdv = stdv.copy()
    stdv[::11] = 0.0
    params = [(1.0, 0.0, 0.0, 1.0, 1.0, 1.0), (1.03, 1.5, 0.004, 1.1, 0.95, 1.2), (0.98, -2.0, -0.002, 0.9, 1.1, 0.8)]
    for k, p in enumerate(params):
        gpu_ctx.put_model(k, na.scaled_model_table(r73t, p))
        gpu_ctx.put_transitions(k, *na.transitions_fast(0.3 - 0.05 * k, 0.1 + 0.01 * k))
    # candidates: every read with parameter sets 0 and 1, the long read also with set 2 (three candidates share its events)
    src, ln, drift, slot = [], [], [], []
    for r, n in enumerate(lens):
        for k in ((0, 1, 2) if n == 4500 else (0, 1)):
            src.append(int(off[r])); ln.append(n); drift.append(params[k][2]); slot.append(k)
    states, logp, status = gpu_ctx.viterbi_raw(mean, stdv, start, src, ln, drift, model_slot=slot, trans_slot=slot)
    assert (status == 0).all()
    pos = 0
    for s0, n, dr, k in zip(src, ln, drift, slot):
        if n == 0:
            continue
        cm, sd, ls = na.events_prepare(mean[s0:s0 + n], stdv[s0:s0 + n], start[s0:s0 + n], dr)
        om = oracle.Model(r73t, params[k])
        ot = oracle.Transitions(0.3 - 0.05 * k, 0.1 + 0.01 * k)
        es, mv, lp = oracle.viterbi(om, ot, cm, sd, ls)
        assert np.array_equal(states[pos:pos + n], es), (s0, n, k)
        pos += n
    # log-probabilities in candidate order
    v = 0
    for s0, n, dr, k in zip(src, ln, drift, slot):
        if n:
            cm, sd, ls = na.events_prepare(mean[s0:s0 + n], stdv[s0:s0 + n], start[s0:s0 + n], dr)
            hs, hl, _ = gpu_ctx.viterbi(np.array([0, n], np.uint64), cm, sd, ls, model_slot=[k], trans_slot=[k])
            assert hl[0].tobytes() == logp[v].tobytes()
        else:
            assert np.isnan(logp[v])
        v += 1


def test_shader_clock_probe_reports_a_plausible_clock(gpu_ctx):
    # nchmm_shader_clock_mhz: gfx950's valid sclk range is 500-2400 MHz (rocm-smi --showsclkrange on the boxes)
    mhz = gpu_ctx.shader_clock_mhz()
    assert 400.0 <= mhz <= 2600.0, mhz
