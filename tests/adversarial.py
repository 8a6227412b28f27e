"""Event streams that are NOT draws from the model they are decoded with -- test inputs for the bit-exact Viterbi contract where
exact float ties are dense (Viterbi.hpp:79-89 strict >, :125-132 arg-max; reads up to --max-ed-events 100000, nanocall.cpp:65).

alpha falls by ~3 per event on model-matched events and by tens to hundreds on events the model finds implausible, so on long or
out-of-model reads the fp32 spacing of alpha grows past the transition weights' differences: whole predecessor classes round to
the same sum, and the decode is decided by the reference's tie rule (the lowest predecessor index) -- the branches of the kernels
that the model-matched 5000-event reads of the benchmark hardly ever take.

kinds (each from a seeded numpy Generator + the SURVEY 8d stream of nanocall_amd.synth, so the same on every box):
  matched      the decoding model's own stream, means mapped through the scaling the model is scaled with
  other_model  the stream of ANOTHER builtin table (a strand decoded with the wrong model: nanocall.cpp:692-782 tries them all)
  uniform      means uniform over the scaled level range +- 5, stdv log-uniform over [0.01, 50]
  runs         matched, with constant runs of 50-500 identical events over ~a fifth of the read (a stalled pore)
  spikes       matched, 1 % of the events moved by +-20 level standard deviations
  stdv_tail    matched means, stdv log-uniform over [0.01, 50]
  zeros        matched, 0.5 % of the events with stdv == 0 (Event::update_logs turns them into 0.01, Event.hpp:39-42)
  abasic       matched, with stretches of 100-300 events ~30 above the highest level (what the hairpin / abasic region looks like)
"""
import numpy as np

from nanocall_amd import synth

KINDS = ("matched", "other_model", "uniform", "runs", "spikes", "stdv_tail", "zeros", "abasic")


def _log_uniform(rng, lo, hi, n):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), n)).astype(np.float32)


def events(kind, table, params, n, seed, other_table=None):
    """-> (mean, stdv, start) float32[n]: raw events (before events_prepare) for a strand decoded with `table` scaled by `params`."""
    assert kind in KINDS, kind
    rng = np.random.default_rng([0x6E63, KINDS.index(kind), int(seed)])
    scale, shift = np.float32(params[0]), np.float32(params[1])
    src = other_table if (kind == "other_model" and other_table is not None) else table
    ev = synth.generate(src, 1, n, first_read=int(seed) & 0x7FFFFFFF)
    mean, stdv, start = ev["mean"][0].copy(), ev["stdv"][0].copy(), ev["start"][0].copy()
    if kind != "other_model":
        mean = mean * scale + shift
    t = np.asarray(table, np.float32).reshape(4096, 4)
    lo, hi = float(t[:, 0].min()) * float(scale) + float(shift), float(t[:, 0].max()) * float(scale) + float(shift)
    sigma = float(t[:, 1].mean())
    if kind == "uniform":
        mean = rng.uniform(lo - 5.0, hi + 5.0, n).astype(np.float32)
        stdv = _log_uniform(rng, 0.01, 50.0, n)
    elif kind == "runs":
        covered = 0
        while covered < n // 5:
            ln = int(rng.integers(50, 501))
            a = int(rng.integers(0, max(1, n - ln)))
            mean[a:a + ln] = mean[a]
            stdv[a:a + ln] = stdv[a]
            covered += ln
    elif kind == "spikes":
        hit = rng.random(n) < 0.01
        mean[hit] += (np.where(rng.random(int(hit.sum())) < 0.5, -20.0, 20.0) * sigma).astype(np.float32)
    elif kind == "stdv_tail":
        stdv = _log_uniform(rng, 0.01, 50.0, n)
    elif kind == "zeros":
        stdv[rng.random(n) < 0.005] = 0.0
    elif kind == "abasic":
        for _ in range(max(1, n // 5000)):
            ln = int(rng.integers(100, 301))
            a = int(rng.integers(0, max(1, n - ln)))
            mean[a:a + ln] = np.float32(hi + 30.0) + rng.normal(0.0, 1.5, min(ln, n - a)).astype(np.float32)
    return mean.astype(np.float32), stdv.astype(np.float32), start.astype(np.float32)


def log_uniform_lengths(rng, n_reads, longest):
    """read lengths log-uniform over [1, longest]"""
    return [max(1, int(x)) for x in np.exp(rng.uniform(0.0, np.log(longest), n_reads))]
