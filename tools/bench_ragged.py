"""Throughput of the host-pointer Viterbi on RAGGED batches (what real runs look like: nanopore reads are log-normally
long) against the uniform batch of the same total size.  Lengths: lognormal(median MEDIAN, sigma SIGMA) clipped to
[200, MAXLEN], seeded.  Reports the one-call form, the streaming form (DEPTH batches in flight, default one per lane) and the lower bound the
longest read sets (a read is sequential: one block, one event after the other)."""
import hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nanocall_amd as na
from nanocall_amd import synth

R = int(os.environ.get("READS", 1024))
MEDIAN = float(os.environ.get("MEDIAN", 4000))
SIGMA = float(os.environ.get("SIGMA", 0.7))
MAXLEN = int(os.environ.get("MAXLEN", 30000))
REPS = int(os.environ.get("REPS", 5))
CHECK = int(os.environ.get("CHECK", 0))   # compare this many reads with the oracle

rng = np.random.default_rng(20261002)
lens = np.clip(np.round(np.exp(rng.normal(np.log(MEDIAN), SIGMA, R))), 200, MAXLEN).astype(np.int64)
t = na.builtin_model("r73.t")
ev = synth.generate(t, R, int(lens.max()))
keep = np.arange(int(lens.max()))[None, :] < lens[:, None]
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
mean, stdv, start = ev["mean"][keep], ev["stdv"][keep], ev["start"][keep]
del ev
cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
total = int(off[-1])
ctx = na.Context(0)
ctx.set_sweep(os.environ.get("SWEEP", "auto"))     # auto | wide | ll: which form of the sweep launches take
ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
out = {"reads": R, "events": total, "len_min": int(lens.min()), "len_median": int(np.median(lens)), "len_max": int(lens.max()),
       "slots": ctx.grid_slots()}


def timed(fn):
    best = 1e9
    for _ in range(REPS):
        t0 = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t0)
    return best, r


best, (st, lp, status) = timed(lambda: ctx.viterbi(off, cm, sd, ls))
out["ragged_one_call_ms"] = round(best * 1e3, 2)
out["ragged_one_call_mevents_s"] = round(total / best / 1e6, 1)
out["kernels_ms"] = [round(x, 2) for x in ctx.last_kernel_ms()[:2]]
out["states_sha"] = hashlib.sha256(np.ascontiguousarray(st).tobytes()).hexdigest()[:16]


DEPTH = int(os.environ.get("DEPTH", 3))     # batches in flight (the library takes as many as it has lanes: three)
NB = int(os.environ.get("BATCHES", 6))


def stream(n_batches):
    tk = []
    for _ in range(n_batches):
        if len(tk) == DEPTH:
            ctx.viterbi_end(tk.pop(0))
        tk.append(ctx.viterbi_begin(off, cm, sd, ls))
    while len(tk) > 1:
        ctx.viterbi_end(tk.pop(0))
    return ctx.viterbi_end(tk.pop(0))


best, (st2, lp2, status2) = timed(lambda: stream(NB))
out["ragged_streaming_mevents_s"] = round(NB * total / best / 1e6, 1)
out["streaming_batches"], out["streaming_depth"] = NB, DEPTH
out["streaming_equal"] = bool(np.array_equal(st, st2) and np.array_equal(lp, lp2))

# the same number of events as equal-length reads
E = total // R
evu = synth.generate(t, R, E)
offu, m_u, s_u, st_u = synth.flat_batch(evu)
cmu, sdu, lsu = na.events_prepare(m_u, s_u, st_u, 0.0)
best, _ = timed(lambda: ctx.viterbi(offu, cmu, sdu, lsu))
out["uniform_events_per_read"] = E
out["uniform_one_call_mevents_s"] = round(R * E / best / 1e6, 1)
out["uniform_kernels_ms"] = [round(x, 2) for x in ctx.last_kernel_ms()[:2]]
out["shader_clock_mhz_under_load"] = round(ctx.shader_clock_mhz())
if CHECK:
    sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
    import nc_oracle as oracle
    om, ot = oracle.Model(t, (1.0, 0.0, 0.0, 1.0, 1.0, 1.0)), oracle.Transitions(0.3, 0.1)
    order = np.argsort(lens)[:CHECK]
    ok = True
    for r in order:
        a, b = int(off[r]), int(off[r + 1])
        s, mv, p = oracle.viterbi(om, ot, cm[a:b], sd[a:b], ls[a:b])
        ok &= bool(np.array_equal(s, st[a:b])) and np.float32(p).tobytes() == np.float32(lp[r]).tobytes()
    out["oracle_checked_reads"] = int(CHECK); out["oracle_equal"] = bool(ok)
out["sweep"] = os.environ.get("SWEEP", "auto")
out["launches_wide_ll_reads_wide_ll"] = list(ctx.sweep_stats())
out["ahead_launches_reads_events"] = list(ctx.ahead_stats())
print(json.dumps(out))
