"""A short host-pointer session for a rocprofv3 timeline (--kernel-trace --memory-copy-trace): 3 one-call batches, then 5
streamed ones (two in flight)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nanocall_amd as na
from nanocall_amd import synth
R, E = int(os.environ.get("READS", 1024)), int(os.environ.get("EVENTS", 5000))
t = na.builtin_model("r73.t")
ev = synth.generate(t, R, E)
off, mean, stdv, start = synth.flat_batch(ev)
cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
ctx = na.Context(0)
ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
total = R * E
for i in range(3):
    t0 = time.perf_counter(); ctx.viterbi(off, cm, sd, ls); print("one-call ms", round((time.perf_counter() - t0) * 1e3, 2))
outs = [(np.empty(total, np.uint16), np.empty(R, np.float32), np.zeros(R, np.int32)) for _ in range(2)]
t0 = time.perf_counter()
tk = ctx.viterbi_begin(off, cm, sd, ls, out=outs[0])
for i in range(1, 5):
    tb = time.perf_counter(); nxt = ctx.viterbi_begin(off, cm, sd, ls, out=outs[i & 1]); te = time.perf_counter()
    ctx.viterbi_end(tk); tk = nxt
    print("begin ms", round((te - tb) * 1e3, 2), "end ms", round((time.perf_counter() - te) * 1e3, 2))
ctx.viterbi_end(tk)
print("streamed 5 batches ms", round((time.perf_counter() - t0) * 1e3, 2))
