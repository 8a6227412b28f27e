import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nanocall_amd as na
from nanocall_amd import synth
t = na.builtin_model("r73.t")
ev = synth.generate(t, 1024, 5000)
off, mean, stdv, start = synth.flat_batch(ev)
cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
ctx = na.Context(0)
ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
for i in range(4):
    t0 = time.perf_counter(); st, lp, status = ctx.viterbi(off, cm, sd, ls); dt = time.perf_counter() - t0
    print(f"nchmm_viterbi host pointers: {dt*1e3:.1f} ms  {5.12/dt:.1f} Mevents/s  kernels {ctx.last_kernel_ms()[:2]}")
