"""Wall time of the host-pointer entry points (nchmm_viterbi, nchmm_viterbi_raw) for 1024 reads x 5000 events: kernels
plus the copies between the caller's pageable memory and the device, with the shader clock of the box."""
import hashlib, json, os, sys, time
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
out = {"reads": R, "events": E}
best = 1e9
for i in range(6):
    t0 = time.perf_counter(); st, lp, status = ctx.viterbi(off, cm, sd, ls); dt = time.perf_counter() - t0
    best = min(best, dt)
out["viterbi_ms"] = round(best * 1e3, 2)
out["viterbi_kernels_ms"] = [round(x, 2) for x in ctx.last_kernel_ms()[:2]]
out["viterbi_mevents_s"] = round(R * E / best / 1e6, 1)
out["states_sha"] = hashlib.sha256(np.ascontiguousarray(st).tobytes()).hexdigest()[:16]
src = off[:-1].astype(np.uint64); ln = np.diff(off).astype(np.uint32); drift = np.zeros(R, np.float32)
best = 1e9
for i in range(6):
    t0 = time.perf_counter(); st2, lp2, status2 = ctx.viterbi_raw(mean, stdv, start, src, ln, drift); dt = time.perf_counter() - t0
    best = min(best, dt)
out["viterbi_raw_ms"] = round(best * 1e3, 2)
out["viterbi_raw_mevents_s"] = round(R * E / best / 1e6, 1)
out["raw_equal"] = bool(np.array_equal(st, st2) and np.array_equal(lp, lp2))
out["shader_clock_mhz_under_load"] = round(ctx.shader_clock_mhz())
print(json.dumps(out))
