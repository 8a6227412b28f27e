"""Why do viterbi_kernel launches of the host-pointer path take ~8 % longer than the back-to-back device-resident ones
?  Same batch, same kernel, per-launch hipEvent time under controlled conditions:
  A  device-resident, back to back           B  + identity `order` array
  C  + an idle gap before every launch        D  on the library's own (non-blocking) stream
  E  host-pointer path (nchmm_viterbi)        F  host-pointer path after a busy spin instead of an idle wait
Prints one JSON object."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import nanocall_amd as na
from nanocall_amd import synth

R, E = int(os.environ.get("READS", 1024)), int(os.environ.get("EVENTS", 5000))
t = na.builtin_model("r73.t")
ev = synth.generate(t, R, E)
off, mean, stdv, start = synth.flat_batch(ev)
cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
dev = torch.device("cuda", 0)
ctx = na.Context(0)
ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
total = R * E
d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
d_cm, d_sd, d_ls = (torch.from_numpy(x).to(dev) for x in (cm, sd, ls))
d_state = torch.empty(total, dtype=torch.int16, device=dev)
d_logp = torch.empty(R, dtype=torch.float32, device=dev)
d_status = torch.zeros(R, dtype=torch.int32, device=dev)
d_order = torch.arange(R, dtype=torch.int32, device=dev)
out = {"reads": R, "events": E}


def dev_leg(name, n=8, order=None, gap_ms=0.0, own=False):
    if own:
        ctx.use_own_stream()
    else:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ks, wall = [], []
    for i in range(n + 2):
        if gap_ms:
            torch.cuda.synchronize(); ctx.synchronize(); time.sleep(gap_ms * 1e-3)
        t0 = time.perf_counter()
        ctx.viterbi_dev(R, E, total, d_off, d_cm, d_sd, d_ls, d_state, d_logp, d_status, d_order=order)
        k = ctx.last_kernel_ms()[0]
        ctx.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        if i >= 2:
            ks.append(k)
    out[name] = {"kernel_ms_mean": round(float(np.mean(ks)), 3), "kernel_ms": [round(x, 2) for x in ks],
                 "wall_ms_median": round(float(np.median(wall[2:])), 3)}


def host_leg(name, n=8, own=False, spin_ms=0.0):
    if own:
        ctx.use_own_stream()
    else:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ks, wall = [], []
    for i in range(n + 2):
        if spin_ms:
            time.sleep(spin_ms * 1e-3)
        t0 = time.perf_counter()
        st, lp, status = ctx.viterbi(off, cm, sd, ls)
        wall.append((time.perf_counter() - t0) * 1e3)
        if i >= 2:
            ks.append(ctx.last_kernel_ms()[0])
    out[name] = {"kernel_ms_mean": round(float(np.mean(ks)), 3), "kernel_ms": [round(x, 2) for x in ks],
                 "wall_ms_median": round(float(np.median(wall[2:])), 3), "mevents_s": round(total / np.median(wall[2:]) / 1e3, 1)}


def stream_leg(name, n=24, warm=6):
    """begin(k+1) before end(k): two batches in flight, output buffers recycled.  The first `warm` batches bring the shader
    clock back up (it takes ~50 ms of uninterrupted load after the gaps of the previous leg) and are not timed."""
    ctx.use_own_stream()
    outs = [(np.empty(total, np.uint16), np.empty(R, np.float32), np.zeros(R, np.int32)) for _ in range(2)]
    tk = ctx.viterbi_begin(off, cm, sd, ls, out=outs[0])
    per = []
    for i in range(1, n + warm + 1):
        t0 = time.perf_counter()
        nxt = ctx.viterbi_begin(off, cm, sd, ls, out=outs[i & 1])
        st, lp, status = ctx.viterbi_end(tk)
        tk = nxt
        per.append((time.perf_counter() - t0) * 1e3)
    st, lp, status = ctx.viterbi_end(tk)
    dt = float(np.sum(per[warm:]))
    out[name] = {"batches_timed": n, "warm_up_batches": warm, "wall_ms_per_batch": round(dt / n, 3), "mevents_s": round(total * n / dt / 1e3, 1),
                 "per_batch_ms": [round(x, 2) for x in per],
                 "note": "one loop iteration = begin(k+1) + end(k); wall over the timed iterations"}
    return st, lp


dev_leg("A_dev_back_to_back")
dev_leg("B_dev_identity_order", order=d_order)
dev_leg("C_dev_gap_2ms", gap_ms=2.0)
dev_leg("C2_dev_gap_20ms", gap_ms=20.0)
dev_leg("D_dev_own_stream", own=True)
host_leg("E_host_null_stream")
host_leg("E2_host_own_stream", own=True)
host_leg("F_host_gap_20ms", own=True, spin_ms=20.0)
st_s, lp_s = stream_leg("G_host_streaming_begin_end")
st_1, lp_1, _ = ctx.viterbi(off, cm, sd, ls)
out["streaming_equals_one_call"] = bool(np.array_equal(st_s, st_1) and lp_s.tobytes() == lp_1.tobytes())
out["one_call_equals_device_resident"] = bool(np.array_equal(st_1, d_state.cpu().numpy().view(np.uint16)))
dev_leg("A2_dev_back_to_back_again")
out["shader_clock_mhz_under_load"] = round(ctx.shader_clock_mhz())
print(json.dumps(out, indent=1))
