"""What, exactly, slows the forward sweep down after an idle gap?  The shader clock alone drops 4 % after 20 ms of idling
(tools/ubench/keep_warm.hip), the sweep 17 %.  Here the gap before each device-resident launch is spent
  idle   nothing queued               fma   torch elementwise FMA chains on a small tensor (VALU busy, no HBM traffic)
  copy   1 GiB device-to-device copies (HBM + fabric busy, VALU almost idle)          both  the two interleaved
and the launch is timed with the library's hipEvents."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import nanocall_amd as na
from nanocall_amd import synth

R, E = 1024, 5000
t = na.builtin_model("r73.t")
ev = synth.generate(t, R, E)
off, mean, stdv, start = synth.flat_batch(ev)
cm, sd, ls = na.events_prepare(mean, stdv, start, 0.0)
dev = torch.device("cuda", 0)
ctx = na.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
total = R * E
d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
d_cm, d_sd, d_ls = (torch.from_numpy(x).to(dev) for x in (cm, sd, ls))
d_state = torch.empty(total, dtype=torch.int16, device=dev)
d_logp = torch.empty(R, dtype=torch.float32, device=dev)
d_status = torch.zeros(R, dtype=torch.int32, device=dev)
small = torch.randn(256 * 1024, device=dev)
big_a = torch.empty(256 << 20, dtype=torch.float32, device=dev)
big_b = torch.empty_like(big_a)


def fill_gap(kind, ms):
    t0 = time.perf_counter()
    if kind == "idle":
        torch.cuda.synchronize(); time.sleep(ms * 1e-3); return
    while (time.perf_counter() - t0) * 1e3 < ms:
        if kind in ("fma", "both"):
            for _ in range(8):
                small.mul_(1.0000001).add_(1e-9)
        if kind in ("copy", "both"):
            big_b.copy_(big_a)
        torch.cuda.synchronize()


def launch():
    ctx.viterbi_dev(R, E, total, d_off, d_cm, d_sd, d_ls, d_state, d_logp, d_status)
    return ctx.last_kernel_ms()[0]


out = {}
for _ in range(8):
    launch()
out["back_to_back_ms"] = round(float(np.mean([launch() for _ in range(8)])), 3)
for gap in (2.0, 20.0):
    for kind in ("idle", "fma", "copy", "both"):
        ks = []
        for rep in range(6):
            for _ in range(4):
                launch()                      # warm again
            torch.cuda.synchronize()
            fill_gap(kind, gap)
            ks.append(launch())
        out[f"gap_{gap:g}ms_{kind}"] = round(float(np.mean(ks)), 3)
out["shader_clock_mhz_under_load"] = round(ctx.shader_clock_mhz())
print(json.dumps(out, indent=1))
