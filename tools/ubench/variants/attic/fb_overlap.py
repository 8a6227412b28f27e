"""Do a forward sweep and a backward sweep of DIFFERENT batches co-run well on one device?  Two contexts, one stream each,
one block per CU each (NCHMM_EXP_FB_BLOCKS_PER_CU=1), the second stream started half a cycle late so that its forward
sweeps fall on the first stream's backward sweeps; compared with one context at two blocks per CU doing the same work."""
import os, sys, time, json
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import nanocall_amd as na
from nanocall_amd import synth

n_reads, n_ev, iters = 1024, 100, int(os.environ.get("ITERS", 12))
dev = torch.device("cuda", 0)
t0, t1 = na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")
e0 = synth.generate(t0, n_reads, 2 * n_ev); e1 = synth.generate(t1, n_reads, 2 * n_ev, first_read=10**6)
mean = np.stack([e0["mean"][:, :n_ev], e0["mean"][:, n_ev:], e1["mean"][:, :n_ev], e1["mean"][:, n_ev:]], 1).reshape(-1)
stdv = np.stack([e0["stdv"][:, :n_ev], e0["stdv"][:, n_ev:], e1["stdv"][:, :n_ev], e1["stdv"][:, n_ev:]], 1).reshape(-1)
cm, sd, ls = na.events_prepare(mean, stdv, None, 0.0)
n_win = n_reads * 4; total = n_win * n_ev
off = (np.arange(n_win + 1) * n_ev).astype(np.int64)
strand = np.tile(np.array([0, 0, 1, 1], np.int32), n_reads)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d_off, d_cm, d_sd, d_ls, d_slot = d(off), d(cm), d(sd), d(ls), d(strand)
d_tr = torch.zeros(n_win, dtype=torch.int32, device=dev)
d_sp = torch.tensor([0.1, 0.3], dtype=torch.float32, device=dev).repeat(n_win, 1).contiguous()


def make(stream):
    ctx = na.Context(0)
    ctx.set_stream(stream.cuda_stream)
    for s, t in enumerate((t0, t1)):
        ctx.put_model(s, na.scaled_model_table(t))
    ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    out = (torch.empty(n_win, dtype=torch.float32, device=dev), torch.empty(total * 6, dtype=torch.float32, device=dev),
           torch.empty(n_win * 3, dtype=torch.float32, device=dev))
    return ctx, out


def step(ctx, out):
    ctx.fwbw_dev(n_win, n_ev, total, d_off, d_cm, d_sd, d_ls, out[0], out[1], out[2], d_scaled_slot=d_slot, d_trans_slot=d_tr, d_st_params=d_sp)


def run(n_ctx, delay_cycles):
    streams = [torch.cuda.Stream(dev) for _ in range(n_ctx)]
    cs = [make(s) for s in streams]
    for c, o in cs:
        step(c, o)
    torch.cuda.synchronize()
    t_0 = time.perf_counter()
    if n_ctx == 2 and delay_cycles:
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(delay_cycles)
    for _ in range(iters):
        for c, o in cs:
            step(c, o)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_0
    lp = [float(o[0].mean().item()) for _, o in cs]
    for c, _ in cs:
        c.close()
    return dt / (iters * n_ctx) * 1e3, lp


mode = os.environ.get("MODE", "one")
if mode == "one":
    ms, lp = run(1, 0)
    print(json.dumps({"mode": "one context", "blocks_per_cu": os.environ.get("NCHMM_EXP_FB_BLOCKS_PER_CU", "2"), "ms_per_batch": round(ms, 3), "lpd": lp}))
else:
    for delay in (0, 2_000_000, 4_000_000, 6_000_000):
        ms, lp = run(2, delay)
        print(json.dumps({"mode": "two contexts, two streams", "blocks_per_cu": os.environ.get("NCHMM_EXP_FB_BLOCKS_PER_CU", "2"),
                          "second_stream_delay_cycles": delay, "ms_per_batch": round(ms, 3), "lpd": lp}))
