import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nanocall_amd as na
from nanocall_amd import synth
dev = torch.device("cuda", 0)
n_reads, n_ev = 1024, 100
t0, t1 = na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")
e0 = synth.generate(t0, n_reads, 2 * n_ev); e1 = synth.generate(t1, n_reads, 2 * n_ev, first_read=10**6)
mean = np.stack([e0["mean"][:, :n_ev], e0["mean"][:, n_ev:], e1["mean"][:, :n_ev], e1["mean"][:, n_ev:]], 1).reshape(-1)
stdv = np.stack([e0["stdv"][:, :n_ev], e0["stdv"][:, n_ev:], e1["stdv"][:, :n_ev], e1["stdv"][:, n_ev:]], 1).reshape(-1)
cm, sd, ls = na.events_prepare(mean, stdv, None, 0.0)
n_win = 4 * n_reads
strand = np.tile(np.array([0, 0, 1, 1], np.int32), n_reads)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def make(part, nparts, frac=None):
    lo, hi = (0, n_win) if nparts == 1 else ((0, int(n_win * frac)) if part == 0 else (int(n_win * frac), n_win))
    nw = hi - lo; tot = nw * n_ev
    ctx = na.Context(0); st = torch.cuda.Stream(); ctx.set_stream(st.cuda_stream)
    for s, t in enumerate((t0, t1)): ctx.put_model(s, na.scaled_model_table(t))
    ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    args = dict(off=d((np.arange(nw + 1) * n_ev).astype(np.int64)), cm=d(cm[lo*n_ev:hi*n_ev]), sd=d(sd[lo*n_ev:hi*n_ev]), ls=d(ls[lo*n_ev:hi*n_ev]),
                slot=d(strand[lo:hi]), tr=torch.zeros(nw, dtype=torch.int32, device=dev), sp=torch.tensor([0.1, 0.3], device=dev).repeat(nw, 1).contiguous(),
                lpd=torch.empty(nw, device=dev), pm=torch.empty(tot * 6, device=dev), stt=torch.empty(nw * 3, device=dev))
    def run():
        ctx.fwbw_dev(nw, n_ev, tot, args["off"], args["cm"], args["sd"], args["ls"], args["lpd"], args["pm"], args["stt"], d_scaled_slot=args["slot"], d_trans_slot=args["tr"], d_st_params=args["sp"])
    return ctx, run
def timeit(runs, reps=10):
    for r in runs: r()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for r in runs: r()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
whole = make(0, 1)
print("one context, all windows: %.2f ms" % timeit([whole[1]]))
for frac in (0.5, 0.35, 0.25):
    a, b = make(0, 2, frac), make(1, 2, frac)
    print("two contexts / streams, split %.2f: %.2f ms" % (frac, timeit([a[1], b[1]])))
