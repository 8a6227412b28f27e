#!/usr/bin/env python3
"""FB/EM kernel throughput (BASELINE config 3 shape): 1024 2D reads x (2 strands x 2 windows x 100 events),
one forward-backward + statistics pass ("event-round") per window, inputs resident in HBM.
Prints one JSON line (event-rounds/s, achieved algorithmic GB/s at 32 780 B/event-round)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                      # noqa: E402
import nanocall_amd as na         # noqa: E402
from nanocall_amd import synth    # noqa: E402

n_reads, n_win_per_read, n_ev = int(os.environ.get("READS", 1024)), 4, 100
steps = int(os.environ.get("STEPS", 5))
dev = torch.device("cuda", 0)
t0, t1 = na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")
e0 = synth.generate(t0, n_reads, 2 * n_ev)
e1 = synth.generate(t1, n_reads, 2 * n_ev, first_read=10**6)
mean = np.stack([e0["mean"][:, :n_ev], e0["mean"][:, n_ev:], e1["mean"][:, :n_ev], e1["mean"][:, n_ev:]], 1).reshape(-1)
stdv = np.stack([e0["stdv"][:, :n_ev], e0["stdv"][:, n_ev:], e1["stdv"][:, :n_ev], e1["stdv"][:, n_ev:]], 1).reshape(-1)
cm, sd, ls = na.events_prepare(mean, stdv, None, 0.0)
n_win = n_reads * n_win_per_read
total = n_win * n_ev
off = (np.arange(n_win + 1) * n_ev).astype(np.int64)
strand = np.tile(np.array([0, 0, 1, 1], np.int32), n_reads)
ctx = na.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for s, t in enumerate((t0, t1)):
    ctx.put_model(s, na.scaled_model_table(t))
ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d_off, d_cm, d_sd, d_ls, d_slot = d(off), d(cm), d(sd), d(ls), d(strand)
d_tr = torch.zeros(n_win, dtype=torch.int32, device=dev)
d_sp = torch.tensor([0.1, 0.3], dtype=torch.float32, device=dev).repeat(n_win, 1).contiguous()
d_lpd = torch.empty(n_win, dtype=torch.float32, device=dev)
d_pm = torch.empty(total * 6, dtype=torch.float32, device=dev)
d_st = torch.empty(n_win * 3, dtype=torch.float32, device=dev)


def step():
    ctx.fwbw_dev(n_win, n_ev, total, d_off, d_cm, d_sd, d_ls, d_lpd, d_pm, d_st, d_scaled_slot=d_slot,
                 d_trans_slot=d_tr, d_st_params=d_sp)


step(); torch.cuda.synchronize()
ks = []
t_0 = time.perf_counter()
for _ in range(steps):
    step()
    ks.append(ctx.last_kernel_ms()[2])
torch.cuda.synchronize()
dt = time.perf_counter() - t_0
k_ms = float(np.mean(ks))
print(json.dumps({"metric": "FB+EM-statistics event-rounds/s", "value": round(total * steps / dt / 1e6, 3), "unit": "Mevent-rounds/s",
                  "windows": n_win, "events_per_window": n_ev, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
                  "kernel": "nchmm::fwbw_forward_scaled_kernel + nchmm::fwbw_backward_scaled_kernel (+ 2 empty redo launches)", "kernel_ms": round(k_ms, 3),
                  "roofline": {"bound": "hbm", "bytes_per_event_round": 32780,
                               "achieved_GBs": round(32780 * total / (k_ms * 1e-3) / 1e9, 2), "peak_GBs": 8000.0,
                               "frac": round(32780 * total / (k_ms * 1e-3) / 1e9 / 8000.0, 5)},
                  "shader_clock_mhz_under_load": round(ctx.shader_clock_mhz()),
                  "log_pr_data_mean": float(d_lpd.mean().item())}))
