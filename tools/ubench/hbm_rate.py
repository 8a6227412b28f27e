#!/usr/bin/env python3
"""Streaming HBM rates of this GPU as torch sees them (calibration for the FB roofline fractions): fill (pure write),
sum (pure read), copy (read + write) over 8 GiB fp32 buffers."""
import time, torch
n = 2 * 1024**3
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
w = t(lambda: a.fill_(1.0)); r = t(lambda: a.sum()); c = t(lambda: b.copy_(a))
print(f"write {n*4/w/1e12:.2f} TB/s   read {n*4/r/1e12:.2f} TB/s   copy {2*n*4/c/1e12:.2f} TB/s (read+write)")
