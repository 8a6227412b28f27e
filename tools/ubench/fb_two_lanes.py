#!/usr/bin/env python3
"""Do the forward sweep (store-bound) and the backward sweep (VALU-bound) of DIFFERENT batches run faster side by side than one
behind the other?  One context with the whole config-3 window batch (4096 windows x 100 events, 2 blocks per CU) against two
contexts on the same device with half the batch each, their launches queued alternately on their own streams -- with the full grid
each (their persistent blocks then take turns: only the tails overlap) and with NCHMM_FB_SLOTS=256 each (one block per CU per
context: a forward block of one batch beside a backward block of the other on every CU).

  python tools/ubench/fb_two_lanes.py      (GPU box)   -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def make(n_reads, n_ev, seed):
    import nanocall_amd as na
    from nanocall_amd import synth
    t0, t1 = na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")
    e0 = synth.generate(t0, n_reads, 2 * n_ev, first_read=seed)
    e1 = synth.generate(t1, n_reads, 2 * n_ev, first_read=10 ** 6 + seed)
    pick = lambda k: np.stack([e0[k][:, :n_ev], e0[k][:, n_ev:], e1[k][:, :n_ev], e1[k][:, n_ev:]], 1).reshape(-1)
    cm, sd, ls = na.events_prepare(pick("mean"), pick("stdv"), None, 0.0)
    n_win = n_reads * 4
    return dict(n_win=n_win, total=n_win * n_ev, n_ev=n_ev, off=(np.arange(n_win + 1) * n_ev).astype(np.int64), cm=cm, sd=sd, ls=ls,
                strand=np.tile(np.array([0, 0, 1, 1], np.int32), n_reads), tables=(t0, t1))


class Lane:
    def __init__(self, batch, slots=None):
        import torch
        import nanocall_amd as na
        if slots:
            os.environ["NCHMM_FB_SLOTS"] = str(slots)
        self.ctx = na.Context(0)
        os.environ.pop("NCHMM_FB_SLOTS", None)
        dev = torch.device("cuda", 0)
        for s, t in enumerate(batch["tables"]):
            self.ctx.put_model(2 + s, na.scaled_model_table(t))
        self.ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        b = batch
        self.b = b
        self.t = [d(b["off"]), d(b["cm"]), d(b["sd"]), d(b["ls"]), torch.empty(b["n_win"], dtype=torch.float32, device=dev),
                  torch.empty(b["total"] * 6, dtype=torch.float32, device=dev), torch.empty(b["n_win"] * 3, dtype=torch.float32, device=dev)]
        self.slot = d(b["strand"] + 2)
        self.tr = torch.zeros(b["n_win"], dtype=torch.int32, device=dev)
        self.sp = torch.tensor([0.1, 0.3], dtype=torch.float32, device=dev).repeat(b["n_win"], 1).contiguous()

    def step(self):
        b = self.b
        self.ctx.fwbw_dev(b["n_win"], b["n_ev"], b["total"], *self.t, d_scaled_slot=self.slot, d_trans_slot=self.tr, d_st_params=self.sp)

    def sync(self):
        self.ctx.synchronize()


def rate(lanes, steps, warm=12):
    for _ in range(warm):
        for l in lanes:
            l.step()
    for l in lanes:
        l.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for l in lanes:
            l.step()
    for l in lanes:
        l.sync()
    dt = time.perf_counter() - t0
    return sum(l.b["total"] for l in lanes) * steps / dt / 1e6


def main():
    import torch
    assert torch.cuda.is_available()
    steps = int(os.environ.get("STEPS", 60))
    whole, half_a, half_b = make(1024, 100, 0), make(512, 100, 0), make(512, 100, 7000)
    out = {}
    one = Lane(whole)
    out["one_context_4096_windows"] = round(rate([one], steps), 2)
    for name, slots in (("two_contexts_2048_each_full_grid", None), ("two_contexts_2048_each_256_slots", 256), ("two_contexts_2048_each_384_slots", 384)):
        a, b = Lane(half_a, slots), Lane(half_b, slots)
        out[name] = round(rate([a, b], steps), 2)
        a.ctx.close(); b.ctx.close()
    a, b = Lane(whole, 256), Lane(make(1024, 100, 9000), 256)
    out["two_contexts_4096_each_256_slots"] = round(rate([a, b], steps), 2)
    a.ctx.close(); b.ctx.close()
    out["one_context_4096_windows_again"] = round(rate([one], steps), 2)
    out["unit"] = "M event-rounds/s (wall, launches queued back to back)"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
