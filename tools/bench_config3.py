#!/usr/bin/env python3
"""BASELINE config 3: template+complement 2D reads with 4-round Parameter_Trainer EM, then Viterbi of both
strands for every candidate model pair.  Host-pointer entry points (includes host prep, PCIe copies).
  READS=1024 python tools/bench_config3.py
Reports wall time of the EM stage (nchmm_train_reads), FB event-rounds/s, and of the decode stage."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanocall_amd as na                    # noqa: E402
from nanocall_amd import api, synth          # noqa: E402

n_reads = int(os.environ.get("READS", 1024))
n_ev = int(os.environ.get("EVENTS", 5000))
names = ["r73.c.p1", "r73.c.p2", "r73.t"]
strands = [1, 1, 0]
tables = [na.builtin_model(n) for n in names]
states = np.stack([na.model_load(t) for t in tables])
e0 = synth.generate(tables[2], n_reads, n_ev)
e1 = synth.generate(tables[0], n_reads, n_ev, first_read=10**6)
mean = np.stack([e0["mean"], e1["mean"]], 1).reshape(-1)
stdv = np.stack([e0["stdv"], e1["stdv"]], 1).reshape(-1)
start = np.stack([e0["start"], e1["start"]], 1).reshape(-1)
_, stdv, _ = na.events_prepare(mean, stdv, None, 0.0)
so = (np.arange(2 * n_reads + 1) * n_ev).astype(np.uint64)
opts = api.train_opts(scaling_max_rounds=2, scaling_min_progress=0.0)   # exactly 4 rounds per 2D pair (nanocall.cpp:420)
jr, j0, j1 = api.train_enumerate(opts, strands, so, np.ones(n_reads, np.uint8))
ctx = na.Context(0)
ctx.train_reads(opts, states, so[:5], mean[:2 * n_ev * 2], stdv[:2 * n_ev * 2], start[:2 * n_ev * 2], jr[:4], j0[:4], j1[:4])  # warm-up
t0 = time.perf_counter()
out = ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
t_em = time.perf_counter() - t0
rounds = out["rounds"].astype(np.int64)
ev_rounds = int((rounds * 4 * (opts.scaling_num_events // 2)).sum())
# decode: both strands of every pair with its trained parameters (basecall_reads, nanocall.cpp:692-712)
t1 = time.perf_counter()
nj = len(jr)
slots_m = np.empty((nj, 2), np.int32)
par = np.repeat(out["pm"], 2, axis=0)
idx = np.stack([j0, j1], 1).reshape(-1)
ctx.put_models_scaled(8, states, idx, par)
ctx.put_transitions_fast(8, out["st"].reshape(-1, 2)[:, 1], out["st"].reshape(-1, 2)[:, 0])
off = np.zeros(2 * nj + 1, np.uint64)
cm = np.empty(2 * nj * n_ev, np.float32); sd = np.empty_like(cm); ls = np.empty_like(cm)
for k in range(nj):
    for s in range(2):
        a = int(so[2 * jr[k] + s]); w = 2 * k + s
        c, d, l = na.events_prepare(mean[a:a + n_ev], stdv[a:a + n_ev], start[a:a + n_ev], float(out["pm"][k, 2]))
        cm[w * n_ev:(w + 1) * n_ev], sd[w * n_ev:(w + 1) * n_ev], ls[w * n_ev:(w + 1) * n_ev] = c, d, l
        off[w + 1] = (w + 1) * n_ev
slot = (8 + np.arange(2 * nj)).astype(np.int32)
t2 = time.perf_counter()
states_out, logp, status = ctx.viterbi(off, cm, sd, ls, model_slot=slot, trans_slot=slot)
t_vit = time.perf_counter() - t2
print(json.dumps({"config": "2D + 4-round EM", "reads": n_reads, "events_per_strand": n_ev, "jobs": int(nj),
                  "em_wall_s": round(t_em, 3), "em_rounds_mean": float(rounds.mean()), "fb_event_rounds": ev_rounds,
                  "fb_Mevent_rounds_per_s_incl_host": round(ev_rounds / t_em / 1e6, 3),
                  "decode_prep_s": round(t2 - t1, 3), "viterbi_wall_s": round(t_vit, 3),
                  "viterbi_Mevents_per_s_incl_pcie": round(2 * nj * n_ev / t_vit / 1e6, 2),
                  "fit_mean": float(out["fit"].mean()), "status_ok": bool((status == 0).all())}))
t3 = time.perf_counter()
states_out, logp, status = ctx.viterbi(off, cm, sd, ls, model_slot=slot, trans_slot=slot)
print(json.dumps({"viterbi_wall_s_second_call": round(time.perf_counter() - t3, 3), "kernel_ms": ctx.last_kernel_ms()}))
t4 = time.perf_counter()
out2 = ctx.train_reads(opts, states, so, mean, stdv, start, jr, j0, j1)
print(json.dumps({"em_wall_s_second_call": round(time.perf_counter() - t4, 3)}))
# the drop-in decode stage: nchmm_basecall_reads (candidate table + device tables + raw copy-in + Viterbi + best-model choice,
# host pointers in and out).  First call into fresh arrays, then the arrays are reused as a chunk loop does.
bc = None
walls = []
for label in ["basecall_reads_wall_s"] + ["basecall_reads_wall_s_call_%d" % i for i in range(2, 7)]:
    t5 = time.perf_counter()
    bc = ctx.basecall_reads(opts, states, so, mean, stdv, start, jr, j0, j1, out["pm"], out["st"], out=bc)
    dt = time.perf_counter() - t5
    walls.append(dt)
    print(json.dumps({label: round(dt, 4), "Mevents_per_s_incl_host": round(2 * nj * n_ev / dt / 1e6, 2),
                      "kernel_ms": ctx.last_kernel_ms(), "reads_with_a_winner": int((bc["best_job"][:, 0] >= 0).sum())}))
med = float(np.median(walls[1:]))
# the same candidates, device-resident and back to back, for the ratio (clock of THIS box, now)
print(json.dumps({"basecall_reads_median_of_calls_2_6_s": round(med, 4), "Mevents_per_s_incl_host": round(2 * nj * n_ev / med / 1e6, 2),
                  "shader_clock_mhz_under_load": round(ctx.shader_clock_mhz())}))
