#!/usr/bin/env python3
"""Host event prep (nchmm_events_prepare: stdv 0 -> .01, log_stdv = logf(stdv), drift correction) throughput, against the
rate the GPUs consume events at -- the question behind SURVEY section 8f rank 4 -- and the device-side replacement
(nchmm_viterbi_raw) on the same batch.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanocall_amd as na            # noqa: E402
from nanocall_amd import synth       # noqa: E402
import bench                          # noqa: E402

n_reads, n_events = int(os.environ.get("READS", 1024)), 5000
table = na.builtin_model("r73.t")
off, mean, stdv, start = bench.generate_shard(table, np.arange(n_reads), n_events, threads=8)
total = n_reads * n_events
# one host thread (chunks below the library's threading threshold), then the library's own threading
t0 = time.perf_counter()
for r in range(0, min(n_reads, 64)):
    a, b = r * n_events, (r + 1) * n_events
    na.events_prepare(mean[a:b], stdv[a:b], start[a:b], 0.001)
t1 = time.perf_counter() - t0
one = min(n_reads, 64) * n_events / t1 / 1e6
t0 = time.perf_counter()
for _ in range(3):
    cm, sd, ls = na.events_prepare(mean, stdv, start, 0.001)
tn = (time.perf_counter() - t0) / 3
many = total / tn / 1e6
res = {"host_prepare_Mevents_per_s_one_thread": round(one, 1), "host_prepare_Mevents_per_s_library_threads": round(many, 1),
       "host_logical_cpus": os.cpu_count(), "events": total}
try:
    ctx = na.Context(0)
    ctx.put_model(0, na.scaled_model_table(table))
    ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
    src = (np.arange(n_reads) * n_events).astype(np.uint64)
    ln = np.full(n_reads, n_events, np.uint32)
    dr = np.full(n_reads, 0.001, np.float32)
    ctx.viterbi_raw(mean, stdv, start, src, ln, dr)
    t0 = time.perf_counter()
    for _ in range(3):
        s_raw, lp_raw, _ = ctx.viterbi_raw(mean, stdv, start, src, ln, dr)
    t_raw = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        cm, sd, ls = na.events_prepare(mean, stdv, start, 0.001)
        s_h, lp_h, _ = ctx.viterbi(off, cm, sd, ls)
    t_host = (time.perf_counter() - t0) / 3
    assert np.array_equal(s_raw, s_h) and np.array_equal(lp_raw.view(np.uint32), lp_h.view(np.uint32))
    res.update({"viterbi_raw_ms": round(t_raw * 1e3, 2), "events_prepare_plus_viterbi_ms": round(t_host * 1e3, 2),
                "viterbi_raw_Mevents_per_s_incl_pcie": round(total / t_raw / 1e6, 1), "bit_identical": True})
    ctx.close()
except na.api.NchmmError as e:
    res["gpu"] = str(e)
print(json.dumps(res))
