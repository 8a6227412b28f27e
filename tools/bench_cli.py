#!/usr/bin/env python3
"""End-to-end throughput of the `nanocall` binary (FAST5 in, FASTA out) on this machine's GPU(s): N synthetic 2D reads as
FAST5 files (a few dozen distinct reads written through tools/make_fast5, copied under new names), one CLI run, wall time
split into the stages the CLI reports.  Prints one JSON line.

  READS=2000 EVENTS=3000 THREADS=32 python tools/bench_cli.py [extra nanocall options]
  RAGGED=1: read lengths log-normal around EVENTS (sigma 0.8, 600 .. 8 x EVENTS events; 48 distinct reads), what real runs
  look like -- a launch lasts as long as its longest read
  WORKERS=N: one worker process per GPU as `nanocall --gpus N` starts them, all N on GPU 0 here (NANOCALL_WORKER_DEVICES): the line
  then carries every worker's stage clock and the rate at which its HOST stages (summary pass, event loading + packing, FASTA) take
  its share of the input -- what one worker's host side sustains beside the others, i.e. whether N GPUs would be fed
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_pipeline as op   # noqa: E402  (only its synthetic read generator: test infrastructure, not the measured path)

n_reads = int(os.environ.get("READS", 2000))
n_events = int(os.environ.get("EVENTS", 3000))
threads = int(os.environ.get("THREADS", min(32, os.cpu_count() or 1)))
ragged = os.environ.get("RAGGED") == "1"
distinct = min(n_reads, 48 if ragged else 24)
import numpy as _np
_rng = _np.random.default_rng(4242)
lens = (_np.clip(_np.round(_np.exp(_rng.normal(_np.log(n_events), 0.8, distinct))), 600, 8 * n_events).astype(int) if ragged
        else _np.full(distinct, n_events))
cli = os.path.join(ROOT, "nanocall_amd", "bin", "nanocall")
tool = os.path.join(ROOT, "tools", "make_fast5")
if not os.path.exists(tool):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "make_fast5"], check=True, capture_output=True)
tmp = tempfile.mkdtemp(prefix="nanocall_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
try:
    t0 = time.perf_counter()
    for k in range(distinct):
        half = int(lens[k]) // 2
        ed = op.synth_ed_table("r73", half, half, seed=100 + k, hairpin=8, complement_model="r73.c.p1.006.ont.model" if k % 2 else "r73.c.p2.006.ont.model",
                               scale=1.0 + 0.01 * (k % 5), shift=float(k % 7) - 3.0, drift=0.002 * (k % 3))
        ev = os.path.join(tmp, f"seed{k}.events")
        op.write_events_table(ev, ed, 4000.0, f"read-{k}")
        subprocess.run([tool, ev, os.path.join(tmp, f"seed{k}.fast5")], check=True)
        os.remove(ev)
    d = os.path.join(tmp, "reads")
    os.mkdir(d)
    for r in range(n_reads):
        shutil.copy(os.path.join(tmp, f"seed{r % distinct}.fast5"), os.path.join(d, f"r{r:06d}.fast5"))
    t_gen = time.perf_counter() - t0
    out = os.path.join(tmp, "out.fa")
    cmd = [cli, "--pore", "r73", "-t", str(threads), "-o", out] + sys.argv[1:] + [d]
    t0 = time.perf_counter()
    e0 = time.time()
    n_workers = int(os.environ.get("WORKERS", 0))
    env = dict(os.environ)
    if n_workers:
        env["NANOCALL_WORKER_DEVICES"] = ",".join(["0"] * n_workers)
    p = subprocess.run(cmd, capture_output=True, text=True, env=env)
    wall = time.perf_counter() - t0
    e1 = time.time()
    assert p.returncode == 0, p.stderr[-2000:]
    if os.environ.get("NCHMM_DEBUG"):
        sys.stderr.write("".join(l + "\n" for l in p.stderr.splitlines() if l.startswith("[nchmm")))
    line = [l for l in p.stderr.splitlines() if "counters reads=" in l and "worker_counters" not in l][-1]
    kv = dict(tok.split("=") for tok in line.split() if "=" in tok)
    stages = [l for l in p.stderr.splitlines() if l.startswith("= nanocall info: stage_wall_secs")][-1].split("stage_wall_secs")[1].split()
    stages = {t.split("=")[0]: round(float(t.split("=")[1]), 3) for t in stages}
    n_rec = sum(1 for l in open(out) if l.startswith(">"))
    marks = {k: float(l.split(k + "=")[1].split()[0]) for l in p.stderr.splitlines() for k in ("epoch_at_main", "epoch_at_exit") if k + "=" in l}
    outside = ({"spawn_to_main_s": round(marks["epoch_at_main"] - e0, 3), "exit_to_reaped_s": round(e1 - marks["epoch_at_exit"], 3)}
               if len(marks) == 2 else None)
    reserve = [l.split("reserve_viterbi_workspace", 1)[1].strip() for l in p.stderr.splitlines() if "reserve_viterbi_workspace" in l]
    ev_in = int(sum(int(lens[r % distinct]) for r in range(int(kv["reads"]))))
    # one worker process per GPU: each worker's stage clock, and what its host stages sustain on its share of the input
    workers = []
    for l in p.stderr.splitlines():
        if l.startswith("= nanocall info: worker ") and "stage_wall_secs" in l:
            head, tail = l.split("stage_wall_secs")
            w = head.split()
            st = {t.split("=")[0]: float(t.split("=")[1]) for t in tail.split()}
            share = ev_in * int(w[8]) / max(1, int(kv["reads"]))
            rate = lambda k: (round(share / st[k] / 1e6, 1) if st.get(k) else None)
            workers.append({"worker": int(w[4]), "device": int(w[6]), "reads": int(w[8]), "input_events": int(share),
                            "stages": {k: round(v, 3) for k, v in st.items()},
                            "host_Mevents_per_s": {"summary_pass": rate("init_reads_s"), "load_events": rate("load_events_s"), "soa_and_jobs": rate("soa_and_jobs_s"),
                                                   "fasta": rate("fasta_s")},
                            "gpu_Mevents_per_s_of_input": (round(share / (st.get("training_total_s", 0.0) + st.get("basecalling_total_s", 0.0)) / 1e6, 1)
                                                           if st.get("basecalling_total_s") else None)})
    print(json.dumps({"reads": int(kv["reads"]), "events_per_read": n_events, "ragged": ragged, "longest_read_events": int(lens.max()), "input_events": ev_in, "fasta_records": n_rec, "bases": int(kv["bases"]),
                      "wall_s": round(wall, 3), "training_s": float(kv["training_secs"]), "basecalling_s": float(kv["basecalling_secs"]),
                      "other_s_(summaries, event loading, FASTA)": round(wall - float(kv["training_secs"]) - float(kv["basecalling_secs"]), 3),
                      "reads_per_s": round(int(kv["reads"]) / wall, 1), "input_Mevents_per_s_end_to_end": round(ev_in / wall / 1e6, 2),
                      "events_decoded": int(kv["events_decoded"]), "decoded_Mevents_per_s_in_basecalling": round(int(kv["events_decoded"]) / float(kv["basecalling_secs"]) / 1e6, 1),
                      "fb_event_rounds": int(kv["fb_event_rounds"]), "host_threads": threads, "gathered_by": kv["gathered_by"], "fixture_generation_s": round(t_gen, 1), "stages": stages, "workers": workers or None, "viterbi_workspace_reserved": reserve[-1] if reserve else None, "outside_main": outside, "stderr_bytes": len(p.stderr), "memory_at_exit": ([l.split("memory_at_exit", 1)[1].strip() for l in p.stderr.splitlines() if "memory_at_exit" in l] or [None])[-1],
                      "cmd": " ".join(cmd[:1] + cmd[1:-1])}))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
