"""The two forms of the Viterbi sweep side by side (viterbi_kernel.hip `wide`: 8 waves per read, two reads per CU;
viterbi_ll_kernel.hip `ll`: 16 waves per read, one read per CU) on batches of R equal reads of E events, from host arrays
through the one-call form (nchmm_viterbi).  Reports the kernel's own time (hipEvents around the launch), microseconds per
event of a read, Mevents/s, and whether the two forms returned the same bits.  SHAPES="R:E,R:E,..." overrides the list."""
import hashlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nanocall_amd as na
from nanocall_amd import synth

SHAPES = [tuple(int(v) for v in s.split(":")) for s in os.environ.get("SHAPES", "1:5000,64:5000,256:5000,384:5000,512:5000,1024:5000,256:50000").split(",")]
REPS = int(os.environ.get("REPS", 3))
MODEL = os.environ.get("MODEL", "r73.t")

t = na.builtin_model(MODEL)
ctx = na.Context(0)
ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
n_cu = ctx.grid_slots() // 2
rows = []
for R, E in SHAPES:
    ev = synth.generate(t, R, E)
    off, m, s, st = synth.flat_batch(ev)
    cm, sd, ls = na.events_prepare(m, s, st, 0.0)
    row = {"reads": R, "events_per_read": E}
    sha = {}
    for mode in ("wide", "ll", "ahead", "auto"):
        ctx.set_sweep(mode)
        before = ctx.sweep_stats()
        best_wall, best_k = 1e9, 1e9
        for _ in range(REPS):
            t0 = time.perf_counter()
            states, logp, status = ctx.viterbi(off, cm, sd, ls, raise_on_numeric=False)    # (experiment builds return wrong results on purpose)
            best_wall = min(best_wall, time.perf_counter() - t0)
            best_k = min(best_k, ctx.last_kernel_ms()[0])
        after = ctx.sweep_stats()
        sha[mode] = hashlib.sha256(np.ascontiguousarray(states).tobytes() + np.ascontiguousarray(logp).tobytes()).hexdigest()[:16]
        waves = -(-R // (2 * n_cu if mode == "wide" else n_cu))      # (kernel_ms of "ahead" = the sweep alone: the emission kernel runs in front of the timed events)
        row[mode] = {"kernel_ms": round(best_k, 3), "wall_ms": round(best_wall * 1e3, 3),
                     "mevents_s_kernel": round(R * E / best_k / 1e3, 1), "mevents_s_wall": round(R * E / best_wall / 1e6, 1),
                     "launches_wide_ll": [after[0] - before[0], after[1] - before[1]]}
        if mode != "auto":
            row[mode]["us_per_event_of_a_read"] = round(best_k * 1e3 / (E * waves), 4)
        row[mode]["ahead_launches_reads_events"] = list(ctx.ahead_stats())
    row["same_bits"] = sha["wide"] == sha["ll"] == sha["auto"] == sha["ahead"]
    rows.append(row)
    print(json.dumps(row), flush=True)
print(json.dumps({"n_cu": n_cu, "shader_clock_mhz_under_load": round(ctx.shader_clock_mhz()), "all_same_bits": all(r["same_bits"] for r in rows)}))
