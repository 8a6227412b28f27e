#!/usr/bin/env python3
"""Where does a wave of the two rescaled forward-backward sweeps spend an event?  Builds an instrumented copy of the tree's
fwbw_scaled_kernel.hip (s_memtime stamps at the phase boundaries of both event loops, summed per wave, added up through the
profile buffer), runs the config-3 window shape (4096 windows x 100 events) and prints cycles per wave-event by phase.  The tree's
source is not touched; the library is rebuilt from it at the end.      python tools/ubench/fb_phases.py   (GPU box, repo root)

forward:   produce  group sums of the previous column, DPP pair swap + wave total, LDS writes (up to the barrier)
           barrier  waiting at s_barrier
           consume  LDS reads, column total -> scale, 8 emissions (exp2) and the 8 cell updates
           store    the 8 row stores (issue only: they complete asynchronously)
backward:  produce  row prefetch issue, 8 emissions x beta, group sums, LDS writes, the 7-way wave reduction of the previous event's sums
           barrier  waiting at s_barrier
           consume  publish of the previous event's 6 sums, LDS reads, 8 x (beta update, posterior, 6 pm sums, 3 transition sums)
The stamps cost a few % themselves; the split, not the total, is the result."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "nanocall_amd", "csrc")


def sub1(s, old, new):
    assert s.count(old) == 1, (s.count(old), old[:70])
    return s.replace(old, new)


def main():
    src = open(os.path.join(CSRC, "fwbw_scaled_kernel.hip")).read()
    s = src
    # ---- forward ----
    s = sub1(s, "        int Ia = 0;\n        bool bad = false;\n        double ref_sum = 0.0;",
             "        int Ia = 0;\n        bool bad = false;\n        double ref_sum = 0.0;\n        unsigned long long ph[4] = {0, 0, 0, 0};")
    s = sub1(s, "                    const float a = (ah[0] + ah[2]) + (ah[4] + ah[6]);   // y = h",
             "                    const unsigned long long tk0 = __builtin_readcyclecounter();\n                    const float a = (ah[0] + ah[2]) + (ah[4] + ah[6]);   // y = h")
    s = sub1(s, "                    if (lane == 63) sZ[buf][wave] = z;\n                    __syncthreads();",
             "                    if (lane == 63) sZ[buf][wave] = z;\n                    const unsigned long long tk1 = __builtin_readcyclecounter();\n                    __syncthreads();\n"
             "                    const unsigned long long tk2 = __builtin_readcyclecounter();")
    s = sub1(s, "                        ah[q] = E * __builtin_fmaf(W1[q], in1[q], __builtin_fmaf(T0[q], ah[q], in2[q])) * sc;\n                    }\n",
             "                        ah[q] = E * __builtin_fmaf(W1[q], in1[q], __builtin_fmaf(T0[q], ah[q], in2[q])) * sc;\n                    }\n"
             "                    asm volatile(\"\" : \"+v\"(ah[0]), \"+v\"(ah[7]));\n                    const unsigned long long tk3 = __builtin_readcyclecounter();\n"
             "                    ph[0] += tk1 - tk0; ph[1] += tk2 - tk1; ph[2] += tk3 - tk2; ph[3] -= tk3;\n")
    s = sub1(s, "                rowp += kStates;\n                if (tau == 0) P.ws_exp[e0 + i] = Ia;",
             "                rowp += kStates;\n                if (i != 0) ph[3] += __builtin_readcyclecounter();\n                if (tau == 0) P.ws_exp[e0 + i] = Ia;")
    s = sub1(s, "        // log_pr_data = log sum_j alpha[n-1][j]  (Forward_Backward.hpp:129-134)",
             "        if (P.prof && lane == 0) for (int q = 0; q < 4; ++q) atomicAdd(&P.prof[q], ph[q]);\n        // log_pr_data = log sum_j alpha[n-1][j]  (Forward_Backward.hpp:129-134)")
    # ---- backward ----
    s = sub1(s, "        float acc_p = 0, acc_stay = 0, acc_p01 = 0;\n",
             "        float acc_p = 0, acc_stay = 0, acc_p01 = 0;\n        unsigned long long phb[3] = {0, 0, 0};\n")
    s = sub1(s, "            unsigned tl = tau;\n            asm volatile(\"\" : \"+v\"(tl));",
             "            const unsigned long long tb0 = __builtin_readcyclecounter();\n            unsigned tl = tau;\n            asm volatile(\"\" : \"+v\"(tl));")
    s = sub1(s, "            __syncthreads();\n            publish((unsigned)i, tl);",
             "            const unsigned long long tb1 = __builtin_readcyclecounter();\n            __syncthreads();\n            const unsigned long long tb2 = __builtin_readcyclecounter();\n            publish((unsigned)i, tl);")
    s = sub1(s, "            pend_ei = (unsigned)(i - 1);\n            pend_kappa = kappa;",
             "            pend_ei = (unsigned)(i - 1);\n            pend_kappa = kappa;\n            asm volatile(\"\" : \"+v\"(bh[0]), \"+v\"(bh[7]), \"+v\"(acc_p));\n"
             "            { const unsigned long long tb3 = __builtin_readcyclecounter(); phb[0] += tb1 - tb0; phb[1] += tb2 - tb1; phb[2] += tb3 - tb2; }")
    s = sub1(s, "        // window totals of the transition statistics",
             "        if (P.prof && lane == 0) for (int q = 0; q < 3; ++q) atomicAdd(&P.prof[4 + q], phb[q]);\n        // window totals of the transition statistics")
    keep = "/tmp/fwbw_scaled_kernel.hip.tree"
    open(keep, "w").write(src)
    path = os.path.join(CSRC, "fwbw_scaled_kernel.hip")
    try:
        open(path, "w").write(s)
        subprocess.run(["make", "-s", "-j8"], cwd=CSRC, check=True)
        # bench_fwbw.py creates its own context; read the ticks through a second tiny script around it
        drv = os.path.join(ROOT, "tools", "_fb_phase_driver.py")      # (bench_fwbw.py finds the package relative to its own path)
        open(drv, "w").write(open(os.path.join(ROOT, "tools", "bench_fwbw.py")).read().replace(
            "print(json.dumps(", "TICKS = ctx.profile_ticks(reset=False)\nprint(json.dumps({'ticks': [int(x) for x in TICKS]}))\nprint(json.dumps(", 1))
        p = subprocess.run([sys.executable, drv], capture_output=True, text=True, env=dict(os.environ, NCHMM_PROFILE="1", STEPS="10"), cwd=ROOT)
        os.remove(drv)
        lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
        if not any("ticks" in l for l in lines):
            sys.stderr.write(p.stdout[-1000:] + p.stderr[-3000:])
            return 1
        ticks = next(l["ticks"] for l in lines if "ticks" in l)
        line = next(l for l in lines if "value" in l)
        launches = 10 + 1                                     # STEPS + the warm-up launch of bench_fwbw.py
        wave_events = launches * 4096 * 99 * 8                # columns 1..99 of 4096 windows, 8 waves
        fwd = dict(zip(["produce", "barrier", "consume", "store"], ticks[0:4]))
        bwd = dict(zip(["produce", "barrier", "consume"], ticks[4:7]))
        out = {"kernel_ms_instrumented": line["kernel_ms"], "shader_clock_mhz": line["shader_clock_mhz_under_load"],
               "forward_cycles_per_wave_event": {k: round(v / wave_events, 1) for k, v in fwd.items()},
               "backward_cycles_per_wave_event": {k: round(v / wave_events, 1) for k, v in bwd.items()}}
        out["forward_cycles_per_wave_event"]["total"] = round(sum(fwd.values()) / wave_events, 1)
        out["backward_cycles_per_wave_event"]["total"] = round(sum(bwd.values()) / wave_events, 1)
        print(json.dumps(out))
        if p.returncode != 0:
            sys.stderr.write(p.stderr[-2000:])
    finally:
        open(path, "w").write(src)
        subprocess.run(["make", "-s", "-j8"], cwd=CSRC, check=True, capture_output=True)


if __name__ == "__main__":
    main()
