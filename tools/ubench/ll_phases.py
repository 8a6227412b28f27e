#!/usr/bin/env python3
"""Where does a wave of viterbi_ll_kernel spend an event?  Builds an instrumented copy of the tree's kernel (s_memtime stamps at
the phase boundaries of column_ll(), summed per wave, added up through the profile buffer), runs 256 reads x 5000 events with
the low-latency form forced and prints cycles per wave-event by phase.  The tree's source is not touched; the library is
rebuilt from it at the end.      python tools/ubench/ll_phases.py          (on the GPU box, from the repo root)

Phases:  scan     step-group scan, the two quad merges, next-float probes (everything up to the exchange writes)
         publish  LDS writes of the group winners up to the barrier
         barrier  waiting at s_barrier for the other fifteen waves
         combine  exchange reads, the four 3-way combines and emissions, the back-pointer store
The stamps cost ~10 % themselves; the split, not the total, is the result."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "nanocall_amd", "csrc")


def sub1(s, old, new):
    assert s.count(old) == 1, (s.count(old), old[:60])
    return s.replace(old, new)


def main():
    flags = subprocess.run(["make", "-s", "-C", CSRC, "print-hipflags"], check=True, capture_output=True, text=True).stdout.split()
    s = open(os.path.join(CSRC, "viterbi_ll_kernel.hip")).read()
    s = sub1(s, "    unsigned n_rescan, n_tie;\n};", "    unsigned n_rescan, n_tie;\n    unsigned long long ph[4];\n};")
    s = sub1(s, "    // ---------------- group scans over the previous column ----------------",
             "    const unsigned long long T0 = __builtin_readcyclecounter();\n    // ---------------- group scans over the previous column ----------------")
    s = sub1(s, "    X.w1[PAR * kV1Pitch] = ValSlot{s1, sl1};",
             "    asm volatile(\"\" : \"+v\"(s1), \"+v\"(s2));\n    const unsigned long long T1 = __builtin_readcyclecounter();\n    X.w1[PAR * kV1Pitch] = ValSlot{s1, sl1};")
    s = sub1(s, "    if (yy == 0) X.w2[PAR * kV2Pitch] = ValSlot{s2, sl2};\n    __syncthreads();",
             "    if (yy == 0) X.w2[PAR * kV2Pitch] = ValSlot{s2, sl2};\n    const unsigned long long T2 = __builtin_readcyclecounter();\n    __syncthreads();\n    const unsigned long long T3 = __builtin_readcyclecounter();")
    s = sub1(s, "    __builtin_nontemporal_store(bpw, reinterpret_cast<unsigned*>(bp_row) + tau);\n}",
             "    __builtin_nontemporal_store(bpw, reinterpret_cast<unsigned*>(bp_row) + tau);\n    asm volatile(\"\" : \"+v\"(S.alpha[0]), \"+v\"(S.alpha[3]));\n"
             "    const unsigned long long T4 = __builtin_readcyclecounter();\n    S.ph[0] += T1 - T0; S.ph[1] += T2 - T1; S.ph[2] += T3 - T2; S.ph[3] += T4 - T3;\n}")
    s = sub1(s, "        S.n_rescan = 0; S.n_tie = 0;\n", "        S.n_rescan = 0; S.n_tie = 0;\n        for (int q = 0; q < 4; ++q) S.ph[q] = 0;\n")
    s = sub1(s, "            if ((tau & 63u) == 0) {   // per wave: the branches are wave-uniform",
             "            if ((tau & 63u) == 0) for (int q = 0; q < 4; ++q) atomicAdd(&P.prof[q], S.ph[q]);\n            if ((tau & 63u) == 0) {   // per wave: the branches are wave-uniform")
    s = sub1(s, "        atomicAdd(&P.prof[0], t_fwd);\n        atomicAdd(&P.prof[1], t_tb);\n        atomicAdd(&P.prof[2], wall_clock64() - t_all0);\n        atomicAdd(&P.prof[3], 1ull);\n", "")
    tmp = "/tmp/viterbi_ll_phases.hip"
    open(tmp, "w").write(s)
    code = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import nanocall_amd as na
from nanocall_amd import synth
R, E = int(os.environ.get("READS", 256)), 5000
t = na.builtin_model("r73.t")
ctx = na.Context(0)
ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1))
ctx.set_sweep("ll")
ev = synth.generate(t, R, E)
off, m, s, st = synth.flat_batch(ev)
cm, sd, ls = na.events_prepare(m, s, st, 0.0)
ctx.viterbi(off, cm, sd, ls)
ctx.profile_ticks(reset=True)
n = 3
for _ in range(n): ctx.viterbi(off, cm, sd, ls)
k = ctx.last_kernel_ms()[0]
ticks = ctx.profile_ticks()
we = n * R * (E - 1) * 16
names = ["scan", "publish", "barrier", "combine + emission + store"]
out = {"reads": R, "kernel_ms_instrumented": round(k, 3), "cycles_per_wave_event": {nm: round(ticks[i] / we, 1) for i, nm in enumerate(names)}}
out["cycles_per_wave_event"]["total"] = round(sum(ticks[:4]) / we, 1)
out["shader_clock_mhz"] = round(ctx.shader_clock_mhz())
print(json.dumps(out))
''' % ROOT
    try:
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-I" + CSRC, "-c", tmp, "-o", "viterbi_ll_kernel.o"], cwd=CSRC, check=True)
        subprocess.run(["make", "-s"], cwd=CSRC, check=True, capture_output=True)
        for reads in (os.environ.get("READS", "256").split(",")):
            p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, NCHMM_PROFILE="1", READS=reads))
            print(p.stdout.strip() or p.stderr[-2000:])
    finally:
        os.remove(os.path.join(CSRC, "viterbi_ll_kernel.o"))
        subprocess.run(["make", "-s"], cwd=CSRC, check=True, capture_output=True)


if __name__ == "__main__":
    main()
