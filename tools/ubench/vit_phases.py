#!/usr/bin/env python3
"""Where does a wave of viterbi_kernel spend an event?  Builds an instrumented copy of the tree's kernel (s_memtime stamps at
the phase boundaries of column(), summed per wave in SGPRs, added up through the profile buffer), times it with bench.py and
prints cycles per wave-event by phase.  The tree's kernel source is not touched; the library is rebuilt from it at the end.

  python tools/ubench/vit_phases.py          (on the GPU box, from the repo root)

Phases:  scan     group scans + next-float probes (everything up to the exchange writes)
         publish  LDS writes of the group winners up to the barrier
         barrier  waiting at s_barrier for the other seven waves
         combine  the eight 3-way combines, the eight emissions, the back-pointer store (hipcc moves the emissions behind the
                  combines, so they cannot be told apart from source positions)
The stamps cost ~10 % themselves; the split, not the total, is the result."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "nanocall_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -fno-slp-vectorize".split()


def sub1(s, old, new):
    assert s.count(old) == 1, (s.count(old), old[:60])
    return s.replace(old, new)


def main():
    src = open(os.path.join(CSRC, "viterbi_kernel.hip")).read()
    s = src
    s = sub1(s, "    unsigned n_rescan, n_tie;\n};", "    unsigned n_rescan, n_tie;\n    unsigned long long ph[4];\n};")
    s = sub1(s, "    // ---------------- group scans over the previous column ----------------",
             "    const unsigned long long T0 = __builtin_readcyclecounter();\n    // ---------------- group scans over the previous column ----------------")
    s = sub1(s, "    sV1[(h << 8) | t] = ValSlot{s1[0], sl1[0]};",
             "    asm volatile(\"\" : \"+v\"(s1[0]), \"+v\"(s1[1]), \"+v\"(s2));\n    const unsigned long long T1 = __builtin_readcyclecounter();\n    sV1[(h << 8) | t] = ValSlot{s1[0], sl1[0]};")
    s = sub1(s, "    if (h == 0) sV2[t] = ValSlot{s2, sl2};\n    __syncthreads();",
             "    if (h == 0) sV2[t] = ValSlot{s2, sl2};\n    const unsigned long long T2 = __builtin_readcyclecounter();\n    __syncthreads();\n    const unsigned long long T3 = __builtin_readcyclecounter();")
    s = sub1(s, "    *reinterpret_cast<uint2*>(bp_row + tau * 8u) = make_uint2(w_lo, w_hi);\n}",
             "    *reinterpret_cast<uint2*>(bp_row + tau * 8u) = make_uint2(w_lo, w_hi);\n    asm volatile(\"\" : \"+v\"(S.alpha[0]), \"+v\"(S.alpha[7]));\n"
             "    const unsigned long long T4 = __builtin_readcyclecounter();\n    S.ph[0] += T1 - T0; S.ph[1] += T2 - T1; S.ph[2] += T3 - T2; S.ph[3] += T4 - T3;\n}")
    s = sub1(s, "        S.n_rescan = 0; S.n_tie = 0;", "        S.n_rescan = 0; S.n_tie = 0;\n        for (int q = 0; q < 4; ++q) S.ph[q] = 0;")
    s = sub1(s, "            if ((tau & 63u) == 0) {   // per wave: the branches are wave-uniform",
             "            if ((tau & 63u) == 0) for (int q = 0; q < 4; ++q) atomicAdd(&P.prof[q], S.ph[q]);\n            if ((tau & 63u) == 0) {   // per wave: the branches are wave-uniform")
    s = sub1(s, "        atomicAdd(&P.prof[0], t_fwd);\n        atomicAdd(&P.prof[1], t_tb);\n        atomicAdd(&P.prof[2], wall_clock64() - t_all0);\n        atomicAdd(&P.prof[3], 1ull);\n", "")
    tmp = "/tmp/viterbi_kernel_phases.hip"
    open(tmp, "w").write(s)
    try:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", tmp, "-o", "viterbi_kernel.o"], cwd=CSRC, check=True)
        subprocess.run(["make", "-s"], cwd=CSRC, check=True, capture_output=True)
        steps, warm = 4, 1
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline",
                            "--no-fwbw", "--no-end-to-end"], capture_output=True, text=True, env=dict(os.environ, NCHMM_PROFILE="1"))
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        m = re.search(r"forward=(\d+) traceback=(\d+) block=(\d+) blocks=(\d+)", p.stderr)
        ticks = [int(x) for x in m.groups()]
        wave_events = (steps + warm) * 1024 * 4999 * 8          # columns 1..4999 of 1024 reads, 8 waves
        names = ["scan", "publish", "barrier", "combine + emission + store"]
        out = {"kernel_ms_instrumented": line["roofline"]["kernel_ms"], "shader_clock_mhz": line["device"]["shader_clock_mhz_under_load"],
               "cycles_per_wave_event": {n: round(t / wave_events, 1) for n, t in zip(names, ticks)}}
        out["cycles_per_wave_event"]["total"] = round(sum(ticks) / wave_events, 1)
        print(json.dumps(out))
    finally:
        os.remove(os.path.join(CSRC, "viterbi_kernel.o"))
        subprocess.run(["make", "-s"], cwd=CSRC, check=True, capture_output=True)


if __name__ == "__main__":
    main()
