#!/usr/bin/env python3
"""ISA budget of a kernel's innermost loops: instruction counts by issue class between a loop head and its back-edge.

  python tools/isa_budget.py <file.s> <kernel-symbol-substring> [--min 100]

Reads hipcc's -S output (gfx950), finds every loop (a label that a LATER branch targets) of the kernel whose body is at
least --min instructions and contains no other such loop, and prints per loop the number of instructions by class:
  valu2   VALU ops that issue at the full fp32 rate (add / sub / mul / fma / mov / and / or / xor / add_u32 ... : measured
          1.06-1.14 ns per wave-instruction per SIMD at 4 waves per SIMD, profiles/r03_sstore_rate.txt)
  valu4   VALU ops at half that rate (compare, cndmask, max / min / med3, shifts-with-or, DPP, lane moves: 1.75-1.92 ns)
  trans   transcendentals / rcp (3.4 ns)
  salu, smem, lds (ds_*), vmem (global_/buffer_/scratch_), branch / waitcnt / other
"""
import collections
import re
import sys

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_pk_add_f32", "v_pk_mul_f32",
        "v_pk_fma_f32", "v_add_co_u32", "v_addc_co_u32", "v_not_b32"}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32"}


def klass(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.startswith("v_"):
        if op.endswith("_dpp") or "readlane" in op or "writelane" in op or "readfirstlane" in op:
            return "valu4"
        if base in TRANS:
            return "trans"
        return "valu2" if base in FULL else "valu4"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith(("s_load", "s_store", "s_buffer", "s_dcache", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_endpgm", "s_sleep")):
        return "ctl"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 100
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*" + re.escape(sym) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] and (o[1] - o[0]) >= min_len for o in loops)]
    for a, b in sorted(set(inner)):
        ops = [m.group(1) for l in body[a:b + 1] for m in [re.match(r"\s+([a-z]\w+)", l)] if m and not l.strip().startswith((";", "."))]
        if len(ops) < min_len:
            continue
        by = collections.Counter(klass(o) for o in ops)
        detail = collections.Counter(o for o in ops if klass(o) in ("valu4", "smem", "vmem", "lds", "trans"))
        print(f"loop {body[a].strip()} .. line {b - a} instructions={len(ops)}  " + "  ".join(f"{k}={by[k]}" for k in ("valu2", "valu4", "trans", "salu", "smem", "lds", "vmem", "ctl", "other") if by[k]))
        print("    half-rate / memory ops: " + ", ".join(f"{k} x{v}" for k, v in sorted(detail.items(), key=lambda kv: -kv[1])))
        full = collections.Counter(o for o in ops if klass(o) == "valu2")
        print("    full-rate VALU: " + ", ".join(f"{k} x{v}" for k, v in sorted(full.items(), key=lambda kv: -kv[1])))
        est = by["valu2"] * 1.09 + by["valu4"] * 1.81 + by["trans"] * 3.4
        print(f"    VALU issue time at the measured per-class rates (4 waves/SIMD): {est:.0f} ns per wave-iteration")


if __name__ == "__main__":
    main()
