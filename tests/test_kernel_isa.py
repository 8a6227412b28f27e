"""Structure of the compiled Viterbi forward kernels (the 8-wave form and the 16-wave low-latency form) that their speed depends on and that no parity test would notice losing
(profiles/r03_viterbi_isa_budget.md section 2c): cross-compiles viterbi_kernel.hip to gfx950 assembly
(no GPU needed) and checks the fast path's column loop for
  * no scratch (spill) traffic inside the loop,
  * the hand-written half-rate + full-rate pairs sitting next to each other (the compiler must not pull them apart),
  * the exchange tables still read 16 bytes wide (narrower reads of the thread-major tables are 8-way LDS bank conflicts),
  * the tie paths out of line (the fast path falls through its eight per-cell branches)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nanocall_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")


def _hipflags():
    """the device-code flags of csrc/Makefile (one source of truth: `make print-hipflags`)"""
    p = subprocess.run(["make", "-s", "-C", CSRC, "print-hipflags"], check=True, capture_output=True, text=True)
    return [f.replace("-I../../include", "-I" + os.path.join(ROOT, "include")).replace("-I.", "-I" + CSRC) if f in ("-I../../include", "-I.") else f
            for f in p.stdout.split()]


KERNEL_FILES = ("viterbi_kernel", "viterbi_ll_kernel", "fwbw_scaled_kernel", "fwbw_kernel", "em_kernel")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    """gfx950 assembly of every kernel file, compiled as the library compiles them"""
    from concurrent.futures import ThreadPoolExecutor
    d = tmp_path_factory.mktemp("isa")
    flags = _hipflags() + ["-S", "--cuda-device-only"]

    def build(name):
        out = d / (name + ".s")
        subprocess.run([HIPCC] + flags + ["-o", str(out), os.path.join(CSRC, name + ".hip")], check=True, capture_output=True, timeout=900)
        return name, open(out).read()

    with ThreadPoolExecutor(4) as ex:
        return dict(ex.map(build, KERNEL_FILES))


@pytest.fixture(scope="module")
def fast_loop(asm):
    lines = asm["viterbi_kernel"].split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*viterbi_kernel\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # the column loops: a label that a LATER conditional branch targets, holding eight v_max3 (one per cell); the fast path's
    # is the one without the true-division sequence
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m:
            continue
        back = [k for k in range(i + 1, len(body)) if re.search(r"s_cbranch\S*\s+" + re.escape(m.group(1)) + r"\b", body[k])]
        if back:
            seg = [x.split(";")[0].strip() for x in body[i:back[-1] + 1]]
            seg = [x for x in seg if x and not x.startswith(".")]
            if sum("v_max3_f32" in x for x in seg) == 8 and not any("v_div_scale" in x for x in seg):
                loops.append(seg)
    assert loops, "no fast column loop found in the assembly"
    return min(loops, key=len)


def test_no_spills_in_the_column_loop(fast_loop):
    assert not [x for x in fast_loop if x.startswith(("scratch_", "buffer_load", "buffer_store"))]
    assert not [x for x in fast_loop if "accvgpr" in x]


def test_hand_written_pairs_stay_adjacent(fast_loop):
    pairs = {"v_max3_f32": ("v_sub_f32", 8), "v_lshl_or_b32": ("v_add_f32", 6)}
    for first, (second, want) in pairs.items():
        got = sum(1 for i, x in enumerate(fast_loop) if x.startswith(first) and fast_loop[i + 1].startswith(second))
        assert got >= want, f"{first} directly followed by {second}: {got} of {want}"
    # the combine's two selects per cell: v_cndmask_b32_e64 + v_subrev_f32 (u0 = c - 3 log y) and + v_fma_f32 (first residual)
    assert sum(1 for i, x in enumerate(fast_loop) if x.startswith("v_cndmask_b32_e64") and fast_loop[i + 1].startswith("v_subrev_f32")) >= 8
    assert sum(1 for i, x in enumerate(fast_loop) if x.startswith("v_cndmask_b32_e64") and fast_loop[i + 1].startswith("v_fma_f32")) >= 8


def test_tables_are_read_sixteen_bytes_wide(fast_loop):
    lds_reads = [x.split()[0] for x in fast_loop if x.startswith("ds_read")]
    assert lds_reads.count("ds_read_b128") >= 6, lds_reads          # 3 tables x 2 chunks (+ the event record)
    assert set(lds_reads) <= {"ds_read_b128", "ds_read_b64", "ds_read_b96"}, sorted(set(lds_reads))


def test_tie_paths_are_out_of_line(fast_loop):
    # fast path: per cell one conditional branch that is NOT taken (to the out-of-line exact path) and no compare-heavy slow code inline
    assert sum(x.startswith("v_max3_f32") for x in fast_loop) == 8
    assert len(fast_loop) < 560, len(fast_loop)                      # 470 today; 740 with the tie paths inline
    assert sum(x.startswith("s_barrier") for x in fast_loop) == 1


# ---- the low-latency form (viterbi_ll_kernel.hip): two columns per trip of its fast loop ----
@pytest.fixture(scope="module")
def ll_fast_loop(asm):
    lines = asm["viterbi_ll_kernel"].split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*viterbi_ll_kernel\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m:
            continue
        # (the trip ends in an unconditional branch to the header; conditional branches further down that target labels inside it
        # come back from the out-of-line rescan / tie blocks, which are not part of the fast path)
        back = [k for k in range(i + 1, len(body)) if re.search(r"s_branch\s+" + re.escape(m.group(1)) + r"\b", body[k])]
        if back:
            seg = [x.split(";")[0].strip() for x in body[i:back[0] + 1]]
            seg = [x for x in seg if x and not x.startswith(".")]
            # two columns: per column four 3-way combines (v_max3) and the step group's maximum (one more v_max3)
            if sum(x.startswith("s_barrier") for x in seg) == 2 and sum("v_max3_f32" in x for x in seg) == 10 and not any("v_div_scale" in x for x in seg):
                loops.append(seg)
    assert loops, "no fast two-column loop found in the low-latency kernel's assembly"
    return min(loops, key=len)


def test_ll_no_spills_in_the_column_loop(ll_fast_loop):
    assert not [x for x in ll_fast_loop if x.startswith(("scratch_", "buffer_load", "buffer_store"))]
    assert not [x for x in ll_fast_loop if "accvgpr" in x]


def test_ll_hand_written_pairs_stay_adjacent(ll_fast_loop):
    n = lambda first, second: sum(1 for i, x in enumerate(ll_fast_loop) if x.startswith(first) and ll_fast_loop[i + 1].startswith(second))
    assert n("v_max3_f32", "v_sub_f32") >= 8                 # four cells x two columns
    assert n("v_cndmask_b32_e64", "v_subrev_f32") >= 8
    assert n("v_cndmask_b32_e64", "v_fma_f32") >= 8
    assert n("v_lshl_or_b32", "v_add_f32") >= 6


def test_ll_exchange_is_five_lds_instructions_per_column(ll_fast_loop):
    """the group winners a thread needs are consecutive entries: two 16-byte reads per exchange array, one write per thread
    (+ one per quad), the event record -- against 25 LDS instructions per column in the 8-wave form"""
    lds = [x.split()[0] for x in ll_fast_loop if x.startswith("ds_")]
    assert lds.count("ds_read_b128") == 10 and lds.count("ds_write_b64") == 4 and len(lds) == 14, lds
    # the quad merge of the skip group: DPP as an operand modifier of v_max_f32 / v_min_u32, no lane-swap moves
    assert sum(x.startswith("v_max_f32_dpp") for x in ll_fast_loop) == 4 and sum(x.startswith("v_min_u32_dpp") for x in ll_fast_loop) == 4
    assert not [x for x in ll_fast_loop if x.startswith("v_mov_b32_dpp")]
    assert not [x for x in ll_fast_loop if x.startswith("v_readfirstlane")]


def test_ll_columns_with_emissions_ahead_carry_the_recurrence_only(asm):
    """the eight-column trip of a read whose emissions were computed ahead (column_ahead): one 16-byte global load per column and
    thread, no model parameter in sight (no v_fma: the only FMAs of a column are the next-float probes' two), no scratch"""
    lines = asm["viterbi_ll_kernel"].split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*viterbi_ll_kernel\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    best = None
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m:
            continue
        back = [k for k in range(i + 1, len(body)) if re.search(r"s_c?branch\S*\s+" + re.escape(m.group(1)) + r"\b", body[k])]
        if back:
            seg = [x.split(";")[0].strip() for x in body[i:back[0] + 1]]
            seg = [x for x in seg if x and not x.startswith(".")]
            if sum(x.startswith("s_barrier") for x in seg) == 8 and sum(x.startswith("global_load_dwordx4") for x in seg) == 8:
                if best is None or len(seg) < len(best):
                    best = seg
    assert best, "no eight-column loop with one row load per column found"
    assert not [x for x in best if x.startswith("scratch_") or "accvgpr" in x]
    assert sum(x.startswith(("v_fma_f32", "v_fmac_f32")) for x in best) <= 8 * 2 + 4, [x for x in best if x.startswith("v_fma")][:6]
    assert sum(x.startswith("v_") for x in best) < 8 * 75          # 61 VALU per column today (137.5 with the emissions in place)
    assert sum(x.startswith("global_store_dword") for x in best) == 8


def test_ll_tie_and_rescan_paths_are_out_of_line(ll_fast_loop):
    assert len(ll_fast_loop) < 460, len(ll_fast_loop)       # 400 today for two columns


# ---- data hazards the compiler cannot see (tools/isa_lint.py) ----
import sys
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


def test_no_data_hazards_in_any_kernel(asm):
    """Every VALU-write -> DPP / v_permlane*_swap / v_readlane read, every transcendental -> dependent op, every EXEC write -> DPP
    in the final assembly of all four kernel files keeps its wait states -- including the ones whose two ends sit in different
    inline-asm statements, which LLVM's hazard recogniser does not connect (viterbi_kernel.hip swap1, fwbw_common.hpp)."""
    seen_lane_readers = 0
    for name, text in asm.items():
        bad, stats = isa_lint.lint(text)
        assert not bad, f"{name}: " + "; ".join(f"{b[1]} line {b[2]}: `{b[3]}` -> `{b[4]}` at {b[5]} wait states, needs {b[6]}" for b in bad[:5])
        seen_lane_readers += sum(s["lane_readers"] for s in stats.values())
    assert seen_lane_readers >= 100, seen_lane_readers      # the scan found the DPP / permlane / readlane sites (244 today)


def test_the_lint_catches_a_lane_swap_that_lost_its_wait_states(asm):
    """Commit 4e99858 wrote `s_nop 1` into the lane-swap asm because the distance between the asm select that produces its input
    and the DPP read had been kept only by what the compiler happened to schedule in between (the kernel BEFORE that commit also
    lints clean: the luck had held -- and whether it holds depends on the build, so the test does not lean on it).  Put the
    producer where the compiler would be free to put it -- a VALU write of the swapped register directly in front of a lane swap
    whose s_nop is gone -- and the lint must object; likewise for the permlane swaps of the forward-backward reductions."""
    m = re.search(r"\ts_nop 1\n\t(v_mov_b32_dpp v\d+, (v\d+) [^\n]*quad_perm[^\n]*)", asm["viterbi_kernel"])
    assert m, "no lane swap with its wait states in the Viterbi kernel?"
    vit = asm["viterbi_kernel"].replace(m.group(0), f"\tv_add_f32_e32 {m.group(2)}, {m.group(2)}, {m.group(2)}\n\t{m.group(1)}", 1)
    bad, _ = isa_lint.lint(vit)
    assert any(b[1] == "R1" and "quad_perm" in b[4] for b in bad), "a lane swap that lost its wait states went unnoticed"
    fb = asm["fwbw_scaled_kernel"]
    m = re.search(r"\t(v_\w+ (v\d+), [^\n]*)\n((?:\t[^\n]*\n){0,6}?)\t(v_permlane(?:16|32)_swap_b32 [^\n]*)", fb)
    assert m, "no v_permlane*_swap in the forward-backward kernels?"
    # a VALU write of a swapped register moved directly in front of the swap
    swap = m.group(4)
    reg = re.search(r"v\d+", swap).group(0)
    poisoned = fb.replace("\t" + swap, f"\tv_add_f32_e32 {reg}, {reg}, {reg}\n\t" + swap, 1)
    bad, _ = isa_lint.lint(poisoned)
    assert any(b[1] == "R1" and "permlane" in b[4] for b in bad)


def test_the_lint_rules_on_hand_written_snippets():
    def run(body):
        return isa_lint.lint("_Z1kv:\n" + "".join("\t" + l + "\n" for l in body) + "\ts_endpgm\n")[0]
    # R1: two wait states between a VALU write and a DPP read; s_nop 1 supplies both
    assert run(["v_add_f32_e32 v1, v2, v3", "v_mov_b32_dpp v4, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"])
    assert run(["v_add_f32_e32 v1, v2, v3", "s_nop 0", "v_mov_b32_dpp v4, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"])
    assert not run(["v_add_f32_e32 v1, v2, v3", "s_nop 1", "v_mov_b32_dpp v4, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"])
    assert not run(["v_add_f32_e32 v1, v2, v3", "s_mov_b32 s0, 0", "v_mul_f32_e32 v9, v8, v8", "v_add_f32_dpp v4, v1, v1 row_shr:1 row_mask:0xf bank_mask:0xf"])
    # ... a register pair written by a 64-bit op, read by a swap
    assert run(["v_lshlrev_b64 v[2:3], 1, v[4:5]", "v_permlane32_swap_b32 v3, v7"])
    # R2: a transcendental result needs one wait state before an ordinary VALU op reads it (another transcendental may follow at once)
    assert run(["v_exp_f32_e32 v1, v2", "v_add_f32_e32 v3, v1, v1"])
    assert not run(["v_exp_f32_e32 v1, v2", "v_mov_b32_e32 v9, v8", "v_add_f32_e32 v3, v1, v1"])
    assert not run(["v_rsq_f32_e32 v1, v2", "v_rcp_f32_e32 v3, v1"])
    # R3: five wait states between a VALU write of EXEC and a DPP op
    assert run(["v_cmpx_gt_f32_e32 v1, v2", "s_nop 3", "v_mov_b32_dpp v4, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"])
    assert not run(["v_cmpx_gt_f32_e32 v1, v2", "s_nop 4", "v_mov_b32_dpp v4, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"])


def test_no_scratch_traffic_in_the_forward_backward_event_loops(asm):
    """The rescaled sweeps spill (39 VGPRs in the forward kernel's prologue); none of it may sit in the per-event loop -- the loop
    with the block barrier -- of either sweep."""
    for pat in ("fwbw_forward_scaled_kernel", "fwbw_backward_scaled_kernel"):
        loops = isa_lint.event_loops(asm["fwbw_scaled_kernel"], pat)
        assert loops, pat
        # the event loop proper: the innermost loop around the barrier (an enclosing window loop contains it and the prologue)
        big = [t for t in loops if len(t[2]) > 150 and any(x.startswith("v_permlane") or "_dpp" in x for x in t[2])]
        assert big, (pat, [(t[1], len(t[2])) for t in loops])
        kernel, label, seg = min(big, key=lambda t: len(t[2]))
        spills = [x for x in seg if x.startswith("scratch_") or "accvgpr" in x]
        assert not spills, f"{pat} {label}: {spills[:4]}"


def test_forward_backward_event_loops_raise_their_priority_up_to_the_barrier(asm):
    """Round 6 (profiles/r06_fb_backward_session.md): the producer phase of an event runs at `s_setprio 2`, back to 0 right in front
    of the block barrier, in both rescaled sweeps -- 3-6 % of each sweep that no parity test would notice losing.  In the event loop:
    one raise and one drop, the drop after the raise, the barrier after the drop."""
    for pat in ("fwbw_forward_scaled_kernel", "fwbw_backward_scaled_kernel"):
        loops = isa_lint.event_loops(asm["fwbw_scaled_kernel"], pat)
        big = [t for t in loops if len(t[2]) > 150 and any(x.startswith("v_permlane") or "_dpp" in x for x in t[2])]
        kernel, label, seg = min(big, key=lambda t: len(t[2]))
        ops = [x.split()[0] + " " + " ".join(x.split()[1:2]) for x in seg if x.startswith(("s_setprio", "s_barrier"))]
        raises = [i for i, x in enumerate(ops) if x.startswith("s_setprio 2")]
        drops = [i for i, x in enumerate(ops) if x.startswith("s_setprio 0")]
        bars = [i for i, x in enumerate(ops) if x.startswith("s_barrier")]
        assert len(raises) >= 1 and len(drops) >= 1 and bars, (pat, ops)
        # (the loop is rotated by the compiler: what matters is the cyclic order raise -> drop -> barrier)
        r, d = raises[0], drops[0]
        b = next((i for i in bars if i > d), bars[0])
        cyc = lambda a, b_: (b_ - a) % len(ops)
        assert 0 < cyc(r, d) and cyc(d, b) == 1, (pat, ops)
