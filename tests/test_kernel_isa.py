"""Structure of the compiled Viterbi forward kernel that its speed depends on and that no parity test would notice losing
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


@pytest.fixture(scope="module")
def fast_loop(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "viterbi_kernel.s"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-fno-slp-vectorize", "-S", "--cuda-device-only"]   # = csrc/Makefile
    subprocess.run([HIPCC] + flags + ["-o", str(out), os.path.join(CSRC, "viterbi_kernel.hip")], check=True, capture_output=True, timeout=600)
    lines = open(out).read().split("\n")
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
