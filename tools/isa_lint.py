"""Data-hazard lint over gfx950 assembly (hipcc -S --cuda-device-only).

Why it exists: the kernels' hot loops are written with inline-asm statements (v_cndmask_b32_e64 on SGPR masks, paired half-rate +
full-rate instructions, DPP lane swaps, v_permlane{16,32}_swap).  LLVM's hazard recogniser inserts the wait states the hardware
needs between ordinary instructions, but it does not look INSIDE an asm statement: a VGPR written by an asm select and read by
an asm DPP move two lines later is invisible to it.  The shipped kernel once had the required distance only by scheduling luck
(fixed in commit 4e99858 by writing the s_nop into the asm).  This lint re-derives the distances from the final assembly, so
that luck cannot silently run out when the surrounding code changes.

Rules (CDNA3/CDNA4 ISA guide, "manually inserted wait states"; LLVM GCNHazardRecognizer for gfx940/gfx950):
  R1  VALU writes a VGPR  ->  a DPP instruction or v_permlane{16,32}_swap reads it: >= 2 wait states
      VALU writes a VGPR  ->  v_readlane / v_readfirstlane reads it: >= 1 wait state (what LLVM itself guarantees on gfx940+,
      VALUWriteVGPRReadlaneRead; the compiler's own schedule sits at exactly 1 in several places)
  R2  a transcendental VALU op (exp, log, rcp, rsq, sqrt, sin, cos) writes a VGPR -> a NON-transcendental VALU op reads it:
      >= 1 wait state
  R3  VALU writes EXEC (v_cmpx*, or any VALU with exec as destination) -> a DPP instruction: >= 5 wait states
A wait state is one issued instruction of the wave; `s_nop N` counts N + 1.  The scan is linear over the function body
(labels do not reset it: the fall-through path is the short one; a taken branch only adds cycles).

lint(asm_text) -> list of (kernel, rule, line_no, writer, reader, distance, needed)
"""
import re
import sys

TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f16|f32|f64|legacy_f32)")
LANE_SWAP = re.compile(r"^v_permlane\d+_swap")
LANE_READ = re.compile(r"^(v_readlane|v_readfirstlane)")


def _vgprs(tok):
    """VGPR numbers named by one operand token: v7, v[4:7], -v3, |v3|, v3.l ..."""
    out = set()
    for m in re.finditer(r"(?<![a-z_0-9])v(\d+)(?![\d\[])", tok):
        out.add(int(m.group(1)))
    for m in re.finditer(r"(?<![a-z_0-9])v\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _split_operands(rest):
    # operands are comma separated; modifiers (quad_perm:[1,0,3,2] row_mask:0xf ...) follow the last operand after a space
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


class Inst:
    __slots__ = ("line_no", "text", "mn", "ops", "is_valu", "is_dpp", "writes", "reads", "writes_exec", "nop")

    def __init__(self, line_no, text):
        self.line_no, self.text = line_no, text
        parts = text.split(None, 1)
        self.mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        self.is_dpp = self.mn.endswith("_dpp") or bool(re.search(r"\b(quad_perm|row_shl|row_shr|row_ror|row_bcast|row_mirror|row_half_mirror|wave_shl|wave_shr|row_newbcast|row_share|row_xmask)\b", rest))
        # cut trailing modifiers off the last operand
        ops = _split_operands(rest)
        if ops:
            ops[-1] = re.split(r"\s+(?=[a-z_]+:|[a-z_]+\b(?!\[))", ops[-1], maxsplit=1)[0] if not ops[-1].startswith("v[") else ops[-1].split(" ")[0]
        self.ops = ops
        self.is_valu = self.mn.startswith("v_") and not self.mn.startswith("v_nop")
        self.nop = 0
        if self.mn == "s_nop":
            try:
                self.nop = int(ops[0], 0) + 1
            except Exception:
                self.nop = 1
        self.writes, self.reads, self.writes_exec = set(), set(), False
        if self.is_valu and ops:
            dst = ops[0]
            n_dst = 1
            if re.match(r"^v_(permlane\d+_swap|swap)", self.mn):     # both operands read and written
                self.writes = _vgprs(ops[0]) | _vgprs(ops[1])
                self.reads = set(self.writes)
                return
            if self.mn.startswith(("v_cmpx",)):
                self.writes_exec = True
                n_dst = 0 if not re.match(r"^(vcc|s\[|s\d|exec)", dst) else 1
            if re.match(r"^exec", dst):
                self.writes_exec = True
            if self.mn.startswith(("v_cmp_", "v_cmpx_")) or re.match(r"^(vcc|s\[|s\d|exec)", dst):
                pass                                                # scalar destination
            else:
                self.writes = _vgprs(dst)
            # v_div_scale / v_add_co / v_mad_u64_u32 ... carry a second (scalar) destination
            srcs = ops[n_dst:] if n_dst else ops
            if len(ops) > 1 and re.match(r"^(vcc|s\[\d+:\d+\])$", ops[1]) and re.match(r"^v_(div_scale|add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|mad_u64_u32|mad_i64_i32)", self.mn):
                srcs = ops[2:]
            for o in srcs:
                self.reads |= _vgprs(o)
            # accumulating forms read their destination: v_fmac, v_mac, DPP `old`, v_writelane, v_cndmask? (no)
            if re.match(r"^v_(fmac|mac|dot\w*c|pk_fmac)", self.mn) or self.is_dpp:
                self.reads |= self.writes
        elif ops:
            # non-VALU readers of VGPRs are not subject to R1-R3
            pass


def parse_kernels(asm_text):
    """{kernel_symbol: [Inst, ...]} for every .amdhsa kernel function in the file"""
    lines = asm_text.split("\n")
    kernels = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
        if m:
            name = m.group(1)
            body = []
            j = i + 1
            while j < len(lines) and not re.match(r"^\s*\.(end_amdhsa_kernel|section|size)\b", lines[j]) and "s_endpgm" not in lines[j]:
                t = lines[j].split(";")[0].strip()
                if t and not t.startswith(".") and not t.endswith(":") and not t.startswith("//"):
                    body.append(Inst(j + 1, t))
                j += 1
            if any(x.mn.startswith(("v_", "s_")) for x in body):
                kernels[name] = body
            i = j
        i += 1
    return kernels


def lint_body(name, body):
    bad = []
    # for each VGPR: (index of its last VALU writer, that writer) ; distance = sum of wait states of the instructions in between
    last_write = {}
    last_exec_write = None
    ws_prefix = [0]
    for ins in body:
        ws_prefix.append(ws_prefix[-1] + (ins.nop if ins.nop else 1))
    for k, ins in enumerate(body):
        def dist(w):
            return ws_prefix[k] - ws_prefix[w + 1]          # wait states strictly between writer w and this instruction
        if ins.is_valu:
            need = 2 if (ins.is_dpp or LANE_SWAP.match(ins.mn)) else (1 if LANE_READ.match(ins.mn) else 0)
            for r in sorted(ins.reads):
                if r in last_write:
                    w, wi = last_write[r]
                    d = dist(w)
                    if d < need:
                        bad.append((name, "R1", ins.line_no, wi.text, ins.text, d, need))
                    if TRANS.match(wi.mn) and not TRANS.match(ins.mn) and d < 1:
                        bad.append((name, "R2", ins.line_no, wi.text, ins.text, d, 1))
            if ins.is_dpp and last_exec_write is not None and dist(last_exec_write[0]) < 5:
                bad.append((name, "R3", ins.line_no, last_exec_write[1].text, ins.text, dist(last_exec_write[0]), 5))
            for r in ins.writes:
                last_write[r] = (k, ins)
            if ins.writes_exec:
                last_exec_write = (k, ins)
    return bad


def lint(asm_text):
    out = []
    stats = {}
    for name, body in parse_kernels(asm_text).items():
        out += lint_body(name, body)
        stats[name] = dict(instructions=len(body),
                           lane_readers=sum(1 for x in body if x.is_valu and (x.is_dpp or LANE_SWAP.match(x.mn) or LANE_READ.match(x.mn))),
                           transcendentals=sum(1 for x in body if TRANS.match(x.mn)))
    return out, stats


def event_loops(asm_text, kernel_pattern, marker=r"s_barrier"):
    """The loops (label ... last backward branch to it) of the kernels matching kernel_pattern that contain `marker`:
    list of (kernel, label, [instruction text])."""
    lines = asm_text.split("\n")
    found = []
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m and re.search(kernel_pattern, m.group(1)):
            j = i + 1
            while j < len(lines) and "s_endpgm" not in lines[j]:
                j += 1
            body = lines[i:j]
            for a, l in enumerate(body):
                lm = re.match(r"^(\.LBB\d+_\d+):", l)
                if not lm:
                    continue
                back = [k for k in range(a + 1, len(body)) if re.search(r"s_cbranch\S*\s+" + re.escape(lm.group(1)) + r"\b|s_branch\s+" + re.escape(lm.group(1)) + r"\b", body[k])]
                if back:
                    seg = [x.split(";")[0].strip() for x in body[a:back[-1] + 1]]
                    seg = [x for x in seg if x and not x.startswith(".")]
                    if any(re.search(marker, x) for x in seg):
                        found.append((m.group(1), lm.group(1), seg))
            i = j
        i += 1
    return found


if __name__ == "__main__":
    for f in sys.argv[1:]:
        bad, stats = lint(open(f).read())
        for k, s in stats.items():
            print(f"{f}: {k[:60]}: {s}")
        for b in bad:
            print("  VIOLATION", b)
        print(f"{f}: {len(bad)} violation(s)")
