"""The `nanocall` command line (nanocall_amd/bin/nanocall: FAST5 / event tables -> segmentation -> EM -> Viterbi ->
FASTA on the GPU) against the reference driver transcribed on the CPU oracle (tests/oracle_pipeline.py).

Two kinds of comparison:
  * no EM in the way (--no-train): the FASTA must be byte-identical to the oracle pipeline's, record for record;
  * with EM: the trained parameters are floats that agree with the oracle's within the EM tolerances of
    test_fwbw_gpu.py (not bit for bit), so the decode is checked TEACHER-FORCED -- the oracle decodes with exactly
    the parameters the CLI reports (--dump-params, hex floats) and that FASTA must be byte-identical -- while the
    control flow (model chosen, rounds) and the parameters themselves are compared with the free-running oracle.
BASELINE config 1 (single FAST5 read, template strand, builtin R7.3 model, 1 EM round) is test_config1_*.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_pipeline as op
from test_fast5_ingest import parse_events

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nanocall_amd", "bin", "nanocall")
G = os.path.join(ROOT, "tests", "golden", "fast5")


def run_cli(args, env=None, expect_rc=0):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([CLI] + args, capture_output=True, text=True, env=e, timeout=1200)
    assert p.returncode == expect_rc, f"rc={p.returncode}\n{p.stderr[-3000:]}"
    return p


def fixture_inputs(names, ext=".fast5"):
    """[(path the CLI is given, table for the oracle pipeline)] -- the table is parsed from the .events twin."""
    out = []
    for n in names:
        rate, rid, ed = parse_events(os.path.join(G, n + ".events"))
        if n == "r9_2d_d" and ext == ".fast5":         # stored as variance in the FAST5
            ed["stdv"] = np.sqrt(ed["stdv"] * ed["stdv"])
        if n == "r73_1d_b" and ext == ".fast5":        # written without a read_id attribute
            rid = ""
        out.append((os.path.join(G, n + ext), dict(sampling_rate=rate, read_id=rid, events=ed)))
    return out


def split_fasta(text):
    recs, name = {}, None
    for line in text.splitlines():
        if line.startswith(">"):
            name = line[1:]
            recs[name] = ""
        else:
            recs[name] += line
    return recs


def check_teacher_forced(o, inputs, fasta, dump):
    forced = {k: (v["model"], v["pm"], v["st"]) for k, v in dump.items()}
    exp, _, recs = op.run(o, inputs, forced=forced)
    assert fasta == exp, "FASTA differs from the oracle decode with the CLI's own parameters"
    for name, seq, info in recs:
        rid, _, st = name.rsplit(":", 2)
        got = dump[(rid, int(st))]["logp"]
        assert np.float32(got).tobytes() == np.float32(info["logp"]).tobytes(), (name, got, info["logp"])
    return recs


def check_free_running(o, inputs, dump, fasta):
    """Control flow and parameters against the oracle's own EM."""
    exp_fasta, reads, recs = op.run(o, inputs)
    assert len(recs) == len(dump)
    for name, seq, info in recs:
        rid, _, st = name.rsplit(":", 2)
        d = dump[(rid, int(st))]
        assert d["model"] == info["model"], (name, d["model"], info["model"])
        pm, got = info["pm"], d["pm"]
        assert abs(got[0] - pm[0]) <= 2e-4 * abs(pm[0]) and abs(got[4] - pm[4]) <= 2e-4 * abs(pm[4]), (name, got, pm)
        assert abs(got[1] - pm[1]) <= 2e-4 * 60 and abs(got[2] - pm[2]) <= 2e-4 * 60 / 5.0, (name, got, pm)
        assert abs(got[3] - pm[3]) <= 1e-3 * abs(pm[3]) and abs(got[5] - pm[5]) <= 1e-3 * abs(pm[5]), (name, got, pm)  # (free-running rounds against the fp32 ORACLE: 1e-3 -- 48 full-size jobs measured at most 6.9e-4 / 2.1e-4, profiles/r06_notes.md section 4; against a float64 evaluation the bound is 5e-4, tests/test_fullsize_gpu.py)
        assert np.allclose(d["st"], info["st"], rtol=5e-4, atol=0), (name, d["st"], info["st"])
        assert abs(d["logp"] - info["logp"]) <= 1e-4 * abs(info["logp"]), (name, d["logp"], info["logp"])
    for r in reads:
        for key, rounds in r.rounds.items():
            for st in (0, 1):
                d = dump.get((r.read_id, st))
                if d is not None and key[st] == d["model"] and (not r.together or all(key[s] == dump[(r.read_id, s)]["model"] for s in (0, 1))):
                    assert d["rounds"] == rounds, (r.read_id, key, d["rounds"], rounds)
                    assert abs(d["fit"] - r.fit[key]) <= 1e-4 * abs(r.fit[key])
    # the free-running FASTA is normally identical too; report how close it is rather than require it
    a, b = split_fasta(fasta), split_fasta(exp_fasta)
    assert a.keys() == b.keys()
    return sum(a[k] == b[k] for k in a), len(a)


def test_no_train_fasta_is_byte_identical_r73_2d_and_1d(tmp_path):
    """No EM: initial scaling (Fast5_Summary.hpp:223-278) straight into Viterbi with every candidate model."""
    names = ["r73_2d_a", "r73_1d_b", "r73_short_c", "r73_2d_e"]
    inputs = fixture_inputs(names)
    out = tmp_path / "out.fa"
    p = run_cli(["--pore", "r73", "--no-train", "-o", str(out)] + [i[0] for i in inputs])
    o = op.Opts(pore="r73", train=False)
    exp, reads, recs = op.run(o, inputs)
    assert [r.together for r in reads] == [False] * 4          # nanocall.cpp:1025-1038: no training -> no 2D scaling
    assert len(recs) == 5                                       # 2 + 1 + 0 + 2 strands
    assert out.read_text() == exp
    assert "rid-1-2d:r73_2d_a:0" in exp and "r73_1d_b:r73_1d_b:0" in exp
    # --1d: the hairpin is ignored, the whole read is one template strand
    p = run_cli(["--pore", "r73", "--no-train", "--1d"] + [i[0] for i in inputs])
    exp, _, recs = op.run(op.Opts(pore="r73", train=False, one_d=True), inputs)
    assert len(recs) == 3 and p.stdout == exp


def test_no_train_r9_preset_and_fasta_line_width(tmp_path):
    inputs = fixture_inputs(["r9_2d_d", "r9_1d_f"])
    p = run_cli(["--no-train", "--fasta-line-width", "60"] + [i[0] for i in inputs])      # --pore defaults to r9
    exp, reads, recs = op.run(op.Opts(pore="r9", train=False, fasta_line_width=60), inputs)
    assert len(recs) == 3 and p.stdout == exp
    assert max(len(l) for l in p.stdout.splitlines() if not l.startswith(">")) == 60


def test_config1_single_fast5_read_template_strand_one_em_round(tmp_path):
    """BASELINE config 1: one FAST5 read, template strand, builtin R7.3 model, 1 EM round, -t 1."""
    inputs = fixture_inputs(["r73_1d_b"])
    dump = tmp_path / "params.tsv"
    args = ["--pore", "r73", "--1d", "--scaling-max-rounds", "1", "-t", "1", "--dump-params", str(dump)]
    p = run_cli(args + [inputs[0][0]])
    o = op.Opts(pore="r73", one_d=True, scaling_max_rounds=1)
    d = op.read_dump(str(dump))
    assert list(d) == [("r73_1d_b", 0)] and d[("r73_1d_b", 0)]["rounds"] == 1
    check_teacher_forced(o, inputs, p.stdout, d)
    same, n = check_free_running(o, inputs, d, p.stdout)
    assert n == 1
    assert p.stdout.startswith(">r73_1d_b:r73_1d_b:0\n")


@pytest.mark.parametrize("pore,names,extra", [
    ("r73", ["r73_2d_a", "r73_1d_b", "r73_short_c"], []),                       # 2D scaling, drift trained (r73 preset)
    ("r9", ["r9_2d_d", "r9_1d_f"], []),                                          # r9 preset: no drift training
    ("r73", ["r73_2d_e", "r73_2d_a"], ["--single-strand-scaling"]),              # per-strand jobs, model selection per strand
])
def test_trained_decode_teacher_forced_and_em_within_tolerance(tmp_path, pore, names, extra):
    inputs = fixture_inputs(names)
    dump, out, stats = tmp_path / "params.tsv", tmp_path / "out.fa", tmp_path / "stats.tsv"
    args = ["--pore", pore, "--scaling-num-events", "120", "--scaling-max-rounds", "2", "--dump-params", str(dump), "-o", str(out),
            "--stats", str(stats), "-t", "4"] + extra
    run_cli(args + [i[0] for i in inputs])
    o = op.Opts(pore=pore, scaling_num_events=120, scaling_max_rounds=2, single_strand_scaling=bool(extra))
    d = op.read_dump(str(dump))
    fasta = out.read_text()
    check_teacher_forced(o, inputs, fasta, d)
    same, n = check_free_running(o, inputs, d, fasta)
    print(f"free-running FASTA records identical to the oracle's: {same}/{n}")
    # --stats: one row per input (skipped reads too), reference column layout (Fast5_Summary.hpp:460-502)
    rows = stats.read_text().splitlines()
    assert rows[0].split("\t")[:4] == ["file_name", "read_name", "num_ed_events", "abasic_level"] and len(rows[0].split("\t")) == 8 + 2 * 9
    assert len(rows) == 1 + len(names)
    for row, n_ in zip(rows[1:], names):
        f = row.split("\t")
        assert f[0] == n_ and len(f) == 26
        for st in (0, 1):
            key = (f[1], st)
            if key in d:
                assert f[8 + 9 * st] == d[key]["model"]
                assert f[9 + 9 * st] == f"{d[key]['pm'][0]:.5f}" and f[15 + 9 * st] == f"{d[key]['st'][0]:.5f}"
            else:
                assert f[8 + 9 * st] == "."


def test_fast5_and_event_table_inputs_fofn_and_directory_agree(tmp_path):
    names = ["r73_2d_a", "r73_2d_e"]
    a = run_cli(["--pore", "r73", "--no-train"] + [os.path.join(G, n + ".fast5") for n in names]).stdout
    b = run_cli(["--pore", "r73", "--no-train"] + [os.path.join(G, n + ".events") for n in names]).stdout
    # same events -> same sequences; the record names differ only by the base file name (".events" is not stripped)
    assert [s for s in split_fasta(a).values()] == [s for s in split_fasta(b).values()]
    fofn = tmp_path / "reads.fofn"
    fofn.write_text("".join(os.path.join(G, n + ".fast5") + "\n" for n in names) + "/nonexistent/file.fast5\n")
    assert run_cli(["--pore", "r73", "--no-train", str(fofn)]).stdout == a
    d = tmp_path / "dir"
    d.mkdir()
    for n in names:
        os.symlink(os.path.join(G, n + ".fast5"), d / (n + ".fast5"))
    (d / "notes.txt").write_text("not a read\n")
    got = split_fasta(run_cli(["--pore", "r73", "--no-train", str(d)]).stdout)
    assert got == split_fasta(a)                                 # readdir order is unspecified: compare per record


def test_two_contexts_shard_the_reads_and_gather_counters(tmp_path):
    """The multi-GPU host path on one GPU: two contexts / host threads on device 0 (NANOCALL_DEVICE_IDS=0,0), reads
    sharded by event count, output in input order -- byte-identical to the single-context run."""
    names = ["r73_2d_a", "r73_1d_b", "r73_short_c", "r73_2d_e", "r73_2d_a", "r73_1d_b"]
    files = [os.path.join(G, n + ".fast5") for n in names]
    base = ["--pore", "r73", "--scaling-num-events", "120", "--scaling-max-rounds", "2", "--log", "info"]
    one = run_cli(base + files)
    two = run_cli(base + files, env={"NANOCALL_DEVICE_IDS": "0,0"})
    assert one.stdout == two.stdout and one.stdout.count(">") == 8
    assert "devices=2" in two.stderr and "gathered_by=host_sum" in two.stderr
    c1 = [l for l in one.stderr.splitlines() if "counters reads=" in l][0]
    c2 = [l for l in two.stderr.splitlines() if "counters reads=" in l][0]
    pick = lambda l, k: l.split(k + "=")[1].split()[0]
    for k in ("reads", "bases", "strands_decoded", "events_decoded", "fb_windows"):
        assert pick(c1, k) == pick(c2, k), k
    # tiny chunks: several batches, same output
    three = run_cli(base + ["--chunk-events", "1500"] + files, env={"NANOCALL_DEVICE_IDS": "0,0"})
    assert three.stdout == one.stdout
    # the RCCL path itself (ncclCommInitAll over the pool's devices + one all-reduce) on the single device
    four = run_cli(base + files, env={"NCHMM_POOL_FORCE_RCCL": "1"})
    assert four.stdout == one.stdout
    c4 = [l for l in four.stderr.splitlines() if "counters reads=" in l][0]
    assert "gathered_by=rccl_allreduce" in c4 and pick(c4, "events_decoded") == pick(c1, "events_decoded")


def test_chunk_stages_side_by_side_give_the_serial_output(tmp_path):
    """Event loading / packing, the GPU stages and FASTA writing run side by side on consecutive chunks of reads
    (process_reads); `--serial-chunks` takes one chunk at a time through all three as rounds 1-4 did.  Same FASTA, same --stats,
    same --dump-params, with training, over many small chunks -- and with the process's destructors run (NANOCALL_FULL_EXIT: the
    default leaves through _Exit once everything is written)."""
    names = ["r73_2d_a", "r73_1d_b", "r73_short_c", "r73_2d_e", "r73_2d_a", "r73_1d_b", "r73_2d_e", "r73_2d_a"]
    files = [os.path.join(G, n + ".fast5") for n in names]
    base = ["--pore", "r73", "--scaling-num-events", "120", "--scaling-max-rounds", "2", "--chunk-events", "900", "-t", "4"]
    out = {}
    for tag, extra, env in (("serial", ["--serial-chunks"], {}), ("staged", [], {}), ("staged_full_exit", [], {"NANOCALL_FULL_EXIT": "1"})):
        fa, st, dp = tmp_path / f"{tag}.fa", tmp_path / f"{tag}.tsv", tmp_path / f"{tag}.params"
        r = run_cli(base + extra + ["-o", str(fa), "--stats", str(st), "--dump-params", str(dp)] + files, env=env)
        out[tag] = (fa.read_text(), st.read_text(), dp.read_text())
        assert out[tag][0].count(">") >= 10, r.stderr[-2000:]
    assert out["staged"] == out["serial"] and out["staged_full_exit"] == out["serial"]


def test_counter_gather_falls_back_to_the_host_sum_when_rccl_cannot_be_used(tmp_path):
    """nchmm_pool_counters' RCCL path (csrc/nchmm_pool.cpp: rccl_sum) must never cost a run its counters: librccl missing
    (dlopen fails), librccl without the entry points, and a library whose ncclCommInitAll returns an error all end in the
    host sum, with the same numbers and the same FASTA as the run that never tried.  NCHMM_RCCL_LIB names the library to load."""
    files = [os.path.join(G, n + ".fast5") for n in ("r73_2d_a", "r73_1d_b")]
    base = ["--pore", "r73", "--no-train", "--log", "info"] + files
    pick = lambda line, key: [w for w in line.split() if w.startswith(key + "=")][0]
    ref = run_cli(base)
    c_ref = [l for l in ref.stderr.splitlines() if "counters reads=" in l][0]
    assert "gathered_by=host_sum" in c_ref
    stub = tmp_path / "stub.c"
    stub.write_text("int ncclCommInitAll(void* c, int n, const int* d) { (void)c; (void)n; (void)d; return 2; }\n"      # ncclSystemError
                    "int ncclCommDestroy(void* c) { (void)c; return 0; }\n"
                    "int ncclAllReduce(const void* a, void* b, unsigned long n, int t, int o, void* c, void* s) { (void)a; (void)b; (void)n; (void)t; (void)o; (void)c; (void)s; return 2; }\n"
                    "int ncclGroupStart(void) { return 0; }\nint ncclGroupEnd(void) { return 0; }\n")
    empty = tmp_path / "empty.c"
    empty.write_text("int nothing_here(void) { return 0; }\n")
    for src in (stub, empty):
        subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(src.with_suffix(".so")), str(src)], check=True)
    for lib in (str(tmp_path / "no_such_librccl.so"), str(empty.with_suffix(".so")), str(stub.with_suffix(".so"))):
        got = run_cli(base, env={"NCHMM_POOL_FORCE_RCCL": "1", "NCHMM_RCCL_LIB": lib})
        assert got.stdout == ref.stdout, lib
        c = [l for l in got.stderr.splitlines() if "counters reads=" in l][0]
        assert "gathered_by=host_sum" in c, (lib, c)
        for k in ("reads", "events_decoded", "strands_decoded"):
            assert pick(c, k) == pick(c_ref, k), (lib, k)


def test_reader_processes_give_the_same_output(tmp_path):
    """--reader-procs K: K forked children read the event tables (file i from child i mod K) and stream them to the
    summary pass.  Same FASTA and --stats as the in-process reader, with more children than some have files, an
    unreadable file in the list, FAST5 and text tables mixed, and the default (>= 64 inputs: on)."""
    names = ["r73_2d_a", "r73_1d_b", "r73_short_c", "r73_2d_e", "r9_2d_d"]
    files = [os.path.join(G, n + ".fast5") for n in names] + [os.path.join(G, "r73_2d_a.events")]
    bad = tmp_path / "broken.fast5"
    bad.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)          # HDF5 signature, nothing behind it
    fofn = tmp_path / "reads.fofn"
    fofn.write_text("".join(f + "\n" for f in files[:3]) + str(bad) + "\n" + "".join(f + "\n" for f in files[3:]))
    base = ["--pore", "r73", "--no-train", "--log", "info", str(fofn)]
    ref = run_cli(base + ["--reader-procs", "0", "--stats", str(tmp_path / "s0.tsv")])
    assert "reader_procs=0" in ref.stderr and ref.stdout.count(">") >= 8
    for k in (2, 3, 16):
        got = run_cli(base + ["--reader-procs", str(k), "--stats", str(tmp_path / f"s{k}.tsv")])
        assert f"reader_procs={min(k, 6)}" in got.stderr        # never more children than files (the broken one is not a valid input)
        assert got.stdout == ref.stdout, k
        assert (tmp_path / f"s{k}.tsv").read_text() == (tmp_path / "s0.tsv").read_text(), k
    # default: on from 64 inputs (min(threads, 16) children)
    many = tmp_path / "many.fofn"
    many.write_text("".join(files[i % 5] + "\n" for i in range(70)))
    a = run_cli(["--pore", "r73", "--no-train", "--log", "info", "-t", "4", str(many)])
    b = run_cli(["--pore", "r73", "--no-train", "--log", "info", "-t", "4", "--reader-procs", "0", str(many)])
    assert "reader_procs=4 threads_at_fork=1" in a.stderr and "reader_procs=0" in b.stderr     # forked while single-threaded
    assert a.stdout == b.stdout and a.stdout.count(">") == 98     # 14 x (2 + 1 + 0 + 2 + 2) records


def test_no_train_random_ragged_reads_byte_identical_over_batches_and_contexts(tmp_path):
    """Thirty synthetic reads of random shape (1D and 2D, 120 to 5000 events per strand, both complement models, random affine
    distortion) as event tables: the FASTA of `nanocall --no-train` must equal the oracle pipeline's byte for byte, with the
    reads cut into many small batches, sharded over two contexts, and with reader processes on and off."""
    rng = np.random.default_rng(20261002)
    inputs = []
    for k in range(30):
        two_d = rng.random() < 0.7
        nt = int(rng.integers(120, 5000))
        nc = int(rng.integers(120, 5000)) if two_d else 0
        ed = op.synth_ed_table("r73", nt, nc, seed=100 + k, hairpin=int(rng.integers(6, 20)),
                               complement_model=("r73.c.p2.006.ont.model" if rng.random() < 0.5 else None),
                               scale=float(rng.uniform(0.93, 1.08)), shift=float(rng.uniform(-5, 5)), drift=float(rng.uniform(-0.01, 0.01)))
        rid = f"rnd-{k}" if k % 3 else None
        path = tmp_path / f"rnd_{k}.events"
        op.write_events_table(str(path), ed, 4000.0, rid)
        inputs.append((str(path), dict(sampling_rate=4000.0, read_id=rid or "", events=ed)))
    exp, reads, recs = op.run(op.Opts(pore="r73", train=False), inputs)
    assert len(recs) >= 35
    files = [i[0] for i in inputs]
    base = ["--pore", "r73", "--no-train"]
    assert run_cli(base + files).stdout == exp
    assert run_cli(base + ["--chunk-events", "9000", "--reader-procs", "3"] + files, env={"NANOCALL_DEVICE_IDS": "0,0"}).stdout == exp
    assert run_cli(base + ["--chunk-events", "2500", "-t", "8", "--reader-procs", "0"] + files).stdout == exp


def test_trained_random_reads_teacher_forced_byte_identical(tmp_path):
    """Twelve random 1D / 2D reads through the default pipeline (EM of both model pairs, selection, decode): whatever
    parameters and models the EM arrived at (--dump-params), the oracle decoding with exactly those must give the same
    FASTA byte for byte and the same path log-probabilities bit for bit."""
    rng = np.random.default_rng(77)
    inputs = []
    for k in range(12):
        two_d = k % 4 != 3
        ed = op.synth_ed_table("r73", int(rng.integers(400, 2500)), int(rng.integers(400, 2500)) if two_d else 0, seed=300 + k,
                               hairpin=int(rng.integers(6, 16)), complement_model=("r73.c.p2.006.ont.model" if k % 2 else None),
                               scale=float(rng.uniform(0.94, 1.07)), shift=float(rng.uniform(-4, 4)), drift=float(rng.uniform(-0.008, 0.008)))
        path = tmp_path / f"tr_{k}.events"
        op.write_events_table(str(path), ed, 4000.0, f"tr-{k}")
        inputs.append((str(path), dict(sampling_rate=4000.0, read_id=f"tr-{k}", events=ed)))
    dump = tmp_path / "params.tsv"
    p = run_cli(["--pore", "r73", "-t", "4", "--chunk-events", "7000", "--dump-params", str(dump)] + [i[0] for i in inputs])
    d = op.read_dump(str(dump))
    assert len(d) >= 12
    recs = check_teacher_forced(op.Opts(pore="r73"), inputs, p.stdout, d)
    assert len(recs) == len(d)


def test_fofn_from_stdin_and_duplicate_inputs():
    """`-` reads the list of file names from standard input (nanocall.cpp:233-257); a file named twice is two reads."""
    names = ["r73_2d_a", "r73_1d_b", "r73_2d_a"]
    files = [os.path.join(G, n + ".fast5") for n in names]
    want = run_cli(["--pore", "r73", "--no-train"] + files).stdout
    e = dict(os.environ)
    p = subprocess.run([CLI, "--pore", "r73", "--no-train", "-"], input="".join(f + "\n" for f in files), capture_output=True, text=True,
                       env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout == want and want.count(">") == 5


def test_validity_pass_over_many_inputs_runs_in_child_processes(tmp_path):
    """From 256 candidates on, the is_valid_file pass over a directory / fofn / argument list is spread over forked children.
    Same file list (order, ignored entries) and output as the in-process check."""
    names = ["r73_2d_a", "r73_1d_b", "r73_short_c", "r73_2d_e"]
    d = tmp_path / "reads"
    d.mkdir()
    lines = []
    for k in range(300):
        if k % 37 == 5:
            p = d / f"junk_{k:03d}.txt"
            p.write_text("not a read\n")
        else:
            p = d / f"read_{k:03d}.fast5"
            os.symlink(os.path.join(G, names[k % 4] + ".fast5"), p)
        lines.append(str(p))
    fofn = tmp_path / "reads.fofn"
    fofn.write_text("".join(l + "\n" for l in lines))
    base = ["--pore", "r73", "--no-train", "--log", "info", "-t", "4"]
    a = run_cli(base + [str(fofn)])
    b = run_cli(base + ["--reader-procs", "0", str(fofn)])
    added = lambda p: [l for l in p.stderr.splitlines() if "adding input file" in l]
    n_junk = sum(1 for k in range(300) if k % 37 == 5)
    assert a.stdout == b.stdout and added(a) == added(b) and len(added(a)) == 300 - n_junk
    # a candidate that kills the child checking it: ignored with a warning, everything else as before
    e = run_cli(base + [str(fofn)], env={"NANOCALL_TEST_VALIDATE_ABORT": "read_123.fast5"})
    assert "read_123.fast5: the process checking this file died; file ignored" in e.stderr
    assert [l for l in added(e)] == [l for l in added(a) if "read_123.fast5" not in l]
    c = run_cli(base + [str(d)])                       # the directory itself: readdir order, compare as sets
    assert split_fasta(c.stdout).keys() == split_fasta(a.stdout).keys()
    assert sum("ignoring file" in l for l in c.stderr.splitlines()) == n_junk


def test_a_file_that_kills_its_reader_process_is_skipped_not_reopened(tmp_path):
    """libhdf5 can crash on a corrupt file.  With reader processes that costs one child: the file is skipped
    with a warning and never opened in the process that holds the GPU, the dead child's later files are read in-process,
    everything else is unchanged.  (NANOCALL_TEST_READER_ABORT makes a child abort on a file name, as such a crash would.)"""
    import shutil
    names = ["r73_2d_a", "r73_1d_b", "r73_2d_e", "r73_2d_a", "r73_2d_e"]
    files = []
    for k, n in enumerate(names):
        dst = tmp_path / (f"killer_{k}.fast5" if k == 1 else f"read_{k}.fast5")
        shutil.copy(os.path.join(G, n + ".fast5"), dst)
        files.append(str(dst))
    base = ["--pore", "r73", "--no-train", "--log", "info"]
    want = run_cli(base + ["--reader-procs", "0"] + [f for k, f in enumerate(files) if k != 1])
    got = run_cli(base + ["--reader-procs", "2"] + files, env={"NANOCALL_TEST_READER_ABORT": "killer_"})
    assert got.stdout == want.stdout and got.stdout.count(">") == 8
    assert "killer_1.fast5: the reader process died on this file; read skipped" in got.stderr
    assert "reader_procs=2" in got.stderr


def test_option_errors_and_help():
    assert "Required argument missing" in run_cli([], expect_rc=1).stderr
    assert "unknown pore type" in run_cli(["--pore", "r10", os.path.join(G, "r73_1d_b.fast5")], expect_rc=1).stderr
    assert "not both" in run_cli(["--train", "--no-train", os.path.join(G, "r73_1d_b.fast5")], expect_rc=1).stderr
    assert "Couldn't find match" in run_cli(["--frobnicate", "x"], expect_rc=1).stderr
    h = run_cli(["--help"]).stdout
    for opt in ("--pore", "--1d", "--scaling-max-rounds", "--pr-skip", "--pr-stay", "--fasta-line-width", "--stats", "--no-train-transitions"):
        assert opt in h
    assert "cannot open" in run_cli(["--pore", "r73", "/nonexistent/x.fast5"], expect_rc=1).stderr


def _kmer(j):
    return "".join("ACGT"[(j >> (2 * (5 - i))) & 3] for i in range(6))


def test_custom_model_files_equal_the_builtin_models(tmp_path):
    """-m strand:file (Pore_Model operator>>, Pore_Model.hpp:251-287; init_models nanocall.cpp:99-153): the three r73 tables
    written as text (rows shuffled, header + comment lines) decode to the same sequences as --pore r73."""
    import nanocall_amd as na
    rng = np.random.default_rng(3)
    args = []
    for fname, name, strand in (("a_c_p1.model", "r73.c.p1", 1), ("b_c_p2.model", "r73.c.p2", 1), ("c_t.model", "r73.t", 0)):   # same order as the builtin names
        t = na.builtin_model(name)
        with open(tmp_path / fname, "w") as f:
            f.write("#model_file written by the test\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
            for j in rng.permutation(4096):
                f.write(_kmer(int(j)) + "\t" + "\t".join(f"{v:.9g}" for v in t[j]) + "\n")
        args += ["-m", f"{strand}:{tmp_path / fname}"]
    files = [os.path.join(G, n + ".fast5") for n in ("r73_2d_a", "r73_1d_b")]
    base = ["--scaling-num-events", "120", "--scaling-max-rounds", "2"]
    a = run_cli(["--pore", "r73"] + base + files).stdout
    b = run_cli(["--pore", "r73"] + base + args + files).stdout
    assert a == b and a.count(">") == 3
    # the same through --model-fofn; and a strand given models on one side only is refused (nanocall.cpp:130-135)
    fofn = tmp_path / "models.fofn"
    fofn.write_text("".join(f"{s}:{tmp_path / f}\n" for f, s in (("a_c_p1.model", 1), ("b_c_p2.model", 1), ("c_t.model", 0))))
    assert run_cli(["--pore", "r73", "--model-fofn", str(fofn)] + base + files).stdout == a
    assert "models were specified only for strand" in run_cli(["-m", f"0:{tmp_path / 'c_t.model'}"] + files, expect_rc=1).stderr
    # gzip-compressed model files read like plain ones (the reference opens models through zstr, nanocall.cpp:113-120)
    import gzip
    gz_args = []
    for fname, strand in (("a_c_p1.model", 1), ("b_c_p2.model", 1), ("c_t.model", 0)):
        with open(tmp_path / fname, "rb") as f, gzip.open(tmp_path / (fname + ".gz"), "wb") as g:
            g.write(f.read())
        gz_args += ["-m", f"{strand}:{tmp_path / (fname + '.gz')}"]
    assert run_cli(["--pore", "r73"] + base + gz_args + files).stdout == a
    (tmp_path / "cut.model.gz").write_bytes((tmp_path / "c_t.model.gz").read_bytes()[:2000])
    assert "damaged gzip stream" in run_cli(["-m", f"2:{tmp_path / 'cut.model.gz'}"] + files, expect_rc=1).stderr


def test_non_default_transition_and_segmentation_options_no_train():
    inputs = fixture_inputs(["r73_2d_a", "r73_2d_e"])
    args = ["--pore", "r73", "--no-train", "--pr-stay", "0.12", "--pr-skip", "0.25", "--min-ed-events", "20", "--trim-ed-sq-start", "30",
            "--trim-ed-sq-end", "70", "--trim-ed-hp-start", "40", "--trim-ed-hp-end", "60", "--max-ed-events", "1300"]
    p = run_cli(args + [i[0] for i in inputs])
    o = op.Opts(pore="r73", train=False, pr_stay=0.12, pr_skip=0.25, min_ed_events=20, trim=(30, 70, 40, 60), max_ed_events=1300)
    exp, reads, recs = op.run(o, inputs)
    assert p.stdout == exp and len(recs) >= 3
    assert [r.num_ed_events for r in reads] == [1300, 1300]          # both tables are longer than the cap


@pytest.mark.parametrize("extra,kw", [
    (["--no-train-transitions"], dict(train_transitions=False)),
    (["--no-train-scaling", "--single-strand-scaling"], dict(train_scaling=False, single_strand_scaling=True)),
    (["--train-drift", "0"], dict(train_drift=False)),
])
def test_partial_training_modes_teacher_forced(tmp_path, extra, kw):
    inputs = fixture_inputs(["r73_2d_a", "r73_1d_b"])
    dump = tmp_path / "params.tsv"
    base = ["--pore", "r73", "--scaling-num-events", "120", "--scaling-max-rounds", "2", "--dump-params", str(dump)]
    p = run_cli(base + extra + [i[0] for i in inputs])
    o = op.Opts(pore="r73", scaling_num_events=120, scaling_max_rounds=2, **kw)
    d = op.read_dump(str(dump))
    check_teacher_forced(o, inputs, p.stdout, d)
    for v in d.values():
        if "--no-train-transitions" in extra:
            assert v["st"].tolist() == [np.float32(0.1), np.float32(0.3)]        # transitions stay at --pr-stay / --pr-skip
        if "--no-train-scaling" in extra:
            assert v["pm"][2] == 0 and v["pm"][3] == 1 and v["pm"][4] == 1 and v["pm"][5] == 1   # only the initial scale / shift
        if "--train-drift" in extra:
            assert v["pm"][2] == 0
    same, n = check_free_running(o, inputs, d, p.stdout)


# ---------------------------------------------------------------------------------------------------------------
# one worker process per GPU (nanocall.cpp: fan_out) -- on the one GPU a test box has, every worker on device 0
# ---------------------------------------------------------------------------------------------------------------
def _worker_inputs(tmp_path, n=26):
    """ragged synthetic reads (1D and 2D, a few hundred to a few thousand events per strand) + two of the FAST5 fixtures"""
    rng = np.random.default_rng(77)
    files = []
    for k in range(n):
        n0 = int(rng.integers(150, 2600))
        n1 = int(rng.integers(150, 2600)) if k % 3 else 0
        ed = op.synth_ed_table("r73", n0, n1, seed=300 + k, hairpin=8 if n1 else 0, complement_model="r73.c.p1.006.ont.model" if k % 2 else "r73.c.p2.006.ont.model",
                               scale=1.0 + 0.01 * (k % 5), shift=float(k % 7) - 3.0, drift=0.002 * (k % 3))
        path = tmp_path / f"w_{k:02d}.events"
        op.write_events_table(str(path), ed, 4000.0, f"w-{k}")
        files.append(str(path))
    files[5:5] = [os.path.join(G, "r73_2d_a.fast5"), os.path.join(G, "r73_short_c.fast5")]
    return files


def _pick(line, key):
    return [w for w in line.split() if w.startswith(key + "=")][0].split("=")[1]


@pytest.mark.parametrize("workers", [1, 2, 4])
def test_worker_processes_give_the_single_process_output(tmp_path, workers):
    """`nanocall --gpus N` starts one worker process per GPU before anything touches the HIP runtime, gives each an LPT share of the
    input files and writes their records in input order (nanocall.cpp:282,611,859-866).  N = 1, 2, 4 workers sharing GPU 0
    (NANOCALL_WORKER_DEVICES): FASTA, --stats and --dump-params byte-identical to the single-process run, with training, over
    several chunks per worker; the counters add up; the parent never creates a device context."""
    files = _worker_inputs(tmp_path)
    base = ["--pore", "r73", "--scaling-num-events", "120", "--scaling-max-rounds", "2", "--chunk-events", "6000", "-t", "8"]
    outs = {}
    for tag, env in (("single", {}), ("workers", {"NANOCALL_WORKER_DEVICES": ",".join(["0"] * workers)})):
        fa, st, dp = tmp_path / f"{tag}.fa", tmp_path / f"{tag}.tsv", tmp_path / f"{tag}.params"
        r = run_cli(base + ["-o", str(fa), "--stats", str(st), "--dump-params", str(dp)] + files, env=env)
        outs[tag] = (fa.read_text(), st.read_text(), dp.read_text(), r.stderr)
    assert outs["single"][0].count(">") >= 30
    for i, what in enumerate(("FASTA", "--stats", "--dump-params")):
        assert outs["workers"][i] == outs["single"][i], what
    err = outs["workers"][3]
    assert f"workers={workers} devices=[{','.join(['0'] * workers)}]" in err
    c1 = [l for l in outs["single"][3].splitlines() if "counters reads=" in l][-1]
    cw = [l for l in err.splitlines() if "counters reads=" in l and "worker_counters" not in l][-1]
    for k in ("reads", "bases", "strands_decoded", "events_decoded", "fb_windows", "fb_event_rounds"):
        assert _pick(c1, k) == _pick(cw, k), k
    assert _pick(cw, "workers") == str(workers) and _pick(cw, "gathered_by") == "host_sum"      # (workers on one device cannot form a communicator)
    # every worker reported its stages; the parent's own stages are the merge and the input list -- no device stage
    assert sum("stage_wall_secs" in l and l.startswith("= nanocall info: worker ") for l in err.splitlines()) == workers
    parent_stages = [l for l in err.splitlines() if l.startswith("= nanocall info: stage_wall_secs")][-1]
    assert "merge_output_s=" in parent_stages and "device_init_s" not in parent_stages and "basecalling_total_s" not in parent_stages
    # stdout instead of -o: the same bytes
    r = run_cli(base + files, env={"NANOCALL_WORKER_DEVICES": ",".join(["0"] * workers)})
    assert r.stdout == outs["single"][0]


def test_one_worker_process_reduces_its_counters_through_rccl(tmp_path):
    """The cross-process reduction itself -- ncclGetUniqueId in rank 0, the id relayed through the parent's pipes,
    ncclCommInitRank + one ncclAllReduce in every worker (nchmm_rccl_unique_id / nchmm_counters_allreduce) -- on the one device a
    test box has: one worker, NCHMM_POOL_FORCE_RCCL=1.  And its fall-back when librccl cannot be loaded."""
    files = [os.path.join(G, n + ".fast5") for n in ("r73_2d_a", "r73_1d_b", "r73_2d_e")]
    base = ["--pore", "r73", "--no-train"] + files
    ref = run_cli(base)
    c_ref = [l for l in ref.stderr.splitlines() if "counters reads=" in l][-1]
    got = run_cli(base, env={"NANOCALL_WORKER_DEVICES": "0", "NCHMM_POOL_FORCE_RCCL": "1"})
    c = [l for l in got.stderr.splitlines() if "counters reads=" in l and "worker_counters" not in l][-1]
    assert got.stdout == ref.stdout and "counters_through=rccl_allreduce" in got.stderr
    assert _pick(c, "gathered_by") == "rccl_allreduce", c
    for k in ("reads", "bases", "strands_decoded", "events_decoded"):
        assert _pick(c, k) == _pick(c_ref, k), k
    got = run_cli(base, env={"NANOCALL_WORKER_DEVICES": "0", "NCHMM_POOL_FORCE_RCCL": "1", "NCHMM_RCCL_LIB": str(tmp_path / "no_such_librccl.so")})
    c = [l for l in got.stderr.splitlines() if "counters reads=" in l and "worker_counters" not in l][-1]
    assert got.stdout == ref.stdout and _pick(c, "gathered_by") == "host_sum" and _pick(c, "events_decoded") == _pick(c_ref, "events_decoded")


def test_a_worker_process_that_dies_is_reported_and_the_others_records_are_all_there(tmp_path):
    """A worker that fails (here: aborts before its first record, NANOCALL_TEST_WORKER_ABORT) is reported with the number of reads it
    leaves undone and is not started again; the run ends with a failure code and the records of every other worker's reads, in
    input order."""
    files = _worker_inputs(tmp_path, n=12)
    base = ["--pore", "r73", "--no-train"] + files
    ref = split_fasta(run_cli(base).stdout)
    bad = run_cli(base, env={"NANOCALL_WORKER_DEVICES": "0,0,0", "NANOCALL_TEST_WORKER_ABORT": "1"}, expect_rc=1)
    assert "worker 1 (device 0) failed with signal 6" in bad.stderr and "reads are not in the output" in bad.stderr
    got = split_fasta(bad.stdout)
    assert 0 < len(got) < len(ref) and all(ref[k] == v for k, v in got.items())
    assert list(got) == [k for k in ref if k in got]                 # input order kept
    lost = int(bad.stderr.split("failed with signal 6: ")[1].split()[0])
    assert lost >= 1 and "workers=3" in bad.stderr


def test_more_worker_processes_than_input_files(tmp_path):
    """Four workers, two files: two workers have nothing to decode -- they still answer the counter reduction, and the output is the
    single-process one."""
    files = [os.path.join(G, n + ".fast5") for n in ("r73_2d_a", "r73_1d_b")]
    base = ["--pore", "r73", "--no-train", "--stats", str(tmp_path / "s.tsv")] + files
    ref = run_cli(base)
    st_ref = (tmp_path / "s.tsv").read_text()
    got = run_cli(base, env={"NANOCALL_WORKER_DEVICES": "0,0,0,0"})
    assert got.stdout == ref.stdout and (tmp_path / "s.tsv").read_text() == st_ref
    assert "files_per_worker=[1,1,0,0]" in got.stderr
    c = [l for l in got.stderr.splitlines() if "counters reads=" in l and "worker_counters" not in l][-1]
    c_ref = [l for l in ref.stderr.splitlines() if "counters reads=" in l][-1]
    assert _pick(c, "events_decoded") == _pick(c_ref, "events_decoded") and _pick(c, "workers") == "4"


def test_gpus_option_beyond_the_visible_devices_is_refused():
    p = run_cli(["--pore", "r73", "--no-train", "--gpus", "64", os.path.join(G, "r73_1d_b.fast5")], expect_rc=1)
    assert "--gpus 64 requested but only" in p.stderr


def _adversarial_ed_table(rng, k):
    """an EventDetection table whose strands are NOT draws from the models they will be decoded with (tests/adversarial.py kinds),
    with a hairpin plateau between them in two reads of three"""
    import adversarial
    import nanocall_amd as na
    from nanocall_amd import api
    t_tab, c_tab = na.builtin_model("r73.t"), na.builtin_model("r73.c.p1" if k % 2 else "r73.c.p2")
    params = (float(rng.uniform(0.95, 1.06)), float(rng.uniform(-4, 4)), 0.0, 1.0, 1.0, 1.0)
    parts = []
    kinds = []
    for s, tab in enumerate((t_tab, c_tab) if k % 3 else (t_tab,)):
        kind = adversarial.KINDS[int(rng.integers(len(adversarial.KINDS)))]
        kinds.append(kind)
        n = int(rng.integers(200, 3500))
        mean, stdv, _ = adversarial.events(kind, tab, params, n, seed=9100 + 10 * k + s, other_table=t_tab if s else c_tab)
        parts.append((mean, stdv))
        if s == 0 and k % 3:
            hp = int(rng.integers(6, 20))
            parts.append((np.float32(rng.normal(float(np.asarray(t_tab).reshape(4096, 4)[:, 0].max()) + 40.0, 2.0, hp)), np.float32(rng.uniform(0.5, 2.0, hp))))
    lead = (np.float32(rng.normal(60, 5, 60)), np.float32(rng.uniform(0.5, 2.0, 60)))
    parts = [lead] + parts + [lead]
    mean = np.concatenate([p[0] for p in parts]); stdv = np.concatenate([p[1] for p in parts])
    ed = np.zeros(len(mean), api.ED_DTYPE)
    ed["mean"], ed["stdv"] = mean, stdv
    ed["length"] = rng.integers(8, 120, len(mean))
    ed["start"] = 1000 + np.cumsum(ed["length"]) - ed["length"]
    return ed, kinds


def test_no_train_adversarial_reads_byte_identical(tmp_path):
    """Twenty-four reads whose strands are runs of identical events, spikes, uniform levels, another model's stream, heavy-tailed or
    zero stdv, abasic stretches (tests/adversarial.py) -- segmentation (stretches at the abasic level inside a strand are islands
    too), initial scalings, every candidate model decoded, the better one chosen, bases and FASTA: byte-identical to the oracle
    pipeline, in one batch and in many small ones."""
    rng = np.random.default_rng(20260608)
    inputs, all_kinds = [], set()
    for k in range(24):
        ed, kinds = _adversarial_ed_table(rng, k)
        all_kinds.update(kinds)
        path = tmp_path / f"adv_{k:02d}.events"
        op.write_events_table(str(path), ed, 4000.0, f"adv-{k}")
        inputs.append((str(path), dict(sampling_rate=4000.0, read_id=f"adv-{k}", events=ed)))
    assert len(all_kinds) >= 6
    exp, reads, recs = op.run(op.Opts(pore="r73", train=False), inputs)
    assert len(recs) >= 20
    files = [i[0] for i in inputs]
    base = ["--pore", "r73", "--no-train"]
    assert run_cli(base + files).stdout == exp
    assert run_cli(base + ["--chunk-events", "4000", "-t", "4"] + files).stdout == exp


def test_trained_adversarial_reads_teacher_forced_byte_identical(tmp_path):
    """Twelve of the adversarial reads through the DEFAULT pipeline (EM of every model pair, selection, decode): the EM may take the
    parameters anywhere (scale 2, var 8 on a strand with an abasic stretch), and whatever it reports (--dump-params, hex floats)
    the oracle decoding with exactly those parameters must give the same FASTA byte for byte and the same path log-probabilities
    bit for bit -- the Viterbi path under parameters no well-behaved read produces (true division outside the validated range)."""
    rng = np.random.default_rng(20260609)
    inputs = []
    for k in range(12):
        ed, kinds = _adversarial_ed_table(rng, k)
        path = tmp_path / f"advt_{k:02d}.events"
        op.write_events_table(str(path), ed, 4000.0, f"advt-{k}")
        inputs.append((str(path), dict(sampling_rate=4000.0, read_id=f"advt-{k}", events=ed)))
    dump = tmp_path / "params.tsv"
    p = run_cli(["--pore", "r73", "-t", "4", "--chunk-events", "6000", "--dump-params", str(dump)] + [i[0] for i in inputs])
    d = op.read_dump(str(dump))
    assert len(d) >= 10
    for v in d.values():
        assert np.isfinite(v["pm"]).all() and np.isfinite(v["st"]).all() and np.isfinite(v["logp"]), v
    recs = check_teacher_forced(op.Opts(pore="r73"), inputs, p.stdout, d)
    assert len(recs) == len(d)
