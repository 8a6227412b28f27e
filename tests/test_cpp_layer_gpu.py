"""The C++ host layer (include/nanocall_amd/nanocall_amd.hpp) through tools/run-viterbi -- the reference's
own debug harness shape (src/nanocall/run-viterbi.cpp): model / transitions / events text files in,
base sequence out.  Must equal the oracle's sequence byte for byte."""
import os
import subprocess

import numpy as np
import pytest

import nanocall_amd as na
import nc_oracle as oracle
from helpers import ragged_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "run-viterbi")


def _kmer(j):
    return "".join("ACGT"[(j >> (2 * (5 - i))) & 3] for i in range(6))


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools")], check=True, capture_output=True)
    return TOOL


def test_run_viterbi_text_formats(tool, tmp_path, r73t):
    params = (1.03, 1.5, 0.0, 1.1, 0.95, 1.2)
    scaled = na.model_scale(na.model_load(r73t), params)
    # model file rows in shuffled order with a header and a comment, as Pore_Model operator>> accepts
    order = np.random.default_rng(0).permutation(4096)
    with open(tmp_path / "model.tsv", "w") as f:
        f.write("#comment\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
        for j in order:
            # 9 significant digits round-trip a float32 exactly
            f.write(_kmer(int(j)) + "\t" + "\t".join(f"{v:.9g}" for v in scaled[j, :4]) + "\n")
    off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, [700], first_read=77)
    with open(tmp_path / "events.tsv", "w") as f:
        for m, s, t in zip(mean, stdv, start):
            f.write(f"{m:.9g}\t{s:.9g}\t{t:.9g}\t0.01\n")
    # expected: the oracle on a model LOADED from those four columns (the tool does load_from_vector on the file)
    om = oracle.Model(scaled[:, :4].copy())
    ot = oracle.Transitions(0.3, 0.1)
    st, mv, lp = oracle.viterbi(om, ot, cm, sd, ls)
    exp_seq = oracle.base_seq(st, mv)
    r = subprocess.run([tool, "-p", str(tmp_path / "model.tsv"), "-e", str(tmp_path / "events.tsv"),
                        "--pr-skip", "0.3", "--pr-stay", "0.1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == exp_seq
    # same through a transitions FILE (to_v order, i.e. by source state) and FASTA output
    rp, idx, w = ot.to_csr()
    with open(tmp_path / "trans.tsv", "w") as f:
        for i in range(4096):
            for a in range(rp[i], rp[i + 1]):
                f.write(f"{_kmer(i)}\t{_kmer(int(idx[a]))}\t{w[a]:.9g}\n")
    r = subprocess.run([tool, "-p", str(tmp_path / "model.tsv"), "-e", str(tmp_path / "events.tsv"),
                        "-s", str(tmp_path / "trans.tsv"), "--fasta", "read1:file:0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == oracle.write_fasta("read1:file:0", exp_seq, 80)


def test_run_viterbi_reports_errors_without_crashing(tool, tmp_path):
    (tmp_path / "m.tsv").write_text("AAAAAA\t1\t1\t1\t1\n")
    (tmp_path / "e.tsv").write_text("60 1 0 0.01\n")
    r = subprocess.run([tool, "-p", str(tmp_path / "m.tsv"), "-e", str(tmp_path / "e.tsv")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "unexpected number of states" in r.stderr


def test_run_fwbw_prints_the_middle_event_posteriors(tmp_path, r73t):
    """tools/run-fwbw, the shape of the reference's run-fwbw (src/nanocall/run-fwbw.cpp:71-88): k-mers whose
    posterior at the middle event is >= 0.1, highest first.  Against the oracle's alpha/beta: same k-mers in the
    same order, posteriors to 1e-4, log Pr(data) to 1e-4; the -o dump holds the matrices."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "run-fwbw"], check=True, capture_output=True)
    params = (0.97, -1.0, 0.0, 1.05, 1.02, 0.9)
    scaled = na.model_scale(na.model_load(r73t), params)
    with open(tmp_path / "model.tsv", "w") as f:
        f.write("kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
        for j in range(4096):
            f.write(_kmer(j) + "\t" + "\t".join(f"{v:.9g}" for v in scaled[j, :4]) + "\n")
    off, mean, stdv, start, cm, sd, ls = ragged_batch(r73t, [61], first_read=31)
    with open(tmp_path / "events.tsv", "w") as f:
        for m, s, t in zip(mean, stdv, start):
            f.write(f"{m:.9g}\t{s:.9g}\t{t:.9g}\t0.01\n")
    om, ot = oracle.Model(scaled[:, :4].copy()), oracle.Transitions(0.25, 0.15)
    lpd, al, be = oracle.fwbw(om, ot, cm, sd, ls)
    post = np.exp(al[30].astype(np.float64) + be[30] - lpd)
    exp = sorted(((p, j) for j, p in enumerate(post) if p >= 0.1), reverse=True)
    r = subprocess.run([os.path.join(ROOT, "tools", "run-fwbw"), "-p", str(tmp_path / "model.tsv"), "-e", str(tmp_path / "events.tsv"),
                        "--pr-skip", "0.25", "--pr-stay", "0.15", "-o", str(tmp_path / "mat.tsv")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [ln.split("\t") for ln in r.stdout.strip().splitlines()]
    assert [g[0] for g in got] == [_kmer(j) for _, j in exp]
    assert np.allclose([float(g[1]) for g in got], [p for p, _ in exp], rtol=1e-4)
    assert abs(float(r.stderr.split()[-1]) - float(lpd)) <= 1e-4 * abs(float(lpd))
    mat = np.loadtxt(tmp_path / "mat.tsv")
    assert mat.shape == (61 * 4096, 4)
    assert np.allclose(mat[:, 2].reshape(61, 4096), al, rtol=1e-4, atol=1e-3)
    assert np.allclose(mat[:, 3].reshape(61, 4096), be, rtol=1e-4, atol=1e-3)


def test_reference_call_sites_train_loop_and_basecall_strand(tmp_path):
    """tests/boundary/reference_call_sites.cpp: the 2D round loop (nanocall.cpp:360-426) and basecall_strand (:645-690)
    as the reference writes them, compiled against nanocall_amd.hpp.  This drives the C++
    Parameter_Trainer::train_one_round / Pore_Model::scale / Viterbi::fill mirrors: control flow and fits against the
    oracle's train_one_round loop (EM tolerances), the decode teacher-forced (the oracle with the program's own final
    parameters) bit for bit."""
    import oracle_pipeline as op
    exe = tmp_path / "reference_call_sites"
    libdir = os.path.join(ROOT, "nanocall_amd")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "boundary", "reference_call_sites.cpp"), "-L", libdir, "-lnanocall_hip",
                        "-Wl,-rpath," + libdir, "-pthread", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    names = ["r73.t.006.ont.model", "r73.c.p2.006.ont.model"]
    tabs = [na.builtin_model(n) for n in names]
    from nanocall_amd import synth
    evs = [synth.generate(tabs[0], 1, 420, first_read=900), synth.generate(tabs[1], 1, 380, first_read=901)]
    paths = []
    for st, e in enumerate(evs):
        p = tmp_path / f"ev{st}.tsv"
        with open(p, "w") as f:
            for m, s, t, ln in zip(e["mean"][0], e["stdv"][0], e["start"][0], e["length"][0]):
                f.write(f"{m:.9g}\t{s:.9g}\t{t:.9g}\t{ln:.9g}\n")
        paths.append(str(p))
    for train_drift in (1, 0):
        r = subprocess.run([str(exe)] + paths + names + ["120", "2", str(train_drift)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.splitlines()
        rounds = [l.split() for l in lines if l.startswith("round ")]
        result = [l.split() for l in lines if l.startswith("result ")][0]
        hx = lambda xs: np.float32([float.fromhex(x) for x in xs])
        # the oracle's loop on the same windows
        o = op.Opts(pore="r73", scaling_num_events=120, scaling_max_rounds=2, train_drift=train_drift)
        rd = op.Read()
        rd.events = [(e["mean"][0], na.events_prepare(e["mean"][0], e["stdv"][0], None, 0.0)[1], e["start"][0], e["length"][0]) for e in evs]
        win, wst = op._windows(o, rd, [0, 1])
        key = (names[0], names[1])
        pm, st, fit, rnd = op.train_job(o, dict(zip(names, tabs)), win, wst, key, [1, 0, 0, 1, 1, 1], [0.1, 0.3, 0.1, 0.3])
        assert int(result[12]) == rnd and len(rounds) in (rnd, rnd + 1)
        got_pm, got_st, got_fit = hx(result[1:7]), hx(result[7:11]), np.float32(float.fromhex(result[11]))
        assert abs(got_fit - fit) <= 1e-4 * abs(fit)
        assert abs(got_pm[0] - pm[0]) <= 2e-4 * abs(pm[0]) and abs(got_pm[4] - pm[4]) <= 2e-4 * abs(pm[4])
        assert abs(got_pm[1] - pm[1]) <= 2e-4 * 60 and abs(got_pm[2] - pm[2]) <= 2e-4 * 60 / 5.0
        assert abs(got_pm[3] - pm[3]) <= 1.5e-3 * abs(pm[3]) and abs(got_pm[5] - pm[5]) <= 1.5e-3 * abs(pm[5])  # (free-running rounds against the fp32 ORACLE, whose own var / var_sd sit 1.5e-4 per round from a float64 evaluation and compound: this read measures 1.0-1.5e-3; against float64 the bound is 5e-4, tests/test_fullsize_gpu.py)
        assert np.allclose(got_st, st, rtol=1e-3, atol=0)
        # ... and against the REAL-NUMBER answer: the same free-running rounds in float64 (tools/fb_truth.py --cpp-layer-read).  The fp32
        # oracle ends 4e-4 from it on p_skip of the template strand here, this library 2e-4: both inside 5e-4, 6e-4 apart.
        import json
        truth = json.load(open(os.path.join(ROOT, "tests", "golden", "cpp_layer_read900_truth64.json")))["rounds_by_train_drift"][str(train_drift)]
        fits = [t["fit"] for t in truth[:rnd]]
        assert rnd >= 1 and all(b >= a for a, b in zip(fits, fits[1:]))            # (no roll-back: the result is the last executed round's)
        t_pm, t_st = np.array(truth[rnd - 1]["pm"]), np.array(truth[rnd - 1]["st"])
        assert np.allclose(got_st, t_st, rtol=5e-4, atol=0) and np.allclose(st, t_st, rtol=5e-4, atol=0), (got_st, st, t_st)
        # (var / var_sd: the closed forms of Parameter_Trainer.hpp:406-426 are differences of sums 1000 x their result, formed from FLOAT
        # products as the reference forms them -- that finish alone is worth 2e-4 per round on 240 events, in the oracle and here alike)
        assert abs(got_pm[3] - t_pm[3]) <= 1e-3 * abs(t_pm[3]) and abs(got_pm[5] - t_pm[5]) <= 1e-3 * abs(t_pm[5]), (got_pm, t_pm)
        assert abs(pm[3] - t_pm[3]) <= 1e-3 * abs(t_pm[3]) and abs(pm[5] - t_pm[5]) <= 1e-3 * abs(t_pm[5]), ("oracle", pm, t_pm)
        assert abs(got_pm[0] - t_pm[0]) <= 2e-4 * abs(t_pm[0]) and abs(got_pm[4] - t_pm[4]) <= 2e-4 * abs(t_pm[4])
        if not train_drift:
            assert got_pm[2] == 0.0
        # basecall_strand, teacher-forced
        for s in (0, 1):
            f = [l.split() for l in lines if l.startswith(f"strand {s} ")][0]
            om = oracle.Model(tabs[s], got_pm)
            ot = oracle.Transitions(float(got_st[2 * s + 1]), float(got_st[2 * s]))
            cm, sd, ls = oracle.events_prepare(rd.events[s][0], rd.events[s][1], rd.events[s][2], float(got_pm[2]))
            states, mv, lp = oracle.viterbi(om, ot, cm, sd, ls)
            assert np.float32(float.fromhex(f[2])).tobytes() == np.float32(lp).tobytes()
            assert f[3] == oracle.base_seq(states, mv)


def test_header_swap_call_sites_from_many_threads_are_exact(tool):
    """The reference's call shapes kept as they are, from pfor-like worker threads (tools/bench_cpp_layer, bench_train_threads):
    Viterbi::fill on one strand per call with the thread's own copy of the model -- every strand's states and path probability
    equal to what fill_batch decoded -- and Parameter_Trainer::train_one_round on one read per call -- every fit equal to the
    single-threaded run (nchmm_viterbi_strand / nchmm_fwbw_windows combine the calls; the kernels treat reads independently)."""
    import json
    subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "bench_cpp_layer", "bench_train_threads"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(ROOT, "tools", "bench_cpp_layer"), "160", "700", "48"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["mismatches_vs_fill_batch"] == 0 and "48 worker threads" in out["what"], out
    r = subprocess.run([os.path.join(ROOT, "tools", "bench_train_threads"), "300", "1", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    assert len(lines) == 2 and all(l["fits_differing_from_first_run"] == 0 for l in lines), lines
