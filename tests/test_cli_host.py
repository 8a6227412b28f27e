"""The part of the `nanocall` command line that runs before any device is touched: option checks and init_models
(nanocall.cpp:97-153, 995-1059).  No GPU needed: where there is none the run ends with the "no usable GPU" message
AFTER the models were loaded, which is all these tests look at."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nanocall_amd", "bin", "nanocall")
FAST5 = os.path.join(ROOT, "tests", "golden", "fast5", "r73_short_c.fast5")

pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="nanocall_amd/bin/nanocall not built (__graft_entry__.build())")


def run(args):
    return subprocess.run([CLI] + args, capture_output=True, text=True, timeout=600)


def kmer(j):
    return "".join("ACGT"[(j >> (2 * (5 - i))) & 3] for i in range(6))


def write_models(tmp_path):
    import nanocall_amd as na
    rng = np.random.default_rng(11)
    out = []
    for fname, name, strand in (("a_c_p1.model", "r73.c.p1", 1), ("b_c_p2.model", "r73.c.p2", 1), ("c_t.model", "r73.t", 0)):
        t = na.builtin_model(name)
        text = "#model_file written by the test\nkmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n"
        text += "".join(kmer(int(j)) + "\t" + "\t".join(f"{v:.9g}" for v in t[j]) + "\n" for j in rng.permutation(4096))
        (tmp_path / fname).write_text(text)
        with gzip.open(tmp_path / (fname + ".gz"), "wb") as g:
            g.write(text.encode())
        out.append((strand, fname))
    return out


def loaded(stderr):
    """{file name without .gz: (strand, mean, stdv)} from the 'loaded module' log lines (nanocall.cpp:147-150)"""
    pat = re.compile(r"loaded module \[(.*?)\] for strand \[(\d)\] statistics \[mean=([^,]+), stdv=([^\]]+)\]")
    return {os.path.basename(m.group(1)).replace(".gz", ""): m.groups()[1:] for m in pat.finditer(stderr)}


def test_gzip_model_files_and_fofn_load_like_plain_ones(tmp_path):
    """The reference opens model files and the model fofn through zstr (nanocall.cpp:122,144): plain or gzip."""
    models = write_models(tmp_path)
    plain = run(sum((["-m", f"{s}:{tmp_path / f}"] for s, f in models), []) + [FAST5])
    packed = run(sum((["-m", f"{s}:{tmp_path / (f + '.gz')}"] for s, f in models), []) + [FAST5])
    a, b = loaded(plain.stderr), loaded(packed.stderr)
    assert len(a) == 3 and a == b, (plain.stderr[-1500:], packed.stderr[-1500:])
    fofn = tmp_path / "models.fofn.gz"
    with gzip.open(fofn, "wb") as g:
        g.write("".join(f"{s}:{tmp_path / (f + '.gz')}\n" for s, f in models).encode())
    assert loaded(run(["--model-fofn", str(fofn), FAST5]).stderr) == a


def test_damaged_and_one_sided_model_arguments_are_refused(tmp_path):
    models = write_models(tmp_path)
    (tmp_path / "cut.model.gz").write_bytes((tmp_path / "c_t.model.gz").read_bytes()[:2000])
    p = run(["-m", f"2:{tmp_path / 'cut.model.gz'}", FAST5])
    assert p.returncode == 1 and "damaged gzip stream" in p.stderr
    p = run(["-m", f"2:{tmp_path / 'absent.model'}", FAST5])
    assert p.returncode == 1 and "cannot open model file" in p.stderr
    (tmp_path / "short.model").write_text("kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\nAAAAAA\t1\t1\t1\t1\n")
    assert run(["-m", f"2:{tmp_path / 'short.model'}", FAST5]).returncode == 1
    # models for one strand only (nanocall.cpp:130-135)
    p = run(["-m", f"0:{tmp_path / models[2][1]}", FAST5])
    assert p.returncode == 1 and "models were specified only for strand" in p.stderr


def test_transition_file_and_fast5_write_back_are_refused_with_a_reason():
    p = run(["-s", "/dev/null", FAST5])
    assert p.returncode == 1 and "--trans" in p.stderr
    p = run(["--write-fast5", FAST5])
    assert p.returncode == 1 and "write-fast5" in p.stderr
