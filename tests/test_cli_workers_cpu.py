"""The one-worker-process-per-GPU machinery of the `nanocall` command line (nanocall_amd/cli/nanocall.cpp: fan_out) WITHOUT a GPU:
`--no-train --no-basecall --stats` is host work only (segmentation and initial scalings, Fast5_Summary.hpp:138-319), so the
partition of the input files, the fork before anything touches the HIP runtime, the framed records up the pipes, the merge in
input order, the host-sum of the counters and the report of a worker that died all run here, on CPU -- the counterpart, for the
command line, of the world-size-2 gloo test of bench.py's ranks (tests/test_shard_dist.py).  With a GPU the same machinery carries
FASTA and --dump-params too (tests/test_cli_gpu.py)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nanocall_amd", "bin", "nanocall")
G = os.path.join(ROOT, "tests", "golden", "fast5")

pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="the command line is not built")


def _run(args, env=None, expect_rc=0):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([CLI] + args, capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == expect_rc, f"rc={p.returncode}\n{p.stderr[-3000:]}"
    return p


def _inputs(tmp_path, n=40):
    import oracle_pipeline as op
    rng = np.random.default_rng(11)
    files = []
    for k in range(n):
        n0 = int(rng.integers(150, 1500))
        n1 = int(rng.integers(150, 1500)) if k % 3 else 0
        ed = op.synth_ed_table("r73", n0, n1, seed=700 + k, hairpin=8 if n1 else 0)
        path = tmp_path / f"c_{k:02d}.events"
        op.write_events_table(str(path), ed, 4000.0, f"c-{k}")
        files.append(str(path))
    files[3:3] = [os.path.join(G, "r73_2d_a.fast5"), os.path.join(G, "r73_short_c.fast5"), os.path.join(G, "r9_1d_f.fast5")]
    return files


@pytest.mark.parametrize("workers", [1, 2, 5])
def test_worker_processes_write_the_stats_of_the_single_process_run(tmp_path, workers):
    files = _inputs(tmp_path)
    base = ["--pore", "r73", "--no-train", "--no-basecall", "-t", "4"]
    one = _run(base + ["--stats", str(tmp_path / "one.tsv")] + files)
    assert "devices=0" in one.stderr                                  # no device was asked for
    got = _run(base + ["--stats", str(tmp_path / "w.tsv")] + files, env={"NANOCALL_WORKER_DEVICES": ",".join(["0"] * workers)})
    assert (tmp_path / "w.tsv").read_text() == (tmp_path / "one.tsv").read_text()
    assert (tmp_path / "one.tsv").read_text().count("\n") == len(files) + 1
    assert f"workers={workers} " in got.stderr and "counters_through=host_sum" in got.stderr
    shares = [int(x) for x in got.stderr.split("files_per_worker=[")[1].split("]")[0].split(",")]
    assert len(shares) == workers and sum(shares) == len(files) and min(shares) >= len(files) // workers - 6      # longest-first by file size
    assert sum(l.startswith("= nanocall info: worker ") and "stage_wall_secs" in l for l in got.stderr.splitlines()) == workers
    assert got.stdout == "" and one.stdout == ""


def test_a_worker_that_dies_costs_its_own_reads_only(tmp_path):
    files = _inputs(tmp_path, n=12)
    base = ["--pore", "r73", "--no-train", "--no-basecall"]
    _run(base + ["--stats", str(tmp_path / "one.tsv")] + files)
    bad = _run(base + ["--stats", str(tmp_path / "w.tsv")] + files, env={"NANOCALL_WORKER_DEVICES": "0,0,0", "NANOCALL_TEST_WORKER_ABORT": "1"}, expect_rc=1)
    assert "worker 1 (device 0) failed with signal 6" in bad.stderr
    lost = int(bad.stderr.split("failed with signal 6: ")[1].split()[0])
    one, w = (tmp_path / "one.tsv").read_text().splitlines(), (tmp_path / "w.tsv").read_text().splitlines()
    assert len(w) == len(one) - lost and all(l in one for l in w)
    assert [l for l in one if l in w] == w                            # input order kept


def test_more_workers_than_files_and_one_file(tmp_path):
    f = [os.path.join(G, "r73_2d_a.fast5"), os.path.join(G, "r73_1d_b.fast5")]
    base = ["--pore", "r73", "--no-train", "--no-basecall"]
    _run(base + ["--stats", str(tmp_path / "one.tsv")] + f)
    got = _run(base + ["--stats", str(tmp_path / "w.tsv")] + f, env={"NANOCALL_WORKER_DEVICES": "0,0,0,0"})
    assert "files_per_worker=[1,1,0,0]" in got.stderr and (tmp_path / "w.tsv").read_text() == (tmp_path / "one.tsv").read_text()


def test_worker_machinery_under_sanitizers():
    """tools/asan_cli.py: the command line's translation unit built with -fsanitize=address,undefined, then with -fsanitize=thread (host C++ only), and the
    scenarios above run against it -- pipes, frames, the merge, a worker that aborts, the ranks with nothing to do."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_cli.py")], capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and any(f"cannot find -l{lib}" in (r.stderr + r.stdout) for lib in ("asan", "ubsan", "tsan")):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0 and r.stdout.count("no sanitizer report") == 2, (r.stdout[-3000:], r.stderr[-3000:])      # ASan+UBSan, TSan
