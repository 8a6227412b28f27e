"""FAST5 (HDF5) ingest through the C ABI (include/nanocall_fast5.h) against the committed fixtures: every
tests/golden/fast5/<name>.fast5 must yield exactly the EventDetection table of <name>.events (the text the same
generator wrote, tests/golden/make_fast5_fixtures.py).  CPU only: HDF5 is loaded at run time from the image."""
import os

import numpy as np
import pytest

from nanocall_amd import api

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fast5")
pytestmark = pytest.mark.skipif(not api.fast5_available(), reason="libhdf5 not found on this machine")


def parse_events(path):
    rate, rid, rows = None, "", []
    for line in open(path):
        if line.startswith("#"):
            f = line[1:].split()
            if f and f[0] == "sampling_rate":
                rate = float(f[1])
            elif f and f[0] == "read_id":
                rid = f[1]
            continue
        f = line.split()
        rows.append((float(f[0]), float(f[1]), int(f[2]), int(f[3])))
    return rate, rid, np.array(rows, api.ED_DTYPE)


@pytest.mark.parametrize("name,group,read_name,variance", [
    ("r73_2d_a", "000", "Read_7", False), ("r73_1d_b", "000", "Read_7", False), ("r73_short_c", "000", "Read_7", False),
    ("r9_2d_d", "001", "Read_113", True), ("r73_2d_e", "000", "Read_7", False), ("r9_1d_f", "000", "Read_7", False)])
def test_fixture_tables_round_trip(name, group, read_name, variance):
    rate, rid, ed = parse_events(os.path.join(G, name + ".events"))
    r = api.fast5_load(os.path.join(G, name + ".fast5"))          # "" = smallest EventDetection group present
    assert r["have_sampling_rate"] and r["sampling_rate"] == rate
    assert r["have_events"] and r["ed_group"] == group and r["read_name"] == read_name
    assert r["read_id"] == ("" if name == "r73_1d_b" else rid)    # written with --no-read-id
    if variance:                                                   # stored as variance: the reader takes the square root
        ed["stdv"] = np.sqrt(ed["stdv"] * ed["stdv"])
    assert np.array_equal(r["events"], ed)
    # asking for the group by name gives the same; asking for an absent group gives "no events", not an error
    assert np.array_equal(api.fast5_load(os.path.join(G, name + ".fast5"), group)["events"], ed)
    assert not api.fast5_load(os.path.join(G, name + ".fast5"), "042")["have_events"]


def test_validity_and_errors(tmp_path):
    assert api.fast5_is_valid_file(os.path.join(G, "r73_2d_a.fast5"))
    assert not api.fast5_is_valid_file(os.path.join(G, "r73_2d_a.events"))
    assert not api.fast5_is_valid_file(str(tmp_path / "missing.fast5"))
    with pytest.raises(api.NchmmError) as e:
        api.fast5_load(str(tmp_path / "missing.fast5"))
    assert e.value.code == -7
    junk = tmp_path / "junk.fast5"
    junk.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)           # HDF5 signature, nothing behind it
    with pytest.raises(api.NchmmError):
        api.fast5_load(str(junk))


def test_fast5_reader_under_sanitizers(tmp_path):
    """tools/asan_fast5.cpp: nchmm_fast5.cpp built with AddressSanitizer + UBSan (libhdf5 itself is the system's build)
    over the fixtures, truncations of them at a dozen lengths, copies with the last third zeroed, a text table and a
    missing path, each with five EventDetection group arguments.  Damaged files must be refused or load without events --
    no out-of-bounds access, no leak.  (Random byte flips are left out on purpose: flipped object-header bytes make
    libhdf5 1.10.6 itself read out of bounds in H5O_attr_shared_decode, which no caller can prevent.)"""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "tools"), "asan-fast5"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -lasan" in (r.stderr + r.stdout):
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    fixtures = sorted(glob.glob(os.path.join(root, "tests", "golden", "fast5", "*.fast5")))
    paths = list(fixtures) + [os.path.join(root, "tests", "golden", "fast5", "r73_2d_a.events"), str(tmp_path / "missing.fast5")]
    for f in fixtures:
        b = open(f, "rb").read()
        n = len(b)
        for k, cut in enumerate([0, 7, 8, 9, 511, 512, n // 7, n // 3, n // 2, (2 * n) // 3, n - 4096, n - 1]):
            if 0 <= cut <= n:
                p = tmp_path / f"{os.path.basename(f)}.cut{k}.fast5"
                p.write_bytes(b[:cut])
                paths.append(str(p))
        p = tmp_path / (os.path.basename(f) + ".zeroed_tail.fast5")
        p.write_bytes(b[:2 * n // 3] + bytes(n - 2 * n // 3))
        paths.append(str(p))
    r = subprocess.run([os.path.join(root, "tools", "asan_fast5")] + paths, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fast5 reader under ASan/UBSan: ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
