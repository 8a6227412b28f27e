#!/usr/bin/env python3
"""Synthesise the FAST5 fixtures under tests/golden/fast5/ (the reference ships none; HACKING.org:12 points at a
lab-private file).  For every read: <name>.events (the EventDetection table as text, "#nanocall-events") and
<name>.fast5 (the same table in the ONT HDF5 layout, written by tools/make_fast5 through the HDF5 C API).

  python tests/golden/make_fast5_fixtures.py          # needs `make -C tools make_fast5`

The events come from tests/oracle_pipeline.synth_ed_table: k-mer walks through the builtin models
(nanocall_amd.synth), an abasic hairpin plateau between the strands, affine level distortion for the EM to undo.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_pipeline as op  # noqa: E402

OUT = os.path.join(HERE, "fast5")
TOOL = os.path.join(ROOT, "tools", "make_fast5")

# name -> (synth_ed_table kwargs, sampling rate, read_id or None, make_fast5 flags)
FIXTURES = {
    "r73_2d_a": (dict(pore="r73", n_template=700, n_complement=600, seed=1, hairpin=8, complement_model="r73.c.p2.006.ont.model",
                      scale=1.04, shift=3.0, drift=0.01), 4000.0, "rid-1-2d", []),
    "r73_1d_b": (dict(pore="r73", n_template=500, n_complement=0, seed=2, scale=0.97, shift=-2.0), 4000.0, None, ["--no-read-id"]),
    "r73_short_c": (dict(pore="r73", n_template=30, n_complement=0, seed=3, lead=20, tail=20), 4000.0, "rid-3-short", []),
    "r9_2d_d": (dict(pore="r9", n_template=650, n_complement=700, seed=4, hairpin=9, scale=1.02, shift=-4.0), 4000.0, "rid-4-r9",
                ["--variance", "--ed-group", "001", "--read-number", "113"]),
    "r73_2d_e": (dict(pore="r73", n_template=620, n_complement=640, seed=5, hairpin=7, scale=0.95, shift=1.5, drift=-0.005), 3012.0,
                 "rid-5-2d", []),
    "r9_1d_f": (dict(pore="r9", n_template=900, n_complement=0, seed=6, scale=1.08, shift=6.0), 4000.0, "rid-6-r9", []),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "make_fast5"], check=True)
    for name, (kw, rate, rid, flags) in FIXTURES.items():
        kw = dict(kw)
        pore = kw.pop("pore")
        ed = op.synth_ed_table(pore, kw.pop("n_template"), kw.pop("n_complement"), kw.pop("seed"), rate=rate, **kw)
        ev = os.path.join(OUT, name + ".events")
        op.write_events_table(ev, ed, rate, rid)
        subprocess.run([TOOL] + flags + [ev, os.path.join(OUT, name + ".fast5")], check=True)
        print(name, len(ed), "events")


if __name__ == "__main__":
    main()
