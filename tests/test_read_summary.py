"""Strand segmentation / event filter / initial scaling (the arithmetic of Fast5_Summary.hpp) in the product
library against the oracle's restatement, on randomised EventDetection tables: bit-identical.

Parity status: both follow Fast5_Summary.hpp:138-370,528-571,653-745; fast5::EventDetection_Event_Entry and
alg::mean_stdv_of live in un-vendored submodules, so the entry layout and the mean/stdv formula are unpinned."""
import numpy as np
import pytest

import nanocall_amd as na
from nanocall_amd import api
import nc_oracle as oracle


def synth_ed(rng, n, hairpin=None, level=60.0, abasic=140.0, extra_islands=(), zero_stdv=True):
    ed = np.zeros(n, api.ED_DTYPE)
    ed["mean"] = rng.normal(level, 6.0, n)
    ed["stdv"] = rng.uniform(0.3, 2.5, n)
    ed["stdv"][rng.random(n) < 0.02] = rng.uniform(4.0, 9.0)       # filtered (> 4)
    if zero_stdv:
        ed["stdv"][rng.random(n) < 0.01] = 0.0                      # Event::update_logs -> 0.01
    ed["length"] = rng.integers(8, 200, n)
    ed["start"] = 1000 + np.cumsum(ed["length"]) - ed["length"]
    for a, b in ([hairpin] if hairpin else []) + list(extra_islands):
        ed["mean"][a:b] = rng.normal(abasic, 3.0, b - a)
    return ed


def same_summary(a, b):
    return (a.num_ed_events == b.num_ed_events and np.float32(a.abasic_level).tobytes() == np.float32(b.abasic_level).tobytes()
            and list(a.strand_bounds) == list(b.strand_bounds) and a.scale_strands_together == b.scale_strands_together
            and np.float32(list(a.time_length)).tobytes() == np.float32(list(b.time_length)).tobytes())


CASES = [
    dict(n=3000, hairpin=(1480, 1500)),                                    # clean 2D read
    dict(n=3000, hairpin=None),                                            # template only: no island
    dict(n=3000, hairpin=(200, 230)),                                      # island outside the middle third: template only
    dict(n=3000, hairpin=(1400, 1420), extra_islands=[(1440, 1450), (1500, 1507)]),   # islands merge (within 50)
    dict(n=3000, hairpin=(1500, 1520), extra_islands=[(20, 40)]),          # leading island moves the template start
    dict(n=3000, hairpin=(1500, 1520), extra_islands=[(2930, 2960)]),      # trailing island moves the complement end
    dict(n=3000, hairpin=(1500, 1504)),                                    # 4 consecutive: not an island
    dict(n=105, hairpin=None),                                             # fewer than trim + min events: skipped
    dict(n=112, hairpin=None),                                             # just enough
    dict(n=200, hairpin=(95, 101)),                                        # complement shorter than min_ed_events
    dict(n=12000, hairpin=(5900, 5960), max_ed_events=8000),               # capped table: hairpin no longer central
    dict(n=3000, hairpin=(1480, 1500), level=-20.0, abasic=0.5),           # abasic level <= 1: skipped
]


@pytest.mark.parametrize("pore", ["r73", "r9"])
@pytest.mark.parametrize("sst", [True, False])
def test_summaries_and_events_equal_the_oracle(pore, sst):
    rng = np.random.default_rng(12345)
    for k, case in enumerate(CASES):
        case = dict(case)
        mx = case.pop("max_ed_events", 100000)
        ed = synth_ed(rng, **case)
        for one_d in (False, True):
            po = na.api.segment_opts(pore, template_only=int(one_d), max_ed_events=mx)
            oo = oracle.f5_opts(pore, template_only=one_d, max_ed_events=mx)
            for rate in (4000.0, 3012.0, 500.0):
                ps = api.read_summarize(po, ed, rate, sst)
                os_ = oracle.f5_summarize(oo, ed, rate, sst)
                assert same_summary(ps, os_), (k, one_d, rate, list(ps.strand_bounds), list(os_.strand_bounds))
                for st in (0, 1):
                    pe = api.read_load_events(ps, ed, rate, st)
                    oe = oracle.f5_load_events(os_, ed, rate, st)
                    for a, b in zip(pe, oe):
                        assert a.tobytes() == b.tobytes(), (k, one_d, rate, st)


def test_expected_shapes_of_the_segmentation():
    """Sanity of the fixtures themselves (so that the equality above is not vacuous)."""
    rng = np.random.default_rng(7)
    o = na.api.segment_opts("r73")
    s = api.read_summarize(o, synth_ed(rng, 3000, hairpin=(1480, 1500)), 4000.0, True)
    assert s.num_ed_events == 3000 and list(s.strand_bounds) == [50, 1430, 1530, 2950] and s.scale_strands_together == 1
    s = api.read_summarize(o, synth_ed(rng, 3000, hairpin=None), 4000.0, True)
    assert list(s.strand_bounds) == [50, 2950, 0, 0] and s.scale_strands_together == 0
    s = api.read_summarize(o, synth_ed(rng, 3000, hairpin=(200, 230)), 4000.0, True)
    assert list(s.strand_bounds) == [50, 2950, 0, 0]
    s = api.read_summarize(o, synth_ed(rng, 3000, hairpin=(1500, 1520), extra_islands=[(20, 40)]), 4000.0, True)
    assert s.strand_bounds[0] == 50 or s.strand_bounds[0] >= 40
    s = api.read_summarize(o, synth_ed(rng, 105, hairpin=None), 4000.0, True)
    assert s.num_ed_events == 0
    s = api.read_summarize(na.api.segment_opts("r73", template_only=1), synth_ed(rng, 3000, hairpin=(1480, 1500)), 4000.0, True)
    assert list(s.strand_bounds) == [50, 2950, 0, 0] and s.scale_strands_together == 0
    ev = api.read_load_events(s, synth_ed(rng, 3000, hairpin=(1480, 1500)), 4000.0, 0)
    assert 2700 < len(ev[0]) < 2900 and (ev[1] > 0).all() and (ev[1] <= 4.0).all()


def test_mean_stdv_and_initial_scaling_equal_the_oracle():
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 10, 5000, 100000):
        v = rng.normal(60, 8, n).astype(np.float32)
        a, b = api.mean_stdv(v), oracle.mean_stdv(v)
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
    for _ in range(50):
        r0, r1, m0, m1 = (np.float32([rng.normal(60, 5), rng.uniform(3, 9)]) for _ in range(4))
        for tg in (0, 1):
            a = api.initial_scaling(tg, r0, r1, m0, m1)
            b = oracle.f5_initial_scaling(tg, r0, r1, m0, m1)
            assert np.float32(a).tobytes() == b.tobytes()


def test_pore_presets():
    assert na.api.segment_opts("r9").abasic_level_top_offset == 0.0      # nanocall.cpp:945-946
    assert na.api.segment_opts("r73").abasic_level_top_offset == 5.0     # :956-957
    with pytest.raises(api.NchmmError):
        na.api.segment_opts("r10")


def test_randomised_tables_islands_and_options_equal_the_oracle():
    """1500 random EventDetection tables: any number of abasic islands (3-60 events long: either side of the five-in-a-row rule,
    Fast5_Summary.hpp:545-571) anywhere, gaps either side of the merge reach (:665-676), tables from below the minimum to past the cap,
    random trim margins / minimum / cap, constant stretches and NaN-free outliers in the levels; summaries and the filtered events of
    both strands bit-identical to the oracle's restatement."""
    rng = np.random.default_rng(20260607)
    shapes = set()
    for case in range(1500):
        n = int(rng.choice([int(rng.integers(60, 400)), int(rng.integers(400, 3000)), int(rng.integers(3000, 9000))]))
        level, abasic = float(rng.uniform(40, 90)), float(rng.uniform(95, 160))
        isl = []
        for _ in range(int(rng.integers(0, 7))):
            a = int(rng.integers(0, max(1, n - 3)))
            isl.append((a, min(n, a + int(rng.integers(3, 61)))))
        if rng.random() < 0.5 and n > 200:                 # one near the middle, so that 2D shapes are common
            a = n // 2 + int(rng.integers(-n // 5, n // 5))
            isl.append((a, min(n, a + int(rng.integers(5, 40)))))
        ed = synth_ed(rng, n, hairpin=None, level=level, abasic=abasic, extra_islands=isl, zero_stdv=bool(rng.random() < 0.5))
        if rng.random() < 0.2:                              # a constant stretch (a stalled pore)
            a = int(rng.integers(0, n - 20)); ed["mean"][a:a + int(rng.integers(5, 200))] = ed["mean"][a]
        if rng.random() < 0.2:                              # a few wild levels
            ed["mean"][rng.integers(0, n, 3)] = rng.choice([-50.0, 0.0, 1.0, 500.0], 3)
        trim = tuple(int(x) for x in rng.choice([0, 1, 10, 50, 80, 200], 4))
        mn, mx = int(rng.choice([1, 10, 40, 150])), int(rng.choice([100, 1000, 5000, 100000]))
        pore, one_d, sst = str(rng.choice(["r73", "r9"])), bool(rng.random() < 0.2), bool(rng.random() < 0.7)
        po = na.api.segment_opts(pore, template_only=int(one_d), max_ed_events=mx, min_ed_events=mn)
        po.trim_margins[:] = list(trim)
        oo = oracle.f5_opts(pore, template_only=one_d, max_ed_events=mx, min_ed_events=mn, trim=trim)
        rate = float(rng.choice([4000.0, 3012.5, 999.0, 10000.0]))
        ps, os_ = api.read_summarize(po, ed, rate, sst), oracle.f5_summarize(oo, ed, rate, sst)
        assert same_summary(ps, os_), (case, n, isl, trim, mn, mx, list(ps.strand_bounds), list(os_.strand_bounds))
        for st in (0, 1):
            pe, oe = api.read_load_events(ps, ed, rate, st), oracle.f5_load_events(os_, ed, rate, st)
            for a, b in zip(pe, oe):
                assert a.tobytes() == b.tobytes(), (case, st)
        sb = list(ps.strand_bounds)
        shapes.add("skipped" if ps.num_ed_events == 0 else ("2d" if sb[3] > sb[2] else "1d"))
    assert shapes == {"skipped", "1d", "2d"}
