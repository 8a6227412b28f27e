#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/nc_oracle.c).

PROVENANCE / PARITY STATUS: the reference cannot be executed here (its hot-path headers include
un-vendored hpptools / fast5 headers; see DESIGN.md "Oracle"), and it ships no tests or golden
vectors.  These fixtures are therefore produced by OUR restatement of the reference algorithm and pin
the *implementation history* (any later change to the oracle, the host prep or the kernels that moves
a bit shows up), not the reference itself -- "parity unpinned" for Viterbi / FB / EM numerics.
The parts that ARE pinned to the real reference (Kmer algebra, builtin model tables, transition
weights) are tested separately in tests/test_reference_pins.py.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import nanocall_amd as na          # noqa: E402  (host prep only; no GPU needed)
from nanocall_amd import synth     # noqa: E402
import nc_oracle as oracle         # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PARAMS = (1.05, 2.5, 0.002, 1.1, 0.9, 1.2)   # SURVEY.md section 8c, fixture G2


def viterbi_case(name, model, n, first_read, params, p_skip, p_stay, mutate=None):
    table = na.builtin_model(model)
    ev = synth.generate(table, 1, n, first_read=first_read)
    mean, stdv, start = ev["mean"][0].copy(), ev["stdv"][0].copy(), ev["start"][0].copy()
    if mutate:
        mutate(mean, stdv, start, table)
    om = oracle.Model(table, params)
    ot = oracle.Transitions(p_skip, p_stay)
    cm, sd, ls = oracle.events_prepare(mean, stdv, start, params[2])
    st, mv, lp = oracle.viterbi(om, ot, cm, sd, ls)
    seq = oracle.base_seq(st, mv)
    fasta = oracle.write_fasta(f"{name}:synthetic:0", seq, 80)
    np.savez_compressed(os.path.join(OUT, f"viterbi_{name}.npz"), model=model, params=np.float32(params),
                        p_skip=np.float32(p_skip), p_stay=np.float32(p_stay), mean=mean, stdv=stdv, start=start,
                        states=st, moves=mv, path_logp_bits=np.float32(lp).view(np.uint32), seq=seq, fasta=fasta)
    print(name, n, "events ->", len(seq), "bases, logp", lp)


def ties(mean, stdv, start, table):
    mean[100:160] = mean[100]     # a run of identical events: exact float ties in the DP
    stdv[100:160] = stdv[100]


def zero_stdv(mean, stdv, start, table):
    stdv[::11] = 0.0              # Event::update_logs turns these into 0.01


def main():
    ident = (1.0, 0.0, 0.0, 1.0, 1.0, 1.0)
    viterbi_case("r73t_300", "r73.t", 300, 0, ident, 0.3, 0.1)
    viterbi_case("r73t_1000_scaled", "r73.t", 1000, 1, PARAMS, 0.28, 0.09)
    viterbi_case("r73t_3000", "r73.t", 3000, 2, ident, 0.3, 0.1)
    viterbi_case("r9t_2000", "r9.t", 2000, 3, ident, 0.3, 0.1)
    viterbi_case("r73t_ties", "r73.t", 400, 4, ident, 0.3, 0.1, ties)
    viterbi_case("r73t_zero_stdv", "r73.t", 300, 5, ident, 0.17, 0.12, zero_stdv)
    # scaled model G2 (CRC-sized: rows 0..15 and 4080..4095 of the S x 10 state array)
    for idx, model in ((0, "r73.t"), (3, "r9.t")):
        m = oracle.Model(na.builtin_model(model), PARAMS).states()
        np.savez_compressed(os.path.join(OUT, f"scaled_model_{idx}.npz"), model=model, params=np.float32(PARAMS),
                            head=m[:16], tail=m[-16:], sum_bits=np.uint64(int(m.view(np.uint32).astype(np.uint64).sum())))
    # forward-backward G4: 2 windows x 100 events
    table = na.builtin_model("r73.t")
    ev = synth.generate(table, 2, 100, first_read=7)
    om = oracle.Model(table, ident)
    ot = oracle.Transitions(0.3, 0.1)
    out = {}
    for w in range(2):
        cm, sd, ls = oracle.events_prepare(ev["mean"][w], ev["stdv"][w], ev["start"][w], 0.0)
        lpd, al, be = oracle.fwbw(om, ot, cm, sd, ls)
        post = al[50] + be[50] - lpd
        top = np.argsort(-post)[:5]
        out[f"w{w}_mean"], out[f"w{w}_stdv"], out[f"w{w}_start"] = ev["mean"][w], ev["stdv"][w], ev["start"][w]
        out[f"w{w}_log_pr_data"] = np.float32(lpd)
        out[f"w{w}_probe_cells"] = np.array([[i, j, al[i, j], be[i, j]] for i, j in ((0, 0), (10, 100), (50, 2048), (99, 4095), (70, 1234))], np.float64)
        out[f"w{w}_top5_states"], out[f"w{w}_top5_logpost"] = top.astype(np.int32), post[top]
        print("fwbw window", w, "log_pr_data", lpd)
    np.savez_compressed(os.path.join(OUT, "fwbw_r73t_2x100.npz"), **out)
    # EM G5: one 2D-style read, template (r73.t) + complement (r73.c.p1), two 100-event windows per strand, 4 rounds
    t0, t1 = na.builtin_model("r73.t"), na.builtin_model("r73.c.p1")
    e0 = synth.generate(t0, 1, 400, first_read=20)
    e1 = synth.generate(t1, 1, 400, first_read=21)
    wins = [(e0, slice(0, 100), 0), (e0, slice(300, 400), 0), (e1, slice(0, 100), 1), (e1, slice(300, 400), 1)]
    mean = np.concatenate([e["mean"][0][s] for e, s, _ in wins])
    stdv = np.concatenate([e["stdv"][0][s] for e, s, _ in wins])
    start = np.concatenate([e["start"][0][s] for e, s, _ in wins])
    strand = np.array([st for _, _, st in wins], np.uint32)
    off = np.arange(5, dtype=np.uint64) * 100
    for drift in (1, 0):
        pm = np.array([1, 0, 0, 1, 1, 1], np.float32)
        stp = np.array([0.1, 0.3, 0.1, 0.3], np.float32)
        rounds = []
        for rnd in range(4):
            r = oracle.train_one_round(off, strand, mean, stdv, start, t0, t1, pm, stp, 0.1, 0.3, drift)
            rounds.append(np.concatenate([[r["fit"]], r["pm"], r["st"], [float(r["done"])]]))
            print("EM drift", drift, "round", rnd, "fit", r["fit"], "pm", r["pm"], "st", r["st"], "done", r["done"])
            if r["done"]:
                break
            pm, stp = r["pm"], r["st"]
        np.savez_compressed(os.path.join(OUT, f"em_2d_drift{drift}.npz"), mean=mean, stdv=stdv, start=start, strand=strand,
                            off=off, rounds=np.array(rounds, np.float64))


if __name__ == "__main__":
    main()
