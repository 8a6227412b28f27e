"""The reference driver (src/nanocall/nanocall.cpp) transcribed in Python on top of the CPU oracle -- TEST
INFRASTRUCTURE: what `nanocall <inputs>` prints as FASTA for a list of EventDetection tables, computed without
any product code.  Used by the CLI tests as the expected output.

  summarize / load_events / initial scaling    oracle.f5_* (Fast5_Summary.hpp:138-370)
  train_reads                                   nanocall.cpp:292-574 on oracle.train_one_round
  basecall_reads                                nanocall.cpp:621-857 on oracle.viterbi / base_seq / write_fasta
"""
import numpy as np

import nc_oracle as oracle
from nanocall_amd import models as builtin


class Opts:
    """The command-line options that reach the hot path, with the reference's defaults (nanocall.cpp:50-95)."""

    def __init__(self, pore="r9", one_d=False, train=True, train_scaling=True, train_transitions=True, train_drift=None,
                 double_strand_scaling=None, single_strand_scaling=False, scaling_num_events=200, scaling_max_rounds=10,
                 scaling_min_progress=1.0, scaling_select_threshold=20.0, min_ed_events=10, max_ed_events=100000,
                 pr_stay=0.1, pr_skip=0.3, fasta_line_width=80, trim=(50, 50, 50, 50)):
        self.pore, self.one_d, self.train = pore, one_d, train
        self.train_scaling, self.train_transitions = train_scaling, train_transitions
        self.train_drift = (pore == "r73") if train_drift is None else bool(train_drift)      # nanocall.cpp:949-963
        # nanocall.cpp:1025-1038: double strand scaling is the default only when training + scaling are on
        if double_strand_scaling is None:
            double_strand_scaling = train and train_scaling and not single_strand_scaling
        self.double_strand_scaling = double_strand_scaling
        self.scaling_num_events, self.scaling_max_rounds = scaling_num_events, scaling_max_rounds
        self.scaling_min_progress, self.scaling_select_threshold = scaling_min_progress, scaling_select_threshold
        self.min_ed_events, self.max_ed_events = min_ed_events, max_ed_events
        self.pr_stay, self.pr_skip = np.float32(pr_stay), np.float32(pr_skip)
        self.fasta_line_width, self.trim = fasta_line_width, trim


def load_models(pore):
    """init_models, nanocall.cpp:157-170: builtin tables whose name starts with '<pore>.', in std::map (name) order."""
    out = []
    for name, strand in zip(builtin.builtin_names(), builtin.builtin_strands()):
        if name.startswith(pore + "."):
            out.append((name, strand, builtin.builtin_model(name)))
    out.sort(key=lambda x: x[0])
    return out


class Read:
    pass


def summarize(o, models, file_name, table):
    """Fast5_Summary::summarize (Fast5_Summary.hpp:138-319) -> Read."""
    r = Read()
    base = file_name.rsplit("/", 1)[-1]
    r.base_file_name = base[:-6] if base.endswith(".fast5") else base
    r.read_id = table.get("read_id") or r.base_file_name
    r.rate = np.float32(table["sampling_rate"])
    r.ed = np.ascontiguousarray(table["events"], oracle.ED_DTYPE)
    fo = oracle.f5_opts(o.pore, template_only=o.one_d, min_ed_events=o.min_ed_events, max_ed_events=o.max_ed_events, trim=o.trim)
    r.summary = oracle.f5_summarize(fo, r.ed, float(r.rate), o.double_strand_scaling)
    r.num_ed_events = r.summary.num_ed_events
    r.together = bool(r.summary.scale_strands_together)
    r.events = [oracle.f5_load_events(r.summary, r.ed, float(r.rate), st) for st in (0, 1)] if r.num_ed_events else [None, None]
    r.pm, r.st = {}, {}
    r.preferred = {0: None, 1: None, 2: None}
    if not r.num_ed_events:
        return r
    stats = [oracle.mean_stdv(r.events[st][0]) if len(r.events[st][0]) >= o.min_ed_events else None for st in (0, 1)]
    mstat = {name: (oracle.Model(t).mean, oracle.Model(t).stdv) for name, _, t in models}
    dflt = (o.pr_stay, o.pr_skip)
    if r.together:
        for n0, s0, _ in models:
            if s0 not in (0, 2):
                continue
            for n1, s1, _ in models:
                if s1 not in (1, 2):
                    continue
                sc = oracle.f5_initial_scaling(1, stats[0], stats[1], mstat[n0], mstat[n1])
                r.pm[(n0, n1)] = np.float32([sc[0], sc[1], 0, 1, 1, 1])
                r.st[(n0, n1)] = np.float32([dflt[0], dflt[1], dflt[0], dflt[1]])
    else:
        for st in (0, 1):
            if len(r.events[st][0]) < o.min_ed_events:
                continue
            for n, s, _ in models:
                if s not in (st, 2):
                    continue
                key = (n, "") if st == 0 else ("", n)
                sc = oracle.f5_initial_scaling(0, stats[st], None, mstat[n], None)
                r.pm[key] = np.float32([sc[0], sc[1], 0, 1, 1, 1])
                r.st[key] = np.float32([dflt[0], dflt[1], dflt[0], dflt[1]])
    return r


def _windows(o, r, strands):
    """train_event_seqs, nanocall.cpp:327-338: first and last num_train_events / 2 events of each strand."""
    win, wst = [], []
    for st in strands:
        mean, stdv, start, _ = r.events[st]
        n = len(mean)
        half = min(o.scaling_num_events, n) // 2
        for sl in (slice(0, half), slice(n - half, n)):
            win.append((mean[sl], stdv[sl], start[sl]))
            wst.append(st)
    return win, wst


def train_job(o, tables, windows, strands, key, pm, st):
    """One iteration of the model loops, nanocall.cpp:360-426 (2D) / :476-542 (1D)."""
    off = np.concatenate([[0], np.cumsum([len(w[0]) for w in windows])]).astype(np.uint64)
    mean = np.concatenate([w[0] for w in windows])
    stdv = np.concatenate([w[1] for w in windows])
    start = np.concatenate([w[2] for w in windows])
    two_d = bool(key[0]) and bool(key[1])
    t0 = tables[key[0] or key[1]]
    t1 = tables[key[1] or key[0]]
    crt_pm, crt_st, crt_fit, rnd = np.float32(pm), np.float32(st), np.float32(-np.inf), 0
    while True:
        old_pm, old_st, old_fit = crt_pm.copy(), crt_st.copy(), crt_fit
        res = oracle.train_one_round(off, np.asarray(strands, np.uint32), mean, stdv, start, t0, t1, old_pm, old_st,
                                     o.pr_stay, o.pr_skip, int(o.train_drift), bool(o.train_scaling), bool(o.train_transitions))
        crt_pm, crt_fit = res["pm"], res["fit"]
        new_st = res["st"].copy()
        for s in range(2):
            if s not in strands:
                new_st[2 * s:2 * s + 2] = old_st[2 * s:2 * s + 2]
        if not o.train_transitions:
            new_st = old_st.copy()
        if not o.train_scaling:
            crt_pm = old_pm.copy()
        crt_st = new_st
        if res["done"]:
            break
        if crt_fit < old_fit:
            crt_pm, crt_st, crt_fit = old_pm, old_st, old_fit
            break
        rnd += 1
        limit = 2 * o.scaling_max_rounds if two_d else o.scaling_max_rounds
        if rnd >= limit or (rnd > 1 and crt_fit < old_fit + o.scaling_min_progress):
            break
    return crt_pm, crt_st, crt_fit, rnd


def train_read(o, models, r):
    """process_item of train_reads, nanocall.cpp:292-574.  Records fits / rounds on the read for the tests."""
    tables = {n: t for n, _, t in models}
    r.fit, r.rounds = {}, {}
    if not r.num_ed_events:
        return
    ok = [len(r.events[st][0]) >= o.min_ed_events for st in (0, 1)]
    mlist = [[n for n, s, _ in models if s in (st, 2)] if ok[st] else [] for st in (0, 1)]
    if r.together:
        win, wst = _windows(o, r, [st for st in (0, 1) if ok[st]])
        fits = {}
        for n0 in mlist[0]:
            for n1 in mlist[1]:
                key = (n0, n1)
                r.pm[key], r.st[key], fits[key], r.rounds[key] = train_job(o, tables, win, wst, key, r.pm[key], r.st[key])
        r.fit.update(fits)
        if fits and o.scaling_select_threshold < np.inf:
            keys = list(fits)                                  # std::map order == insertion order here (names sorted)
            best = max(keys, key=lambda k: (fits[k], -keys.index(k)))      # alg::max_of: first maximum
            if all(k == best or fits[k] + np.float32(o.scaling_select_threshold) < fits[best] for k in keys):
                r.preferred[2] = best
    else:
        for st in (0, 1):
            if not ok[st]:
                continue
            win, wst = _windows(o, r, [st])
            fits = {}
            for n in mlist[st]:
                key = (n, "") if st == 0 else ("", n)
                r.pm[key], r.st[key], fits[key], r.rounds[key] = train_job(o, tables, win, wst, key, r.pm[key], r.st[key])
            r.fit.update(fits)
            if fits and o.scaling_select_threshold < np.inf:
                keys = list(fits)
                best = max(keys, key=lambda k: (fits[k], -keys.index(k)))
                if all(k == best or fits[k] + np.float32(o.scaling_select_threshold) < fits[best] for k in keys):
                    r.preferred[st] = best[st]


def basecall_strand(o, tables, r, st, name, pm, st_params, trans_cache):
    """nanocall.cpp:645-690 -> (path_probability, states, moves)."""
    om = oracle.Model(tables[name], pm)
    tk = (np.float32(st_params[1]).tobytes(), np.float32(st_params[0]).tobytes())
    if tk not in trans_cache:
        trans_cache[tk] = oracle.Transitions(float(st_params[1]), float(st_params[0]))
    mean, stdv, start, _ = r.events[st]
    cm, sd, ls = oracle.events_prepare(mean, stdv, start, float(pm[2]))
    states, moves, lp = oracle.viterbi(om, trans_cache[tk], cm, sd, ls)
    return lp, states, moves


def basecall_read(o, models, r, forced=None, trans_cache=None):
    """process_item of basecall_reads, nanocall.cpp:621-857 -> list of (seq_name, sequence, info) per decoded strand.
    forced: {(read_id, strand): (model_name, pm[6], (p_stay, p_skip))} decodes exactly that instead of choosing."""
    tables = {n: t for n, _, t in models}
    trans_cache = {} if trans_cache is None else trans_cache
    out = []
    if not r.num_ed_events:
        return out
    if forced is not None:
        for st in (0, 1):
            f = forced.get((r.read_id, st))
            if f is None:
                continue
            lp, states, moves = basecall_strand(o, tables, r, st, f[0], np.float32(f[1]), np.float32(f[2]), trans_cache)
            out.append((f"{r.read_id}:{r.base_file_name}:{st}", oracle.base_seq(states, moves), dict(model=f[0], logp=lp, states=states)))
        return out
    if r.together:
        cand = [r.preferred[2]] if r.preferred[2] else [k for k in r.pm if k[0] and k[1]]
        results = []
        for key in cand:
            part = [basecall_strand(o, tables, r, st, key[st], r.pm[key], r.st[key][2 * st:2 * st + 2], trans_cache) for st in (0, 1)]
            results.append((np.float32(part[0][0] + part[1][0]), key, part))
        # sort(...) by the sum, back(): the highest; the later candidate among exact ties (stable order assumed)
        best = max(range(len(results)), key=lambda i: (results[i][0], i))
        _, key, part = results[best]
        for st in (0, 1):
            out.append((f"{r.read_id}:{r.base_file_name}:{st}", oracle.base_seq(part[st][1], part[st][2]),
                        dict(model=key[st], logp=part[st][0], states=part[st][1], pm=r.pm[key], st=r.st[key][2 * st:2 * st + 2])))
    else:
        for st in (0, 1):
            if len(r.events[st][0]) < o.min_ed_events:
                continue
            if r.preferred[st]:
                cand = [(r.preferred[st], "") if st == 0 else ("", r.preferred[st])]
            else:
                cand = [k for k in r.pm if k[st] and not k[1 - st]]
            results = [(basecall_strand(o, tables, r, st, key[st], r.pm[key], r.st[key][2 * st:2 * st + 2], trans_cache), key) for key in cand]
            best = max(range(len(results)), key=lambda i: (results[i][0][0], i))
            (lp, states, moves), key = results[best]
            out.append((f"{r.read_id}:{r.base_file_name}:{st}", oracle.base_seq(states, moves),
                        dict(model=key[st], logp=lp, states=states, pm=r.pm[key], st=r.st[key][2 * st:2 * st + 2])))
    return out


def run(o, inputs, forced=None):
    """inputs: list of (file_name, table dict(sampling_rate, read_id, events)).  -> (fasta text, reads, records)"""
    models = load_models(o.pore)
    reads = [summarize(o, models, fn, t) for fn, t in inputs]
    if o.train and forced is None:
        for r in reads:
            train_read(o, models, r)
    fasta, records = "", []
    cache = {}
    for r in reads:
        for name, seq, info in basecall_read(o, models, r, forced, cache):
            fasta += oracle.write_fasta(name, seq, o.fasta_line_width)
            records.append((name, seq, info))
    return fasta, reads, records


# ------------------------------------------------------------------------------------------------
# synthetic reads as EventDetection tables
# ------------------------------------------------------------------------------------------------
def synth_ed_table(pore, n_template, n_complement, seed, rate=4000.0, complement_model=None, lead=60, tail=60, hairpin=25,
                   drift=0.0, scale=1.0, shift=0.0):
    """An EventDetection table of a (2D when n_complement > 0) read: leader events, the template strand emitted by the
    pore's template model along a random k-mer walk (nanocall_amd.synth), an abasic hairpin plateau, the complement strand
    from `complement_model`, trailing events.  Levels are scaled / shifted / drifted so that the EM has something to find."""
    from nanocall_amd import synth
    import nanocall_amd as na
    names = [n for n in builtin.builtin_names() if n.startswith(pore + ".")]
    t_name = [n for n in names if ".t." in n][0]
    c_name = complement_model or [n for n in names if ".c.p1." in n][0]
    rng = np.random.default_rng(seed)
    parts = []
    tab_t = na.builtin_model(t_name)
    parts.append(("t", synth.generate(tab_t, 1, lead + n_template, first_read=seed * 7 + 1)))
    if n_complement > 0:
        parts.append(("h", hairpin))
        parts.append(("c", synth.generate(na.builtin_model(c_name), 1, n_complement + tail, first_read=seed * 7 + 2)))
    else:
        parts.append(("t2", synth.generate(tab_t, 1, tail, first_read=seed * 7 + 3)))
    mean, stdv, length = [], [], []
    for kind, p in parts:
        if kind == "h":
            mean.append(rng.normal(float(tab_t[:, 0].max()) * scale + shift + 60.0, 1.5, p))
            stdv.append(rng.uniform(0.8, 2.0, p))
            length.append(rng.uniform(0.01, 0.03, p))
        else:
            mean.append(p["mean"][0].astype(np.float64))
            stdv.append(p["stdv"][0].astype(np.float64))
            length.append(p["length"][0].astype(np.float64))
    mean, stdv, length = np.concatenate(mean), np.concatenate(stdv), np.concatenate(length)
    ed = np.zeros(len(mean), oracle.ED_DTYPE)
    ed["length"] = np.maximum(1, np.round(length * rate)).astype(np.int64)
    ed["start"] = 5000 + np.cumsum(ed["length"]) - ed["length"]
    t = (ed["start"] - ed["start"][0]) / rate
    is_h = np.zeros(len(mean), bool)
    if n_complement > 0:
        is_h[lead + n_template: lead + n_template + hairpin] = True
    ed["mean"] = np.where(is_h, mean, mean * scale + shift + drift * t)
    ed["stdv"] = stdv
    return ed


def write_events_table(path, ed, rate, read_id=None):
    with open(path, "w") as f:
        f.write("#nanocall-events v1\n")
        f.write(f"#sampling_rate {rate!r}\n")
        if read_id:
            f.write(f"#read_id {read_id}\n")
        for e in ed:
            f.write(f"{float(e['mean'])!r} {float(e['stdv'])!r} {int(e['start'])} {int(e['length'])}\n")


def read_dump(path):
    """--dump-params file -> {(read_id, strand): dict(model, pm[6], st(p_stay, p_skip), logp, rounds, fit)}"""
    out = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        v = [float.fromhex(x) for x in f[3:12]]
        out[(f[0], int(f[1]))] = dict(model=f[2], pm=np.float32(v[:6]), st=np.float32(v[6:8]), logp=np.float32(v[8]),
                                      rounds=int(f[12]), fit=np.float32(float.fromhex(f[13])))
    return out
