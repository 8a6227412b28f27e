"""ctypes binding of the CPU oracle (oracle/libnc_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg.  The
product package (nanocall_amd) never imports this module.  See nc_oracle.h for parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "libnc_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libnc_ref.so")
S = 4096
vp = C.c_void_p
_lib = None
_ref = None


def build():
    subprocess.run(["make", "-C", _HERE, "all", "ref"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        # NC_ORACLE_LIB: another build of nc_oracle.c (tests/test_oracle_builds.py checks that the goldens do not depend on the
        # compiler or its optimisation level)
        so = os.environ.get("NC_ORACLE_LIB") or SO
        if so == SO and not os.path.exists(SO):
            build()
        L = C.CDLL(so)
        L.nco_transitions_fast.restype = vp
        L.nco_transitions_fast.argtypes = [C.c_float, C.c_float]
        L.nco_transitions_free.argtypes = [vp]
        L.nco_transitions_n_arcs.restype = C.c_uint32
        L.nco_transitions_n_arcs.argtypes = [vp]
        L.nco_transitions_export_from.argtypes = [vp, vp, vp, vp]
        L.nco_transitions_export_to.argtypes = [vp, vp, vp, vp]
        L.nco_trans_prob.restype = C.c_float
        L.nco_trans_prob.argtypes = [C.c_uint, C.c_uint, C.c_float, C.c_float, C.c_float]
        L.nco_model_load_from_vector.argtypes = [vp, vp]
        L.nco_model_scale.argtypes = [vp, vp]
        L.nco_model_export6.argtypes = [vp, vp]
        L.nco_log_pr_corrected_emission.restype = C.c_float
        L.nco_log_pr_corrected_emission.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        L.nco_viterbi_soa.restype = C.c_float
        L.nco_viterbi_soa.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, vp, vp]
        L.nco_viterbi_tie_cells.restype = C.c_uint64
        L.nco_viterbi_tie_cells.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
        L.nco_fwbw_soa.restype = C.c_float
        L.nco_fwbw_soa.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, vp, vp]
        L.nco_logsumset_val.restype = C.c_float
        L.nco_logsumset_val.argtypes = [vp, C.c_size_t]
        L.nco_kmer_min_skip.restype = C.c_uint
        L.nco_kmer_min_skip.argtypes = [C.c_uint, C.c_uint]
        L.nco_kmer_prefix.restype = C.c_uint
        L.nco_kmer_prefix.argtypes = [C.c_uint, C.c_uint]
        L.nco_kmer_suffix.restype = C.c_uint
        L.nco_kmer_suffix.argtypes = [C.c_uint, C.c_uint]
        L.nco_kmer_max_self_overlap.restype = C.c_uint
        L.nco_kmer_max_self_overlap.argtypes = [C.c_uint]
        L.nco_kmer_neighbour_list.argtypes = [C.c_uint, C.c_uint, vp]
        L.nco_kmer_to_string.argtypes = [C.c_uint, C.c_char_p]
        L.nco_kmer_to_int.restype = C.c_uint
        L.nco_kmer_to_int.argtypes = [C.c_char_p]
        L.nco_st_train_kmers.restype = C.c_uint
        L.nco_st_train_kmers.argtypes = [vp]
        L.nco_event_init.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
        L.nco_events_apply_drift_correction.argtypes = [vp, C.c_size_t, C.c_float]
        L.nco_events_get_base_seq.restype = C.c_size_t
        L.nco_events_get_base_seq.argtypes = [vp, C.c_size_t, C.c_char_p]
        L.nco_write_fasta.restype = C.c_size_t
        L.nco_write_fasta.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_uint]
        L.nco_train_one_round_soa.restype = C.c_float
        L.nco_train_one_round_soa.argtypes = [C.c_size_t, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_int,
                                              vp, vp, vp, vp, vp, C.c_int, C.c_int]
        L.nco_logf_mismatches.restype = C.c_size_t
        L.nco_logf_mismatches.argtypes = [vp, vp, C.c_size_t, vp]
        L.nco_mean_stdv.argtypes = [vp, C.c_size_t, vp, vp]
        L.nco_f5_summarize.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_int, vp]
        L.nco_f5_load_events.restype = C.c_size_t
        L.nco_f5_load_events.argtypes = [vp, vp, C.c_float, C.c_uint, vp, vp, vp, vp]
        L.nco_f5_initial_scaling.argtypes = [C.c_int, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def ref():
    """The real reference's Kmer.hpp + Builtin_Model.cpp (oracle/_ref), or None if not built."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        R = C.CDLL(REF_SO)
        for n in ("ref_kmer_min_skip", "ref_kmer_prefix", "ref_kmer_suffix"):
            getattr(R, n).restype = C.c_uint
            getattr(R, n).argtypes = [C.c_uint, C.c_uint]
        R.ref_kmer_max_self_overlap.restype = C.c_uint
        R.ref_kmer_max_self_overlap.argtypes = [C.c_uint]
        R.ref_kmer_to_int.restype = C.c_uint
        R.ref_kmer_to_int.argtypes = [C.c_char_p]
        R.ref_kmer_to_string.argtypes = [C.c_uint, C.c_char_p]
        R.ref_kmer_neighbour_list.restype = C.c_uint
        R.ref_kmer_neighbour_list.argtypes = [C.c_uint, C.c_uint, vp]
        R.ref_builtin_num.restype = C.c_uint
        R.ref_builtin_strand.restype = C.c_uint
        R.ref_builtin_strand.argtypes = [C.c_uint]
        R.ref_builtin_name.restype = C.c_char_p
        R.ref_builtin_name.argtypes = [C.c_uint]
        R.ref_builtin_size.restype = C.c_uint
        R.ref_builtin_size.argtypes = [C.c_uint]
        R.ref_builtin_table.restype = C.POINTER(C.c_float)
        R.ref_builtin_table.argtypes = [C.c_uint]
        _ref = R
    return _ref


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


# sizeof(nco_model) = 4096 * 10 floats + 2 floats
_MODEL_BYTES = S * 10 * 4 + 8


class Model:
    """nco_model: load_from_vector + scale."""

    def __init__(self, table_Sx4, params=None):
        self.buf = np.zeros(_MODEL_BYTES // 4, np.float32)
        t = np.ascontiguousarray(table_Sx4, np.float32).reshape(S, 4)
        lib().nco_model_load_from_vector(_p(self.buf), _p(t))
        if params is not None:
            self.scale(params)

    def scale(self, params):
        p = np.ascontiguousarray(params, np.float32).reshape(6)
        lib().nco_model_scale(_p(self.buf), _p(p))

    @property
    def ptr(self):
        return _p(self.buf)

    @property
    def mean(self):
        """Pore_Model::mean() (Pore_Model.hpp:307-313)"""
        return np.float32(self.buf[S * 10])

    @property
    def stdv(self):
        return np.float32(self.buf[S * 10 + 1])

    def states(self):
        """S x 10 in Pore_Model_State field order."""
        return self.buf[: S * 10].reshape(S, 10).copy()

    def table6(self):
        out = np.empty((S, 6), np.float32)
        lib().nco_model_export6(self.ptr, _p(out))
        return out

    def emission(self, j, cmean, stdv, log_stdv):
        return lib().nco_log_pr_corrected_emission(C.c_void_p(self.buf.ctypes.data + j * 40),
                                                   C.c_float(cmean), C.c_float(stdv), C.c_float(log_stdv))


class Transitions:
    def __init__(self, p_skip, p_stay):
        self.h = lib().nco_transitions_fast(C.c_float(p_skip), C.c_float(p_stay))
        self.n_arcs = lib().nco_transitions_n_arcs(self.h)

    def __del__(self):
        try:
            lib().nco_transitions_free(self.h)
        except Exception:
            pass

    def _export(self, fn):
        rp = np.empty(S + 1, np.uint32)
        idx = np.empty(self.n_arcs, np.uint32)
        w = np.empty(self.n_arcs, np.float32)
        fn(self.h, _p(rp), _p(idx), _p(w))
        return rp, idx, w

    def from_csr(self):
        return self._export(lib().nco_transitions_export_from)

    def to_csr(self):
        return self._export(lib().nco_transitions_export_to)


def viterbi(model, trans, cmean, stdv, log_stdv):
    """Reference-layout Viterbi on one read -> (states u16, moves i32, path_probability)."""
    cm = np.ascontiguousarray(cmean, np.float32)
    sd = np.ascontiguousarray(stdv, np.float32)
    ls = np.ascontiguousarray(log_stdv, np.float32)
    n = cm.shape[0]
    st = np.empty(n, np.uint16)
    mv = np.empty(n, np.int32)
    lp = lib().nco_viterbi_soa(model.ptr, trans.h, n, _p(cm), _p(sd), _p(ls), _p(st), _p(mv))
    return st, mv, np.float32(lp)


def viterbi_tie_cells(model, trans, cmean, stdv, log_stdv):
    """cells of the Viterbi matrix whose maximum two or more predecessors attain with exactly equal floats"""
    cm = np.ascontiguousarray(cmean, np.float32)
    sd = np.ascontiguousarray(stdv, np.float32)
    ls = np.ascontiguousarray(log_stdv, np.float32)
    return int(lib().nco_viterbi_tie_cells(model.ptr, trans.h, cm.shape[0], _p(cm), _p(sd), _p(ls)))


def fwbw(model, trans, cmean, stdv, log_stdv, want_matrices=True):
    cm = np.ascontiguousarray(cmean, np.float32)
    sd = np.ascontiguousarray(stdv, np.float32)
    ls = np.ascontiguousarray(log_stdv, np.float32)
    n = cm.shape[0]
    al = np.empty((n, S), np.float32) if want_matrices else None
    be = np.empty((n, S), np.float32) if want_matrices else None
    lpd = lib().nco_fwbw_soa(model.ptr, trans.h, n, _p(cm), _p(sd), _p(ls), _p(al), _p(be))
    return np.float32(lpd), al, be


def events_prepare(mean, stdv, start, drift):
    """Event init (update_logs) + apply_drift_correction -> (corrected_mean, stdv, log_stdv)."""
    n = len(mean)
    # nco_event: 8 floats + unsigned + int = 40 bytes
    ev = np.zeros(n * 10, np.float32)
    L = lib()
    for i in range(n):
        L.nco_event_init(C.c_void_p(ev.ctypes.data + 40 * i), C.c_float(mean[i]), C.c_float(stdv[i]),
                         C.c_float(start[i]), C.c_float(0.0))
    L.nco_events_apply_drift_correction(_p(ev), n, C.c_float(drift))
    e = ev.reshape(n, 10)
    return e[:, 1].copy(), e[:, 2].copy(), e[:, 7].copy()


def base_seq(states, moves):
    n = len(states)
    ev = np.zeros(n * 10, np.float32)
    iv = ev.view(np.int32).reshape(n, 10)
    iv[:, 8] = np.asarray(states, np.int64)
    iv[:, 9] = np.asarray(moves, np.int64)
    buf = C.create_string_buffer(6 * max(n, 1) + 1)
    ln = lib().nco_events_get_base_seq(_p(ev), n, buf)
    return buf.raw[:ln].decode()


def write_fasta(name, seq, width=80):
    cap = len(name) + len(seq) + len(seq) // max(width, 1) + 16
    buf = C.create_string_buffer(cap)
    n = lib().nco_write_fasta(buf, cap, name.encode(), seq.encode(), width)
    return buf.raw[:n].decode()


def st_train_kmers():
    out = np.empty(S, np.uint32)
    n = lib().nco_st_train_kmers(_p(out))
    return out[:n].copy()


def logsumset(vals):
    v = np.ascontiguousarray(vals, np.float32).copy()
    return np.float32(lib().nco_logsumset_val(_p(v), v.shape[0]))


def train_one_round(off, strand, mean, stdv, start, model0_Sx4, model1_Sx4, crt_pm, crt_st,
                    default_p_stay=0.1, default_p_skip=0.3, train_drift=1, train_scaling=True,
                    train_transitions=True):
    """Parameter_Trainer::train_one_round -> dict(fit, pm[6], st[4], done)."""
    off = np.ascontiguousarray(off, np.uint64)
    strand = np.ascontiguousarray(strand, np.uint32)
    mean = np.ascontiguousarray(mean, np.float32)
    stdv = np.ascontiguousarray(stdv, np.float32)
    start = np.ascontiguousarray(start, np.float32)
    m0 = np.ascontiguousarray(model0_Sx4, np.float32)
    m1 = np.ascontiguousarray(model1_Sx4 if model1_Sx4 is not None else model0_Sx4, np.float32)
    cp = np.ascontiguousarray(crt_pm, np.float32).reshape(6)
    cs = np.ascontiguousarray(crt_st, np.float32).reshape(4)
    npm = np.empty(6, np.float32)
    nst = np.empty(4, np.float32)
    done = C.c_int(0)
    fit = lib().nco_train_one_round_soa(off.shape[0] - 1, _p(off), _p(strand), _p(mean), _p(stdv), _p(start),
                                        _p(m0), _p(m1), C.c_float(default_p_stay), C.c_float(default_p_skip),
                                        int(train_drift), _p(cp), _p(cs), _p(npm), _p(nst), C.byref(done),
                                        int(train_scaling), int(train_transitions))
    return dict(fit=np.float32(fit), pm=npm, st=nst, done=bool(done.value))


# ------------------------------------------------------------------------------------------------
# Fast5_Summary: segmentation, filter, initial scaling (parity unpinned, see nc_oracle.h)
# ------------------------------------------------------------------------------------------------
ED_DTYPE = np.dtype([("mean", "<f8"), ("stdv", "<f8"), ("start", "<i8"), ("length", "<i8")])


class F5Opts(C.Structure):
    _fields_ = [("min_ed_events", C.c_uint), ("max_ed_events", C.c_uint), ("abasic_level_top_percent", C.c_double),
                ("abasic_level_top_offset", C.c_double), ("template_only", C.c_uint), ("trim_margins", C.c_uint * 4)]


class F5Summary(C.Structure):
    _fields_ = [("num_ed_events", C.c_uint), ("abasic_level", C.c_float), ("strand_bounds", C.c_uint * 4),
                ("scale_strands_together", C.c_int), ("time_length", C.c_float * 2)]


def f5_opts(pore="r73", template_only=False, min_ed_events=10, max_ed_events=100000, trim=(50, 50, 50, 50)):
    """The option singletons main() sets (nanocall.cpp:925-964)."""
    o = F5Opts()
    o.min_ed_events, o.max_ed_events = min_ed_events, max_ed_events
    o.abasic_level_top_percent = 1.0
    o.abasic_level_top_offset = {"r9": 0.0, "r73": 5.0}[pore]
    o.template_only = int(template_only)
    o.trim_margins[:] = list(trim)
    return o


def mean_stdv(v):
    v = np.ascontiguousarray(v, np.float32)
    m, s = C.c_float(0), C.c_float(0)
    lib().nco_mean_stdv(_p(v), v.shape[0], C.byref(m), C.byref(s))
    return np.float32(m.value), np.float32(s.value)


def f5_summarize(opts, ed, sampling_rate, sst):
    ed = np.ascontiguousarray(ed, ED_DTYPE)
    out = F5Summary()
    lib().nco_f5_summarize(C.byref(opts), _p(ed), ed.shape[0], C.c_float(sampling_rate), int(sst), C.byref(out))
    return out


def f5_load_events(summary, ed, sampling_rate, st):
    """-> (mean, stdv, start, length) float32 arrays of the filtered events of strand st."""
    ed = np.ascontiguousarray(ed, ED_DTYPE)
    cap = max(0, int(summary.strand_bounds[2 * st + 1]) - int(summary.strand_bounds[2 * st]))
    bufs = [np.empty(cap, np.float32) for _ in range(4)]
    n = lib().nco_f5_load_events(C.byref(summary), _p(ed), C.c_float(sampling_rate), st, *[_p(b) for b in bufs]) if cap else 0
    return tuple(b[:n].copy() for b in bufs)


def f5_initial_scaling(together, r0, r1, m0, m1):
    a = [np.ascontiguousarray(x if x is not None else (0, 1), np.float32) for x in (r0, r1, m0, m1)]
    out = np.empty(2, np.float32)
    lib().nco_f5_initial_scaling(int(together), *[_p(x) for x in a], _p(out))
    return out


def logf_mismatches(x, y):
    """Number of elements where the host libm's logf(x) differs bitwise from y (both float32), and the first index."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    first = C.c_longlong(-1)
    n = lib().nco_logf_mismatches(_p(x), _p(y), x.shape[0], C.byref(first))
    return int(n), int(first.value)
