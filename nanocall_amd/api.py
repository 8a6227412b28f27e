"""numpy / torch level wrappers over the C ABI (include/nanocall_hip.h).

Host-prep wrappers take and return numpy arrays.  `Context` owns an nchmm_ctx; its `viterbi` /
`fwbw` methods take host (numpy) buffers, its `*_dev` methods take torch CUDA tensors that are
already resident in HBM (what bench.py times).
"""
import ctypes as C

import numpy as np

from ._lib import lib, check, NchmmError, TrainOpts, SegmentOpts, ReadSummary, Fast5Read  # noqa: F401

S = 4096
MAX_ARCS = S * 21


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------------------------------------
# host prep
# ------------------------------------------------------------------------------------------------
def model_load(table_Sx4):
    """Pore_Model::load_from_vector -> S x 10 state array (field order of Pore_Model_State)."""
    t = _f32(table_Sx4).reshape(S, 4)
    st = np.empty((S, 10), np.float32)
    check(lib().nchmm_model_load(_p(t), _p(st)), "nchmm_model_load")
    return st


def model_scale(state_Sx10, params):
    """Pore_Model::scale; params = (scale, shift, drift, var, scale_sd, var_sd). Returns a new array."""
    st = _f32(state_Sx10).reshape(S, 10).copy()
    p = _f32(params).reshape(6)
    check(lib().nchmm_model_scale(_p(st), _p(p)), "nchmm_model_scale")
    return st


def model_pack6(state_Sx10):
    st = _f32(state_Sx10).reshape(S, 10)
    t6 = np.empty((S, 6), np.float32)
    check(lib().nchmm_model_pack6(_p(st), _p(t6)), "nchmm_model_pack6")
    return t6


def scaled_model_table(table_Sx4, params=(1.0, 0.0, 0.0, 1.0, 1.0, 1.0)):
    """load_from_vector + scale + pack6: the S x 6 table nchmm_put_model takes."""
    return model_pack6(model_scale(model_load(table_Sx4), params))


def transitions_fast(p_skip, p_stay):
    """State_Transitions::compute_transitions_fast -> (row_ptr[S+1] u32, pred[n] u16, logw[n] f32)."""
    rp = np.empty(S + 1, np.uint32)
    pred = np.empty(MAX_ARCS, np.uint16)
    w = np.empty(MAX_ARCS, np.float32)
    n = C.c_uint32(0)
    check(lib().nchmm_transitions_fast(C.c_float(p_skip), C.c_float(p_stay), _p(rp), _p(pred), _p(w), C.byref(n)),
          "nchmm_transitions_fast")
    return rp, pred[: n.value].copy(), w[: n.value].copy()


def events_prepare(mean, stdv, start=None, drift=0.0):
    """Event::update_logs + apply_drift_correction -> (corrected_mean, stdv, log_stdv)."""
    mean = _f32(mean)
    stdv = _f32(stdv).copy()
    start = None if start is None else _f32(start)
    n = mean.shape[0]
    cm = np.empty(n, np.float32)
    ls = np.empty(n, np.float32)
    check(lib().nchmm_events_prepare(n, _p(mean), _p(stdv), _p(start), C.c_float(drift), _p(cm), _p(ls)),
          "nchmm_events_prepare")
    return cm, stdv, ls


def base_seq(states):
    """fill_move_seq + get_base_seq -> (moves int32[n], sequence str)."""
    st = np.ascontiguousarray(states, dtype=np.uint16)
    n = st.shape[0]
    mv = np.empty(n, np.int32)
    buf = C.create_string_buffer(6 * max(n, 1) + 1)
    ln = C.c_size_t(0)
    check(lib().nchmm_base_seq(n, _p(st), _p(mv), buf, C.byref(ln)), "nchmm_base_seq")
    return mv, buf.raw[: ln.value].decode()


def write_fasta(name, seq, line_width=80):
    cap = len(name) + len(seq) + len(seq) // max(line_width, 1) + 16
    buf = C.create_string_buffer(cap)
    wr = C.c_size_t(0)
    check(lib().nchmm_write_fasta(name.encode(), seq.encode(), line_width, buf, cap, C.byref(wr)), "nchmm_write_fasta")
    return buf.raw[: wr.value].decode()


def st_train_kmers():
    out = np.empty(S, np.uint16)
    n = C.c_uint32(0)
    check(lib().nchmm_st_train_kmers(_p(out), C.byref(n)), "nchmm_st_train_kmers")
    return out[: n.value].copy()


def train_pm_finish(pm_sums, mean, stdv, start, crt_pm, train_drift=True):
    """train_pm_params' host half -> (new_pm[6], done)."""
    pm = _f32(pm_sums).reshape(-1, 6)
    n = pm.shape[0]
    mean, stdv = _f32(mean), _f32(stdv)
    start = None if start is None else _f32(start)
    new = np.empty(6, np.float32)
    done = C.c_int(0)
    check(lib().nchmm_train_pm_finish(n, _p(pm), _p(mean), _p(stdv), _p(start), int(train_drift),
                                      _p(_f32(crt_pm).reshape(6)), _p(new), C.byref(done)), "nchmm_train_pm_finish")
    return new, bool(done.value)


def train_pm_solve(n_events, acc13, crt_pm, train_drift=True):
    """The solve half of train_pm_finish from the thirteen outer sums -> (new_pm[6], done)."""
    acc = np.ascontiguousarray(acc13, np.float64).reshape(13)
    new = np.empty(6, np.float32)
    done = C.c_int(0)
    check(lib().nchmm_train_pm_solve(int(n_events), _p(acc), int(train_drift), _p(_f32(crt_pm).reshape(6)), _p(new), C.byref(done)),
          "nchmm_train_pm_solve")
    return new, bool(done.value)


def train_st_finish(st_sums):
    """train_st_params' host half for one strand -> (p_stay, p_skip)."""
    st = _f32(st_sums).reshape(-1, 3)
    a, b = C.c_float(0), C.c_float(0)
    check(lib().nchmm_train_st_finish(st.shape[0], _p(st), C.byref(a), C.byref(b)), "nchmm_train_st_finish")
    return a.value, b.value


def train_opts(**kw):
    """nchmm_train_opts with the reference's CLI defaults, overridden by keyword."""
    o = TrainOpts()
    check(lib().nchmm_train_opts_default(C.byref(o)), "nchmm_train_opts_default")
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def train_enumerate(opts, model_strand, strand_off, together):
    """Job list of train_reads -> (job_read, job_m0, job_m1) int32 arrays."""
    ms = np.ascontiguousarray(model_strand, np.int32)
    so = np.ascontiguousarray(strand_off, np.uint64)
    tg = np.ascontiguousarray(together, np.uint8)
    n_reads = tg.shape[0]
    n = C.c_size_t(0)
    check(lib().nchmm_train_enumerate(C.byref(opts), ms.shape[0], _p(ms), n_reads, _p(so), _p(tg), C.byref(n), None, None, None),
          "nchmm_train_enumerate")
    jr, j0, j1 = (np.empty(n.value, np.int32) for _ in range(3))
    check(lib().nchmm_train_enumerate(C.byref(opts), ms.shape[0], _p(ms), n_reads, _p(so), _p(tg), C.byref(n), _p(jr), _p(j0), _p(j1)),
          "nchmm_train_enumerate")
    return jr, j0, j1


# ------------------------------------------------------------------------------------------------
# read summary (Fast5_Summary arithmetic on an EventDetection table)
# ------------------------------------------------------------------------------------------------
ED_DTYPE = np.dtype([("mean", "<f8"), ("stdv", "<f8"), ("start", "<i8"), ("length", "<i8")])   # nchmm_ed_event


def segment_opts(pore="r9", **kw):
    o = SegmentOpts()
    check(lib().nchmm_segment_opts_default(C.byref(o), pore.encode()), "nchmm_segment_opts_default")
    for k, v in kw.items():
        if k == "trim_margins":
            o.trim_margins[:] = list(v)
        else:
            setattr(o, k, v)
    return o


def mean_stdv(v):
    v = _f32(v)
    m, s = C.c_float(0), C.c_float(0)
    check(lib().nchmm_mean_stdv(v.shape[0], _p(v), C.byref(m), C.byref(s)), "nchmm_mean_stdv")
    return np.float32(m.value), np.float32(s.value)


def read_summarize(opts, ed, sampling_rate, double_strand_scaling=True):
    ed = np.ascontiguousarray(ed, ED_DTYPE)
    out = ReadSummary()
    check(lib().nchmm_read_summarize(C.byref(opts), ed.shape[0], _p(ed), C.c_float(sampling_rate), int(double_strand_scaling),
                                     C.byref(out)), "nchmm_read_summarize")
    return out


def read_load_events(summary, ed, sampling_rate, st):
    """-> (mean, stdv, start, length) float32 arrays of the filtered events of strand st."""
    ed = np.ascontiguousarray(ed, ED_DTYPE)
    cap = max(0, int(summary.strand_bounds[2 * st + 1]) - int(summary.strand_bounds[2 * st]))
    bufs = [np.empty(max(cap, 1), np.float32) for _ in range(4)]
    n = C.c_size_t(0)
    check(lib().nchmm_read_load_events(C.byref(summary), _p(ed), C.c_float(sampling_rate), st, *[_p(b) for b in bufs], C.byref(n)),
          "nchmm_read_load_events")
    return tuple(b[: n.value].copy() for b in bufs)


def initial_scaling(together, r0, r1, m0, m1):
    a = [None if x is None else _f32(x) for x in (r0, r1, m0, m1)]
    sc, sh = C.c_float(0), C.c_float(0)
    check(lib().nchmm_initial_scaling(int(together), *[_p(x) for x in a], C.byref(sc), C.byref(sh)), "nchmm_initial_scaling")
    return np.float32(sc.value), np.float32(sh.value)


# ------------------------------------------------------------------------------------------------
# FAST5 ingest (include/nanocall_fast5.h)
# ------------------------------------------------------------------------------------------------
def fast5_available():
    return bool(lib().nchmm_fast5_available())


def fast5_is_valid_file(path):
    return bool(lib().nchmm_fast5_is_valid_file(str(path).encode()))


def fast5_load(path, ed_group=""):
    """-> dict(have_sampling_rate, have_events, sampling_rate, ed_group, read_name, read_id, events[ED_DTYPE])"""
    r = Fast5Read()
    check(lib().nchmm_fast5_load(str(path).encode(), ed_group.encode(), C.byref(r)), "nchmm_fast5_load")
    try:
        ev = np.zeros(r.n_events, ED_DTYPE)
        if r.n_events:
            C.memmove(ev.ctypes.data, r.events, r.n_events * ED_DTYPE.itemsize)
        return dict(have_sampling_rate=bool(r.have_sampling_rate), have_events=bool(r.have_events), sampling_rate=r.sampling_rate,
                    ed_group=r.ed_group.decode(), read_name=r.read_name.decode(), read_id=r.read_id.decode(), events=ev)
    finally:
        lib().nchmm_fast5_release(C.byref(r))


# ------------------------------------------------------------------------------------------------
# device context
# ------------------------------------------------------------------------------------------------
def _dp(t):
    """device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


class Context:
    """Owns one nchmm_ctx (one GPU).  Raises NchmmError when there is no usable device."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().nchmm_create(C.byref(self._h), int(device)), "nchmm_create")
        self.device = int(device)

    def close(self):
        if self._h:
            lib().nchmm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- tables --
    def put_model(self, slot, table_Sx6):
        t = _f32(table_Sx6).reshape(S, 6)
        check(lib().nchmm_put_model(self._h, slot, _p(t)), "nchmm_put_model")

    def put_transitions(self, slot, row_ptr, pred, logw):
        rp = np.ascontiguousarray(row_ptr, np.uint32)
        pr = np.ascontiguousarray(pred, np.uint16)
        w = _f32(logw)
        check(lib().nchmm_put_transitions(self._h, slot, _p(rp), _p(pr), _p(w)), "nchmm_put_transitions")

    def reserve_slots(self, n):
        check(lib().nchmm_reserve_slots(self._h, int(n)), "nchmm_reserve_slots")

    def put_models_scaled(self, first_slot, states, table_idx, params):
        """Batched scale + upload: states = [n_tables, S, 10] (model_load outputs), params = [n, 6]."""
        st = _f32(states).reshape(-1, S, 10)
        idx = np.ascontiguousarray(table_idx, np.int32)
        par = _f32(params).reshape(-1, 6)
        assert idx.shape[0] == par.shape[0] and idx.max(initial=0) < st.shape[0]
        check(lib().nchmm_put_models_scaled(self._h, int(first_slot), idx.shape[0], _p(st), _p(idx), _p(par)),
              "nchmm_put_models_scaled")

    def put_transitions_fast(self, first_slot, p_skip, p_stay):
        ps, pt = _f32(np.atleast_1d(p_skip)), _f32(np.atleast_1d(p_stay))
        check(lib().nchmm_put_transitions_fast(self._h, int(first_slot), ps.shape[0], _p(ps), _p(pt)),
              "nchmm_put_transitions_fast")

    def set_stream(self, hip_stream_ptr):
        """Launch on the caller's hipStream_t; 0 is the legacy default stream (torch's default stream)."""
        check(lib().nchmm_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)), "nchmm_set_stream")

    def use_own_stream(self):
        check(lib().nchmm_use_own_stream(self._h), "nchmm_use_own_stream")

    def synchronize(self):
        check(lib().nchmm_synchronize(self._h), "nchmm_synchronize")

    # -- Viterbi --
    def viterbi(self, off, cmean, stdv, log_stdv, model_slot=None, trans_slot=None, raise_on_numeric=True):
        """Host-buffer batch Viterbi.  Returns (states u16[total], path_logp f32[n_reads], status i32[n_reads])."""
        off = np.ascontiguousarray(off, np.uint64)
        n = off.shape[0] - 1
        total = int(off[-1]) if n > 0 else 0
        cm, sd, ls = _f32(cmean), _f32(stdv), _f32(log_stdv)
        assert cm.shape[0] >= total and sd.shape[0] >= total and ls.shape[0] >= total
        ms = None if model_slot is None else np.ascontiguousarray(model_slot, np.int32)
        ts = None if trans_slot is None else np.ascontiguousarray(trans_slot, np.int32)
        states = np.empty(total, np.uint16)
        logp = np.empty(max(n, 0), np.float32)
        status = np.zeros(max(n, 0), np.int32)
        rc = lib().nchmm_viterbi(self._h, n, _p(off), _p(cm), _p(sd), _p(ls), _p(ms), _p(ts),
                                 _p(states), _p(logp), _p(status))
        if rc != 0 and not (rc == -6 and not raise_on_numeric):
            check(rc, "nchmm_viterbi")
        return states, logp, status

    def viterbi_begin(self, off, cmean, stdv, log_stdv, model_slot=None, trans_slot=None, out=None):
        """nchmm_viterbi_begin: enqueue one batch (copy-in + kernels) and return a ticket; at most three may be in flight.
        `out` = (states, logp, status) arrays to reuse (a streaming caller recycles them: fresh pages cost a fault each).
        The ticket keeps every array alive until viterbi_end."""
        off = np.ascontiguousarray(off, np.uint64)
        n = off.shape[0] - 1
        total = int(off[-1]) if n > 0 else 0
        cm, sd, ls = _f32(cmean), _f32(stdv), _f32(log_stdv)
        assert cm.shape[0] >= total and sd.shape[0] >= total and ls.shape[0] >= total
        ms = None if model_slot is None else np.ascontiguousarray(model_slot, np.int32)
        ts = None if trans_slot is None else np.ascontiguousarray(trans_slot, np.int32)
        if out is None:
            out = (np.empty(total, np.uint16), np.empty(max(n, 0), np.float32), np.zeros(max(n, 0), np.int32))
        states, logp, status = out
        assert states.dtype == np.uint16 and states.shape[0] >= total and logp.shape[0] >= n and status.shape[0] >= n
        check(lib().nchmm_viterbi_begin(self._h, n, _p(off), _p(cm), _p(sd), _p(ls), _p(ms), _p(ts), _p(states), _p(logp), _p(status)),
              "nchmm_viterbi_begin")
        return {"keep": (off, cm, sd, ls, ms, ts), "out": (states, logp, status)}

    def viterbi_end(self, ticket, raise_on_numeric=True):
        """nchmm_viterbi_end: complete the OLDEST batch in flight (tickets end in the order they began)."""
        rc = lib().nchmm_viterbi_end(self._h)
        if rc != 0 and not (rc == -6 and not raise_on_numeric):
            check(rc, "nchmm_viterbi_end")
        return ticket["out"]

    def viterbi_in_flight(self):
        return int(lib().nchmm_viterbi_in_flight(self._h))

    def viterbi_raw(self, mean, stdv, start, src, length, drift, model_slot=None, trans_slot=None, raise_on_numeric=True):
        """nchmm_viterbi_raw: candidates over raw events, host prep on the device.
        Returns (states u16[sum(length)] packed by candidate, path_logp f32[n_cand], status i32[n_cand])."""
        mean, stdv, start = _f32(mean), _f32(stdv), _f32(start)
        src = np.ascontiguousarray(src, np.uint64)
        ln = np.ascontiguousarray(length, np.uint32)
        dr = _f32(drift)
        n = src.shape[0]
        ms = None if model_slot is None else np.ascontiguousarray(model_slot, np.int32)
        ts = None if trans_slot is None else np.ascontiguousarray(trans_slot, np.int32)
        states = np.empty(int(ln.sum()), np.uint16)
        logp = np.empty(n, np.float32)
        status = np.zeros(n, np.int32)
        rc = lib().nchmm_viterbi_raw(self._h, mean.shape[0], _p(mean), _p(stdv), _p(start), n, _p(src), _p(ln), _p(dr), _p(ms), _p(ts),
                                     _p(states), _p(logp), _p(status))
        if rc != 0 and not (rc == -6 and not raise_on_numeric):
            check(rc, "nchmm_viterbi_raw")
        return states, logp, status

    def logf(self, x):
        """Device logf (bit-identical port of glibc's) on a host array."""
        x = _f32(x)
        out = np.empty_like(x)
        check(lib().nchmm_logf(self._h, x.shape[0], _p(x), _p(out)), "nchmm_logf")
        return out

    def viterbi_dev(self, n_reads, max_events, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_out_state,
                    d_out_logp, d_out_status=None, d_model_slot=None, d_trans_slot=None, d_order=None):
        """Device-resident batch Viterbi on torch CUDA tensors; asynchronous on the context's stream."""
        check(lib().nchmm_viterbi_dev(self._h, n_reads, max_events, total_events, _dp(d_off), _dp(d_cmean),
                                      _dp(d_stdv), _dp(d_lstdv), _dp(d_model_slot), _dp(d_trans_slot),
                                      _dp(d_order), _dp(d_out_state), _dp(d_out_logp), _dp(d_out_status)),
              "nchmm_viterbi_dev")

    def viterbi_dev_enqueue(self, n_reads, max_events, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_out_state,
                            d_out_logp, d_out_status=None, d_model_slot=None, d_trans_slot=None, d_order=None):
        """nchmm_viterbi_dev_enqueue: queue the batch on one of the context's internal streams (it may run beside the
        batch queued before it); outputs are complete after viterbi_dev_join() / synchronize()."""
        check(lib().nchmm_viterbi_dev_enqueue(self._h, n_reads, max_events, total_events, _dp(d_off), _dp(d_cmean),
                                              _dp(d_stdv), _dp(d_lstdv), _dp(d_model_slot), _dp(d_trans_slot),
                                              _dp(d_order), _dp(d_out_state), _dp(d_out_logp), _dp(d_out_status)),
              "nchmm_viterbi_dev_enqueue")

    def viterbi_strand(self, table_Sx6, p_skip, p_stay, cmean, stdv, log_stdv):
        """nchmm_viterbi_strand: one strand, blocking, thread-safe on one context (concurrent calls are combined into launches;
        ctypes releases the GIL for the duration).  Returns (states u16[n], path_logp, status)."""
        t6 = _f32(table_Sx6)
        cm, sd, ls = _f32(cmean), _f32(stdv), _f32(log_stdv)
        n = cm.shape[0]
        states = np.empty(n, np.uint16)
        lp = C.c_float(0.0)
        rc = lib().nchmm_viterbi_strand(self._h, _p(t6), C.c_float(p_skip), C.c_float(p_stay), n, _p(cm), _p(sd), _p(ls), _p(states), C.byref(lp))
        if rc not in (0, -6):
            check(rc, "nchmm_viterbi_strand")
        return states, np.float32(lp.value), rc

    def viterbi_strand_scaled(self, unscaled_states, pm_params, p_skip, p_stay, cmean, stdv, log_stdv):
        """nchmm_viterbi_strand_scaled: as viterbi_strand, the model given as (unscaled S x 10 states from model_load, parameters)."""
        un, pm6 = _f32(unscaled_states), _f32(pm_params)
        cm, sd, ls = _f32(cmean), _f32(stdv), _f32(log_stdv)
        n = cm.shape[0]
        states = np.empty(n, np.uint16)
        lp = C.c_float(0.0)
        rc = lib().nchmm_viterbi_strand_scaled(self._h, _p(un), _p(pm6), C.c_float(p_skip), C.c_float(p_stay), n, _p(cm), _p(sd), _p(ls), _p(states), C.byref(lp))
        if rc not in (0, -6):
            check(rc, "nchmm_viterbi_strand_scaled")
        return states, np.float32(lp.value), rc

    def viterbi_dev_join(self):
        check(lib().nchmm_viterbi_dev_join(self._h), "nchmm_viterbi_dev_join")

    def fwbw_windows(self, unscaled_states, pm_params, p_skip, p_stay, off, cmean, stdv, log_stdv, win_model, st_params=None):
        """nchmm_fwbw_windows: one read's training windows over its own models (unscaled S x 10 states from model_load, all scaled
        by pm_params on the device), blocking, thread-safe on one context (concurrent calls are combined into launches).
        Returns dict(log_pr_data, pm_sums, st_sums) like fwbw."""
        tabs = [_f32(t) for t in unscaled_states]
        ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
        pm6 = _f32(pm_params)
        ps, pt = _f32(p_skip), _f32(p_stay)
        off = np.ascontiguousarray(off, np.uint64)
        n = off.shape[0] - 1
        total = int(off[-1]) if n > 0 else 0
        cm, sd, ls = _f32(cmean), _f32(stdv), _f32(log_stdv)
        wm = np.ascontiguousarray(win_model, np.int32)
        sp = None if st_params is None else _f32(st_params).reshape(n, 2)
        lpd = np.empty(n, np.float32)
        pm = np.empty((total, 6), np.float32)
        stt = np.empty((n, 3), np.float32)
        check(lib().nchmm_fwbw_windows(self._h, len(tabs), ptrs, _p(pm6), _p(ps), _p(pt), n, _p(off), _p(cm), _p(sd), _p(ls), _p(wm), _p(sp),
                                       _p(lpd), _p(pm), _p(stt)), "nchmm_fwbw_windows")
        return dict(log_pr_data=lpd, pm_sums=pm, st_sums=stt)

    # -- forward-backward --
    def fwbw(self, off, cmean, stdv, log_stdv, scaled_slot=None, pm_params=None, trans_slot=None,
             st_params=None, want_matrices=False):
        """Host-buffer batch forward-backward + EM sums.
        pm_params (n_win x 6, or one row for all windows) are the scaling parameters behind each window's
        scaled model; the pm sums are taken over the corresponding unscaled model (None: identity).
        Returns dict(log_pr_data[n_win], pm_sums[total,6], st_sums[n_win,3], alpha, beta)."""
        off = np.ascontiguousarray(off, np.uint64)
        n = off.shape[0] - 1
        total = int(off[-1]) if n > 0 else 0
        cm, sd, ls = _f32(cmean), _f32(stdv), _f32(log_stdv)
        ss = None if scaled_slot is None else np.ascontiguousarray(scaled_slot, np.int32)
        us = None if pm_params is None else np.ascontiguousarray(np.broadcast_to(_f32(pm_params).reshape(-1, 6), (n, 6)))
        ts = None if trans_slot is None else np.ascontiguousarray(trans_slot, np.int32)
        sp = None if st_params is None else _f32(st_params).reshape(n, 2)
        lpd = np.empty(n, np.float32)
        pm = np.empty((total, 6), np.float32)
        stt = np.empty((n, 3), np.float32)
        al = np.empty((total, S), np.float32) if want_matrices else None
        be = np.empty((total, S), np.float32) if want_matrices else None
        check(lib().nchmm_fwbw(self._h, n, _p(off), _p(cm), _p(sd), _p(ls), _p(ss), _p(us), _p(ts), _p(sp),
                               _p(lpd), _p(pm), _p(stt), _p(al), _p(be)), "nchmm_fwbw")
        return dict(log_pr_data=lpd, pm_sums=pm, st_sums=stt, alpha=al, beta=be)

    def fwbw_dev(self, n_win, max_events, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_out_lpd, d_out_pm,
                 d_out_st, d_scaled_slot=None, d_pm_params=None, d_trans_slot=None, d_st_params=None,
                 d_out_alpha=None, d_out_beta=None):
        check(lib().nchmm_fwbw_dev(self._h, n_win, max_events, total_events, _dp(d_off), _dp(d_cmean), _dp(d_stdv),
                                   _dp(d_lstdv), _dp(d_scaled_slot), _dp(d_pm_params), _dp(d_trans_slot),
                                   _dp(d_st_params), _dp(d_out_lpd), _dp(d_out_pm), _dp(d_out_st),
                                   _dp(d_out_alpha), _dp(d_out_beta)), "nchmm_fwbw_dev")

    # -- one EM round with resident events --
    def em_load_events(self, mean, stdv, start, log_stdv):
        mean, stdv, start, ls = _f32(mean), _f32(stdv), _f32(start), _f32(log_stdv)
        check(lib().nchmm_em_load_events(self._h, mean.shape[0], _p(mean), _p(stdv), _p(start), _p(ls)), "nchmm_em_load_events")

    def em_round(self, win_src, win_len, win_drift, win_pm, scaled_slot, trans_slot, st_params, job_first_win, train_drift=True):
        """-> dict(log_pr_data[n_win], st_sums[n_win,3], acc[n_jobs,13])"""
        ws = np.ascontiguousarray(win_src, np.uint64)
        n = ws.shape[0]
        wl = np.ascontiguousarray(win_len, np.uint32)
        wd = _f32(win_drift)
        wp = np.ascontiguousarray(np.broadcast_to(_f32(win_pm).reshape(-1, 6), (n, 6)))
        ss = np.ascontiguousarray(scaled_slot, np.int32)
        ts = np.ascontiguousarray(trans_slot, np.int32)
        sp = _f32(st_params).reshape(n, 2)
        jf = np.ascontiguousarray(job_first_win, np.uint32)
        nj = jf.shape[0] - 1
        lpd = np.empty(n, np.float32)
        st = np.empty((n, 3), np.float32)
        acc = np.empty((nj, 13), np.float64)
        check(lib().nchmm_em_round(self._h, n, _p(ws), _p(wl), _p(wd), _p(wp), _p(ss), _p(ts), _p(sp), nj, _p(jf), int(train_drift),
                                   _p(lpd), _p(st), _p(acc)), "nchmm_em_round")
        return dict(log_pr_data=lpd, st_sums=st, acc=acc)

    # -- EM driver --
    def train_reads(self, opts, model_states, strand_off, mean, stdv, start, job_read, job_m0, job_m1, init_pm=None,
                    init_st=None):
        """nchmm_train_reads -> dict(pm[n_jobs,6], st[n_jobs,4], fit, rounds, preferred[n_reads,3])."""
        st10 = _f32(model_states).reshape(-1, S, 10)
        so = np.ascontiguousarray(strand_off, np.uint64)
        n_reads = (so.shape[0] - 1) // 2
        jr, j0, j1 = (np.ascontiguousarray(a, np.int32) for a in (job_read, job_m0, job_m1))
        nj = jr.shape[0]
        pm = np.tile(np.float32([1, 0, 0, 1, 1, 1]), (nj, 1)) if init_pm is None else _f32(init_pm).reshape(nj, 6).copy()
        st = (np.tile(np.float32([opts.default_p_stay, opts.default_p_skip] * 2), (nj, 1)) if init_st is None
              else _f32(init_st).reshape(nj, 4).copy())
        fit = np.empty(nj, np.float32)
        rounds = np.empty(nj, np.uint32)
        pref = np.empty((n_reads, 3), np.int32)
        check(lib().nchmm_train_reads(self._h, C.byref(opts), st10.shape[0], _p(st10), n_reads, _p(so), _p(_f32(mean)),
                                      _p(_f32(stdv)), _p(_f32(start)), nj, _p(jr), _p(j0), _p(j1), _p(pm), _p(st), _p(fit),
                                      _p(rounds), _p(pref)), "nchmm_train_reads")
        return dict(pm=pm, st=st, fit=fit, rounds=rounds, preferred=pref)

    def basecall_reads(self, opts, model_states, strand_off, mean, stdv, start, job_read, job_m0, job_m1, job_pm, job_st,
                       preferred=None, out=None):
        """nchmm_basecall_reads -> dict(states u16[total], best_job i32[n_reads,2], best_logp f32[n_reads,2]).
        `out` = a previous result to write into (a caller that decodes chunk after chunk reuses its arrays, as the command
        line does; fresh zero pages cost a fault each when the winners are copied in)."""
        st10 = _f32(model_states).reshape(-1, S, 10)
        so = np.ascontiguousarray(strand_off, np.uint64)
        n_reads = (so.shape[0] - 1) // 2
        jr, j0, j1 = (np.ascontiguousarray(a, np.int32) for a in (job_read, job_m0, job_m1))
        pm, st = _f32(job_pm).reshape(-1, 6), _f32(job_st).reshape(-1, 4)
        pref = None if preferred is None else np.ascontiguousarray(preferred, np.int32).reshape(n_reads, 3)
        if out is not None:
            states, bj, bl = out["states"], out["best_job"], out["best_logp"]
            assert states.shape[0] >= int(so[-1]) and bj.shape == (n_reads, 2) and bl.shape == (n_reads, 2)
        else:
            states = np.zeros(int(so[-1]), np.uint16)
            bj = np.empty((n_reads, 2), np.int32)
            bl = np.empty((n_reads, 2), np.float32)
        rc = lib().nchmm_basecall_reads(self._h, C.byref(opts), st10.shape[0], _p(st10), n_reads, _p(so), _p(_f32(mean)),
                                        _p(_f32(stdv)), _p(_f32(start)), jr.shape[0], _p(jr), _p(j0), _p(j1), _p(pm), _p(st),
                                        _p(pref), _p(states), _p(bj), _p(bl))
        if rc not in (0, -6):
            check(rc, "nchmm_basecall_reads")
        if out is not None:
            # the library writes the states of strands that HAVE a winner; a reused array still holds the previous chunk's
            # decode under the others (fresh arrays are zero there): clear those strands, so that the result does not depend
            # on what the array was used for before
            for r, s in zip(*np.nonzero(bj < 0)):
                states[int(so[2 * r + s]):int(so[2 * r + s + 1])] = 0
        return dict(states=states, best_job=bj, best_logp=bl)

    # -- introspection --
    def counters(self):
        out = np.zeros(8, np.uint64)
        check(lib().nchmm_counters(self._h, _p(out)), "nchmm_counters")
        return out

    def last_kernel_ms(self):
        """(viterbi forward ms, traceback ms, fwbw ms, 0) of the most recent launches (hipEvents)."""
        out = np.zeros(4, np.float32)
        check(lib().nchmm_last_kernel_ms(self._h, _p(out)), "nchmm_last_kernel_ms")
        return tuple(float(x) for x in out)

    def shader_clock_mhz(self):
        """nchmm_shader_clock_mhz: the shader clock the device sustains under a full-chip VALU load, measured now."""
        out = C.c_double(0.0)
        check(lib().nchmm_shader_clock_mhz(self._h, C.byref(out)), "nchmm_shader_clock_mhz")
        return float(out.value)

    def profile_ticks(self, reset=True):
        out = np.zeros(8, np.uint64)
        check(lib().nchmm_profile_ticks(self._h, _p(out), int(reset)), "nchmm_profile_ticks")
        return out

    def profile_blocks(self):
        out = np.zeros(6144, np.uint64)
        check(lib().nchmm_profile_blocks(self._h, _p(out)), "nchmm_profile_blocks")
        return out

    def mem_stats(self):
        """(device bytes the context holds now, at its high-water mark)"""
        out = np.zeros(2, np.uint64)
        check(lib().nchmm_mem_stats(self._h, _p(out)), "nchmm_mem_stats")
        return int(out[0]), int(out[1])

    def reserve_workspaces(self, fb_events=0, viterbi_longest=None):
        """take the FB alpha-row workspace (for batches of up to fb_events window events) and / or the Viterbi back-pointer regions
        (for reads of up to viterbi_longest events; 0 = as long as a full pool fits in the budget) now instead of at first use"""
        if fb_events:
            check(lib().nchmm_reserve_fb_workspace(self._h, int(fb_events)), "nchmm_reserve_fb_workspace")
        if viterbi_longest is not None:
            check(lib().nchmm_reserve_viterbi_workspace(self._h, int(viterbi_longest)), "nchmm_reserve_viterbi_workspace")

    def grid_slots(self):
        v = C.c_int(0)
        check(lib().nchmm_grid_slots(self._h, C.byref(v)), "nchmm_grid_slots")
        return v.value

    def set_sweep(self, mode):
        """which form of the Viterbi sweep launches take: "auto" (per launch, from the read lengths), "wide" (two reads per
        CU on 8 waves each), "ll" (one read per CU on 16 waves), "ahead" (ll with the emissions of the longest reads computed
        ahead by the whole device) -- bit-identical results"""
        check(lib().nchmm_set_sweep(self._h, {"auto": 0, "wide": 1, "ll": 2, "ahead": 3}[mode]), "nchmm_set_sweep")

    def ahead_stats(self):
        """(low-latency launches with emissions ahead, reads ahead, events ahead) so far"""
        out = (C.c_uint64 * 3)()
        check(lib().nchmm_ahead_stats(self._h, out), "nchmm_ahead_stats")
        return tuple(int(v) for v in out)

    def sweep_stats(self):
        """(launches wide, launches ll, reads wide, reads ll) so far"""
        out = (C.c_uint64 * 4)()
        check(lib().nchmm_sweep_stats(self._h, out), "nchmm_sweep_stats")
        return tuple(int(v) for v in out)


# ------------------------------------------------------------------------------------------------
# device pool (one context + host thread per GPU, reads sharded by event count)
# ------------------------------------------------------------------------------------------------
def device_count():
    n = C.c_int(0)
    lib().nchmm_device_count(C.byref(n))
    return n.value


def rccl_unique_id():
    """nchmm_rccl_unique_id -> 128 bytes (rank 0 of a one-process-per-GPU run makes it, the others receive it)"""
    b = np.zeros(128, np.uint8)
    check(lib().nchmm_rccl_unique_id(_p(b)), "nchmm_rccl_unique_id")
    return b


def counters_allreduce(device_id, n_ranks, rank, unique_id, counters8):
    """nchmm_counters_allreduce: ncclCommInitRank + one all-reduce (sum) of eight uint64 across the ranks -> the sums"""
    c = np.ascontiguousarray(counters8, np.uint64).copy()
    assert c.shape == (8,)
    check(lib().nchmm_counters_allreduce(int(device_id), int(n_ranks), int(rank), _p(np.ascontiguousarray(unique_id, np.uint8)), _p(c)),
          "nchmm_counters_allreduce")
    return c


def device_mem_info(device_id=0):
    """nchmm_device_mem_info -> (free, total) bytes of one GPU, through the library's own HIP runtime."""
    f, t = C.c_uint64(0), C.c_uint64(0)
    check(lib().nchmm_device_mem_info(int(device_id), C.byref(f), C.byref(t)), "nchmm_device_mem_info")
    return f.value, t.value


def lpt_partition(weights, n_shards):
    """nchmm_lpt_partition -> shard index per item (int32)."""
    w = np.ascontiguousarray(weights, np.uint64)
    out = np.empty(w.shape[0], np.int32)
    check(lib().nchmm_lpt_partition(w.shape[0], _p(w), int(n_shards), _p(out)), "nchmm_lpt_partition")
    return out


class Pool:
    """nchmm_pool: device_ids may repeat (several contexts on one GPU) to exercise the sharding on one device."""

    def __init__(self, device_ids):
        ids = np.ascontiguousarray(device_ids, np.int32)
        self._h = C.c_void_p()
        check(lib().nchmm_pool_create(C.byref(self._h), ids.shape[0], _p(ids)), "nchmm_pool_create")

    def close(self):
        if self._h:
            lib().nchmm_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __len__(self):
        return lib().nchmm_pool_size(self._h)

    def train_reads(self, opts, model_states, strand_off, mean, stdv, start, job_read, job_m0, job_m1, init_pm=None, init_st=None):
        st10 = _f32(model_states).reshape(-1, S, 10)
        so = np.ascontiguousarray(strand_off, np.uint64)
        n_reads = (so.shape[0] - 1) // 2
        jr, j0, j1 = (np.ascontiguousarray(a, np.int32) for a in (job_read, job_m0, job_m1))
        nj = jr.shape[0]
        pm = np.tile(np.float32([1, 0, 0, 1, 1, 1]), (nj, 1)) if init_pm is None else _f32(init_pm).reshape(nj, 6).copy()
        st = (np.tile(np.float32([opts.default_p_stay, opts.default_p_skip] * 2), (nj, 1)) if init_st is None
              else _f32(init_st).reshape(nj, 4).copy())
        fit = np.empty(nj, np.float32)
        rounds = np.empty(nj, np.uint32)
        pref = np.empty((n_reads, 3), np.int32)
        check(lib().nchmm_pool_train_reads(self._h, C.byref(opts), st10.shape[0], _p(st10), n_reads, _p(so), _p(_f32(mean)),
                                           _p(_f32(stdv)), _p(_f32(start)), nj, _p(jr), _p(j0), _p(j1), _p(pm), _p(st), _p(fit),
                                           _p(rounds), _p(pref)), "nchmm_pool_train_reads")
        return dict(pm=pm, st=st, fit=fit, rounds=rounds, preferred=pref)

    def basecall_reads(self, opts, model_states, strand_off, mean, stdv, start, job_read, job_m0, job_m1, job_pm, job_st,
                       preferred=None):
        st10 = _f32(model_states).reshape(-1, S, 10)
        so = np.ascontiguousarray(strand_off, np.uint64)
        n_reads = (so.shape[0] - 1) // 2
        jr, j0, j1 = (np.ascontiguousarray(a, np.int32) for a in (job_read, job_m0, job_m1))
        pm, st = _f32(job_pm).reshape(-1, 6), _f32(job_st).reshape(-1, 4)
        pref = None if preferred is None else np.ascontiguousarray(preferred, np.int32).reshape(n_reads, 3)
        states = np.zeros(int(so[-1]), np.uint16)
        bj = np.empty((n_reads, 2), np.int32)
        bl = np.empty((n_reads, 2), np.float32)
        rc = lib().nchmm_pool_basecall_reads(self._h, C.byref(opts), st10.shape[0], _p(st10), n_reads, _p(so), _p(_f32(mean)),
                                             _p(_f32(stdv)), _p(_f32(start)), jr.shape[0], _p(jr), _p(j0), _p(j1), _p(pm), _p(st),
                                             _p(pref), _p(states), _p(bj), _p(bl))
        if rc not in (0, -6):
            check(rc, "nchmm_pool_basecall_reads")
        return dict(states=states, best_job=bj, best_logp=bl)

    def counters(self):
        """-> (summed counters uint64[8], used_rccl bool)"""
        out = np.zeros(8, np.uint64)
        used = C.c_int(0)
        check(lib().nchmm_pool_counters(self._h, _p(out), C.byref(used)), "nchmm_pool_counters")
        return out, bool(used.value)
