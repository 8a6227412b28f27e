/*
 * nanocall_hip.h -- C ABI of the MI355X-native HMM basecalling core for nanocall.
 *
 * The reference (mateidavid/nanocall) has no FFI seam: its DP layer is header-only templates
 * instantiated inside nanocall.cpp.  The seam this library replaces is therefore defined by the
 * reference's call sites; every entry point below names the reference interface it stands in for
 * (paths relative to the reference root).  INTEGRATION.md shows the C++ binding a nanocall
 * maintainer would add (the headers under include/nanocall_amd/ are drop-in mirrors of the reference classes
 * that forward to these functions).
 *
 * Conventions
 *   - plain C types only; all functions return 0 on success or a negative NCHMM_E_* code;
 *     nothing throws, nothing calls exit().  nchmm_strerror() describes a code.
 *   - "host" functions are pure CPU prep the reference also does on the host (they use libm so
 *     that logs are bit-identical to the reference's); they never touch the GPU.
 *   - "device" functions need a context; a context is bound to one GPU and is not thread-safe
 *     (use one context per host thread, as the reference uses one DP object per pfor worker,
 *     src/nanocall/nanocall.cpp:611-621).
 *   - S = 4096 states (6-mers, 2 bits per base, first base in the top bits, Kmer.hpp:41-50).
 *   - There is NO CPU fallback: device entry points fail with NCHMM_E_NO_DEVICE / a HIP error when
 *     no gfx950 device is usable.
 */
#ifndef NANOCALL_HIP_H
#define NANOCALL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NCHMM_N_STATES 4096
#define NCHMM_KMER_SIZE 6
#define NCHMM_MAX_ARCS (4096 * 21)

enum {
    NCHMM_OK = 0,
    NCHMM_E_INVALID = -1,      /* bad argument (null pointer, slot out of range, ragged offsets) */
    NCHMM_E_NO_DEVICE = -2,    /* no usable HIP device / device id out of range */
    NCHMM_E_HIP = -3,          /* a HIP runtime call failed; see nchmm_last_hip_error() */
    NCHMM_E_TOPOLOGY = -4,     /* transitions are not the stay/step/skip-1 6-mer graph that
                                  State_Transitions::compute_transitions_fast produces */
    NCHMM_E_NOMEM = -5,        /* host or device allocation failed */
    NCHMM_E_NUMERIC = -6       /* a read decoded to -INF/NaN everywhere (the reference reads out of
                                  bounds there, Viterbi.hpp:125-141); per-read status says which */
};

typedef struct nchmm_ctx nchmm_ctx;

const char* nchmm_strerror(int code);
int nchmm_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Host prep (no GPU).  Bit-exact with the reference's host code.
 * ---------------------------------------------------------------------------------------- */

/* Builtin_Model (src/nanocall/Builtin_Model.hpp:7-13, Builtin_Model.cpp:3-17): num, names[i], strands[i] and
 * init_lists[i] = S x 4 {level_mean, level_stdv, sd_mean, sd_stdv} in k-mer order.  The tables are data carried inside
 * the library (nanocall_amd/data/builtin_models.f32). */
int nchmm_builtin_count(void);
const char* nchmm_builtin_name(int i);
int nchmm_builtin_strand(int i);
const float* nchmm_builtin_table(int i);

/* Pore_Model::load_from_vector, src/nanocall/Pore_Model.hpp:220-239.
 * table: S x 4 {level_mean, level_stdv, sd_mean, sd_stdv}.
 * state: S x 10 {level_mean, level_stdv, sd_mean, sd_stdv, sd_lambda, log_level_mean,
 *                log_level_stdv, log_sd_mean, log_sd_stdv(=0, unset in the reference), log_sd_lambda}
 * i.e. the field order of Pore_Model_State (Pore_Model.hpp:85-96). */
int nchmm_model_load(const float* table_Sx4, float* state_Sx10);

/* Pore_Model::scale, src/nanocall/Pore_Model.hpp:190-201 / Pore_Model_State::scale :126-138.
 * params: {scale, shift, drift, var, scale_sd, var_sd} (Pore_Model_Parameters, :46-51). In place. */
int nchmm_model_scale(float* state_Sx10, const float params[6]);

/* Pack the six fields the emission reads (Pore_Model.hpp:145-149) for nchmm_put_model:
 * S x 6 {level_mean, level_stdv, log_level_stdv, sd_mean, sd_lambda, log_sd_lambda}. */
int nchmm_model_pack6(const float* state_Sx10, float* table_Sx6);

/* State_Transitions::compute_transitions_fast(p_skip, p_stay), State_Transitions.hpp:181-224, then
 * update_fields :79-104.  Output is the from_v view as CSR by destination state: row_ptr[S+1],
 * pred[n_arcs] ascending within a row, logw[n_arcs].  Buffers must hold NCHMM_MAX_ARCS entries. */
int nchmm_transitions_fast(float p_skip, float p_stay, uint32_t* row_ptr, uint16_t* pred, float* logw,
                           uint32_t* n_arcs);

/* Event::update_logs (Event.hpp:35-45: stdv == 0 -> 0.01, log_stdv = log(stdv)) followed by
 * Event_Sequence::apply_drift_correction (Event.hpp:77-84: corrected_mean = mean - drift * start).
 * stdv is updated in place. */
int nchmm_events_prepare(size_t n, const float* mean, float* stdv, const float* start, float drift,
                         float* corrected_mean, float* log_stdv);

/* Viterbi::fill_move_seq (Viterbi.hpp:144-150, Kmer::min_skip Kmer.hpp:51-68) and
 * Event_Sequence::get_base_seq (Event.hpp:85-99).  seq must hold 6*n + 1 bytes; *seq_len excludes
 * the terminating NUL.  move may be NULL. */
int nchmm_base_seq(size_t n, const uint16_t* state, int32_t* move, char* seq, size_t* seq_len);

/* write_fasta, src/nanocall/nanocall.cpp:584-591.  Returns bytes written via *written. */
int nchmm_write_fasta(const char* name, const char* seq, unsigned line_width, char* out, size_t cap,
                      size_t* written);

/* Parameter_Trainer::init / st_train_kmers, Parameter_Trainer.hpp:30-63.  out holds <= 4096. */
int nchmm_st_train_kmers(uint16_t* out, uint32_t* count);

/* Parameter_Trainer::train_pm_params, Parameter_Trainer.hpp:297-427, from the per-event inner sums
 * {s0,s1,s2,l0,l1,l2} that nchmm_fwbw produces (concatenated over the training windows of a round, in
 * window order) and the UNCORRECTED events of those windows.  new_pm = {scale, shift, drift, var,
 * scale_sd, var_sd}; *done = 1 when the normal matrix is singular (new_pm = crt_pm, :355-360). */
int nchmm_train_pm_finish(size_t n_events, const float* pm_sums_nx6, const float* mean, const float* stdv,
                          const float* start, int train_drift, const float crt_pm[6], float new_pm[6], int* done);

/* Parameter_Trainer::train_st_params, Parameter_Trainer.hpp:516-530, for one strand: combines the
 * {denom, stay_num, skip_num} log-sums of that strand's windows and applies the [.05, .4] reset. */
int nchmm_train_st_finish(size_t n_win, const float* st_sums_nx3, float* p_stay, float* p_skip);

/* The solve half of nchmm_train_pm_finish (Parameter_Trainer.hpp:314-427) from the thirteen outer sums
 * {A00, A01, A11, B0, B1, A02, A12, A22, B2, D, V_numer, V_denom, U_pos} of :297-312, which nchmm_em_round
 * accumulates on the device. */
int nchmm_train_pm_solve(size_t n_events, const double acc[13], int train_drift, const float crt_pm[6],
                         float new_pm[6], int* done);

/* ------------------------------------------------------------------------------------------
 * Read summary: strand segmentation, event filter, initial scaling (host).
 * Replaces the arithmetic of Fast5_Summary (src/nanocall/Fast5_Summary.hpp) on the EventDetection table of a
 * read, wherever that table came from (FAST5 through nanocall_fast5.h, or the events text form).
 * fast5::EventDetection_Event_Entry is declared in the un-vendored fast5 submodule; the layout here (two
 * doubles, two 64-bit integers: level mean, level stdv, start and length in samples) is what the FAST5
 * EventDetection dataset stores.
 * ---------------------------------------------------------------------------------------- */
typedef struct nchmm_ed_event {
    double mean, stdv;
    int64_t start, length;
} nchmm_ed_event;

typedef struct nchmm_segment_opts {
    uint32_t min_ed_events;             /* --min-ed-events 10        Fast5_Summary.hpp:74-78 */
    uint32_t max_ed_events;             /* --max-ed-events 100000    :80-84 */
    double abasic_level_top_percent;    /* :93-97, set per pore at nanocall.cpp:943-964 */
    double abasic_level_top_offset;     /* :100-104 */
    uint32_t template_only;             /* --1d                      :121-125 */
    uint32_t trim_margins[4];           /* --trim-ed-{sq-start,sq-end,hp-start,hp-end} 50   :128-132 */
} nchmm_segment_opts;

typedef struct nchmm_read_summary {
    uint32_t num_ed_events;             /* 0 = the read is skipped (Fast5_Summary.hpp:185-209) */
    float abasic_level;
    uint32_t strand_bounds[4];          /* [template begin, end, complement begin, end) in the EventDetection table */
    int32_t scale_strands_together;
    float time_length[2];
} nchmm_read_summary;

/* pore = "r9" or "r73": the presets of nanocall.cpp:943-964; anything else is NCHMM_E_INVALID */
int nchmm_segment_opts_default(nchmm_segment_opts* opts, const char* pore);

/* alg::mean_stdv_of<float> as the callers use it (Fast5_Summary.hpp:225-230,256-258, Pore_Model.hpp:307-313,
 * nanocall.cpp:633-635).  hpptools is un-vendored: float accumulation of sum and sum of squares, sample (n-1) stdv. */
int nchmm_mean_stdv(size_t n, const float* v, float* mean, float* stdv);

/* Fast5_Summary::summarize, Fast5_Summary.hpp:160-219: event cap, abasic level (:528-543), strand detection
 * (:545-571,653-731), scale_strands_together (:210-212), time lengths.  double_strand_scaling = the constructor's
 * `sst` argument (opts::double_strand_scaling, nanocall.cpp:269). */
int nchmm_read_summarize(const nchmm_segment_opts* opts, size_t n_ed, const nchmm_ed_event* ed, float sampling_rate,
                         int double_strand_scaling, nchmm_read_summary* out);

/* Fast5_Summary::load_events, :348-365: the filtered (:734-745) events of strand st as Event fields (stdv after
 * Event::update_logs' 0 -> .01; start relative to the strand's -- or, scaled together, the read's -- first event,
 * in seconds).  The arrays must hold strand_bounds[2st+1] - strand_bounds[2st] entries; *n receives the count. */
int nchmm_read_load_events(const nchmm_read_summary* s, const nchmm_ed_event* ed, float sampling_rate, int st, float* mean,
                           float* stdv, float* start, float* length, size_t* n);

/* Initial model scaling, :223-278.  r0 / r1 = {mean, stdv} of the strand's event means, m0 / m1 = {mean, stdv} of the
 * models' level means.  together != 0: the 2D form (:237-241); else the single-strand form on r0 / m0 (:265-267). */
int nchmm_initial_scaling(int together, const float r0[2], const float r1[2], const float m0[2], const float m1[2],
                          float* scale, float* shift);

/* ------------------------------------------------------------------------------------------
 * Device context
 * ---------------------------------------------------------------------------------------- */

/* device_id: HIP device ordinal.  Creates its own non-blocking stream. */
int nchmm_create(nchmm_ctx** out, int device_id);
int nchmm_destroy(nchmm_ctx* ctx);
int nchmm_last_hip_error(const nchmm_ctx* ctx); /* raw hipError_t of the last failure */
/* Run later launches on a caller-owned hipStream_t, taken as is: the handle 0 IS the legacy default
 * (null) stream -- which is what torch.cuda.current_stream().cuda_stream returns unless the caller made
 * its own stream.  Ordering contract of the *_dev entry points: they enqueue on the context's current
 * stream and return; their inputs must be complete on that stream (or synchronised) before the call and
 * their outputs are valid once that stream reaches the end of the enqueued work.  The context's own stream
 * is hipStreamNonBlocking, i.e. NOT ordered against the null stream: either hand the library the stream your
 * producers / consumers use, or call nchmm_synchronize().  nchmm_use_own_stream() goes back to the private one. */
int nchmm_set_stream(nchmm_ctx* ctx, void* hip_stream);
int nchmm_use_own_stream(nchmm_ctx* ctx);
int nchmm_synchronize(nchmm_ctx* ctx);

/* Register a scaled pore model in `slot` (0 <= slot < reserved slots, 64 by default): what basecall_strand builds with
 * `Pore_Model_Type pm(models.at(m_name)); pm.scale(pm_params);` (nanocall.cpp:649-650).
 * table_Sx6 as produced by nchmm_model_pack6. */
int nchmm_put_model(nchmm_ctx* ctx, int slot, const float* table_Sx6);

/* Register transitions in `slot` (0 <= slot < reserved slots): the `*transitions_ptr` of basecall_strand
 * (nanocall.cpp:651-661).  CSR by destination state, predecessors ascending (from_v order,
 * State_Transitions.hpp:85-94).  Fails with NCHMM_E_TOPOLOGY unless the graph is exactly the
 * stay/step/skip-1 graph of compute_transitions_fast and its weights factor per DESIGN.md
 * section "Transition factorisation" (always true for compute_transitions_fast output). */
int nchmm_put_transitions(nchmm_ctx* ctx, int slot, const uint32_t* row_ptr_S1, const uint16_t* pred,
                          const float* logw);

/* Slots 0..63 exist from the start; this grows the model and transition slot tables (contents kept). */
int nchmm_reserve_slots(nchmm_ctx* ctx, int n_slots);

/* Batched `Pore_Model pm(models.at(name)); pm.scale(pm_params);` (nanocall.cpp:649-650,
 * Parameter_Trainer.hpp:105-114): slot first_slot + k receives table table_idx[k] (an S x 10 state array
 * from nchmm_model_load, tables concatenated in states_Sx10) scaled by params_nx6[k].  Multi-threaded on
 * the host, one upload.  Grows the slot tables as needed. */
int nchmm_put_models_scaled(nchmm_ctx* ctx, int first_slot, size_t n, const float* states_Sx10, const int32_t* table_idx,
                            const float* params_nx6);

/* Batched `custom_transitions.compute_transitions_fast(st_params)` (nanocall.cpp:653-657,
 * Parameter_Trainer.hpp:123-127): slot first_slot + k receives the transitions of (p_skip[k], p_stay[k]). */
int nchmm_put_transitions_fast(nchmm_ctx* ctx, int first_slot, size_t n, const float* p_skip, const float* p_stay);

/* ------------------------------------------------------------------------------------------
 * Viterbi  -- replaces `Viterbi_Type vit; vit.fill(pm, *transitions_ptr, corrected_events);
 *             vit.path_probability()`  (src/nanocall/nanocall.cpp:687-689, Viterbi.hpp:44-99,
 *             fill_state_seq :120-142), batched over reads.
 *
 * Events are SoA: corrected_mean, stdv, log_stdv (the three fields log_pr_corrected_emission
 * reads, Pore_Model.hpp:145-149).  Read r owns events [off[r], off[r+1]).  model_slot/trans_slot
 * give the per-read slots (NULL = slot 0 for every read).
 * Outputs: out_state[e] = Event::model_state_idx of event e (Viterbi.hpp:136); out_path_logp[r] =
 * path_probability(); out_status[r] = 0 or NCHMM_E_NUMERIC (may be NULL).
 * Reads with zero events get out_path_logp = NaN and status 0.
 * ---------------------------------------------------------------------------------------- */

/* host-pointer form: copies in, runs, copies out, synchronises (pipelined over read ranges, see nchmm_viterbi_begin). */
int nchmm_viterbi(nchmm_ctx* ctx, size_t n_reads, const uint64_t* off, const float* corrected_mean,
                  const float* stdv, const float* log_stdv, const int32_t* model_slot,
                  const int32_t* trans_slot, uint16_t* out_state, float* out_path_logp,
                  int32_t* out_status);

/* device-pointer form: every pointer is device memory on the context's GPU; enqueues on the
 * context's stream and returns without synchronising.  max_events = max_r (off[r+1]-off[r]) and
 * total_events = off[n_reads] must be supplied by the caller (max_events sizes the back-pointer workspace: one
 * region of 4 KiB x max_events per resident thread block, whatever n_reads is; the offsets themselves stay on
 * the device).  order (may be NULL) is a permutation of reads giving the processing order (longest first
 * balances the work queue). */
int nchmm_viterbi_dev(nchmm_ctx* ctx, size_t n_reads, size_t max_events, size_t total_events,
                      const uint64_t* d_off, const float* d_corrected_mean, const float* d_stdv,
                      const float* d_log_stdv, const int32_t* d_model_slot, const int32_t* d_trans_slot,
                      const uint32_t* d_order, uint16_t* d_out_state, float* d_out_path_logp,
                      int32_t* d_out_status);

/* nchmm_viterbi_dev in two halves, for callers that keep several batches going (the reference keeps its cores busy with
 * one strand per pfor thread, nanocall.cpp:611-621; a GPU is kept busy by letting the thread blocks of the next batch
 * start where those of the previous one run out of reads).
 *   enqueue  queues the batch behind whatever is on the context's stream NOW, on one of the context's internal
 *            streams (three, taken in turn), and returns.  The batch may run beside the batch enqueued before it.
 *   join     makes the context's stream wait for every batch enqueued so far: what is queued on that stream
 *            afterwards sees their outputs.  (nchmm_synchronize joins, waits, and reports.)
 * Between enqueue and join the caller leaves the batch's inputs and outputs alone, and must not hand the same output
 * arrays to a second batch.  nchmm_viterbi_dev is enqueue followed by join.
 * The lengths are only on the device, so the plan the host-pointer forms make on the host is made there (plan_kernel.hip): a
 * batch whose stated longest read is well above its mean is handed out longest first (a kernel in front of the sweep, nothing
 * waited for); and in ONE case the call does wait -- for the lanes and for one 32-byte read-back: when max_events is so long
 * that a full pool of back-pointer regions of that length does not fit NCHMM_WS_BUDGET_MB, the few reads that long are found on
 * the device and get regions of their own beside the pooled launch (else the whole batch would run on as many blocks as the
 * budget has regions of that length for).  d_order != NULL switches both off: the caller's order is taken as it is. */
int nchmm_viterbi_dev_enqueue(nchmm_ctx* ctx, size_t n_reads, size_t max_events, size_t total_events,
                              const uint64_t* d_off, const float* d_corrected_mean, const float* d_stdv,
                              const float* d_log_stdv, const int32_t* d_model_slot, const int32_t* d_trans_slot,
                              const uint32_t* d_order, uint16_t* d_out_state, float* d_out_path_logp,
                              int32_t* d_out_status);
int nchmm_viterbi_dev_join(nchmm_ctx* ctx);

/* ONE strand per call, from MANY host threads at once -- the reference's own call shape: basecall_strand builds the scaled model
 * and the transitions of a strand and runs `vit.fill(pm, *transitions_ptr, corrected_events)` (nanocall.cpp:645-690), inside a
 * pfor whose worker threads each hold one strand (:611-621).  Thread-safe on one context: calls that are in progress at the
 * same time are combined into batched launches (two batches alternate: one collects callers while the other is on the device),
 * each caller staging its own strand and taking its own results; a call returns when its strand is decoded.  With T calling
 * threads about T/2 strands go into every launch, so the GPU fills up at T of the order of a thousand (worker threads that
 * sleep in this call cost nothing); one caller alone gets a launch to itself.
 *   table_Sx6   the scaled model as nchmm_put_model takes it;  (p_skip, p_stay): transitions = compute_transitions_fast of them
 *   returns 0, NCHMM_E_NUMERIC for this strand (every state -INF/NaN in the last column), or the error of its batch.
 * While threads are inside this call the context must not be used through any other entry point -- and every combined batch
 * OVERWRITES the context's model and transition slots 0 .. n-1 (strand k of a batch decodes with slot k; the slot tables may
 * grow and move): tables the caller uploaded with nchmm_put_model / nchmm_put_transitions* before are not there afterwards.
 * The same holds for nchmm_viterbi_strand_scaled and nchmm_fwbw_windows.  Staging (pinned host memory: 128 KiB + 14 B per
 * event a strand) follows demand: it starts at 1/64 of a full batch and doubles when batches fill up.
 * nchmm_model_image / nchmm_put_model_images are its building blocks (the device image of a model built on the caller's
 * thread; many images uploaded into consecutive slots with one copy). */
int nchmm_viterbi_strand(nchmm_ctx* ctx, const float* table_Sx6, float p_skip, float p_stay, size_t n_events,
                         const float* corrected_mean, const float* stdv, const float* log_stdv, uint16_t* out_state,
                         float* out_path_logp);
/* the same with the strand's model given as where it came from: an UNSCALED table (nchmm_model_load layout, valid during the call)
 * and the Pore_Model_Parameters it is scaled by -- `pm = models.at(name); pm.scale(pm_params)` in basecall_strand (nanocall.cpp:653-657)
 * -- so that 32 bytes travel instead of a 128 KiB image and Pore_Model::scale runs on the device (bit for bit) */
int nchmm_viterbi_strand_scaled(nchmm_ctx* ctx, const float* unscaled_Sx10, const float* pm_params, float p_skip, float p_stay,
                                size_t n_events, const float* corrected_mean, const float* stdv, const float* log_stdv,
                                uint16_t* out_state, float* out_path_logp);
int nchmm_model_image(const float* table_Sx6, float* image_8xS, int32_t* fast);
int nchmm_put_model_images(nchmm_ctx* ctx, int first_slot, size_t n, const float* images_nx8xS, const int32_t* fast_n);

/* Viterbi from RAW events, with the host prep of basecall_strand on the device:
 * candidate v decodes raw events [src[v], src[v] + len[v]) of the uploaded (mean, stdv, start) arrays -- several
 * candidates may share a range (one strand, several models / parameter sets, nanocall.cpp:715-732,809-818) -- after
 *   Event::update_logs            stdv == 0 -> 0.01, log_stdv = log(stdv)                (Event.hpp:39-43)
 *   apply_drift_correction        corrected_mean = mean - drift[v] * start               (Event.hpp:77-84, nanocall.cpp:685-686)
 * done by a gather kernel; log() there is a port of glibc 2.35 logf that is bit-identical to the host libm over all
 * binary32 inputs (nchmm_logf; tests/test_logf_gpu.py), so the decoded path is the same as with host prep.
 * The raw arrays go up once (12 B per event however many candidates).  out_state is packed by candidate:
 * candidate v's states start at sum_{u < v} len[u].  model_slot / trans_slot / out_status as nchmm_viterbi. */
int nchmm_viterbi_raw(nchmm_ctx* ctx, size_t n_raw_events, const float* mean, const float* stdv, const float* start,
                      size_t n_cand, const uint64_t* src, const uint32_t* len, const float* drift, const int32_t* model_slot,
                      const int32_t* trans_slot, uint16_t* out_state, float* out_path_logp, int32_t* out_status);

/* The two host-pointer forms above, split for callers that stream batches (the reference hides the latency of a strand
 * behind its pfor worker threads, nanocall.cpp:611-621; here the unit is a batch and the latencies are the PCIe copies and
 * the tail of a launch, when its blocks run out of reads one by one).  A batch is cut into read ranges: range k+1 is
 * copied in while range k computes, on a copy-in stream and three compute lanes taken in turn, results written
 * by the kernels straight into pinned host memory -- the one-call forms are begin followed by end.
 *   begin  validates, stages and enqueues copy-in + kernels of a batch, and returns (it holds the thread for the
 *          duration of the H2D copies only).  At most THREE batches may be in flight per context (one per compute lane; NCHMM_E_INVALID beyond).
 *   end    completes the OLDEST batch in flight: copies its states out range by range as they finish, waits, fills
 *          out_path_logp / out_status, returns 0 or NCHMM_E_NUMERIC as the one-call form does.
 * Every array passed to begin -- inputs and outputs -- must stay valid and untouched until the matching end has
 * returned, and the model / transition slots the batch names must not be rewritten before that.  While a batch is in
 * flight nchmm_viterbi, nchmm_viterbi_raw and nchmm_viterbi_dev return NCHMM_E_INVALID.
 * begin(0); begin(1); end(0); begin(2); end(1); ... keeps the GPU busy: batch k+1 goes up and starts while batch k
 * computes, batch k comes down under the kernels of batch k+1.  With reads of very different lengths keep three going
 * (begin(0); begin(1); begin(2); end(0); begin(3); ...): a launch lasts as long as its longest read, and the long reads of
 * two batches then finish behind the bulk of the third. */
int nchmm_viterbi_begin(nchmm_ctx* ctx, size_t n_reads, const uint64_t* off, const float* corrected_mean,
                        const float* stdv, const float* log_stdv, const int32_t* model_slot,
                        const int32_t* trans_slot, uint16_t* out_state, float* out_path_logp, int32_t* out_status);
int nchmm_viterbi_raw_begin(nchmm_ctx* ctx, size_t n_raw_events, const float* mean, const float* stdv, const float* start,
                            size_t n_cand, const uint64_t* src, const uint32_t* len, const float* drift,
                            const int32_t* model_slot, const int32_t* trans_slot, uint16_t* out_state,
                            float* out_path_logp, int32_t* out_status);
int nchmm_viterbi_end(nchmm_ctx* ctx);
/* batches begun and not yet ended (0 .. 3) */
int nchmm_viterbi_in_flight(const nchmm_ctx* ctx);

/* logf on the device, bit-identical to glibc 2.35 logf as x86-64 CPUs with FMA run it: out[i] = log(in[i]), host
 * buffers, any n (chunked).  Exists so that the claim above can be checked exhaustively. */
int nchmm_logf(nchmm_ctx* ctx, size_t n, const float* in, float* out);

/* ------------------------------------------------------------------------------------------
 * Forward-backward + EM sufficient statistics -- replaces Forward_Backward::fill
 * (Forward_Backward.hpp:46-135) as called from Parameter_Trainer::fill_train_data
 * (Parameter_Trainer.hpp:141-155), fused with the per-event inner sums of train_pm_params
 * (:273-296) and the log-sums of train_st_params (:451-515).
 *
 * Windows are SoA like Viterbi reads.  For window w:
 *   out_log_pr_data[w]            = Forward_Backward::log_pr_data()
 *   out_pm_sums[e*6 .. e*6+5]     = {s0,s1,s2,l0,l1,l2} of event e (Parameter_Trainer.hpp:273-296),
 *                                   taken over the UNSCALED model: the one that Pore_Model::scale
 *                                   (Pore_Model.hpp:126-138,190-201) with pm_params[w] = {scale, shift,
 *                                   drift, var, scale_sd, var_sd} turned into scaled_slot[w].  The
 *                                   kernel holds only the scaled states and undoes the (affine) scaling
 *                                   on the six block sums; pm_params NULL = identity (sums over the
 *                                   scaled model itself)
 *   out_st_sums[w*3 .. w*3+2]     = {denom, stay_num, skip_num} log-sums of train_st_params over the
 *                                   events of this window (log_p_stay / log_p_step_4 from st_params)
 *   out_alpha / out_beta          = optional full matrices (n x S, log space), NULL to skip
 * The host finishes the round (3x3 solve etc.) -- see nanocall_amd/Parameter_Trainer.hpp.
 * ---------------------------------------------------------------------------------------- */
int nchmm_fwbw(nchmm_ctx* ctx, size_t n_win, const uint64_t* off, const float* corrected_mean,
               const float* stdv, const float* log_stdv, const int32_t* scaled_slot,
               const float* pm_params /* n_win x 6 or NULL */, const int32_t* trans_slot,
               const float* st_params /* n_win x 2 {p_stay, p_skip} or NULL */,
               float* out_log_pr_data, float* out_pm_sums, float* out_st_sums,
               float* out_alpha, float* out_beta);

/* One read's training windows per call, from MANY host threads at once -- the reference's call shape again:
 * Parameter_Trainer::train_one_round (Parameter_Trainer.hpp:541-579) scales the one or two models of a read by the current
 * parameters, runs forward-backward over its 2-4 windows and sums, inside the pfor of train_reads (nanocall.cpp:282-579).
 * Thread-safe on one context: calls in progress at the same time are combined into batched nchmm_fwbw launches exactly as
 * nchmm_viterbi_strand combines strands (each caller stages its own windows and takes its own sums).
 *   unscaled_Sx10[m]   the UNSCALED model of the call's m-th strand (nchmm_model_load layout; stays valid during the call),
 *   pm_params          the Pore_Model_Parameters {scale, shift, drift, var, scale_sd, var_sd} every model of the call is scaled
 *                      by (Pore_Model::scale, on the device: a model travels as 32 bytes, not as a 128 KiB image)
 *   p_skip[m], p_stay[m]                   its transitions: compute_transitions_fast(p_skip[m], p_stay[m]),  m < n_models
 *   off / corrected_mean / stdv / log_stdv the windows, SoA like nchmm_fwbw; win_model[w] < n_models
 *   st_params (n_win x 2 {p_stay, p_skip}, NULL = those of the window's model); outputs as nchmm_fwbw (sums may be NULL).
 * Calls of nchmm_viterbi_strand and nchmm_fwbw_windows may be in progress on one context at the same time (their batches take turns on
 * the device); while threads are inside either, the context must not be used through any OTHER entry point. */
int nchmm_fwbw_windows(nchmm_ctx* ctx, size_t n_models, const float* const* unscaled_Sx10, const float* pm_params,
                       const float* p_skip, const float* p_stay, size_t n_win, const uint64_t* off, const float* corrected_mean,
                       const float* stdv, const float* log_stdv, const int32_t* win_model, const float* st_params,
                       float* out_log_pr_data, float* out_pm_sums, float* out_st_sums);

int nchmm_fwbw_dev(nchmm_ctx* ctx, size_t n_win, size_t max_events, size_t total_events,
                   const uint64_t* d_off, const float* d_corrected_mean, const float* d_stdv,
                   const float* d_log_stdv, const int32_t* d_scaled_slot, const float* d_pm_params,
                   const int32_t* d_trans_slot, const float* d_st_params,
                   float* d_out_log_pr_data, float* d_out_pm_sums, float* d_out_st_sums,
                   float* d_out_alpha, float* d_out_beta);

/* ------------------------------------------------------------------------------------------
 * One EM round with the events resident on the device -- Parameter_Trainer::fill_train_data
 * (Parameter_Trainer.hpp:99-155) + the event loops of train_pm_params (:273-312), for many jobs at once.
 * nchmm_em_load_events uploads the raw events of all reads once (stdv after Event::update_logs' 0 -> .01,
 * log_stdv from the host libm; only the entries training windows cover are ever read).  nchmm_em_round then
 * takes window w = raw events [win_src[w], win_src[w] + win_len[w]) with its drift, scaling parameters
 * (win_pm, 6 floats: what produced scaled_slot[w]), model / transition slots and {p_stay, p_skip}; job k owns
 * windows [job_first_win[k], job_first_win[k+1]).  On the device: drift correction + packing, forward-backward
 * + inner sums (as nchmm_fwbw_dev), then the outer sums of :297-312 per job.  Back come log Pr(data) per
 * window, {denom, stay_num, skip_num} per window (nchmm_train_st_finish) and thirteen doubles per job
 * (nchmm_train_pm_solve).
 * ---------------------------------------------------------------------------------------- */
int nchmm_em_load_events(nchmm_ctx* ctx, size_t n_events, const float* mean, const float* stdv, const float* start,
                         const float* log_stdv);
int nchmm_em_round(nchmm_ctx* ctx, size_t n_win, const uint64_t* win_src, const uint32_t* win_len, const float* win_drift,
                   const float* win_pm, const int32_t* scaled_slot, const int32_t* trans_slot, const float* st_params,
                   size_t n_jobs, const uint32_t* job_first_win, int train_drift, float* out_log_pr_data,
                   float* out_st_sums, double* out_job_acc);

/* ------------------------------------------------------------------------------------------
 * EM driver loop -- replaces the body of train_reads (src/nanocall/nanocall.cpp:292-574): window
 * extraction :327-338, the round loop with its stop / roll-back rules :367-426 (2D) and :483-542 (1D),
 * and threshold model selection :437-459 / :552-570 -- batched: every round, all jobs still training go
 * through one forward-backward launch, or (64 jobs and more) through one per part of the jobs, the parts
 * taking turns on two lanes of the context so that one part's host steps run behind another's kernels;
 * a job's rounds and results do not depend on the arrangement (NCHMM_EM_LANES=1: always one part).
 *
 * A job is one iteration of the reference's model loops: (read, m0, m1) with both >= 0 when the read's
 * strands are scaled together, or (read, m, -1) / (read, -1, m) for a single strand.  Reads are given as
 * 2 strands each: strand st of read r owns events [strand_off[2r+st], strand_off[2r+st+1]) of mean / stdv /
 * start (uncorrected; stdv after Event::update_logs).  model_states_Sx10 = n_models tables from
 * nchmm_model_load (unscaled).  job_pm (n_jobs x 6) and job_st (n_jobs x 4 = {p_stay, p_skip} per strand)
 * hold the initial parameters on entry (Fast5_Summary's pm_params_m / st_params_m) and the trained ones on
 * return; job_fit / job_rounds receive the final fit and round count.  read_preferred (may be NULL),
 * n_reads x 3: the job selected for strand 0, strand 1 and the 2D pair (preferred_model), or -1.
 * Uses model slots [0, n_models + 2*jobs) and transition slots [0, 1 + 2*jobs) of the context at most (in parts:
 * four per job of the largest part); its second lane computes on the stream of Viterbi lane 1 (a batch queued with
 * nchmm_viterbi_dev_enqueue before the call runs first; the call itself returns with nothing of its own in flight). */
typedef struct nchmm_train_opts {
    uint32_t scaling_num_events;       /* --scaling-num-events      200  nanocall.cpp:72 */
    uint32_t scaling_max_rounds;       /* --scaling-max-rounds       10  :71 (2D jobs run up to twice this, :420) */
    float scaling_min_progress;        /* --scaling-min-progress    1.0  :70 */
    float scaling_select_threshold;    /* --scaling-select-threshold 20  :69 (INFINITY: no selection) */
    uint32_t min_ed_events;            /* --min-ed-events            10  :66 */
    int32_t train_scaling;             /* !--no-train-scaling */
    int32_t train_transitions;         /* !--no-train-transitions */
    int32_t train_drift;               /* Parameter_Trainer::pm_train_drift() (r73: 1, r9: 0, :943-970) */
    float default_p_stay, default_p_skip;   /* --pr-stay .1 / --pr-skip .3: what is_default() compares with */
} nchmm_train_opts;

int nchmm_train_opts_default(nchmm_train_opts* opts);

/* The job list the reference's loops would visit (models in the order given; pass them sorted by name to
 * mirror std::map).  model_strand[a] in {0, 1, 2}.  On entry *n_jobs = capacity of the arrays (which may
 * be NULL to only count), on return the number of jobs. */
int nchmm_train_enumerate(const nchmm_train_opts* opts, size_t n_models, const int32_t* model_strand, size_t n_reads,
                          const uint64_t* strand_off, const uint8_t* scale_strands_together, size_t* n_jobs,
                          int32_t* job_read, int32_t* job_m0, int32_t* job_m1);

int nchmm_train_reads(nchmm_ctx* ctx, const nchmm_train_opts* opts, size_t n_models, const float* model_states_Sx10,
                      size_t n_reads, const uint64_t* strand_off, const float* mean, const float* stdv, const float* start,
                      size_t n_jobs, const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1, float* job_pm,
                      float* job_st, float* job_fit, uint32_t* job_rounds, int32_t* read_preferred);

/* basecall_reads (src/nanocall/nanocall.cpp:593-868) for the same reads / jobs: every candidate job (the
 * preferred one where read_preferred names one, else all jobs of the read) is Viterbi-decoded with its
 * trained parameters -- `pm.scale(pm_params)`, custom transitions unless default, drift-corrected events
 * (:645-690) -- in one batched launch; 2D jobs are ranked by the float sum of their two path
 * log-probabilities (:725-739), single-strand jobs by their own (:807-825), highest wins (the later
 * candidate on an exact tie, as `sort ... back()`).  out_state is indexed like mean/stdv/start and
 * receives the winner's Event::model_state_idx for every decoded strand; out_best_job / out_best_logp are
 * n_reads x 2 (per strand; -1 / NaN where nothing was decoded). */
int nchmm_basecall_reads(nchmm_ctx* ctx, const nchmm_train_opts* opts, size_t n_models, const float* model_states_Sx10,
                         size_t n_reads, const uint64_t* strand_off, const float* mean, const float* stdv, const float* start,
                         size_t n_jobs, const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1,
                         const float* job_pm, const float* job_st, const int32_t* read_preferred, uint16_t* out_state,
                         int32_t* out_best_job, float* out_best_logp);

/* ------------------------------------------------------------------------------------------
 * Counters (what the 8-GPU run gathers with one RCCL all-reduce)
 * out[0]=reads decoded, [1]=events decoded, [2]=back-pointer bytes written, [3]=kernel launches,
 * [4]=windows (FB), [5]=FB event-rounds, [6]=device bytes allocated,
 * [7]=FB windows the rescaled kernels handed to the exact log-space redo (synchronises the stream)
 * ---------------------------------------------------------------------------------------- */
int nchmm_counters(const nchmm_ctx* ctx, uint64_t out[8]);

/* hipEvent times (ms) of the kernels most recently launched through this context, measured on the
 * stream they ran on: out[0] = the most recent Viterbi launch (sweep + the traceback each block does when
 * its read ends; a kernel time only when no other launch ran beside it), out[1] = 0 (there is no separate
 * traceback kernel), out[2] = forward-backward kernel, out[3] reserved.  Blocks until those kernels have finished. */
int nchmm_last_kernel_ms(nchmm_ctx* ctx, float out[4]);

/* The shader clock (MHz) the device sustains under a full-chip VALU load, measured now by a ~3 ms probe kernel on the
 * context's stream: shader-clock ticks over constant-rate wall-clock ticks.  The hot kernels are VALU-issue bound, so
 * their duration scales with 1 / this clock; bench.py reports it next to the throughput (boxes of one pool differ).
 * No reference counterpart (measurement aid). */
int nchmm_shader_clock_mhz(nchmm_ctx* ctx, double* out_mhz);

/* Phase counters of the Viterbi kernel, accumulated over launches while the environment variable
 * NCHMM_PROFILE=1 was set at nchmm_create time: out[0] = forward-sweep ticks summed over blocks,
 * [1] = arg-max + traceback ticks, [2] = whole-block ticks, [3] = blocks (100 MHz wall_clock64 ticks), [4] = traceback
 * segments that had to be re-walked, [5] = speculative traceback segments, [6] = wave-columns of the forward sweep
 * that took the exact sum-by-sum group rescan (a smaller alpha could round to the winner's sum), [7] = 3-way combines
 * (per wave and cell) that took the exact lowest-predecessor-index rule because two class winners were equal. */
int nchmm_profile_ticks(nchmm_ctx* ctx, uint64_t out[8], int reset);
/* (start, end) wall_clock64 ticks of the first 2048 blocks of the last profiled Viterbi launch */
int nchmm_profile_blocks(nchmm_ctx* ctx, uint64_t* out_2x2048);

/* number of resident thread-block slots (persistent grid size) the Viterbi kernel launches */
int nchmm_grid_slots(const nchmm_ctx* ctx, int* viterbi_slots);

/* Which form of the Viterbi sweep launches take.  A read is sequential (Viterbi.hpp:72-96 is a column-by-column
 * recurrence), so there are two ways to put reads on a CU: two reads side by side on 8 waves each (NCHMM_SWEEP_WIDE: the most
 * events per second) or one read on 16 waves (NCHMM_SWEEP_LL: about half the time per event for that read) -- what a batch
 * with few or very unequal reads, and the reference's one-strand-per-call shape (nanocall.cpp:687-689), need.  Results are
 * bit-identical.  NCHMM_SWEEP_AUTO (default; environment NCHMM_VIT_SWEEP=auto|wide|ll|ahead at nchmm_create) decides per launch from
 * the read lengths.  No reference counterpart (the reference's parallelism is pfor over reads, nanocall.cpp:611). */
#define NCHMM_SWEEP_AUTO 0
#define NCHMM_SWEEP_WIDE 1
#define NCHMM_SWEEP_LL 2
/* NCHMM_SWEEP_AHEAD: the low-latency form with the emission log-densities (Pore_Model.hpp:145-149: 60 % of a column's arithmetic,
 * independent of the recurrence) of the launch's longest reads computed AHEAD by every CU of the device, 16 KiB per event in a
 * buffer of NCHMM_EM_BUDGET_MB (default 256: what stays in the memory-side cache): those reads' columns carry the max-plus
 * recurrence only -- 0.59 us per event against 0.80 for a strand decoded on its own.  Forced, it applies to one-call batches (as many of their longest reads as the buffer holds) and to
 * nchmm_viterbi_dev batches that fit the buffer whole; AUTO takes it where the plan prices it cheaper (nchmm_plan.hpp). */
#define NCHMM_SWEEP_AHEAD 3
int nchmm_set_sweep(nchmm_ctx* ctx, int mode);
/* out[0] = launches of the wide form so far, [1] = of the low-latency form (with or without emissions ahead), [2] / [3] = reads
 * they decoded */
int nchmm_sweep_stats(const nchmm_ctx* ctx, uint64_t out[4]);
/* out[0] = low-latency launches that had emissions computed ahead, [1] = reads, [2] = events (rows of the buffer) ahead */
int nchmm_ahead_stats(const nchmm_ctx* ctx, uint64_t out[3]);

/* Device memory the context holds now (out[0]) and at its high-water mark (out[1]), in bytes: tables, staging and the
 * back-pointer workspace -- one region of 4 KiB per event of the LONGEST read for every thread block that can be resident
 * (2 per CU, a few to spare), independent of the number of reads: a block walks its read back as soon as it has swept
 * it and reuses the region (the reference's Viterbi matrix is 32 KiB per event, one per pfor thread, Viterbi.hpp:50).
 * NCHMM_WS_BUDGET_MB bounds the workspace: reads too long for the full set of regions run on as many blocks as the
 * budget has regions for. */
int nchmm_mem_stats(const nchmm_ctx* ctx, uint64_t out[2]);

/* ------------------------------------------------------------------------------------------
 * Device pool -- the reference's read-parallel pfor loops (src/nanocall/nanocall.cpp:282-579, :611-866; `-t`
 * worker threads, one read each) become: one context + one host thread per GPU of the node, reads assigned to
 * devices by total event count (longest-processing-time first), every device working through its shard with the
 * batched entry points above.  Reads are independent, so there is no collective on the data path; the only
 * exchange is the counter reduction of nchmm_pool_counters.
 * ---------------------------------------------------------------------------------------- */
typedef struct nchmm_pool nchmm_pool;

int nchmm_device_count(int* n);   /* HIP devices visible to this process (0 and NCHMM_E_NO_DEVICE when none) */

/* Free / total device memory of one GPU in bytes (hipMemGetInfo), asked through the library's own HIP runtime, after
 * the device has drained.  What a long-running host checks the ownership contract with: the reference's DP object frees
 * its matrix on scope exit (src/nanocall/Viterbi.hpp:50 -- a std::vector member), so every byte a context allocated
 * must be free again after nchmm_destroy. */
int nchmm_device_mem_info(int device_id, uint64_t* free_bytes, uint64_t* total_bytes);

/* device_ids NULL = 0 .. n_devices-1.  An id may repeat (several contexts on one GPU: the host-thread / sharding logic
 * can then be exercised on a single-GPU machine; the counter reduction then runs on the host instead of RCCL). */
int nchmm_pool_create(nchmm_pool** out, int n_devices, const int* device_ids);
int nchmm_pool_destroy(nchmm_pool* pool);
int nchmm_pool_size(const nchmm_pool* pool);
nchmm_ctx* nchmm_pool_ctx(nchmm_pool* pool, int i);

/* LPT: items in descending weight (stable) each go to the least-loaded shard (lowest index on ties);
 * equal weights throughout (BASELINE config 4) give contiguous slices.  shard_of_item[i] in [0, n_shards). */
int nchmm_lpt_partition(size_t n_items, const uint64_t* weight, int n_shards, int32_t* shard_of_item);

/* nchmm_train_reads / nchmm_basecall_reads over the pool: same arguments and results (job and read indices are the
 * caller's), the reads sharded across the devices by their event counts, one host thread per device. */
int nchmm_pool_train_reads(nchmm_pool* pool, const nchmm_train_opts* opts, size_t n_models, const float* model_states_Sx10,
                           size_t n_reads, const uint64_t* strand_off, const float* mean, const float* stdv, const float* start,
                           size_t n_jobs, const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1, float* job_pm,
                           float* job_st, float* job_fit, uint32_t* job_rounds, int32_t* read_preferred);
int nchmm_pool_basecall_reads(nchmm_pool* pool, const nchmm_train_opts* opts, size_t n_models, const float* model_states_Sx10,
                              size_t n_reads, const uint64_t* strand_off, const float* mean, const float* stdv, const float* start,
                              size_t n_jobs, const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1,
                              const float* job_pm, const float* job_st, const int32_t* read_preferred, uint16_t* out_state,
                              int32_t* out_best_job, float* out_best_logp);

/* Take the forward-backward alpha-row workspace (16 KiB per window event, bounded by NCHMM_FB_BUDGET_MB) for batches of up to
 * `events` window events now instead of in the first EM round: a host that knows a long run is coming calls it while it still
 * reads its input (the first multi-GiB allocation on a device whose memory is not mapped yet costs ~20 ms per GiB).  No
 * reference counterpart (`Forward_Backward::fill` allocates its matrix per call, Forward_Backward.hpp:52). */
int nchmm_reserve_fb_workspace(nchmm_ctx* ctx, size_t events);
int nchmm_pool_reserve_fb_workspace(nchmm_pool* pool, size_t events_per_device);
/* The same for the Viterbi back-pointer regions (`Viterbi::fill` allocates its matrix per call, Viterbi.hpp:50): a full pool of
 * regions for reads of up to longest_events (0: the longest a full pool fits in NCHMM_WS_BUDGET_MB). */
int nchmm_reserve_viterbi_workspace(nchmm_ctx* ctx, size_t longest_events);
int nchmm_pool_reserve_viterbi_workspace(nchmm_pool* pool, size_t longest_events);

/* nchmm_counters summed over the pool's contexts.  With two or more DISTINCT devices the sum is one RCCL all-reduce
 * (ncclCommInitAll over the pool's devices, single process; librccl is loaded at run time) and *used_rccl = 1;
 * otherwise (one device, repeated ids, librccl missing) it is a host sum and *used_rccl = 0.  NCHMM_POOL_FORCE_RCCL=1
 * takes the RCCL path even for a single device.  used_rccl may be NULL.
 * Threading: a pool keeps one host thread per device for its whole life and runs ONE batched call at a time (concurrent
 * callers queue).  RCCL prints a banner on stdout when its communicators are created; for that moment this call points
 * the process's fd 1 at stderr (a host may be streaming FASTA on stdout) -- so do not write to stdout from another
 * thread while nchmm_pool_counters runs (the command line calls it once, after its last record is flushed). */
int nchmm_pool_counters(nchmm_pool* pool, uint64_t out[8], int* used_rccl);

/* One PROCESS per GPU (what `nanocall --gpus N` starts: the reference's unit of parallelism is a pfor worker inside one
 * process, nanocall.cpp:282,611; here a worker process per device, started before anything touches the HIP runtime).  The only
 * exchange between the workers is the same counter reduction, as one RCCL all-reduce across processes:
 *   nchmm_rccl_unique_id        rank 0 makes the communicator's id (ncclGetUniqueId; 128 bytes) and hands it to the others by any
 *                               means (the command line: over its pipes, through the parent)
 *   nchmm_counters_allreduce    every rank, with that id: ncclCommInitRank(n_ranks, id, rank) on `device_id` (the rank's own
 *                               device, as ITS process numbers it), one ncclAllReduce(sum) of the eight counters in place, the
 *                               communicator destroyed again.  All ranks must call it; their devices must be distinct.
 * NCHMM_E_NO_DEVICE when librccl (NCHMM_RCCL_LIB) cannot be loaded or lacks an entry point, NCHMM_E_HIP when RCCL or HIP
 * fails (nchmm_last_hip_error is not set: no context is involved) -- the caller then sums on the host, as nchmm_pool_counters
 * does.  n_ranks = 1 is allowed (a communicator of one: the whole call sequence on a machine with one GPU). */
#define NCHMM_RCCL_ID_BYTES 128
int nchmm_rccl_unique_id(uint8_t id[NCHMM_RCCL_ID_BYTES]);
int nchmm_counters_allreduce(int device_id, int n_ranks, int rank, const uint8_t id[NCHMM_RCCL_ID_BYTES], uint64_t inout[8]);

#ifdef __cplusplus
}
#endif
#endif
