/*
 * nc_oracle.h -- CPU restatement of nanocall's HMM hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity checker for the HIP path: only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it.  The product library (nanocall_amd/csrc) never
 * links, loads or calls anything in this directory.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - Kmer algebra and the builtin model tables are PINNED against the real reference, compiled
 *     unmodified from /root/reference into oracle/_ref (oracle/Makefile target `ref`).
 *   - State_Transitions weights are pinned against the 18 mask->weight values, arc count and
 *     degree histogram that SURVEY.md section 8a-4 records from a run of the reference.
 *   - Viterbi / Pore_Model / Event / Forward_Backward / Parameter_Trainer: "parity unpinned".
 *     Their reference headers #include hpptools/fast5 headers that are absent from
 *     /root/reference (empty submodules), so they cannot be compiled here without stand-ins,
 *     and the reference has no tests or golden vectors.  These functions follow the reference
 *     source line by line (citations on every function) but have not been run against it.
 *   - logsumset (hpptools, un-vendored, no pinned version): restated from its call sites; any
 *     log-sum-exp is inside the 1e-4 relative tolerance north_star sets for FB/EM.
 *
 * All arithmetic is IEEE binary32 unless a comment says double; build with -ffp-contract=off and
 * no -march flags so the float behaviour equals the reference's generic x86-64 Release build
 * (src/CMakeLists.txt:144,162).
 */
#ifndef NC_ORACLE_H
#define NC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NCO_KMER_SIZE 6
#define NCO_N_STATES 4096u

/* ---- Kmer (src/nanocall/Kmer.hpp) ---- */
void     nco_kmer_to_string(unsigned k, char out[7]);
unsigned nco_kmer_to_int(const char* s);
unsigned nco_kmer_min_skip(unsigned k1, unsigned k2);
unsigned nco_kmer_prefix(unsigned i, unsigned k);
unsigned nco_kmer_suffix(unsigned i, unsigned k);
unsigned nco_kmer_max_self_overlap(unsigned i);
void     nco_kmer_neighbour_list(unsigned i, unsigned d, unsigned* out /* 4 or 16 */);

/* ---- State_Transitions (src/nanocall/State_Transitions.hpp) ---- */
typedef struct {
    /* CSR by destination state (from_v) and by source state (to_v); both ascending. */
    uint32_t from_ptr[NCO_N_STATES + 1];
    uint32_t to_ptr[NCO_N_STATES + 1];
    uint32_t n_arcs;
    uint32_t* from_idx; float* from_logw;
    uint32_t* to_idx;   float* to_logw;
} nco_transitions;

float nco_trans_prob(unsigned i, unsigned j, float p_stay, float p_step, float p_skip_1);
nco_transitions* nco_transitions_fast(float p_skip, float p_stay);
void  nco_transitions_free(nco_transitions* t);
/* flat accessors for ctypes */
uint32_t nco_transitions_n_arcs(const nco_transitions* t);
void nco_transitions_export_from(const nco_transitions* t, uint32_t* row_ptr, uint32_t* idx, float* logw);
void nco_transitions_export_to(const nco_transitions* t, uint32_t* row_ptr, uint32_t* idx, float* logw);

/* ---- Pore_Model (src/nanocall/Pore_Model.hpp) ---- */
typedef struct {
    float level_mean, level_stdv, sd_mean, sd_stdv, sd_lambda;
    float log_level_mean, log_level_stdv, log_sd_mean, log_sd_stdv, log_sd_lambda;
} nco_state;

typedef struct { float scale, shift, drift, var, scale_sd, var_sd; } nco_pm_params;

typedef struct {
    nco_state st[NCO_N_STATES];
    float mean, stdv;
} nco_model;

void  nco_model_load_from_vector(nco_model* m, const float* v /* 4096*4 */);
void  nco_model_scale(nco_model* m, const nco_pm_params* p);
/* 4096 x 6 {level_mean, level_stdv, log_level_stdv, sd_mean, sd_lambda, log_sd_lambda} */
void  nco_model_export6(const nco_model* m, float* out);
float nco_log_pr_corrected_emission(const nco_state* s, float corrected_mean, float stdv, float log_stdv);

/* ---- Event (src/nanocall/Event.hpp) ---- */
typedef struct {
    float mean, corrected_mean, stdv, start, length;
    float log_mean, log_corrected_mean, log_stdv;
    unsigned model_state_idx;
    int move;
} nco_event;

void nco_event_init(nco_event* e, float mean, float stdv, float start, float length);
void nco_events_apply_drift_correction(nco_event* ev, size_t n, float drift);
/* returns length written (no NUL counted); out must hold 6 + 6*(n-1) + 1 chars */
size_t nco_events_get_base_seq(const nco_event* ev, size_t n, char* out);
/* write_fasta (src/nanocall/nanocall.cpp:584-591); returns bytes written */
size_t nco_write_fasta(char* out, size_t cap, const char* name, const char* seq, unsigned line_width);

/* ---- Viterbi (src/nanocall/Viterbi.hpp) ---- */
/* Fills ev[i].model_state_idx / move, returns path probability.  Allocates the reference's full
 * n_events x 4096 matrix of {float alpha; unsigned beta} (Viterbi.hpp:26-30,50). Returns NAN and
 * leaves events untouched when n == 0 or allocation fails. */
float nco_viterbi_fill(const nco_model* pm, const nco_transitions* st, nco_event* ev, size_t n);
/* convenience for ctypes: SoA in, arrays out */
uint64_t nco_viterbi_tie_cells(const nco_model* pm, const nco_transitions* st, size_t n, const float* corrected_mean,
                               const float* stdv, const float* log_stdv);
float nco_viterbi_soa(const nco_model* pm, const nco_transitions* st, size_t n,
                      const float* corrected_mean, const float* stdv, const float* log_stdv,
                      uint16_t* out_state, int32_t* out_move);

/* ---- Forward_Backward (src/nanocall/Forward_Backward.hpp) ---- */
typedef struct {
    size_t n_events;
    float* alpha;  /* n x 4096 */
    float* beta;   /* n x 4096 */
    float log_pr_data;
} nco_fwbw;

nco_fwbw* nco_fwbw_fill(const nco_model* pm, const nco_transitions* st, const nco_event* ev, size_t n);
void  nco_fwbw_free(nco_fwbw* f);
float nco_fwbw_soa(const nco_model* pm, const nco_transitions* st, size_t n,
                   const float* corrected_mean, const float* stdv, const float* log_stdv,
                   float* out_alpha /* n*4096 or NULL */, float* out_beta /* n*4096 or NULL */);

/* logsumset restatement (hpptools include/logsumset.hpp, un-vendored; see header note) */
float nco_logsumset_val(float* vals, size_t n); /* destroys vals */

/* ---- Parameter_Trainer (src/nanocall/Parameter_Trainer.hpp) ---- */
typedef struct { float p_stay, p_skip; } nco_st_params;

typedef struct {
    /* inputs */
    size_t n_seqs;
    const nco_event* const* seqs;  /* uncorrected training windows */
    const size_t* seq_len;
    const unsigned* seq_strand;
    const nco_model* model[2];     /* unscaled */
    float default_p_stay, default_p_skip; /* is_default() reference values (State_Transitions.hpp:20-37) */
    int train_drift;               /* Parameter_Trainer::pm_train_drift() */
} nco_train_input;

/* One EM round (Parameter_Trainer.hpp:541-579). Returns fit; writes new params, *done. */
float nco_train_one_round(const nco_train_input* in,
                          const nco_pm_params* crt_pm, const nco_st_params crt_st[2],
                          nco_pm_params* new_pm, nco_st_params new_st[2],
                          int* done, int train_scaling, int train_transitions);
/* st_train_kmers (Parameter_Trainer.hpp:30-57); returns count, fills out (cap 4096) */
unsigned nco_st_train_kmers(unsigned* out);

/* flat ctypes entry for one round on SoA windows (mean, stdv, start per event) */
float nco_train_one_round_soa(size_t n_seqs, const uint64_t* off, const unsigned* strand,
                              const float* mean, const float* stdv, const float* start,
                              const float* model0_4096x4, const float* model1_4096x4,
                              float default_p_stay, float default_p_skip, int train_drift,
                              const float crt_pm[6], const float crt_st[4],
                              float new_pm[6], float new_st[4], int* done,
                              int train_scaling, int train_transitions);

/* The host libm's logf over an array (what Event::update_logs' std::log(float) calls, Event.hpp:43): the reference value
 * for the device port, compared bit for bit.  Returns the number of i with bits(logf(x[i])) != bits(y[i]) (NaNs compare
 * equal to each other), and the first such index through *first_bad (or -1). */
size_t nco_logf_mismatches(const float* x, const float* y, size_t n, long long* first_bad);

/* ---- Fast5_Summary (src/nanocall/Fast5_Summary.hpp): strand segmentation, event filter, initial scaling ----
 * PARITY UNPINNED: fast5::EventDetection_Event_Entry comes from the un-vendored fast5 submodule ([recalled]:
 * {double mean; double stdv; long long start; long long length;}) and alg::mean_stdv_of from hpptools
 * (restated as nco_mean_stdv: float accumulation, sample (n-1) stdv -- the same restatement the oracle's
 * Pore_Model statistics use). */
typedef struct { double mean, stdv; long long start, length; } nco_ed_event;

typedef struct {
    unsigned min_ed_events;               /* Fast5_Summary.hpp:74-78   (10) */
    unsigned max_ed_events;               /* :80-84                     (100000) */
    double abasic_level_top_percent;      /* :93-97                     (1.0) */
    double abasic_level_top_offset;       /* :100-104                   (r9: 0.0, r73: 5.0; nanocall.cpp:943-964) */
    unsigned template_only;               /* :121-125 */
    unsigned trim_margins[4];             /* :128-132  sq_start, sq_end, hp_start, hp_end (50 each) */
} nco_f5_opts;

typedef struct {
    unsigned num_ed_events;               /* 0 = read skipped */
    float abasic_level;
    unsigned strand_bounds[4];
    int scale_strands_together;
    float time_length[2];
} nco_f5_summary;

void nco_mean_stdv(const float* v, size_t n, float* mean, float* stdv);
/* summarize (:138-319) from the EventDetection table; n_ed_file = events in the file */
void nco_f5_summarize(const nco_f5_opts* o, const nco_ed_event* ed, size_t n_ed_file, float sampling_rate, int sst,
                      nco_f5_summary* out);
/* load_events (:321-370) for strand st: filtered events -> mean/stdv/start/length (stdv after update_logs); returns count.
 * Arrays must hold strand_bounds[2st+1] - strand_bounds[2st] entries. */
size_t nco_f5_load_events(const nco_f5_summary* s, const nco_ed_event* ed, float sampling_rate, unsigned st, float* mean,
                          float* stdv, float* start, float* length);
/* initial model scaling (:223-278): together != 0 -> pair (r0, r1 = strand mean/stdv; m0, m1 = model mean/stdv),
 * else single strand (r0, m0 only).  out = {scale, shift}. */
void nco_f5_initial_scaling(int together, const float r0[2], const float r1[2], const float m0[2], const float m1[2],
                            float out[2]);

#ifdef __cplusplus
}
#endif
#endif
