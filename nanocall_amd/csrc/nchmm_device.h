// nchmm_device.h -- shared declarations between the HIP kernels and the C-ABI host layer.
#ifndef NCHMM_DEVICE_H
#define NCHMM_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nchmm {

constexpr int kStates = 4096;
constexpr int kThreads = 512;         // 8 waves; thread 2t+h owns 8 of the 16 states whose low 8 bits == t
constexpr int kStatesPerThread = 8;
constexpr int kModelFloats = 8 * kStates;               // SoA [field][state]
constexpr int kTransFloats = kStates + 1024 + 256;      // w0[4096] | w1[1024] | w2[256]
constexpr int kMaxSlots = 64;
constexpr unsigned kNoState = 0xFFFFu;

// device image of a pore model: field-major so thread t reads field f of state t+256k at [f][t+256k]
enum ModelField {
    MF_MU = 0, MF_SIGMA, MF_RSIGMA /* RN(1/sigma) */, MF_NEG_LOG_SIGMA, MF_ETA, MF_RETA /* RN(1/eta) */, MF_LAMBDA,
    MF_C /* log_lambda - log_2pi */
};

struct ViterbiArgs {
    const float* cmean;        // SoA events
    const float* stdv;
    const float* lstdv;
    const uint64_t* off;       // n_reads + 1
    const int32_t* model_slot; // per read or null
    const int32_t* trans_slot; // per read or null
    const uint32_t* order;     // processing order or null
    const float* models;       // [kMaxSlots][kModelFloats]
    const float* trans;        // [kMaxSlots][kTransFloats]
    const int32_t* model_fast; // [kMaxSlots] 1 = parameters inside the range the reciprocal division is proven for
    unsigned long long* prof;  // optional [4]: forward ticks, traceback ticks, block ticks, blocks (wall_clock64)
    uint8_t* ws;               // back-pointer workspace: one 4 KiB row per event of the (sub-)batch
    uint64_t ev_base;          // off[first_read]: event index of the first row of the workspace
    unsigned first_read;       // reads [first_read, first_read + n_reads) form this (sub-)batch
    unsigned* cu_progress;     // [4096] per (CU, block slot) events done in this launch; zeroed before launch
    unsigned* last_state;      // [n_reads_total] arg-max state of the last column (forward -> traceback)
    uint16_t* out_state;
    float* out_logp;
    int32_t* out_status;
    unsigned* queue;           // work-queue head, zeroed before launch
    unsigned n_reads;
    int tb_margin;             // events a speculative traceback segment runs before its first owned event
    float log_n_states;        // std::log(4096.f) from the host libm (Viterbi.hpp:51)
    float log_2pi;             // (float)std::log(2.0 * M_PI) (Pore_Model.hpp:28,37)
};

void launch_viterbi(const ViterbiArgs& a, int grid, hipStream_t stream);
void launch_traceback(const ViterbiArgs& a, hipStream_t stream);
int viterbi_blocks_per_cu();

constexpr int kFbTransFloats = 6 * kStates;   // forward c0|c1|c2 then backward c0b|c1b|c2b, per state, log space

struct FwbwArgs {
    const float* cmean;
    const float* stdv;
    const float* lstdv;
    const uint64_t* off;
    const int32_t* scaled_slot;
    const float* pm_params;     // n_win x 6 {scale, shift, drift, var, scale_sd, var_sd} behind scaled_slot[w], or null (identity)
    const int32_t* trans_slot;
    const float* st_params;     // n_win x 2 {p_stay, p_skip} or null
    const float* models;        // [kMaxSlots][kModelFloats]
    const float* trans_fb;      // [kMaxSlots][kFbTransFloats]
    const float* trans;         // [kMaxSlots][kTransFloats] (the group weights w2 are read from here)
    const uint8_t* train_mask;  // [512] bit u of byte tau: state 8*tau+u is a transition-training k-mer
    float* ws_alpha;            // alpha rows, one per event of the batch (4096 floats each)
    float* ws_lpd2;             // [n_win] log2 Pr(data | window), forward kernel -> backward kernel
    int alpha_natural;          // rows are natural logs (caller's out_alpha buffer) instead of the internal base 2
    // rescaled linear-space fast path (fwbw_scaled_kernel.hip) and its exact fallback
    int32_t* ws_exp;            // [total events] cumulative power-of-two exponent taken out of the alpha rows up to each event
    float* ws_zfin;             // [n_win] sum of the last (rescaled) alpha row
    uint8_t* fb_flag;           // [n_win] window left the range the rescaled kernels vouch for -> redo in log space
    unsigned* fb_list;          // [n_win] those windows, appended by the scaled kernels
    unsigned* fb_count;         // [1] their number
    unsigned long long* fb_total;   // [1] running total over the context's lifetime (nchmm_counters)
    const unsigned* win_list;   // log-space kernels: process windows win_list[0 .. *n_list) instead of [0, n_win)
    const unsigned* n_list;
    float* out_log_pr_data;
    float* out_pm_sums;
    float* out_st_sums;
    float* out_beta;
    unsigned* queue;            // [2] work-queue heads: forward kernel, backward kernel
    unsigned n_win;
    float log_n_states;
    float log_2pi;
};

void launch_scale_models(const float* d_states, const int32_t* d_table_idx, const float* d_params, float* d_models,
                         int32_t* d_model_fast, int first_slot, size_t n, float log_2pi, hipStream_t stream);
void launch_expand_transitions(const float* d_wm, const uint8_t* d_masks, float* d_trans, float* d_trans_fb, int first_slot,
                               size_t n, hipStream_t stream);
struct EmGatherArgs {
    const float* mean; const float* stdv; const float* start; const float* lstdv;   // resident raw events
    const uint64_t* win_src;   // [n_win] first raw event of each window
    const uint64_t* off;       // [n_win + 1] packed offsets
    const float* win_drift;    // [n_win]
    float* cmean; float* out_stdv; float* out_lstdv;                                 // packed SoA for the FB kernels
};
struct EmReduceArgs {
    const float* mean; const float* stdv; const float* start;
    const uint64_t* win_src; const uint64_t* off;
    const uint32_t* job_first_win;   // [n_jobs + 1]
    const float* pm_sums;            // packed [events][6]
    int train_drift;
    double* out;                     // [n_jobs][13]
};
void launch_em_gather(const EmGatherArgs& a, unsigned n_win, hipStream_t stream);
void launch_em_reduce(const EmReduceArgs& a, unsigned n_jobs, hipStream_t stream);
void launch_fwbw(const FwbwArgs& a, int grid, hipStream_t stream, bool scaled);
void launch_fwbw_scaled(const FwbwArgs& a, int grid, hipStream_t stream);
int fwbw_scaled_blocks_per_cu();
int fwbw_blocks_per_cu();

}  // namespace nchmm
#endif
