// nchmm_device.h -- shared declarations between the HIP kernels and the C-ABI host layer.
#ifndef NCHMM_DEVICE_H
#define NCHMM_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nchmm {

constexpr int kStates = 4096;
constexpr int kThreads = 512;         // 8 waves; thread 2t+h owns 8 of the 16 states whose low 8 bits == t
constexpr int kStatesPerThread = 8;
constexpr int kLlThreads = 1024;      // the low-latency sweep (viterbi_ll_kernel.hip): 16 waves, thread 4t+y owns the 4 states t + 256(4x+y)
constexpr int kModelFloats = 8 * kStates;               // SoA [field][state]
constexpr int kTransFloats = kStates + 1024 + 256;      // w0[4096] | w1[1024] | w2[256]
constexpr int kMaxSlots = 64;
constexpr unsigned kNoState = 0xFFFFu;

// Back-pointer row of one event: one byte per state (viterbi_kernel.hip)
constexpr unsigned kBpRowBytes = kStates;

// device image of a pore model: field-major so thread t reads field f of state t+256k at [f][t+256k]
enum ModelField {
    MF_MU = 0, MF_SIGMA, MF_RSIGMA /* RN(1/sigma) */, MF_NEG_LOG_SIGMA, MF_ETA, MF_RETA /* RN(1/eta) */, MF_LAMBDA,
    MF_C /* log_lambda - log_2pi */
};

// Back-pointer workspace: one REGION per resident block ("slot"), not per read.  A block writes the rows of the read it is
// sweeping into its region, walks them back itself as soon as the last column is done (in the shadow of the co-resident
// block's sweep), and reuses the region for its next read.  Regions are handed out per XCD (XCC_ID): a region is only ever
// touched through one L2, so a block that takes over a region sees exactly what it wrote itself.
constexpr unsigned kXcds = 8;
constexpr unsigned kNoSlot = 0xFFFFFFFFu;
constexpr uint64_t kNoEmRow = ~(uint64_t)0;

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
    uint8_t* ws;               // back-pointer regions: region s at ws + s * slot_bytes, row i of the current read at + i * kBpRowBytes
    uint64_t slot_bytes;       // >= kBpRowBytes * (longest read of the launch)
    unsigned* slot_owner;      // [kXcds][slots_per_xcd] 0 = free; null: region = blockIdx.x (launches do not overlap then)
    unsigned slots_per_xcd;
    unsigned* host_err;        // pinned host word: set to 1 by a block that found no region (cannot happen with a sane pool)
    unsigned first_read;       // reads [first_read, first_read + n_reads) form this launch
    unsigned* cu_progress;     // [4096] per (CU, block slot): launch tag << 20 | events done in this launch; zeroed on exit
    unsigned launch_tag;       // 1 .. 4095, distinguishes co-resident blocks of different launches
    uint16_t* out_state;
    float* out_logp;
    int32_t* out_status;
    unsigned* queue;           // work-queue head of this launch's lane: never reset -- tickets start at queue_base
    unsigned queue_base;       // value of *queue when the launch starts (host-tracked: every launch adds n_reads + grid)
    unsigned n_reads;
    // emissions computed ahead (emission_kernel.hip, read by viterbi_ll_kernel.hip): row i of read r -- 4096 floats, thread-major for
    // the low-latency ownership map (thread tau's four states at [4 tau .. 4 tau + 3]) -- at em + (em_row0[r] + i) * 4096;
    // em_row0[r] = kNoEmRow: the read computes its emissions itself.  em == nullptr: nobody's are ahead.
    const float* em;
    const uint64_t* em_row0;   // indexed by read; null with em != null: row of read r's event i = off[r] + i
    uint64_t em_rows;          // rows the buffer at em holds: a read whose rows would not all lie inside computes its emissions itself
    int tb_margin;             // events a speculative traceback segment runs before its first owned event
    float log_n_states;        // std::log(4096.f) from the host libm (Viterbi.hpp:51)
    float log_2pi;             // (float)std::log(2.0 * M_PI) (Pore_Model.hpp:28,37)
};

void launch_viterbi(const ViterbiArgs& a, int grid, hipStream_t stream);
int viterbi_blocks_per_cu();
// the low-latency form: one read per CU on 16 waves (viterbi_ll_kernel.hip); same arguments, same results, cu_progress unused
void launch_viterbi_ll(const ViterbiArgs& a, int grid, hipStream_t stream);
// The emissions of the first n_ahead reads of the launch's order (a.order / a.first_read as in the sweep) into a.em, so that the
// sweep of those reads carries only the recurrence (emission_kernel.hip).  max_events = the longest of them.
void launch_emissions(const ViterbiArgs& a, unsigned n_ahead, uint64_t max_events, float* em, hipStream_t stream);

// plan_kernel.hip: longest-first order (and the reads longer than `outlier_above`) of a batch whose offsets are on the device
void launch_plan_order(const uint64_t* d_off, unsigned n, uint64_t span, uint64_t outlier_above, uint32_t* d_order, uint32_t* d_outlier,
                       unsigned long long* d_counts, hipStream_t stream);

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
    unsigned long long* prof;   // [8] NCHMM_PROFILE=1 only; the shipped kernels do not touch it (tools/ubench/fb_phases.py builds a copy that does)
    unsigned n_win;
    float log_n_states;
    float log_2pi;
};

void launch_scale_models(const float* d_states, const int32_t* d_table_idx, const float* d_params, float* d_models,
                         int32_t* d_model_fast, int first_slot, size_t n, float log_2pi, hipStream_t stream);
void launch_expand_transitions(const float* d_wm, const uint8_t* d_masks, float* d_trans, float* d_trans_fb, int first_slot,
                               size_t n, hipStream_t stream);
// glibc 2.35 logf (sysdeps/ieee754/flt-32/e_logf.c, table + cubic in double) as x86-64 machines with FMA execute it
// (the ifunc-selected __logf_fma: every  a * b + c  below is one fused operation there -- read off the disassembly of
// libm.so.6, and off the published source for the constants).  Device fp64 FMA is IEEE, so the double result and its
// rounding to float are the host's, bit for bit: tests/test_logf_gpu.py checks all 2^31 non-negative inputs (and the
// negative / NaN classes) against the host libm.  This is what lets Event::update_logs' log(stdv) (Event.hpp:43) move to
// the device without touching the bit-exact contract of the Viterbi path.
__device__ __forceinline__ float glibc_logf(float x)
{
    // T[i] = {1/c_i, log(c_i)}, c_i near the centre of the i-th of 16 sub-intervals of [0x1.66p-1, 0x1.66p0)
    const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                             0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                             0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                             0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                             -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                             -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                             0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    unsigned ix = __builtin_bit_cast(unsigned, x);
    if (ix == 0x3f800000u) return 0.0f;                                   // log(1) = +0
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {                  // x < 0x1p-126, or inf, or nan
        if (ix * 2u == 0u) return -__builtin_inff();                      // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;                                  // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");   // x < 0 or NaN
        ix = __builtin_bit_cast(unsigned, x * 0x1p23f);                   // subnormal: normalise
        ix -= 23u << 23;
    }
    const unsigned tmp = ix - 0x3f330000u;
    const unsigned i = (tmp >> 19) & 15u;
    const int k = (int)tmp >> 23;                                         // arithmetic shift
    const unsigned iz = ix - (tmp & 0xff800000u);
    const double z = (double)__builtin_bit_cast(float, iz);
    const double r = __builtin_fma(z, invc[i], -1.0);
    const double y0 = __builtin_fma((double)k, Ln2, logc[i]);
    const double r2 = r * r;
    double y = __builtin_fma(A1, r, A2);
    y = __builtin_fma(A0, r2, y);
    y = __builtin_fma(y, r2, y0 + r);
    return (float)y;
}

struct EmGatherArgs {
    const float* mean; const float* stdv; const float* start;
    const float* lstdv;   // resident raw events; lstdv == nullptr: log(stdv) is computed here (glibc_logf), and
                          // stdv == 0 becomes 0.01 first (Event::update_logs, Event.hpp:39-43)
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
void launch_em_gather(const EmGatherArgs& a, unsigned n_win, hipStream_t stream, unsigned max_events);
void launch_logf(const float* in, float* out, size_t n, hipStream_t stream);
void launch_em_reduce(const EmReduceArgs& a, unsigned n_jobs, hipStream_t stream);
void launch_fwbw(const FwbwArgs& a, int grid, hipStream_t stream, bool scaled);
void launch_fwbw_scaled(const FwbwArgs& a, int grid, hipStream_t stream);
int fwbw_scaled_blocks_per_cu();
int fwbw_blocks_per_cu();

}  // namespace nchmm
#endif
