// nchmm_ctx.hpp -- the device context behind the opaque nchmm_ctx of include/nanocall_hip.h, shared by the translation
// units that enqueue device work (nchmm_api.cpp: tables, device-pointer entry points, forward-backward;
// nchmm_pipeline.cpp: the host-pointer Viterbi entry points).
#ifndef NCHMM_CTX_HPP
#define NCHMM_CTX_HPP

#include "nanocall_hip.h"
#include "nchmm_device.h"

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

namespace nchmm {
struct PipeState;
}

namespace nchmm {
// A compute lane of the Viterbi path: consecutive launches go to the lanes in turn so that the blocks of one launch start in
// the places the previous launches' blocks vacate (viterbi_kernel.hip).  Lane 0 runs on the context's own stream.  Three: a
// launch lasts as long as its longest read (a read is sequential), on log-normally long reads 2.5 x as long as its share of
// the work -- with three in turn the stragglers of two launches hide behind the bulk of the third (same box, 1024 ragged
// reads per batch: 220 Mevents/s on two lanes, 301 on three).  Four lanes + the copy-in stream + the caller's are more
// streams than the runtime has hardware queues for (4): streams then share a queue and wait for each other (267; config 2
// drops from 343 to 324) -- profiles/r04_lanes_ab.txt.
constexpr int kVitLanes = 3;
struct VitLaneState {
    hipStream_t stream = nullptr;
    unsigned vq_base = 0;            // what this lane's queue head will read when its next launch starts (never reset)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the lane's most recent launch (timing)
    hipEvent_t done = nullptr;       // behind the lane's most recent launch (ordering)
    bool pending = false;            // work queued that no HOST-side wait has seen finish (cleared by a stream / event synchronise only)
    bool joined = true;              // ... and a stream already waits for it (viterbi_join queued the stream wait on joined_to)
    hipStream_t joined_to = nullptr;
};
// What one EM round uses on the device and must keep to itself while it is in flight: a stream, the staging buffers, the alpha
// rows, the work-queue words, a pinned host arena.  The context holds two sets -- its own fields and `em_other` -- and
// em_lane_select swaps them, so that every function that queues forward-backward work runs unchanged on either.
// nchmm_train_reads keeps the rounds of one half of its jobs on the device while the host finishes and prepares the other half's.
struct EmLaneRes {
    hipStream_t stream = nullptr;
    void* d_stage = nullptr; size_t stage_bytes = 0;
    float* d_fb_ws = nullptr; size_t fb_ws_floats = 0;
    void* d_fb_aux = nullptr; size_t fb_aux_bytes = 0;
    unsigned* d_queue = nullptr;
    void* d_tab_stage = nullptr; size_t tab_stage_bytes = 0;
    void* h_pin = nullptr; size_t h_pin_bytes = 0, pin_cursor = 0;
    hipEvent_t ev_fb0 = nullptr, ev_fb1 = nullptr;
    bool fb_timed = false;
};
}  // namespace nchmm

struct nchmm_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool external_stream = false;   // stream was set by the caller (0 is then the legacy default stream)
    int last_hip = 0;
    int n_cu = 0;
    int vit_slots = 0;              // block slots of the wide sweep (viterbi_kernel.hip: two per CU)
    int sweep_mode = 0;             // nchmm::Sweep: 0 = per launch by nchmm_plan.hpp choose_sweep, 1 = wide always, 2 = ll always (NCHMM_VIT_SWEEP, nchmm_set_sweep)
    uint64_t sweep_stats[4] = {0, 0, 0, 0};   // launches wide, launches ll, reads wide, reads ll
    uint64_t ahead_stats[3] = {0, 0, 0};      // low-latency launches with emissions ahead, reads ahead, rows (events) ahead
    int fb_slots = 0;
    float* d_models = nullptr;      // [kMaxSlots][kModelFloats]
    float* d_trans = nullptr;       // [kMaxSlots][kTransFloats]   log-space w0|w1|w2
    float* d_trans_fb = nullptr;    // [kMaxSlots][kFbTransFloats] per-state forward/backward weights for FB
    uint8_t* d_train_mask = nullptr; // [512] transition-training k-mers, one bit per state
    unsigned* d_queue = nullptr;    // forward-backward work-queue heads [1..4], redo count [5]
    unsigned* d_vq = nullptr;       // Viterbi work-queue heads [lane] and per-(CU, block slot) progress words [16..]
    nchmm::VitLaneState lane[nchmm::kVitLanes];
    int next_lane = 0;
    int last_lane = -1;             // lane of the most recent launch
    unsigned launch_seq = 0;        // launch tags (1 .. 4095, cyclic)
    hipEvent_t ev_entry = nullptr;  // what was on the caller's stream when a batch was queued
    unsigned* h_err = nullptr;      // pinned host word a block sets when it finds no back-pointer region
    int32_t* d_model_fast = nullptr; // [kMaxSlots]
    unsigned long long* d_prof = nullptr; // [4] phase counters when NCHMM_PROFILE=1
    bool profile = false;
    int tb_margin = 64;             // NCHMM_TB_MARGIN overrides (test hook: 0 forces the re-walk path); profiles/r04_inblock_tb_params_ab.txt:
                                    // 0 of 698 368 speculative segments un-merged at 128, 11 at 64, 3 190 at 32 -- and a miss only costs a re-walk
    uint8_t* d_ws = nullptr;        // viterbi back-pointer workspace: ws_regions regions of slot_bytes (4 KiB per event of the
    size_t ws_bytes = 0;            // longest read), one per resident block
    size_t slot_bytes = 0;
    unsigned ws_regions = 0;
    bool ws_pooled = false;         // regions are handed out by the blocks themselves (per XCD): launches may overlap
    uint8_t* d_ws_big = nullptr;    // regions for the outliers of a batch (reads too long for the pool): big_regions x big_slot_bytes
    size_t big_slot_bytes = 0;
    unsigned big_regions = 0;
    hipEvent_t ev_big = nullptr;    // behind the most recent launch of outliers (they use regions 0 .. grid-1: one at a time)
    bool big_pending = false;
    // emissions computed ahead of a low-latency sweep (emission_kernel.hip): rows of 4096 floats, one buffer, one launch at a time
    float* d_em = nullptr;
    size_t em_bytes = 0, em_budget = 0;      // NCHMM_EM_BUDGET_MB (default 256)
    hipEvent_t ev_em = nullptr;     // behind the most recent sweep that read d_em
    double plan_clock_mhz = 0.0;    // the sustained shader clock the sweep plan prices with (0: the clock its rates were measured at): NCHMM_PLAN_CLOCK_MHZ, or the last nchmm_shader_clock_mhz
    size_t tight_skip_reads = 0, tight_skip_longest = 0;   // nchmm_viterbi_dev*: the batch shape whose device-side plan found most reads long
    bool em_pending = false;
    unsigned* d_slot_owner = nullptr;   // [kXcds][slots_per_xcd]
    unsigned slots_per_xcd = 0;     // capacity of the owner table per XCD
    unsigned ws_per_xcd = 0;        // regions per XCD the workspace holds now (<= slots_per_xcd)
    // device-side plan of the device-pointer forms (plan_kernel.hip): per lane [order n | outliers n] and four counts
    uint32_t* d_plan[nchmm::kVitLanes + 1] = {nullptr, nullptr, nullptr, nullptr};     // (the last: the tight path's, used with every lane idle)
    size_t plan_cap[nchmm::kVitLanes + 1] = {0, 0, 0, 0};
    unsigned long long* d_plan_counts = nullptr;   // [kVitLanes][4]
    unsigned long long* h_plan_counts = nullptr;   // pinned [4]: read back only when the pool does not fit the longest read
    hipStream_t s_in = nullptr;     // copy-in stream of the host-pointer pipeline (nchmm_pipeline.cpp); it computes on own_stream
    nchmm::PipeState* pipe = nullptr;   // batches in flight (nchmm_pipeline.cpp)
    void* combiner = nullptr;           // nchmm_viterbi_strand's batcher (nchmm_combine.cpp), created on first use
    void* win_combiner = nullptr;       // nchmm_fwbw_windows' batcher
    std::mutex combine_run;             // one combined batch on the device at a time, strands or windows (the context is not re-entrant)
    size_t peak_bytes = 0;          // high-water mark of counters[6] (device bytes held)
    size_t ws_budget = 0;           // largest workspace we are willing to allocate (bytes)
    size_t fb_budget = 0;           // same for the forward-backward alpha rows (16 KiB per event)
    float* d_fb_ws = nullptr;       // FB alpha workspace
    size_t fb_ws_floats = 0;
    void* d_fb_aux = nullptr;       // FB per-call scratch: lpd2 | last-row totals | redo list | redo flags | row exponents
    size_t fb_aux_bytes = 0;
    unsigned long long* d_fb_total = nullptr;   // windows redone in log space, running total
    void* d_em_events = nullptr;    // resident raw events of an EM run: mean | stdv | start | log_stdv (nchmm_em_load_events)
    size_t em_events_bytes = 0, em_n_events = 0;
    bool fb_force_log = false;      // NCHMM_FB_FORCE_LOG: never take the rescaled linear-space kernels
    // staging buffers of the host-pointer entry points
    void* d_stage = nullptr;
    size_t stage_bytes = 0;
    uint8_t* d_masks = nullptr;     // overlap-mask ids of the stay / step-group / skip-group arcs (5376 bytes)
    void* d_tab_stage = nullptr;    // device staging of the loaded (not yet scaled) tables + per-slot parameters
    size_t tab_stage_bytes = 0;
    void* h_pin = nullptr;          // pinned host buffer for the batched table uploads (reused across calls)
    size_t h_pin_bytes = 0;
    size_t pin_cursor = 0;          // em_async: what of the pinned buffer this round has handed out
    hipEvent_t ev_fb0 = nullptr, ev_fb1 = nullptr;
    bool vit_timed = false, fb_timed = false;
    nchmm::EmLaneRes em_other;      // the resources of the EM lane that is NOT selected (lane 1 until em_lane_select(c, 1))
    int em_lane = 0;                // which lane's resources the fields above are
    bool em_other_made = false;
    bool em_async = false;          // table uploads and EM rounds queue their work and return (nchmm_train_reads waits per lane);
                                    // the pinned buffer is handed out piecewise, not reused, until the lane has been waited for
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n_slots = 0;                // capacity of the model / transition slot tables (grows on demand)
    std::vector<char> model_set, trans_set;
};

namespace nchmm {

constexpr size_t kQueueWords = 16 + 4096;

void pipe_destroy(nchmm_ctx* c);
void combine_destroy(nchmm_ctx* c);
int pipe_in_flight(const nchmm_ctx* c);
// Size the back-pointer regions for launches of up to `count` reads of up to `longest` events (reallocates only when every
// lane is idle).  budget_share > 0: the part of the workspace budget the pool may take (the rest belongs to the outliers' regions).
int viterbi_ws_prepare(nchmm_ctx* c, uint64_t longest, size_t count, size_t budget_share = 0);
// Queue one launch (sweep + in-block traceback) for reads [first, first + count) on the next lane; *lane_out = that lane.
// The launch starts after `after` (an event, may be null) and after the lane's previous launch.  viterbi_ws_prepare first.
// Emissions ahead (sweep == kSweepAhead): the first `n` reads of the launch's order, `rows` rows of the buffer in total, the longest
// of them `longest` events; d_row0 = per read (indexed by read) its first row or kNoEmRow, null = row of read r's event i is
// off[r] + i (every read of the batch ahead).
struct AheadArgs { size_t n = 0; uint64_t rows = 0, longest = 0; const uint64_t* d_row0 = nullptr; };
uint64_t viterbi_em_budget_rows(nchmm_ctx* c);
// sweep: nchmm::kSweepWide / kSweepLl / kSweepAhead (the caller has decided, nchmm_plan.hpp); the context's sweep_mode overrides it
// when forced (forced "ahead" with ahead == nullptr: plain low-latency).
int launch_viterbi_range(nchmm_ctx* c, hipEvent_t after, size_t first, size_t count, uint64_t ev_count,
                         const uint64_t* d_off, const float* d_cmean, const float* d_stdv, const float* d_lstdv,
                         const int32_t* d_model_slot, const int32_t* d_trans_slot, const uint32_t* d_order, uint16_t* d_out_state,
                         float* d_out_logp, int32_t* d_out_status, int* lane_out, int sweep, const AheadArgs* ahead = nullptr);
int viterbi_big_prepare(nchmm_ctx* c, uint64_t longest, size_t n_long, size_t budget_big);
int launch_viterbi_outliers(nchmm_ctx* c, hipEvent_t after, size_t count, uint64_t ev_count, const uint64_t* d_off, const float* d_cmean,
                            const float* d_stdv, const float* d_lstdv, const int32_t* d_model_slot, const int32_t* d_trans_slot,
                            const uint32_t* d_order, uint16_t* d_out_state, float* d_out_logp, int32_t* d_out_status, int* lane_out, int sweep,
                            const AheadArgs* ahead = nullptr);
// The stream of the lane the NEXT launch_viterbi_range will use (for kernels that must run in front of it).
hipStream_t viterbi_next_lane_stream(nchmm_ctx* c);
// Make `s` wait for everything queued on the lanes; then 1 if a block reported a pool failure.
int viterbi_join(nchmm_ctx* c, hipStream_t s);
int viterbi_check_err(nchmm_ctx* c);
int viterbi_ws_budget(nchmm_ctx* c, size_t* out);
void mask_weights(float p_skip, float p_stay, float wm[64]);   // nchmm_api.cpp

// (the EM lanes' functions: nchmm_internal.hpp -- their user, nchmm_train.cpp, is compiled without HIP)

#define HIP_TRY(ctx, expr)                                  \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                             \
            (ctx)->last_hip = (int)e_;                      \
            return e_ == hipErrorOutOfMemory ? NCHMM_E_NOMEM : NCHMM_E_HIP; \
        }                                                   \
    } while (0)

inline int dev_alloc(nchmm_ctx* c, void** p, size_t bytes)
{
    HIP_TRY(c, hipMalloc(p, bytes));
    c->counters[6] += bytes;
    if (c->counters[6] > c->peak_bytes) c->peak_bytes = c->counters[6];
    return NCHMM_OK;
}

inline int ensure(nchmm_ctx* c, void** p, size_t* have, size_t need)
{
    if (*have >= need) return NCHMM_OK;
    if (*p) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(*p));
        c->counters[6] -= *have;
        *p = nullptr; *have = 0;
    }
    need = need + need / 8;  // head-room so slowly growing batches do not reallocate every call
    int rc = dev_alloc(c, p, need);
    if (rc != NCHMM_OK) return rc;
    *have = need;
    return NCHMM_OK;
}

inline int check_offsets(size_t n, const uint64_t* off, size_t* max_events, size_t* total)
{
    size_t mx = 0;
    if (n && !off) return NCHMM_E_INVALID;
    for (size_t r = 0; r < n; ++r) {
        if (off[r + 1] < off[r]) return NCHMM_E_INVALID;
        mx = std::max<size_t>(mx, off[r + 1] - off[r]);
    }
    *max_events = mx;
    *total = n ? (size_t)(off[n] - off[0]) : 0;
    if (n && off[0] != 0) return NCHMM_E_INVALID;
    return NCHMM_OK;
}


}  // namespace nchmm
#endif
