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
#include <vector>

namespace nchmm {
struct PipeState;
}

struct nchmm_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool external_stream = false;   // stream was set by the caller (0 is then the legacy default stream)
    int last_hip = 0;
    int n_cu = 0;
    int vit_slots = 0;
    int fb_slots = 0;
    float* d_models = nullptr;      // [kMaxSlots][kModelFloats]
    float* d_trans = nullptr;       // [kMaxSlots][kTransFloats]   log-space w0|w1|w2
    float* d_trans_fb = nullptr;    // [kMaxSlots][kFbTransFloats] per-state forward/backward weights for FB
    uint8_t* d_train_mask = nullptr; // [512] transition-training k-mers, one bit per state
    unsigned* d_queue = nullptr;    // forward-backward work-queue heads [1..4], redo count [5]
    unsigned* d_vq = nullptr;       // Viterbi work-queue head [0] and per-(CU, block slot) progress words [16..]
    unsigned vq_base = 0;           // what the queue head will read when the next launch starts (zeroed once, never reset)
    int32_t* d_model_fast = nullptr; // [kMaxSlots]
    unsigned long long* d_prof = nullptr; // [4] phase counters when NCHMM_PROFILE=1
    bool profile = false;
    int tb_margin = 128;            // NCHMM_TB_MARGIN overrides (test hook: 0 forces the re-walk path); profiles/r03_tb_margin.txt:
                                    // 0 of 86 016 speculative segments un-merged at 64, 384 at 32 -- and a miss only costs a re-walk
    uint8_t* d_ws = nullptr;        // viterbi back-pointer workspace (4 KiB per event of a launch)
    size_t ws_bytes = 0;
    hipStream_t s_in = nullptr;     // copy-in stream of the host-pointer pipeline (nchmm_pipeline.cpp); it computes on own_stream
    nchmm::PipeState* pipe = nullptr;   // batches in flight (nchmm_pipeline.cpp)
    size_t peak_bytes = 0;          // high-water mark of counters[6] (device bytes held)
    size_t ws_budget = 0;           // largest workspace we are willing to allocate (bytes)
    size_t fb_budget = 0;           // same for the forward-backward alpha rows (16 KiB per event)
    unsigned* d_last_state = nullptr; // per read: arg-max state of the last column
    size_t last_state_bytes = 0;
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
    hipEvent_t ev_vit0 = nullptr, ev_vit1 = nullptr, ev_vit2 = nullptr, ev_fb0 = nullptr, ev_fb1 = nullptr;
    bool vit_timed = false, fb_timed = false;
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n_slots = 0;                // capacity of the model / transition slot tables (grows on demand)
    std::vector<char> model_set, trans_set;
};

namespace nchmm {

constexpr size_t kQueueWords = 16 + 4096;

// Where a forward + traceback launch pair runs and what it writes into.
struct VitLane {
    hipStream_t stream;
    unsigned* last_state;   // [reads of the batch] arg-max state of the last column, forward -> traceback
};
void pipe_destroy(nchmm_ctx* c);
int pipe_in_flight(const nchmm_ctx* c);
int launch_viterbi_range(nchmm_ctx* c, const VitLane& L, size_t first, size_t count, uint64_t ev_base, uint64_t ev_count,
                         const uint64_t* d_off, const float* d_cmean, const float* d_stdv, const float* d_lstdv,
                         const int32_t* d_model_slot, const int32_t* d_trans_slot, const uint32_t* d_order, uint16_t* d_out_state,
                         float* d_out_logp, int32_t* d_out_status);
int viterbi_ws_budget(nchmm_ctx* c, size_t* out);
void mask_weights(float p_skip, float p_stay, float wm[64]);   // nchmm_api.cpp

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
