// nchmm_pipeline.cpp -- the host-pointer Viterbi entry points (nchmm_viterbi, nchmm_viterbi_raw, and their
// begin / end halves): copy-in on one stream, kernels on three compute lanes taken in turn, up to three batches in flight.
//
// The reference's caller hands over one strand at a time and hides every latency behind pfor worker threads
// (nanocall.cpp:611-621, 645-690).  Here a call carries a batch of reads and the latencies to hide are the PCIe copies
// either side of the kernels (12 B per event in, 2 B out, against ~3 ns of kernel time per event), and -- less
// obviously -- the GPU's own clock management: after an idle gap of 2 ms the next 15 ms forward sweep runs 9 % slower,
// after 20 ms 17 % slower (profiles/r04_hostpath_gap.json).  A caller that wants the device-resident rate from host
// arrays must therefore keep kernels back to back, which one synchronous call per batch cannot do.
//
// What the timelines showed (profiles/r04_pipeline_timeline.md) and what follows from it:
//   * A sweep is resident with every VGPR of every SIMD taken.  Anything the runtime implements as a shader kernel --
//     memsets, copies under 16 KiB, every device-to-host copy -- then waits for a block to exit.  Only SDMA copies
//     (host-to-device of >= 16 KiB) run beside a sweep.
//       - no memset in front of a launch: queue tickets count up across launches (ViterbiArgs::queue_base), the
//         back-pointer regions are taken and returned by the blocks themselves
//       - the small inputs travel as ONE padded block from pinned memory (SDMA), on the copy-in stream
//       - streaming outputs are not copied at all: the kernel writes states / log-probs / status straight into a pinned
//         host block (2 B per event over PCIe, posted writes); `end` moves them into the caller's arrays on the CPU
//   * A launch lasts as long as its longest read and its blocks run out of reads one by one.  Launches do not depend on each
//     other (each block walks its read back itself and owns its back-pointer region meanwhile, viterbi_kernel.hip), so
//     consecutive launches -- ranges of a batch, batches of a stream -- go to the context's three compute lanes in turn and
//     roll into each other.  (Round 3 had a separate traceback kernel per range; a second lane then bought nothing, because
//     the traceback of range k starved until the sweep of range k+1 had drained.)
//   * HIP streams share four hardware queues; a stream that lands on another's queue waits behind its kernels.  Three lanes
//     (lane 0 is the context's own stream) + the copy-in stream are all the pipeline owns.
//
//   begin(k+1) queues the H2D copies of batch k+1 (they run under the kernel of batch k) and its launches behind them;
//   end(k) waits for batch k range by range and hands its results over while batch k+1 computes.
#include "nanocall_hip.h"
#include "nchmm_ctx.hpp"
#include "nchmm_device.h"
#include "nchmm_internal.hpp"
#include "nchmm_pipe.hpp"
#include "nchmm_plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

using namespace nchmm;

namespace nchmm {

constexpr int kPipeDepth = kVitLanes;   // batches in flight: one per compute lane

// One batch in flight.
struct PipeCall {
    bool direct = false;              // one-call form: outputs in device memory, copied straight into the caller's arrays
    size_t n = 0, total = 0;
    std::vector<PipeRange> ranges;
    std::vector<hipEvent_t> done;     // per range: kernels finished
    std::vector<hipEvent_t> ev;       // this slot's events (created once, reused by every batch that takes the slot)
    size_t ev_used = 0;
    std::vector<uint64_t> h_off;
    std::vector<uint32_t> h_order;
    char* d = nullptr;                // this slot's device staging
    char* h = nullptr;                // this slot's pinned host block: [small inputs | logp | status | states]
    size_t h_bytes = 0;
    size_t o_state = 0, o_logp = 0, o_status = 0;     // device staging offsets (direct form)
    size_t ho_logp = 0, ho_status = 0, ho_state = 0;  // pinned block offsets (streaming form)
    uint16_t* out_state = nullptr;
    float* out_logp = nullptr;
    int32_t* out_status = nullptr;
    std::vector<int32_t> status;
};

struct PipeState {
    PipeCall call[kPipeDepth];
    void* d_stage[kPipeDepth] = {};
    size_t stage_bytes[kPipeDepth] = {};
    unsigned next_begin = 0, next_end = 0, in_flight = 0;
};

void pipe_destroy(nchmm_ctx* c)
{
    if (!c->pipe) return;
    for (int s = 0; s < kPipeDepth; ++s) {
        if (c->pipe->d_stage[s]) (void)hipFree(c->pipe->d_stage[s]);
        if (c->pipe->call[s].h) (void)hipHostFree(c->pipe->call[s].h);
        for (hipEvent_t e : c->pipe->call[s].ev) (void)hipEventDestroy(e);
    }
    delete c->pipe;
    c->pipe = nullptr;
}

int pipe_in_flight(const nchmm_ctx* c) { return c->pipe ? (int)c->pipe->in_flight : 0; }

}  // namespace nchmm

namespace {

constexpr size_t kMinCopy = 64 << 10;      // copies below 16 KiB become shader kernels; stay well clear of the threshold

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

int pipe_init(nchmm_ctx* c)
{
    if (!c->pipe) {
        c->pipe = new (std::nothrow) PipeState();
        if (!c->pipe) return NCHMM_E_NOMEM;
    }
    if (!c->s_in) HIP_TRY(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    return NCHMM_OK;
}

int pipe_event(nchmm_ctx* c, PipeCall& K, hipEvent_t* out)
{
    if (K.ev_used == K.ev.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        K.ev.push_back(e);
    }
    *out = K.ev[K.ev_used++];
    return NCHMM_OK;
}

struct PreparedIn { const float* cmean; const float* stdv; const float* lstdv; };
struct RawIn { size_t n_raw; const float* mean; const float* stdv; const float* start; const uint64_t* src; const float* drift; };

// Enqueue one batch.  off = n + 1 packed offsets (host).  direct: the one-call form (nothing else in flight).
int pipe_begin(nchmm_ctx* c, size_t n, const uint64_t* off, size_t total, const int32_t* model_slot, const int32_t* trans_slot,
               const PreparedIn* prep, const RawIn* raw, uint16_t* out_state, float* out_logp, int32_t* out_status, bool direct,
               const PipeTables* tab = nullptr)
{
    int rc = pipe_init(c);
    if (rc != NCHMM_OK) return rc;
    PipeState* P = c->pipe;
    if (P->in_flight >= (unsigned)kPipeDepth || (tab && P->in_flight)) return NCHMM_E_INVALID;   // (a batch with tables rewrites slots 0 .. n-1)
    const unsigned slot = P->next_begin;
    PipeCall& K = P->call[slot];
    K.ev_used = 0;
    hipStream_t si = c->s_in;

    uint64_t longest = 1;
    for (size_t r = 0; r < n; ++r) longest = std::max<uint64_t>(longest, off[r + 1] - off[r]);
    // (the back-pointer workspace is one region per resident block: it does not bound a range)
    // Which form of the sweep (nchmm_plan.hpp).  A batch that goes up alone and whose duration its longest reads would set
    // takes the low-latency form, one read per CU: in one range (one launch hands out the longest reads of the whole batch
    // first) unless it is large.  A streaming caller's batches and the ranges of a large batch run beside each other: those
    // are decided range by range, with the tail of a launch covered by its neighbours.
    const bool alone = (direct || tab) && P->in_flight == 0;
    int batch_sweep = kSweepWide;
    size_t n_ahead = 0;             // leading reads of the (single) range's longest-first order whose emissions are computed ahead
    {
        size_t forced = 0;
        if (const char* e = std::getenv("NCHMM_PIPE_READS")) {      // test hook: ranges of that many reads
            const long v = std::atol(e);
            if (v > 0) forced = (size_t)v;
        }
        if (alone && c->sweep_mode != kSweepWide) {
            std::vector<uint64_t> lens(n);
            for (size_t r = 0; r < n; ++r) lens[r] = off[r + 1] - off[r];
            const uint64_t em_rows = viterbi_em_budget_rows(c);
            if (c->sweep_mode == kSweepAuto) {
                batch_sweep = choose_sweep(lens, (size_t)c->n_cu, (size_t)c->vit_slots, false, rates_at_clock(c->plan_clock_mhz), em_rows, &n_ahead);
            } else {
                // a forced low-latency form: "ahead" takes as many of the longest reads ahead as the buffer holds
                batch_sweep = c->sweep_mode;
                if (batch_sweep == kSweepAhead) {
                    std::vector<uint64_t> desc(lens);
                    std::sort(desc.begin(), desc.end(), std::greater<uint64_t>());
                    uint64_t rows = 0;
                    while (n_ahead < n && n_ahead < kMaxAheadReads && rows + desc[n_ahead] <= em_rows) rows += desc[n_ahead++];
                }
            }
        }
        const bool low_latency = batch_sweep == kSweepLl || batch_sweep == kSweepAhead;
        if (low_latency && !forced && n <= 2 * (size_t)c->vit_slots) forced = n;
        cut_ranges(off, n, low_latency ? (size_t)c->n_cu : (size_t)c->vit_slots, direct, forced, &K.ranges);
        if (K.ranges.size() != 1) n_ahead = 0;      // (emissions ahead: one launch hands out the batch's longest reads first)
    }
    const size_t n_ranges = K.ranges.size();

    // longest-first processing order inside each range (the device work queue hands reads out in this order)
    K.h_off.assign(off, off + n + 1);
    off = K.h_off.data();
    std::vector<uint32_t>& order = K.h_order;
    order_ranges(off, n, K.ranges, &order);
    // Outliers (nchmm_plan.hpp): a few reads too long for a full pool of regions within the budget go through regions of their
    // own as one more launch (launch_viterbi_outliers), and the pool is sized for the rest.
    size_t biggest = 1;
    for (const PipeRange& g : K.ranges) biggest = std::max(biggest, g.r1 - g.r0);
    size_t budget = 0;
    if ((rc = viterbi_ws_budget(c, &budget))) return rc;
    OutlierPlan plan = plan_outliers(off, n, K.ranges, order, (size_t)kXcds * std::min<size_t>(c->slots_per_xcd, biggest * kVitLanes), budget, kBpRowBytes);
    const uint64_t pool_longest = plan.pool_longest;          // what the pool's regions must hold
    const std::vector<uint32_t>& outliers = plan.outliers;    // read indices, longest first
    const std::vector<size_t>& n_out = plan.n_out;            // per range: how many of its (sorted) reads are outliers -- a prefix of its order
    if (!outliers.empty() && (rc = viterbi_big_prepare(c, longest, outliers.size(), plan.budget_big))) return rc;
    if (raw) {
        uint64_t hi = 0;
        for (PipeRange& g : K.ranges) {
            for (size_t v = g.r0; v < g.r1; ++v) hi = std::max<uint64_t>(hi, raw->src[v] + (off[v + 1] - off[v]));
            g.raw_hi = hi;
        }
    }

    // Device staging of this slot:
    //   [ small: off | mslot | tslot | order | src | drift ]    one copy from the pinned block
    //   [ cm | sd | ls | raw mean | stdv | start ]
    //   [ logp | status | state ]                               direct form only
    // Pinned host block:  [ small (as above) | logp | status | state ]   (the outputs: streaming form only)
    const size_t n_raw = raw ? raw->n_raw : 0;
    size_t o_off = 0, o_ms = o_off + al256(8 * (n + 1)), o_ts = o_ms + al256(4 * n), o_or = o_ts + al256(4 * n);
    size_t o_ol = o_or + al256(4 * n);        // the outliers' list
    size_t o_er = o_ol + al256(4 * outliers.size());      // per read: first row of its emissions computed ahead (or kNoEmRow)
    size_t o_src = o_er + (n_ahead ? al256(8 * n) : 0), o_dr = o_src + (raw ? al256(8 * n) : 0);
    const size_t small_bytes = std::max<size_t>(o_dr + (raw ? al256(4 * n) : 0), kMinCopy);
    size_t o_cm = small_bytes, o_sd = o_cm + al256(4 * total), o_ls = o_sd + al256(4 * total);
    size_t o_rm = o_ls + al256(4 * total), o_rs = o_rm + al256(4 * n_raw), o_rt = o_rs + al256(4 * n_raw);
    size_t o_lp = o_rt + al256(4 * n_raw), o_ss = o_lp + al256(4 * n), o_st = o_ss + al256(4 * n);
    const size_t out_end = direct ? o_st + al256(2 * total) : o_lp;
    const size_t ho_lp = small_bytes, ho_ss = ho_lp + al256(4 * n), ho_st = ho_ss + al256(4 * n);
    const size_t h_out_end = direct ? small_bytes : ho_st + al256(2 * total);
    // tables (when the batch builds its own): the loaded models once, then per range one block [idx | par | wm] of its candidates
    std::vector<size_t> tb_off(n_ranges + 1, 0);
    size_t tb_states = 0;
    if (tab) {
        tb_states = al256(sizeof(float) * tab->n_tables * kStates * 10);
        for (size_t k = 0; k < n_ranges; ++k) {
            const size_t m = K.ranges[k].r1 - K.ranges[k].r0;
            tb_off[k + 1] = tb_off[k] + std::max<size_t>(al256(4 * m) + al256(32 * m) + al256(256 * m), kMinCopy);
        }
    }
    const size_t o_tst = out_end, o_tb = o_tst + tb_states, need = o_tb + tb_off[n_ranges];
    const size_t ho_tb = h_out_end, h_need = ho_tb + tb_off[n_ranges];
    // (this slot's buffers: nothing in flight uses them)
    if ((rc = ensure(c, &P->d_stage[slot], &P->stage_bytes[slot], need))) return rc;
    if (K.h_bytes < h_need) {
        if (K.h) { HIP_TRY(c, hipHostFree(K.h)); K.h = nullptr; K.h_bytes = 0; }
        void* hp = nullptr;
        HIP_TRY(c, hipHostMalloc(&hp, h_need + h_need / 8, hipHostMallocDefault));
        K.h = (char*)hp; K.h_bytes = h_need + h_need / 8;
    }
    if ((rc = viterbi_ws_prepare(c, pool_longest, biggest, outliers.empty() ? 0 : budget - plan.budget_big))) return rc;   // (waits for the batch in flight if the regions must grow)
    char* const d = (char*)P->d_stage[slot];
    K.direct = direct; K.n = n; K.total = total; K.d = d;
    K.o_state = o_st; K.o_logp = o_lp; K.o_status = o_ss;
    K.ho_logp = ho_lp; K.ho_status = ho_ss; K.ho_state = ho_st;
    K.out_state = out_state; K.out_logp = out_logp; K.out_status = out_status;
    K.done.assign(n_ranges, nullptr);
    // where the kernels write: device staging (direct) or the pinned block (hipHostMalloc memory is device-visible at its
    // host address)
    uint16_t* const k_state = direct ? (uint16_t*)(d + o_st) : (uint16_t*)(K.h + ho_st);
    float* const k_logp = direct ? (float*)(d + o_lp) : (float*)(K.h + ho_lp);
    int32_t* const k_status = direct ? (int32_t*)(d + o_ss) : (int32_t*)(K.h + ho_ss);

    // the batch starts after whatever the caller queued on the context's stream (table uploads, device-pointer calls)
    // (the context's own stream is lane 0: what is on it are this context's earlier launches -- every other entry point that
    // uses it returns synchronised -- and waiting for those is what the lanes are there to avoid)
    hipEvent_t ev_entry = nullptr;
    if (c->external_stream) {
        if ((rc = pipe_event(c, K, &ev_entry))) return rc;
        HIP_TRY(c, hipEventRecord(ev_entry, c->stream));
    }

    std::memcpy(K.h + o_off, off, 8 * (n + 1));
    std::vector<int32_t> ident;
    if (tab) {
        ident.resize(n);
        std::iota(ident.begin(), ident.end(), 0);
        model_slot = trans_slot = ident.data();
    }
    if (model_slot) std::memcpy(K.h + o_ms, model_slot, 4 * n);
    if (trans_slot) std::memcpy(K.h + o_ts, trans_slot, 4 * n);
    std::memcpy(K.h + o_or, order.data(), 4 * n);
    if (!outliers.empty()) std::memcpy(K.h + o_ol, outliers.data(), 4 * outliers.size());
    AheadArgs ahead;
    if (n_ahead) {
        // with outliers (reads with regions of their own: the longest of all) only those are taken ahead -- one launch reads the
        // buffer at a time
        if (!outliers.empty()) n_ahead = std::min(n_ahead, outliers.size());
        uint64_t* row0 = (uint64_t*)(K.h + o_er);
        for (size_t r = 0; r < n; ++r) row0[r] = kNoEmRow;
        const uint32_t* lead = outliers.empty() ? order.data() : outliers.data();
        for (size_t k = 0; k < n_ahead; ++k) {
            const uint32_t r = lead[k];
            row0[r] = ahead.rows;
            ahead.rows += off[r + 1] - off[r];
            ahead.longest = std::max<uint64_t>(ahead.longest, off[r + 1] - off[r]);
        }
        ahead.n = n_ahead;
        ahead.d_row0 = (const uint64_t*)(d + o_er);
    }
    if (raw) {
        std::memcpy(K.h + o_src, raw->src, 8 * n);
        std::memcpy(K.h + o_dr, raw->drift, 4 * n);
    }
    HIP_TRY(c, hipMemcpyAsync(d, K.h, small_bytes, hipMemcpyHostToDevice, si));
    if (tab) {
        // model / transition slot v belongs to candidate v
        if ((rc = nchmm_reserve_slots(c, (int)n))) return rc;
        HIP_TRY(c, hipMemcpyAsync(d + o_tst, tab->states_Sx10, sizeof(float) * tab->n_tables * kStates * 10, hipMemcpyHostToDevice, si));
        for (size_t v = 0; v < n; ++v) { c->model_set[v] = 1; c->trans_set[v] = 1; }
    }
    uint64_t raw_up = 0;
    std::vector<int> range_lane(n_ranges, 0);
    std::vector<hipEvent_t> range_ev(n_ranges, nullptr);
    for (size_t k = 0; k < n_ranges; ++k) {
        const PipeRange& g = K.ranges[k];
        const size_t ne = (size_t)(g.e1 - g.e0);
        const size_t m = g.r1 - g.r0;
        if (tab) {
            // Pore_Model::scale parameters (Pore_Model.hpp:190-201) and the 64 mask weights of compute_transitions_fast
            // (State_Transitions.hpp:198-224) of this range's candidates: host libm here, expansion on the device
            char* hb = K.h + ho_tb + tb_off[k];
            int32_t* h_idx = (int32_t*)hb;
            float* h_par = (float*)(hb + al256(4 * m));
            float* h_wm = (float*)(hb + al256(4 * m) + al256(32 * m));
            auto fill = [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    const size_t v = g.r0 + i;
                    h_idx[i] = tab->table_idx[v];
                    const float* pp = tab->params_nx6 + 6 * v;
                    std::memcpy(h_par + 8 * i, pp, 6 * sizeof(float));
                    h_par[8 * i + 6] = std::log(pp[3]);   // log_params.var, Pore_Model.hpp:193
                    h_par[8 * i + 7] = std::log(pp[5]);   // log_params.var_sd :195
                    if (i > lo && tab->p_skip[v] == tab->p_skip[v - 1] && tab->p_stay[v] == tab->p_stay[v - 1])
                        std::memcpy(h_wm + 64 * i, h_wm + 64 * (i - 1), 64 * sizeof(float));
                    else
                        mask_weights(tab->p_skip[v], tab->p_stay[v], h_wm + 64 * i);
                }
            };
            if (m >= 256) parallel_for(m, fill); else fill(0, m);
            HIP_TRY(c, hipMemcpyAsync(d + o_tb + tb_off[k], hb, tb_off[k + 1] - tb_off[k], hipMemcpyHostToDevice, si));
        }
        // pageable sources: the runtime pins the pages in place and the SDMA engines read them at PCIe rate; the call
        // returns when the copy is done, so the launches below are queued range by range right behind their data
        if (raw) {
            if (g.raw_hi > raw_up) {
                const size_t mr = (size_t)(g.raw_hi - raw_up);
                HIP_TRY(c, hipMemcpyAsync(d + o_rm + 4 * raw_up, raw->mean + raw_up, 4 * mr, hipMemcpyHostToDevice, si));
                HIP_TRY(c, hipMemcpyAsync(d + o_rs + 4 * raw_up, raw->stdv + raw_up, 4 * mr, hipMemcpyHostToDevice, si));
                HIP_TRY(c, hipMemcpyAsync(d + o_rt + 4 * raw_up, raw->start + raw_up, 4 * mr, hipMemcpyHostToDevice, si));
                raw_up = g.raw_hi;
            }
        } else if (ne) {
            HIP_TRY(c, hipMemcpyAsync(d + o_cm + 4 * g.e0, prep->cmean + g.e0, 4 * ne, hipMemcpyHostToDevice, si));
            HIP_TRY(c, hipMemcpyAsync(d + o_sd + 4 * g.e0, prep->stdv + g.e0, 4 * ne, hipMemcpyHostToDevice, si));
            HIP_TRY(c, hipMemcpyAsync(d + o_ls + 4 * g.e0, prep->lstdv + g.e0, 4 * ne, hipMemcpyHostToDevice, si));
        }
        hipEvent_t ev_in;
        if ((rc = pipe_event(c, K, &ev_in))) return rc;
        HIP_TRY(c, hipEventRecord(ev_in, si));
        // consecutive ranges run on alternating lanes: the blocks of range k+1 start where those of range k run out of reads
        hipStream_t sl = viterbi_next_lane_stream(c);
        HIP_TRY(c, hipStreamWaitEvent(sl, ev_in, 0));
        if (ev_entry && sl != c->stream) HIP_TRY(c, hipStreamWaitEvent(sl, ev_entry, 0));
        if (tab) {
            const char* db = d + o_tb + tb_off[k];
            HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)(c->d_model_fast + g.r0), 1, m, sl));   // the scale kernel clears it for an out-of-range model
            launch_scale_models((const float*)(d + o_tst), (const int32_t*)db, (const float*)(db + al256(4 * m)), c->d_models, c->d_model_fast,
                                (int)g.r0, m, static_cast<float>(std::log(2.0 * M_PI)), sl);
            HIP_TRY(c, hipGetLastError());
            launch_expand_transitions((const float*)(db + al256(4 * m) + al256(32 * m)), c->d_masks, c->d_trans, c->d_trans_fb, (int)g.r0, m, sl);
            HIP_TRY(c, hipGetLastError());
        }
        if (raw) {
            EmGatherArgs ga;
            ga.mean = (const float*)(d + o_rm); ga.stdv = (const float*)(d + o_rs); ga.start = (const float*)(d + o_rt); ga.lstdv = nullptr;
            ga.win_src = (const uint64_t*)(d + o_src) + g.r0; ga.off = (const uint64_t*)(d + o_off) + g.r0;
            ga.win_drift = (const float*)(d + o_dr) + g.r0;
            ga.cmean = (float*)(d + o_cm); ga.out_stdv = (float*)(d + o_sd); ga.out_lstdv = (float*)(d + o_ls);
            launch_em_gather(ga, (unsigned)(g.r1 - g.r0), sl, (unsigned)g.max_events);
            HIP_TRY(c, hipGetLastError());
        }
        if (!outliers.empty()) {
            // what the outliers' launch waits for: this range's copy-in, gather and tables (not its launch)
            if ((rc = pipe_event(c, K, &range_ev[k]))) return rc;
            HIP_TRY(c, hipEventRecord(range_ev[k], sl));
        }
        int lane = (int)(k % kVitLanes);
        if (g.r1 - g.r0 > n_out[k]) {
            int sweep = batch_sweep;
            if (!alone && c->sweep_mode == kSweepAuto) {
                std::vector<uint64_t> lens(g.r1 - g.r0 - n_out[k]);
                for (size_t i = 0; i < lens.size(); ++i) { const uint32_t r = order[g.r0 + n_out[k] + i]; lens[i] = off[r + 1] - off[r]; }
                sweep = choose_sweep(lens, (size_t)c->n_cu, (size_t)c->vit_slots, true);
            }
            rc = launch_viterbi_range(c, nullptr, g.r0, g.r1 - g.r0 - n_out[k], g.e1 - g.e0,
                                      (const uint64_t*)(d + o_off), (const float*)(d + o_cm), (const float*)(d + o_sd),
                                      (const float*)(d + o_ls), model_slot ? (const int32_t*)(d + o_ms) : nullptr,
                                      trans_slot ? (const int32_t*)(d + o_ts) : nullptr, (const uint32_t*)(d + o_or) + g.r0 + n_out[k], k_state, k_logp,
                                      k_status, &lane, sweep, (ahead.n && outliers.empty()) ? &ahead : nullptr);
            if (rc != NCHMM_OK) return rc;
        } else {
            // (a range of outliers only: what was queued in front of its launch -- tables, gather -- is on stream sl)
            for (int l = 0; l < kVitLanes; ++l) if (c->lane[l].stream == sl) lane = l;
        }
        range_lane[k] = lane;
        if (outliers.empty()) {
            if ((rc = pipe_event(c, K, &K.done[k]))) return rc;
            HIP_TRY(c, hipEventRecord(K.done[k], c->lane[lane].stream));
        }
    }
    if (!outliers.empty()) {
        // one more launch for the outliers of all ranges, then the ranges' completion events behind it (a range is complete when
        // its outliers are)
        hipStream_t so = viterbi_next_lane_stream(c);
        for (size_t k = 0; k < n_ranges; ++k) HIP_TRY(c, hipStreamWaitEvent(so, range_ev[k], 0));   // (the outliers' events, gathered or copied in, and tables)
        int lane_o = 0;
        uint64_t ev_out = 0;
        for (uint32_t r : outliers) ev_out += off[r + 1] - off[r];
        rc = launch_viterbi_outliers(c, nullptr, outliers.size(), ev_out, (const uint64_t*)(d + o_off), (const float*)(d + o_cm), (const float*)(d + o_sd),
                                     (const float*)(d + o_ls), model_slot ? (const int32_t*)(d + o_ms) : nullptr,
                                     trans_slot ? (const int32_t*)(d + o_ts) : nullptr, (const uint32_t*)(d + o_ol), k_state, k_logp, k_status, &lane_o, batch_sweep, ahead.n ? &ahead : nullptr);
        if (rc != NCHMM_OK) return rc;
        hipEvent_t ev_o;
        if ((rc = pipe_event(c, K, &ev_o))) return rc;
        HIP_TRY(c, hipEventRecord(ev_o, c->lane[lane_o].stream));
        for (size_t k = 0; k < n_ranges; ++k) {
            hipStream_t sk = c->lane[range_lane[k]].stream;
            if (n_out[k]) HIP_TRY(c, hipStreamWaitEvent(sk, ev_o, 0));
            if ((rc = pipe_event(c, K, &K.done[k]))) return rc;
            HIP_TRY(c, hipEventRecord(K.done[k], sk));
        }
    }
    c->counters[0] += n;
    c->counters[1] += total;
    P->next_begin = (P->next_begin + 1u) % (unsigned)kPipeDepth;
    P->in_flight += 1;
    return NCHMM_OK;
}

}  // namespace

namespace nchmm {

size_t pipe_n_ranges(const nchmm_ctx* c)
{
    const PipeState* P = c->pipe;
    return P && P->in_flight ? P->call[P->next_end].ranges.size() : 0;
}

int pipe_wait_range(nchmm_ctx* c, size_t k, size_t* r0, size_t* r1)
{
    PipeState* P = c->pipe;
    if (!P || P->in_flight == 0) return NCHMM_E_INVALID;
    PipeCall& K = P->call[P->next_end];
    if (k >= K.ranges.size() || K.direct) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(K.done[k]));
    *r0 = K.ranges[k].r0; *r1 = K.ranges[k].r1;
    return NCHMM_OK;
}

void pipe_results(const nchmm_ctx* c, const uint16_t** states, const float** logp, const int32_t** status)
{
    const PipeCall& K = c->pipe->call[c->pipe->next_end];
    *states = (const uint16_t*)(K.h + K.ho_state);
    *logp = (const float*)(K.h + K.ho_logp);
    *status = (const int32_t*)(K.h + K.ho_status);
}

int pipe_release(nchmm_ctx* c)
{
    PipeState* P = c->pipe;
    if (!P || P->in_flight == 0) return NCHMM_E_INVALID;
    PipeCall& K = P->call[P->next_end];
    HIP_TRY(c, hipSetDevice(c->device));
    hipError_t e = hipSuccess;
    for (hipEvent_t d : K.done)      // (consecutive ranges sit on different lanes: the last one to be queued need not be the last to finish)
        if (e == hipSuccess) e = hipEventSynchronize(d);
    P->next_end = (P->next_end + 1u) % (unsigned)kPipeDepth;
    P->in_flight -= 1;
    HIP_TRY(c, e);
    return viterbi_check_err(c);      // a block that found no back-pointer region: NCHMM_E_HIP here, not a stale flag for the next call
}

}  // namespace nchmm

namespace {

// Hand over the oldest batch in flight.
int pipe_end(nchmm_ctx* c)
{
    PipeState* P = c->pipe;
    if (!P || P->in_flight == 0) return NCHMM_E_INVALID;
    PipeCall& K = P->call[P->next_end];
    hipStream_t sr = c->own_stream;
    auto body = [&]() -> int {
        K.status.resize(K.n);
        if (K.direct) {
            // nothing is queued behind a lone batch, so the runtime's shader copies into the caller's pageable arrays start
            // at once (they hold this thread until they are done)
            for (hipEvent_t e : K.done) HIP_TRY(c, hipStreamWaitEvent(sr, e, 0));
            HIP_TRY(c, hipMemcpyAsync(K.out_logp, K.d + K.o_logp, 4 * K.n, hipMemcpyDeviceToHost, sr));
            HIP_TRY(c, hipMemcpyAsync(K.status.data(), K.d + K.o_status, 4 * K.n, hipMemcpyDeviceToHost, sr));
            if (K.total) HIP_TRY(c, hipMemcpyAsync(K.out_state, K.d + K.o_state, 2 * K.total, hipMemcpyDeviceToHost, sr));
            HIP_TRY(c, hipStreamSynchronize(sr));
            return NCHMM_OK;
        }
        // range by range as each finishes: pinned block -> the caller's arrays on this thread, under the kernels of the
        // ranges and the batch behind it
        for (size_t k = 0; k < K.ranges.size(); ++k) {
            const PipeRange& g = K.ranges[k];
            HIP_TRY(c, hipEventSynchronize(K.done[k]));
            if (g.e1 > g.e0) std::memcpy(K.out_state + g.e0, K.h + K.ho_state + 2 * g.e0, 2 * (size_t)(g.e1 - g.e0));
            std::memcpy(K.out_logp + g.r0, K.h + K.ho_logp + 4 * g.r0, 4 * (g.r1 - g.r0));
            std::memcpy(K.status.data() + g.r0, K.h + K.ho_status + 4 * g.r0, 4 * (g.r1 - g.r0));
        }
        return NCHMM_OK;
    };
    int rc = body();
    if (rc == NCHMM_OK) rc = viterbi_check_err(c);
    if (rc != NCHMM_OK) {
        // leave nothing of this batch running behind the caller's back
        (void)hipStreamSynchronize(c->s_in);
        for (int l = 0; l < kVitLanes; ++l) (void)hipStreamSynchronize(c->lane[l].stream);
    }
    P->next_end = (P->next_end + 1u) % (unsigned)kPipeDepth;
    P->in_flight -= 1;
    if (rc != NCHMM_OK) return rc;
    int worst = NCHMM_OK;
    for (size_t r = 0; r < K.n; ++r) {
        if (K.out_status) K.out_status[r] = K.status[r];
        if (K.status[r] != 0) worst = NCHMM_E_NUMERIC;
    }
    return worst;
}

int check_slots(const nchmm_ctx* c, size_t n, const int32_t* model_slot, const int32_t* trans_slot)
{
    for (size_t r = 0; r < n; ++r) {
        const int ms = model_slot ? model_slot[r] : 0, ts = trans_slot ? trans_slot[r] : 0;
        if (ms < 0 || ms >= c->n_slots || ts < 0 || ts >= c->n_slots || !c->model_set[ms] || !c->trans_set[ts]) return NCHMM_E_INVALID;
    }
    return NCHMM_OK;
}

// a begin that failed half-way may have queued work: drain it so the slot's buffers can be reused
int fail_drain(nchmm_ctx* c, int rc)
{
    if (c->s_in) (void)hipStreamSynchronize(c->s_in);
    for (int l = 0; l < kVitLanes; ++l) (void)hipStreamSynchronize(c->lane[l].stream);
    return rc;
}

int begin_prepared(nchmm_ctx* c, size_t n_reads, const uint64_t* off, const float* cmean, const float* stdv, const float* lstdv,
                   const int32_t* model_slot, const int32_t* trans_slot, uint16_t* out_state, float* out_logp, int32_t* out_status,
                   bool direct)
{
    if (!c || n_reads == 0 || n_reads > 0xFFFFFFF0ull) return NCHMM_E_INVALID;
    size_t max_events = 0, total = 0;
    int rc = check_offsets(n_reads, off, &max_events, &total);
    if (rc != NCHMM_OK) return rc;
    if (max_events > 0x7FFFFFF0ull) return NCHMM_E_INVALID;
    if (!out_logp || (total && (!cmean || !stdv || !lstdv || !out_state))) return NCHMM_E_INVALID;
    if ((rc = check_slots(c, n_reads, model_slot, trans_slot))) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const PreparedIn in{cmean, stdv, lstdv};
    rc = pipe_begin(c, n_reads, off, total, model_slot, trans_slot, &in, nullptr, out_state, out_logp, out_status, direct);
    return rc == NCHMM_OK ? rc : fail_drain(c, rc);
}

int begin_raw(nchmm_ctx* c, size_t n_raw, const float* mean, const float* stdv, const float* start, size_t n_cand, const uint64_t* src,
              const uint32_t* len, const float* drift, const int32_t* model_slot, const int32_t* trans_slot, uint16_t* out_state,
              float* out_logp, int32_t* out_status, bool direct)
{
    if (!c || n_cand == 0 || n_cand > 0xFFFFFFF0ull) return NCHMM_E_INVALID;
    if (!src || !len || !drift || !out_logp || (n_raw && (!mean || !stdv || !start))) return NCHMM_E_INVALID;
    std::vector<uint64_t> off(n_cand + 1, 0);
    for (size_t v = 0; v < n_cand; ++v) {
        if (src[v] > n_raw || len[v] > n_raw - src[v]) return NCHMM_E_INVALID;
        off[v + 1] = off[v] + len[v];
    }
    const size_t total = (size_t)off[n_cand];
    if (total && !out_state) return NCHMM_E_INVALID;
    int rc = check_slots(c, n_cand, model_slot, trans_slot);
    if (rc != NCHMM_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const RawIn in{n_raw, mean, stdv, start, src, drift};
    rc = pipe_begin(c, n_cand, off.data(), total, model_slot, trans_slot, nullptr, &in, out_state, out_logp, out_status, direct);
    return rc == NCHMM_OK ? rc : fail_drain(c, rc);
}

}  // namespace

int nchmm::pipe_raw_tables_begin(nchmm_ctx* c, size_t n_raw, const float* mean, const float* stdv, const float* start, size_t n_cand,
                                 const uint64_t* src, const uint32_t* len, const float* drift, const PipeTables& tab)
{
    if (!c || n_cand == 0 || n_cand > 0x7FFFFFF0ull) return NCHMM_E_INVALID;
    if (!src || !len || !drift || (n_raw && (!mean || !stdv || !start))) return NCHMM_E_INVALID;
    if (!tab.states_Sx10 || !tab.table_idx || !tab.params_nx6 || !tab.p_skip || !tab.p_stay || tab.n_tables == 0) return NCHMM_E_INVALID;
    std::vector<uint64_t> off(n_cand + 1, 0);
    for (size_t v = 0; v < n_cand; ++v) {
        if (src[v] > n_raw || len[v] > n_raw - src[v] || tab.table_idx[v] < 0 || (size_t)tab.table_idx[v] >= tab.n_tables) return NCHMM_E_INVALID;
        off[v + 1] = off[v] + len[v];
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const RawIn in{n_raw, mean, stdv, start, src, drift};
    const int rc = pipe_begin(c, n_cand, off.data(), (size_t)off[n_cand], nullptr, nullptr, nullptr, &in, nullptr, nullptr, nullptr, false, &tab);
    return rc == NCHMM_OK ? rc : fail_drain(c, rc);
}

extern "C" {

int nchmm_viterbi_begin(nchmm_ctx* c, size_t n_reads, const uint64_t* off, const float* cmean, const float* stdv,
                        const float* lstdv, const int32_t* model_slot, const int32_t* trans_slot, uint16_t* out_state,
                        float* out_logp, int32_t* out_status)
{
    return begin_prepared(c, n_reads, off, cmean, stdv, lstdv, model_slot, trans_slot, out_state, out_logp, out_status, false);
}

int nchmm_viterbi_raw_begin(nchmm_ctx* c, size_t n_raw, const float* mean, const float* stdv, const float* start, size_t n_cand,
                            const uint64_t* src, const uint32_t* len, const float* drift, const int32_t* model_slot,
                            const int32_t* trans_slot, uint16_t* out_state, float* out_logp, int32_t* out_status)
{
    return begin_raw(c, n_raw, mean, stdv, start, n_cand, src, len, drift, model_slot, trans_slot, out_state, out_logp, out_status, false);
}

int nchmm_viterbi_end(nchmm_ctx* c)
{
    if (!c) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    return pipe_end(c);
}

int nchmm_viterbi_in_flight(const nchmm_ctx* c) { return c ? pipe_in_flight(c) : 0; }

int nchmm_viterbi(nchmm_ctx* c, size_t n_reads, const uint64_t* off, const float* cmean, const float* stdv,
                  const float* lstdv, const int32_t* model_slot, const int32_t* trans_slot,
                  uint16_t* out_state, float* out_logp, int32_t* out_status)
{
    if (!c) return NCHMM_E_INVALID;
    if (n_reads == 0) return NCHMM_OK;
    if (pipe_in_flight(c)) return NCHMM_E_INVALID;   // the one-call form returns THIS batch's results
    const int rc = begin_prepared(c, n_reads, off, cmean, stdv, lstdv, model_slot, trans_slot, out_state, out_logp, out_status, true);
    return rc == NCHMM_OK ? nchmm_viterbi_end(c) : rc;
}

int nchmm_viterbi_raw(nchmm_ctx* c, size_t n_raw, const float* mean, const float* stdv, const float* start, size_t n_cand,
                      const uint64_t* src, const uint32_t* len, const float* drift, const int32_t* model_slot,
                      const int32_t* trans_slot, uint16_t* out_state, float* out_logp, int32_t* out_status)
{
    if (!c) return NCHMM_E_INVALID;
    if (n_cand == 0) return NCHMM_OK;
    if (pipe_in_flight(c)) return NCHMM_E_INVALID;
    const int rc = begin_raw(c, n_raw, mean, stdv, start, n_cand, src, len, drift, model_slot, trans_slot, out_state, out_logp,
                             out_status, true);
    return rc == NCHMM_OK ? nchmm_viterbi_end(c) : rc;
}

}  // extern "C"
