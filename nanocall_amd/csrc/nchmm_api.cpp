// nchmm_api.cpp -- C-ABI device layer: context, model/transition upload, batched launches.
// Compiled with hipcc; see include/nanocall_hip.h for the contract of every entry point.
#include "nanocall_hip.h"
#include "nchmm_ctx.hpp"
#include "nchmm_device.h"
#include "nchmm_internal.hpp"
#include "nchmm_kmer.hpp"
#include "nchmm_plan.hpp"
#include "nchmm_probe.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <thread>
#include <vector>

using namespace nchmm;

namespace {

// Split the from_v CSR of compute_transitions_fast into w0[j] (stay), w1[r] (step group r = low 10
// bits of the predecessor) and w2[q] (skip group q = low 8 bits of the predecessor).
//
// Why this is exact: get_trans_prob (State_Transitions.hpp:125-144) is a function of the overlap
// mask m(p,j) = [p==j] | sum_l [suffix(p,6-l)==prefix(j,6-l)] << l.  For a skip arc bits 3..5 of m
// depend only on q, for a step arc bits 2..5 depend only on r; bit 0 (p==j) and bit 1 (p is also a
// step predecessor) are the only per-arc extras, and an arc that has them is ALSO a member of the
// more specific class (stay, or step) where it carries its full weight.  In the looser class it is
// evaluated with a weight that is <= its true weight, so it can never win there with a larger
// value, and on an exact tie it names the same predecessor index.  We verify all of this on the
// actual numbers instead of trusting the algebra; anything else is NCHMM_E_TOPOLOGY.
int factor_transitions(const uint32_t* row_ptr, const uint16_t* pred, const float* logw, float* out)
{
    float* w0 = out; float* w1 = out + kStates; float* w2 = out + kStates + 1024;
    std::vector<char> have1(1024, 0), have2(256, 0);
    if (row_ptr[0] != 0) return NCHMM_E_TOPOLOGY;
    for (unsigned j = 0; j < (unsigned)kStates; ++j) {
        unsigned cand[21], m = 0;
        cand[m++] = j;
        for (unsigned x = 0; x < 4; ++x) cand[m++] = Kmer6::step_pred(j, x);
        for (unsigned xy = 0; xy < 16; ++xy) cand[m++] = Kmer6::skip_pred(j, xy);
        std::sort(cand, cand + m);
        m = (unsigned)(std::unique(cand, cand + m) - cand);
        const uint32_t b = row_ptr[j], e = row_ptr[j + 1];
        if (e < b || e - b != m || e > (uint32_t)NCHMM_MAX_ARCS) return NCHMM_E_TOPOLOGY;
        for (unsigned k = 0; k < m; ++k) {
            if (pred[b + k] != cand[k]) return NCHMM_E_TOPOLOGY;
            const unsigned p = cand[k];
            const float w = logw[b + k];
            if (std::isnan(w)) return NCHMM_E_TOPOLOGY;
            const bool is_stay = p == j;
            const bool is_step = (p & 1023u) == (j >> 2);
            if (is_stay) { w0[j] = w; continue; }
            if (is_step) {
                const unsigned r = p & 1023u;
                if (!have1[r]) { w1[r] = w; have1[r] = 1; }
                else if (std::memcmp(&w1[r], &w, 4) != 0) return NCHMM_E_TOPOLOGY;
                continue;
            }
            const unsigned q = p & 255u;
            if (!have2[q]) { w2[q] = w; have2[q] = 1; }
            else if (std::memcmp(&w2[q], &w, 4) != 0) return NCHMM_E_TOPOLOGY;
        }
    }
    for (unsigned r = 0; r < 1024; ++r) if (!have1[r]) return NCHMM_E_TOPOLOGY;
    for (unsigned q = 0; q < 256; ++q) if (!have2[q]) return NCHMM_E_TOPOLOGY;
    // arcs evaluated in a looser class must not be over-weighted there
    for (unsigned j = 0; j < (unsigned)kStates; ++j) {
        const uint32_t b = row_ptr[j], e = row_ptr[j + 1];
        for (uint32_t a = b; a < e; ++a) {
            const unsigned p = pred[a];
            const float w = logw[a];
            const bool is_stay = p == j;
            const bool in_step = (p & 1023u) == (j >> 2);
            const bool in_skip = (p & 255u) == (j >> 4);
            if (is_stay && in_step && !(w1[p & 1023u] <= w)) return NCHMM_E_TOPOLOGY;
            if ((is_stay || in_step) && in_skip && !(w2[p & 255u] <= w)) return NCHMM_E_TOPOLOGY;
        }
    }
    return NCHMM_OK;
}

// Per-state weights for the sum-product recursions (log space).  With the arcs factorised as
// stay w0[j] / step group w1[r] / skip group w2[q], a state's incoming probability mass is
//   T0 a[j] + W1 S1[j>>2] + W2 S2[j>>4]   (S1, S2 = sums over the 4 / 16 group members)
// except that for 28 low-complexity k-mers some predecessor is a member of two classes and must be
// counted once, with its most specific weight: fold the correction into the coefficients
//   c2 = W2,  c1 = W1 - [step preds are skip preds] W2,
//   c0 = T0 - [j is its own step pred] W1 - [j is its own skip pred but not step pred] W2.
// The outgoing (backward) direction has the same form with the groups taken over successors.
void fb_weights(const float* w, float* out)
{
    const float* w0 = w; const float* w1 = w + kStates; const float* w2 = w + kStates + 1024;
    auto fold = [](float lw0, float lw1, float lw2, bool stay_in_step, bool stay_in_skip, bool step_in_skip, float* c0,
                   float* c1, float* c2) {
        if (!stay_in_step && !stay_in_skip && !step_in_skip) {   // the common case: nothing to fold
            *c0 = lw0; *c1 = lw1; *c2 = lw2;
            return;
        }
        const double T0 = std::exp((double)lw0), W1 = std::exp((double)lw1), W2 = std::exp((double)lw2);
        *c0 = (float)std::log(T0 - (stay_in_step ? W1 : 0.0) - (stay_in_skip && !stay_in_step ? W2 : 0.0));
        *c1 = (float)std::log(W1 - (step_in_skip ? W2 : 0.0));
        *c2 = lw2;
    };
    for (unsigned j = 0; j < (unsigned)kStates; ++j) {
        // forward: groups of predecessors, indexed by the consumer's high bits
        fold(w0[j], w1[j >> 2], w2[j >> 4], (j & 1023u) == (j >> 2), (j & 255u) == (j >> 4), ((j >> 2) & 255u) == (j >> 4),
             &out[0 * kStates + j], &out[1 * kStates + j], &out[2 * kStates + j]);
        // backward: groups of successors, indexed by the source's low bits
        fold(w0[j], w1[j & 1023u], w2[j & 255u], (j >> 2) == (j & 1023u), (j >> 4) == (j & 255u),
             ((j & 1023u) >> 2) == (j & 255u), &out[3 * kStates + j], &out[4 * kStates + j], &out[5 * kStates + j]);
    }
}

// One state of the device model image (nchmm_device.h ModelField) from the six emission fields.
// The kernel divides by sigma, eta (per state) and stdv (per event) through correctly rounded
// reciprocals + two FMA residual steps; that equals IEEE division when no intermediate leaves the
// normal range, which holds for parameters inside these (very wide) bounds.  A model outside them is
// still decoded, with true divisions (*fast = 0).
inline void model_image_row(float* img, int j, float level_mean, float level_stdv, float log_level_stdv, float sd_mean,
                            float sd_lambda, float log_sd_lambda, int32_t* fast)
{
    static const float log_2pi = static_cast<float>(std::log(2.0 * M_PI));  // Pore_Model.hpp:28,37
    auto in = [](float v, float lo, float hi) { return std::isfinite(v) && v >= lo && v <= hi; };
    img[MF_MU * kStates + j] = level_mean;
    img[MF_SIGMA * kStates + j] = level_stdv;
    img[MF_RSIGMA * kStates + j] = static_cast<float>(1.0 / static_cast<double>(level_stdv));
    img[MF_NEG_LOG_SIGMA * kStates + j] = -log_level_stdv;
    img[MF_ETA * kStates + j] = sd_mean;
    img[MF_RETA * kStates + j] = static_cast<float>(1.0 / static_cast<double>(sd_mean));
    img[MF_LAMBDA * kStates + j] = sd_lambda;
    img[MF_C * kStates + j] = log_sd_lambda - log_2pi;  // first subtraction of log_invgauss_pdf, Pore_Model.hpp:39
    if (!(in(std::fabs(level_mean), 0x1p-10f, 0x1p20f) && in(level_stdv, 0x1p-10f, 0x1p10f) && in(sd_mean, 0x1p-6f, 0x1p10f)
          && in(sd_lambda, 0x1p-10f, 0x1p14f)))
        *fast = 0;
}

// w0 | w1 | w2 of compute_transitions_fast(p_skip, p_stay).  get_trans_prob is a function of the 6-bit
// overlap mask of (i, j) only, so one representative arc per class gives the class's mask, and the
// weight is evaluated once per distinct mask (18 of them) instead of once per arc.
struct MaskTables {
    uint8_t m0[kStates], m1[1024], m2[256];   // overlap mask of the stay arc / step-group arc / skip-group arc
    uint16_t rep_i[64], rep_j[64];            // one arc (i -> j) having each mask
    bool used[64];
    MaskTables()
    {
        std::memset(used, 0, sizeof(used));
        auto mask = [](unsigned i, unsigned j) {
            unsigned m = i == j ? 1u : 0u;
            for (unsigned l = 1; l < 6; ++l)
                if (Kmer6::suffix(i, 6 - l) == Kmer6::prefix(j, 6 - l)) m |= 1u << l;
            return m;
        };
        auto note = [&](unsigned i, unsigned j) {
            const unsigned m = mask(i, j);
            if (!used[m]) { used[m] = true; rep_i[m] = (uint16_t)i; rep_j[m] = (uint16_t)j; }
            return (uint8_t)m;
        };
        for (unsigned j = 0; j < (unsigned)kStates; ++j) m0[j] = note(j, j);
        for (unsigned r = 0; r < 1024; ++r) {
            // a step arc p -> j with p & 1023 == r == j >> 2 that is not the stay arc
            unsigned p = r, j = r << 2;            // x = 0, a = 0
            if (p == j) j |= 1u;                   // r == 0: p = j = 0
            m1[r] = note(p, j);
        }
        for (unsigned q = 0; q < 256; ++q) {
            // a skip arc p -> j with p & 255 == q == j >> 4 that is neither a step arc nor the stay arc
            bool found = false;
            for (unsigned xy = 0; xy < 16 && !found; ++xy)
                for (unsigned ab = 0; ab < 16 && !found; ++ab) {
                    const unsigned p = (xy << 8) | q, j = (q << 4) | ab;
                    if (p == j || (p & 1023u) == (j >> 2)) continue;
                    m2[q] = note(p, j);
                    found = true;
                }
        }
    }
};

const MaskTables& mask_tables()
{
    static const MaskTables T;
    return T;
}

}  // namespace

// log weight of every overlap mask that occurs, for compute_transitions_fast(p_skip, p_stay)
void nchmm::mask_weights(float p_skip, float p_stay, float wm[64])
{
    const MaskTables& T = mask_tables();
    float p_step, p_skip_1;
    step_params(p_skip, p_stay, p_step, p_skip_1);
    const TransPow pw(p_skip_1);
    for (unsigned m = 0; m < 64; ++m)
        wm[m] = T.used[m] ? std::log(trans_prob(T.rep_i[m], T.rep_j[m], p_stay, p_step, pw)) : 0.0f;
}

namespace {

int pinned(nchmm_ctx* c, size_t bytes, void** out)
{
    if (c->em_async) {
        // work queued on this lane still reads what was handed out before: piecewise, from the arena em_lanes_prepare sized
        const size_t a = (bytes + 255) & ~(size_t)255;
        if (c->pin_cursor + a > c->h_pin_bytes) return NCHMM_E_INVALID;
        *out = (char*)c->h_pin + c->pin_cursor;
        c->pin_cursor += a;
        return NCHMM_OK;
    }
    if (c->h_pin_bytes < bytes) {
        if (c->h_pin) { HIP_TRY(c, hipHostFree(c->h_pin)); c->h_pin = nullptr; c->h_pin_bytes = 0; }
        bytes += bytes / 8;
        HIP_TRY(c, hipHostMalloc(&c->h_pin, bytes, hipHostMallocDefault));
        c->h_pin_bytes = bytes;
    }
    *out = c->h_pin;
    return NCHMM_OK;
}

// Grow the slot tables (models, Viterbi weights, FB weights, fast flags) to hold n slots, keeping
// what is already uploaded.
int reserve_slots(nchmm_ctx* c, int n)
{
    if (n <= c->n_slots) return NCHMM_OK;
    if (c->own_stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int l = 0; l < kVitLanes; ++l)      // a launch on a lane may still be reading the tables that are about to move
        if (c->lane[l].stream) HIP_TRY(c, hipStreamSynchronize(c->lane[l].stream));
    auto grow = [&](void** p, size_t elem_bytes) -> int {
        void* q = nullptr;
        int rc = dev_alloc(c, &q, elem_bytes * (size_t)n);
        if (rc != NCHMM_OK) return rc;
        if (*p) {
            HIP_TRY(c, hipMemcpy(q, *p, elem_bytes * (size_t)c->n_slots, hipMemcpyDeviceToDevice));
            HIP_TRY(c, hipFree(*p));
            c->counters[6] -= elem_bytes * (size_t)c->n_slots;
        }
        *p = q;
        return NCHMM_OK;
    };
    int rc;
    if ((rc = grow((void**)&c->d_models, sizeof(float) * kModelFloats))) return rc;
    if ((rc = grow((void**)&c->d_trans, sizeof(float) * kTransFloats))) return rc;
    if ((rc = grow((void**)&c->d_trans_fb, sizeof(float) * kFbTransFloats))) return rc;
    if ((rc = grow((void**)&c->d_model_fast, sizeof(int32_t)))) return rc;
    HIP_TRY(c, hipMemset(c->d_model_fast + c->n_slots, 0, sizeof(int32_t) * (size_t)(n - c->n_slots)));
    c->model_set.resize(n, 0); c->trans_set.resize(n, 0);
    c->n_slots = n;
    return NCHMM_OK;
}

}  // namespace

extern "C" {

int nchmm_create(nchmm_ctx** out, int device_id)
{
    if (!out) return NCHMM_E_INVALID;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return NCHMM_E_NO_DEVICE;
    if (device_id < 0 || device_id >= n_dev) return NCHMM_E_NO_DEVICE;
    nchmm_ctx* c = new (std::nothrow) nchmm_ctx();
    if (!c) return NCHMM_E_NOMEM;
    c->device = device_id;
    int rc = NCHMM_OK;
    auto fail = [&](int code) { nchmm_destroy(c); return code; };
    if (hipSetDevice(device_id) != hipSuccess) return fail(NCHMM_E_NO_DEVICE);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(NCHMM_E_NO_DEVICE);
    c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(NCHMM_E_HIP);
    c->stream = c->own_stream;
    c->lane[0].stream = c->own_stream;
    for (int l = 1; l < kVitLanes; ++l)
        if (hipStreamCreateWithFlags(&c->lane[l].stream, hipStreamNonBlocking) != hipSuccess) return fail(NCHMM_E_HIP);
    for (int l = 0; l < kVitLanes; ++l)
        if (hipEventCreate(&c->lane[l].ev0) != hipSuccess || hipEventCreate(&c->lane[l].ev1) != hipSuccess
            || hipEventCreateWithFlags(&c->lane[l].done, hipEventDisableTiming) != hipSuccess)
            return fail(NCHMM_E_HIP);
    if (hipEventCreateWithFlags(&c->ev_entry, hipEventDisableTiming) != hipSuccess) return fail(NCHMM_E_HIP);
    if (hipEventCreateWithFlags(&c->ev_big, hipEventDisableTiming) != hipSuccess) return fail(NCHMM_E_HIP);
    if (hipEventCreateWithFlags(&c->ev_em, hipEventDisableTiming) != hipSuccess) return fail(NCHMM_E_HIP);
    {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocDefault) != hipSuccess) return fail(NCHMM_E_HIP);
        c->h_err = (unsigned*)hp;
        *c->h_err = 0;
    }
    if ((rc = reserve_slots(c, kMaxSlots))) return fail(rc);
    if ((rc = dev_alloc(c, (void**)&c->d_train_mask, 512))) return fail(rc);
    if ((rc = dev_alloc(c, (void**)&c->d_masks, kTransFloats))) return fail(rc);
    {
        const MaskTables& T = mask_tables();
        std::vector<uint8_t> mk(kTransFloats);
        std::memcpy(mk.data(), T.m0, kStates); std::memcpy(mk.data() + kStates, T.m1, 1024); std::memcpy(mk.data() + kStates + 1024, T.m2, 256);
        if (hipMemcpy(c->d_masks, mk.data(), kTransFloats, hipMemcpyHostToDevice) != hipSuccess) return fail(NCHMM_E_HIP);
    }
    {
        // Parameter_Trainer::init, Parameter_Trainer.hpp:30-57
        std::vector<uint16_t> km(kStates);
        uint32_t nk = 0;
        nchmm_st_train_kmers(km.data(), &nk);
        std::vector<uint8_t> mask(512, 0);
        for (uint32_t i = 0; i < nk; ++i) mask[km[i] >> 3] |= (uint8_t)(1u << (km[i] & 7));
        if (hipMemcpy(c->d_train_mask, mask.data(), 512, hipMemcpyHostToDevice) != hipSuccess) return fail(NCHMM_E_HIP);
    }
    if ((rc = dev_alloc(c, (void**)&c->d_queue, sizeof(unsigned) * 16))) return fail(rc);
    if ((rc = dev_alloc(c, (void**)&c->d_vq, sizeof(unsigned) * kQueueWords))) return fail(rc);
    if (hipMemset(c->d_vq, 0, sizeof(unsigned) * kQueueWords) != hipSuccess) return fail(NCHMM_E_HIP);
    if ((rc = dev_alloc(c, (void**)&c->d_prof, sizeof(unsigned long long) * 6200))) return fail(rc);
    if (hipMemset(c->d_prof, 0, sizeof(unsigned long long) * 6200) != hipSuccess) return fail(NCHMM_E_HIP);
    if ((rc = dev_alloc(c, (void**)&c->d_fb_total, sizeof(unsigned long long)))) return fail(rc);
    if (hipMemset(c->d_fb_total, 0, sizeof(unsigned long long)) != hipSuccess) return fail(NCHMM_E_HIP);
    {
        const char* e = std::getenv("NCHMM_PROFILE");
        c->profile = e && e[0] == '1';
        const char* m = std::getenv("NCHMM_TB_MARGIN");
        if (m) c->tb_margin = std::max(0, std::atoi(m));
        const char* f = std::getenv("NCHMM_FB_FORCE_LOG");
        c->fb_force_log = f && f[0] == '1';
        if (const char* w = std::getenv("NCHMM_VIT_SWEEP"))     // wide | ll | auto: which form of the sweep launches take
            c->sweep_mode = !std::strcmp(w, "wide") ? kSweepWide : !std::strcmp(w, "ll") ? kSweepLl : !std::strcmp(w, "ahead") ? kSweepAhead : kSweepAuto;
        if (const char* k = std::getenv("NCHMM_PLAN_CLOCK_MHZ")) c->plan_clock_mhz = std::atof(k);      // the sustained shader clock the sweep plan prices with
        const char* b = std::getenv("NCHMM_FB_BUDGET_MB");
        if (b) c->fb_budget = std::max<size_t>((size_t)std::strtoull(b, nullptr, 10) << 20, (size_t)16 << 20);
    }
    if (hipEventCreate(&c->ev_fb0) != hipSuccess || hipEventCreate(&c->ev_fb1) != hipSuccess)
        return fail(NCHMM_E_HIP);
    c->vit_slots = c->n_cu * viterbi_blocks_per_cu();
    // back-pointer regions are handed out per XCD: as many as blocks can be resident there, and some to spare
    c->slots_per_xcd = (unsigned)((c->vit_slots + (int)kXcds - 1) / (int)kXcds) + 8u;
    if ((rc = dev_alloc(c, (void**)&c->d_slot_owner, sizeof(unsigned) * kXcds * c->slots_per_xcd))) return fail(rc);
    // NCHMM_TEST_POISON_POOL=1 (test hook): every region marked taken, as a kernel that died with its regions would leave
    // them -- launches must then fail loudly after their bounded wait, not hang and not return stale outputs
    const int fill = std::getenv("NCHMM_TEST_POISON_POOL") ? 1 : 0;
    if (hipMemset(c->d_slot_owner, fill, sizeof(unsigned) * kXcds * c->slots_per_xcd) != hipSuccess) return fail(NCHMM_E_HIP);
    c->fb_slots = c->n_cu * fwbw_blocks_per_cu();
    // NCHMM_FB_SLOTS (measurement hook, tools/ubench/fb_two_lanes.py): a smaller grid for the forward-backward kernels of this context
    if (const char* e = std::getenv("NCHMM_FB_SLOTS")) { const int v = std::atoi(e); if (v > 0 && v < c->fb_slots) c->fb_slots = v; }
    *out = c;
    return NCHMM_OK;
}

int nchmm_destroy(nchmm_ctx* c)
{
    if (!c) return NCHMM_E_INVALID;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    em_lane_select(c, 0);
    if (c->own_stream) (void)hipStreamSynchronize(c->stream);
    for (int l = 0; l < kVitLanes; ++l)
        if (c->lane[l].stream) (void)hipStreamSynchronize(c->lane[l].stream);
    if (c->d_models) (void)hipFree(c->d_models);
    if (c->d_trans) (void)hipFree(c->d_trans);
    if (c->d_trans_fb) (void)hipFree(c->d_trans_fb);
    if (c->d_train_mask) (void)hipFree(c->d_train_mask);
    if (c->d_queue) (void)hipFree(c->d_queue);
    if (c->d_vq) (void)hipFree(c->d_vq);
    if (c->s_in) { (void)hipStreamSynchronize(c->s_in); (void)hipStreamDestroy(c->s_in); }
    combine_destroy(c);
    pipe_destroy(c);
    if (c->d_model_fast) (void)hipFree(c->d_model_fast);
    if (c->d_prof) (void)hipFree(c->d_prof);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_slot_owner) (void)hipFree(c->d_slot_owner);
    if (c->d_ws_big) (void)hipFree(c->d_ws_big);
    if (c->d_em) (void)hipFree(c->d_em);
    for (int l = 0; l <= kVitLanes; ++l) if (c->d_plan[l]) (void)hipFree(c->d_plan[l]);
    if (c->d_plan_counts) (void)hipFree(c->d_plan_counts);
    if (c->h_plan_counts) (void)hipHostFree(c->h_plan_counts);
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->d_fb_ws) (void)hipFree(c->d_fb_ws);
    if (c->d_fb_aux) (void)hipFree(c->d_fb_aux);
    if (c->d_fb_total) (void)hipFree(c->d_fb_total);
    if (c->d_em_events) (void)hipFree(c->d_em_events);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->d_tab_stage) (void)hipFree(c->d_tab_stage);
    {   // the second EM lane's set (its stream is Viterbi lane 1's)
        EmLaneRes& o = c->em_other;
        if (o.d_stage) (void)hipFree(o.d_stage);
        if (o.d_fb_aux) (void)hipFree(o.d_fb_aux);
        if (o.d_queue) (void)hipFree(o.d_queue);
        if (o.d_tab_stage) (void)hipFree(o.d_tab_stage);
        if (o.h_pin) (void)hipHostFree(o.h_pin);
        if (o.ev_fb0) (void)hipEventDestroy(o.ev_fb0);
        if (o.ev_fb1) (void)hipEventDestroy(o.ev_fb1);
    }
    if (c->d_masks) (void)hipFree(c->d_masks);
    for (int l = 0; l < kVitLanes; ++l) {
        if (c->lane[l].ev0) (void)hipEventDestroy(c->lane[l].ev0);
        if (c->lane[l].ev1) (void)hipEventDestroy(c->lane[l].ev1);
        if (c->lane[l].done) (void)hipEventDestroy(c->lane[l].done);
        if (l > 0 && c->lane[l].stream) (void)hipStreamDestroy(c->lane[l].stream);
    }
    if (c->ev_entry) (void)hipEventDestroy(c->ev_entry);
    if (c->ev_big) (void)hipEventDestroy(c->ev_big);
    if (c->ev_em) (void)hipEventDestroy(c->ev_em);
    if (c->ev_fb0) (void)hipEventDestroy(c->ev_fb0);
    if (c->ev_fb1) (void)hipEventDestroy(c->ev_fb1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return NCHMM_OK;
}

int nchmm_last_hip_error(const nchmm_ctx* c) { return c ? c->last_hip : 0; }

int nchmm_set_stream(nchmm_ctx* c, void* s)
{
    if (!c) return NCHMM_E_INVALID;
    c->stream = (hipStream_t)s;   // 0 = the legacy default stream, not "mine" (nchmm_use_own_stream)
    c->external_stream = true;
    return NCHMM_OK;
}

int nchmm_use_own_stream(nchmm_ctx* c)
{
    if (!c) return NCHMM_E_INVALID;
    c->stream = c->own_stream;
    c->external_stream = false;
    return NCHMM_OK;
}

int nchmm_synchronize(nchmm_ctx* c)
{
    if (!c) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    for (int l = 0; l < kVitLanes; ++l) {
        if (c->lane[l].pending) HIP_TRY(c, hipStreamSynchronize(c->lane[l].stream));
        c->lane[l].pending = false; c->lane[l].joined = true;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return viterbi_check_err(c);
}

int nchmm_put_model(nchmm_ctx* c, int slot, const float* t6)
{
    if (!c || !t6 || slot < 0 || slot >= c->n_slots) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<float> img(kModelFloats);
    int32_t fast = 1;
    for (int j = 0; j < kStates; ++j) {
        const float* s = t6 + (size_t)j * 6;
        model_image_row(img.data(), j, s[0], s[1], s[2], s[3], s[4], s[5], &fast);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // a running kernel may still read this slot
    HIP_TRY(c, hipMemcpy(c->d_models + (size_t)slot * kModelFloats, img.data(), sizeof(float) * kModelFloats,
                         hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_model_fast + slot, &fast, sizeof(int32_t), hipMemcpyHostToDevice));
    c->model_set[slot] = true;
    return NCHMM_OK;
}

// The device image of a scaled model (what nchmm_put_model builds from the S x 6 table), on the host: callers that stage many
// models in parallel (nchmm_viterbi_strand) build their own and upload them together.
int nchmm_model_image(const float* t6, float* image, int32_t* fast)
{
    if (!t6 || !image || !fast) return NCHMM_E_INVALID;
    *fast = 1;
    for (int j = 0; j < kStates; ++j) {
        const float* s = t6 + (size_t)j * 6;
        model_image_row(image, j, s[0], s[1], s[2], s[3], s[4], s[5], fast);
    }
    return NCHMM_OK;
}

int nchmm_put_model_images(nchmm_ctx* c, int first_slot, size_t n, const float* images, const int32_t* fast)
{
    if (!c || first_slot < 0 || (n && (!images || !fast)) || n > (1u << 20)) return NCHMM_E_INVALID;
    if (n == 0) return NCHMM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = reserve_slots(c, first_slot + (int)n);
    if (rc != NCHMM_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // a running kernel may still read these slots
    HIP_TRY(c, hipMemcpyAsync(c->d_models + (size_t)first_slot * kModelFloats, images, sizeof(float) * kModelFloats * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_model_fast + first_slot, fast, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < n; ++k) c->model_set[first_slot + k] = 1;
    return NCHMM_OK;
}

int nchmm_reserve_slots(nchmm_ctx* c, int n_slots)
{
    if (!c || n_slots < 0 || n_slots > (1 << 20)) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    return reserve_slots(c, n_slots);
}

// Batched Pore_Model::scale + upload: slot first_slot + k gets table table_idx[k] scaled by params[k].
// The float operations are those of Pore_Model_State::scale (Pore_Model.hpp:126-138) on the fields the
// emission reads (sd_stdv is not one of them and is skipped).
int nchmm_put_models_scaled(nchmm_ctx* c, int first_slot, size_t n, const float* states_Sx10, const int32_t* table_idx,
                            const float* params_nx6)
{
    if (!c || first_slot < 0 || (n && (!states_Sx10 || !params_nx6))) return NCHMM_E_INVALID;
    if (n == 0) return NCHMM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = reserve_slots(c, first_slot + (int)n);
    if (rc != NCHMM_OK) return rc;
    // tables + per-slot parameters go up (small); the images are built on the device (tables_kernel.hip)
    int32_t max_idx = 0;
    for (size_t k = 0; k < n; ++k) {
        const int32_t ti = table_idx ? table_idx[k] : 0;
        if (ti < 0) return NCHMM_E_INVALID;
        max_idx = std::max(max_idx, ti);
    }
    const size_t n_tables = (size_t)max_idx + 1;
    const size_t b_states = sizeof(float) * n_tables * kStates * 10, b_idx = sizeof(int32_t) * n, b_par = sizeof(float) * 8 * n;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    void* hp = nullptr;
    if ((rc = pinned(c, al(b_idx) + al(b_par) + al(sizeof(int32_t) * n), &hp))) return rc;
    int32_t* h_idx = (int32_t*)hp;
    float* h_par = (float*)((char*)hp + al(b_idx));
    int32_t* h_one = (int32_t*)((char*)hp + al(b_idx) + al(b_par));
    for (size_t k = 0; k < n; ++k) {
        h_idx[k] = table_idx ? table_idx[k] : 0;
        const float* p = params_nx6 + 6 * k;
        std::memcpy(h_par + 8 * k, p, 6 * sizeof(float));
        h_par[8 * k + 6] = std::log(p[3]);   // log_params.var, Pore_Model.hpp:193
        h_par[8 * k + 7] = std::log(p[5]);   // log_params.var_sd :195
        h_one[k] = 1;
    }
    void* dp = c->d_tab_stage;
    rc = ensure(c, &dp, &c->tab_stage_bytes, al(b_states) + al(b_idx) + al(b_par));
    c->d_tab_stage = dp;
    if (rc != NCHMM_OK) return rc;
    char* d = (char*)c->d_tab_stage;
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(d, states_Sx10, b_states, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d + al(b_states), h_idx, b_idx, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d + al(b_states) + al(b_idx), h_par, b_par, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_model_fast + first_slot, h_one, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    launch_scale_models((const float*)d, (const int32_t*)(d + al(b_states)), (const float*)(d + al(b_states) + al(b_idx)),
                        c->d_models, c->d_model_fast, first_slot, n, static_cast<float>(std::log(2.0 * M_PI)), st);
    HIP_TRY(c, hipGetLastError());
    if (!c->em_async) HIP_TRY(c, hipStreamSynchronize(st));   // the pinned staging buffer is reused by the next call
    for (size_t k = 0; k < n; ++k) c->model_set[first_slot + k] = 1;
    return NCHMM_OK;
}

// Batched State_Transitions::compute_transitions_fast + upload, without materialising the 85 936
// arcs: the factorised weights w0/w1/w2 are read off one representative arc per stay / step-group /
// skip-group (same get_trans_prob arithmetic), which is what factor_transitions would extract from
// the full CSR (tests compare the two routes).
int nchmm_put_transitions_fast(nchmm_ctx* c, int first_slot, size_t n, const float* p_skip, const float* p_stay)
{
    if (!c || first_slot < 0 || (n && (!p_skip || !p_stay))) return NCHMM_E_INVALID;
    if (n == 0) return NCHMM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = reserve_slots(c, first_slot + (int)n);
    if (rc != NCHMM_OK) return rc;
    void* hp = nullptr;
    if ((rc = pinned(c, sizeof(float) * 64 * n, &hp))) return rc;
    float* const wm = (float*)hp;
    // ~100 pow / log calls per slot: worth the host cores when an EM round brings thousands of (job, strand) tables
    if (n >= 256) parallel_for(n, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) mask_weights(p_skip[k], p_stay[k], wm + 64 * k); });
    else for (size_t k = 0; k < n; ++k) mask_weights(p_skip[k], p_stay[k], wm + 64 * k);
    void* dp = c->d_tab_stage;
    rc = ensure(c, &dp, &c->tab_stage_bytes, sizeof(float) * 64 * n);
    c->d_tab_stage = dp;
    if (rc != NCHMM_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_tab_stage, wm, sizeof(float) * 64 * n, hipMemcpyHostToDevice, c->stream));
    launch_expand_transitions((const float*)c->d_tab_stage, c->d_masks, c->d_trans, c->d_trans_fb, first_slot, n, c->stream);
    HIP_TRY(c, hipGetLastError());
    if (!c->em_async) HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < n; ++k) c->trans_set[first_slot + k] = 1;
    return NCHMM_OK;
}

int nchmm_put_transitions(nchmm_ctx* c, int slot, const uint32_t* row_ptr, const uint16_t* pred, const float* logw)
{
    if (!c || !row_ptr || !pred || !logw || slot < 0 || slot >= c->n_slots) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<float> w(kTransFloats), fb(kFbTransFloats);
    int rc = factor_transitions(row_ptr, pred, logw, w.data());
    if (rc != NCHMM_OK) return rc;
    fb_weights(w.data(), fb.data());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_trans + (size_t)slot * kTransFloats, w.data(), sizeof(float) * kTransFloats,
                         hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_trans_fb + (size_t)slot * kFbTransFloats, fb.data(), sizeof(float) * kFbTransFloats,
                         hipMemcpyHostToDevice));
    c->trans_set[slot] = true;
    return NCHMM_OK;
}

}  // extern "C" (reopened below)

namespace nchmm {

// Largest back-pointer workspace the context is willing to hold: NCHMM_WS_BUDGET_MB, else 60 % of what is free at first use
int viterbi_ws_budget(nchmm_ctx* c, size_t* out)
{
    if (c->ws_budget == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
        const char* e = std::getenv("NCHMM_WS_BUDGET_MB");
        c->ws_budget = e ? (size_t)std::strtoull(e, nullptr, 10) << 20 : (free_b / 10) * 6;
        if (c->ws_budget < ((size_t)16 << 20)) c->ws_budget = (size_t)16 << 20;
    }
    *out = c->ws_budget;
    return NCHMM_OK;
}

// One region of kBpRowBytes x (longest read) per resident block.  Normal case: a pool of kXcds x p regions -- p = as many
// blocks as can be resident on an XCD and a few to spare, fewer when the launches are smaller than that -- fits the budget:
// blocks take and return regions themselves and launches may overlap.  Reads so long that the pool would not fit: as many
// regions as the budget holds (at least one: a read longer than the budget still runs, alone), one block per region,
// region = block index, launches one after the other.
int viterbi_ws_prepare(nchmm_ctx* c, uint64_t longest, size_t count, size_t budget_share)
{
    size_t budget = 0;
    int rc = viterbi_ws_budget(c, &budget);
    if (rc != NCHMM_OK) return rc;
    // (a batch with outliers keeps 30 % of the budget for their regions: the pool and its head-room get the rest, nchmm_plan.hpp)
    if (budget_share && budget_share < budget) budget = budget_share;
    const size_t need = (size_t)std::max<uint64_t>(longest, 1) * kBpRowBytes;
    // blocks of up to kVitLanes launches can sit on one XCD at a time
    const unsigned per_xcd = (unsigned)std::min<size_t>(c->slots_per_xcd, std::max<size_t>(count, 1) * kVitLanes);
    const bool want_pooled = need <= budget / ((size_t)kXcds * per_xcd);
    if (c->d_ws && need <= c->slot_bytes && ((c->ws_pooled && per_xcd <= c->ws_per_xcd) || (!c->ws_pooled && !want_pooled))) return NCHMM_OK;
    // (re)allocate: nothing may be using the regions.  Unconditionally: a join only queues a stream wait, the kernels of a
    // joined batch may still be running
    for (int l = 0; l < kVitLanes; ++l) {
        HIP_TRY(c, hipStreamSynchronize(c->lane[l].stream));
        c->lane[l].pending = false; c->lane[l].joined = true;
    }
    const unsigned per_xcd_new = c->ws_pooled ? std::max(per_xcd, c->ws_per_xcd) : per_xcd;
    const size_t keep_slot = c->ws_pooled ? c->slot_bytes : 0;
    if (c->d_ws) {
        HIP_TRY(c, hipFree(c->d_ws));
        c->counters[6] -= c->ws_bytes;
        c->d_ws = nullptr; c->ws_bytes = 0; c->slot_bytes = 0; c->ws_regions = 0; c->ws_per_xcd = 0;
    }
    // The budget was taken from the memory that was free when the context first sized a workspace.  Other contexts and other
    // processes on the device (four worker processes sharing one GPU; four contexts in one test process) may have taken it since:
    // an allocation that does not fit is retried with what is free NOW -- fewer regions, down to one -- instead of failing the
    // batch (round 6: a sweep with four contexts of 60 000-event reads in one process ended in NCHMM_E_NOMEM).
    for (int attempt = 0;; ++attempt) {
        const bool pooled = need <= budget / ((size_t)kXcds * per_xcd);
        size_t slot = need, regions = (size_t)kXcds * per_xcd_new;
        if (pooled && std::max(need, keep_slot) <= budget / regions) {
            slot = std::max(need, keep_slot);
            slot = std::min(slot + slot / 8, budget / regions);   // head-room so slowly growing batches do not reallocate every call
        } else if (pooled) {
            regions = (size_t)kXcds * per_xcd;                    // (the pool had grown past what this read length allows)
        } else {
            regions = std::min<size_t>(std::max<size_t>(budget / need, 1), (size_t)std::max(c->vit_slots, 1));
        }
        slot = (slot + 4095) & ~(size_t)4095;
        void* p = nullptr;
        rc = dev_alloc(c, &p, slot * regions);
        if (rc == NCHMM_OK) {
            c->d_ws = (uint8_t*)p; c->ws_bytes = slot * regions; c->slot_bytes = slot; c->ws_regions = (unsigned)regions;
            c->ws_pooled = pooled;
            c->ws_per_xcd = pooled ? (unsigned)(regions / kXcds) : 0;
            return NCHMM_OK;
        }
        if (rc != NCHMM_E_NOMEM || attempt >= 4 || slot * regions <= need + 4096) return rc;      // (one region did not fit either)
        (void)hipGetLastError();
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = slot * regions / 2;
        budget = std::max<size_t>(need, std::min<size_t>(slot * regions / 2, free_b / 10 * 8));
        if (!budget_share) c->ws_budget = std::min(c->ws_budget, budget);     // (later batches start from what the device really has)
    }
}

// rows (events) the buffer of emissions computed ahead may hold: NCHMM_EM_BUDGET_MB, default 256 MiB = 16 384 events -- what the
// memory-side cache holds.  From there the recurrence-only column runs at its own pace, 0.59 us per event (a 5000-event strand:
// 2.9 ms against 4.0 in the plain low-latency form; the same loop without any stream: 0.577); once the rows come from HBM a
// column costs 0.77-0.83 whatever the prefetch schedule (8 x 30 000 events: 23.0 against 24.7 ms; 96 x 10 000: slower;
// profiles/r05_fed_reads.md) -- not worth 16 KiB per event.
uint64_t viterbi_em_budget_rows(nchmm_ctx* c)
{
    if (c->em_budget == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)32 << 30;
        const char* e = std::getenv("NCHMM_EM_BUDGET_MB");
        c->em_budget = e ? (size_t)std::strtoull(e, nullptr, 10) << 20 : std::min<size_t>((size_t)256 << 20, free_b / 4);
        // a caller that bounds the back-pointer workspace (NCHMM_WS_BUDGET_MB) is short of memory: the buffer follows, a quarter of
        // that at most (unless it was set on its own)
        size_t ws = 0;
        if (!e && viterbi_ws_budget(c, &ws) == NCHMM_OK) c->em_budget = std::min(c->em_budget, ws / 4);
        if (c->em_budget < ((size_t)1 << 20)) c->em_budget = (size_t)1 << 20;
    }
    return (uint64_t)(c->em_budget / ((size_t)kStates * sizeof(float)));
}

namespace {
int viterbi_em_prepare(nchmm_ctx* c, uint64_t rows)
{
    const size_t need = (size_t)std::max<uint64_t>(rows, 1) * kStates * sizeof(float);
    if (c->d_em && c->em_bytes >= need) return NCHMM_OK;
    for (int l = 0; l < kVitLanes; ++l) HIP_TRY(c, hipStreamSynchronize(c->lane[l].stream));     // (a sweep may be reading the old one)
    c->em_pending = false;
    if (c->d_em) { HIP_TRY(c, hipFree(c->d_em)); c->counters[6] -= c->em_bytes; c->d_em = nullptr; c->em_bytes = 0; }
    const size_t want = std::min<size_t>(need + need / 4, std::max(need, c->em_budget));
    void* p = nullptr;
    const int rc = dev_alloc(c, &p, want);
    if (rc != NCHMM_OK) return rc;
    c->d_em = (float*)p; c->em_bytes = want;
    return NCHMM_OK;
}
}  // namespace

hipStream_t viterbi_next_lane_stream(nchmm_ctx* c) { return c->lane[c->ws_pooled ? c->next_lane : 0].stream; }

// The outliers of a batch.  Reads too long for regions of which the whole pool fits the budget get a few regions of their own
// (one block per region, region = block index), so that one 300 000-event read does not put the whole batch on a tenth of the
// GPU: the rest goes through the pool as usual and the outliers run beside it as one more launch (launch_viterbi_outliers).
int viterbi_big_prepare(nchmm_ctx* c, uint64_t longest, size_t n_long, size_t budget_big)
{
    const size_t need = ((size_t)std::max<uint64_t>(longest, 1) * kBpRowBytes + 4095) & ~(size_t)4095;
    const size_t regions = std::min<size_t>(std::min<size_t>(std::max<size_t>(n_long, 1), (size_t)std::max(c->vit_slots, 1)), std::max<size_t>(budget_big / need, 1));
    if (c->d_ws_big && need <= c->big_slot_bytes && regions <= c->big_regions) return NCHMM_OK;
    for (int l = 0; l < kVitLanes; ++l) {    // (re)allocate: a launch of outliers may be using the regions
        HIP_TRY(c, hipStreamSynchronize(c->lane[l].stream));
        c->lane[l].pending = false; c->lane[l].joined = true;
    }
    if (c->d_ws_big) {
        HIP_TRY(c, hipFree(c->d_ws_big));
        c->counters[6] -= c->big_slot_bytes * c->big_regions;
        c->d_ws_big = nullptr; c->big_slot_bytes = 0; c->big_regions = 0;
    }
    void* p = nullptr;
    const int rc = dev_alloc(c, &p, need * regions);
    if (rc != NCHMM_OK) return rc;
    c->d_ws_big = (uint8_t*)p; c->big_slot_bytes = need; c->big_regions = (unsigned)regions;
    return NCHMM_OK;
}

namespace {

// One launch on the next lane.  pooled: regions from the pool (ws / slot_bytes / per_xcd of the context); else region = block
// index in `ws`, at most `regions` blocks, and `serial_after` (may be null) = the launch that used those regions before.
int launch_on_next_lane(nchmm_ctx* c, bool pooled, uint8_t* ws, size_t slot_bytes, size_t regions, hipEvent_t after, hipEvent_t serial_after,
                        size_t first, size_t count, uint64_t ev_count, const uint64_t* d_off, const float* d_cmean, const float* d_stdv,
                        const float* d_lstdv, const int32_t* d_model_slot, const int32_t* d_trans_slot, const uint32_t* d_order,
                        uint16_t* d_out_state, float* d_out_logp, int32_t* d_out_status, int li, int* lane_out, int sweep, const AheadArgs* ahead)
{
    VitLaneState& L = c->lane[li];
    const int form = c->sweep_mode == kSweepAuto ? sweep : c->sweep_mode;
    const bool ll = form == kSweepLl || form == kSweepAhead;
    // emissions ahead: only where the caller planned them (a forced "ahead" without a plan is the plain low-latency form), and
    // not when the context is forced to another form
    bool em = ahead && ahead->n && form == kSweepAhead;
    if (em) {
        // (no memory for the rows: the reads compute their emissions in place -- the same bits, a slower launch)
        const int rc = viterbi_em_prepare(c, ahead->rows);
        if (rc == NCHMM_E_NOMEM) { (void)hipGetLastError(); em = false; }
        else if (rc != NCHMM_OK) return rc;
    }
    ViterbiArgs a;
    a.cmean = d_cmean; a.stdv = d_stdv; a.lstdv = d_lstdv; a.off = d_off;
    a.model_slot = d_model_slot; a.trans_slot = d_trans_slot; a.order = d_order;
    a.models = c->d_models; a.trans = c->d_trans; a.model_fast = c->d_model_fast;
    a.prof = c->profile ? c->d_prof : nullptr;
    a.ws = ws; a.slot_bytes = slot_bytes;
    a.slot_owner = pooled ? c->d_slot_owner : nullptr; a.slots_per_xcd = c->ws_per_xcd;
    a.host_err = c->h_err;
    a.first_read = (unsigned)first;
    a.out_state = d_out_state; a.out_logp = d_out_logp; a.out_status = d_out_status;
    a.queue = c->d_vq + li; a.cu_progress = c->d_vq + 16; a.n_reads = (unsigned)count;
    c->launch_seq = c->launch_seq % 4095u + 1u;
    a.launch_tag = c->launch_seq;
    a.tb_margin = c->tb_margin;
    a.em = em ? c->d_em : nullptr; a.em_row0 = em ? ahead->d_row0 : nullptr;
    a.em_rows = em ? (uint64_t)(c->em_bytes / ((size_t)kStates * sizeof(float))) : 0;
    a.log_n_states = std::log(static_cast<float>(kStates));           // Viterbi.hpp:51
    a.log_2pi = static_cast<float>(std::log(2.0 * M_PI));
    // the wide sweep: two blocks per CU; the low-latency sweep: one
    const size_t slots = std::min<size_t>(pooled ? (size_t)c->vit_slots : regions, ll ? (size_t)c->n_cu : (size_t)-1);
    const int grid = (int)std::min<size_t>(slots, count);
    a.queue_base = L.vq_base;
    if (after) HIP_TRY(c, hipStreamWaitEvent(L.stream, after, 0));
    if (serial_after) HIP_TRY(c, hipStreamWaitEvent(L.stream, serial_after, 0));
    if (em && c->em_pending) HIP_TRY(c, hipStreamWaitEvent(L.stream, c->ev_em, 0));   // the buffer is one: behind the sweep that read it last
    HIP_TRY(c, hipEventRecord(L.ev0, L.stream));      // (the launch's duration -- nchmm_last_kernel_ms -- includes its emission kernel)
    if (em) {
        // the emissions of the first ahead->n reads of the order, by every CU that is free (the device is this launch's: nothing
        // else is in flight when a plan asks for them)
        launch_emissions(a, (unsigned)ahead->n, ahead->longest, c->d_em, L.stream);
        HIP_TRY(c, hipGetLastError());
    }
    if (ll) launch_viterbi_ll(a, grid, L.stream); else launch_viterbi(a, grid, L.stream);
    HIP_TRY(c, hipGetLastError());
    if (em) { HIP_TRY(c, hipEventRecord(c->ev_em, L.stream)); c->em_pending = true; c->ahead_stats[0] += 1; c->ahead_stats[1] += ahead->n; c->ahead_stats[2] += ahead->rows; }
    // every read takes a ticket, every block one more to find the queue empty -- counted only once the launch is in the queue
    // (a failure above must leave the lane's ticket base where the device's queue head will be)
    L.vq_base += (unsigned)count + (unsigned)grid;
    c->sweep_stats[ll ? 1 : 0] += 1;
    c->sweep_stats[ll ? 3 : 2] += count;
    HIP_TRY(c, hipEventRecord(L.ev1, L.stream));
    HIP_TRY(c, hipEventRecord(L.done, L.stream));
    L.pending = true; L.joined = false;
    c->last_lane = li;
    c->next_lane = (li + 1) % kVitLanes;
    c->vit_timed = true;
    c->counters[2] += (uint64_t)(ev_count > count ? ev_count - count : 0) * kBpRowBytes;
    c->counters[3] += 1;
    if (lane_out) *lane_out = li;
    return NCHMM_OK;
}

}  // namespace

int launch_viterbi_range(nchmm_ctx* c, hipEvent_t after, size_t first, size_t count, uint64_t ev_count,
                         const uint64_t* d_off, const float* d_cmean, const float* d_stdv, const float* d_lstdv,
                         const int32_t* d_model_slot, const int32_t* d_trans_slot, const uint32_t* d_order, uint16_t* d_out_state,
                         float* d_out_logp, int32_t* d_out_status, int* lane_out, int sweep, const AheadArgs* ahead)
{
    if (!c->d_ws) return NCHMM_E_INVALID;
    // without a pool every launch owns regions 0 .. grid-1: one lane, strictly one launch after the other
    const int li = c->ws_pooled ? c->next_lane : 0;
    hipEvent_t serial = (!c->ws_pooled && c->last_lane >= 0 && c->last_lane != li) ? c->lane[c->last_lane].done : nullptr;
    return launch_on_next_lane(c, c->ws_pooled, c->d_ws, c->slot_bytes, c->ws_regions, after, serial, first, count, ev_count, d_off, d_cmean, d_stdv,
                               d_lstdv, d_model_slot, d_trans_slot, d_order, d_out_state, d_out_logp, d_out_status, li, lane_out, sweep, ahead);
}

// The outliers of a batch (d_order lists them): one block per region of d_ws_big, behind the previous launch of outliers.
int launch_viterbi_outliers(nchmm_ctx* c, hipEvent_t after, size_t count, uint64_t ev_count, const uint64_t* d_off, const float* d_cmean,
                            const float* d_stdv, const float* d_lstdv, const int32_t* d_model_slot, const int32_t* d_trans_slot,
                            const uint32_t* d_order, uint16_t* d_out_state, float* d_out_logp, int32_t* d_out_status, int* lane_out, int sweep,
                            const AheadArgs* ahead)
{
    if (!c->d_ws_big || !d_order || !c->ws_pooled) return NCHMM_E_INVALID;
    int li = 0;
    const int rc = launch_on_next_lane(c, false, c->d_ws_big, c->big_slot_bytes, c->big_regions, after, c->big_pending ? c->ev_big : nullptr, 0, count, ev_count,
                                       d_off, d_cmean, d_stdv, d_lstdv, d_model_slot, d_trans_slot, d_order, d_out_state, d_out_logp, d_out_status,
                                       c->next_lane, &li, sweep, ahead);
    if (rc != NCHMM_OK) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_big, c->lane[li].stream));
    c->big_pending = true;
    if (lane_out) *lane_out = li;
    return NCHMM_OK;
}

int viterbi_join(nchmm_ctx* c, hipStream_t s)
{
    for (int l = 0; l < kVitLanes; ++l) {
        VitLaneState& L = c->lane[l];
        if (!L.pending || (L.joined && L.joined_to == s)) continue;
        if (L.stream != s) HIP_TRY(c, hipStreamWaitEvent(s, L.done, 0));
        L.joined = true; L.joined_to = s;      // (pending stays set: no host-side wait has seen the launch finish -- nchmm_synchronize and the reallocations do)
    }
    return NCHMM_OK;
}

int viterbi_check_err(nchmm_ctx* c)
{
    if (c->h_err && *(volatile unsigned*)c->h_err) {
        *c->h_err = 0;
        c->last_hip = (int)hipErrorLaunchFailure;   // (the launch ran, but a block left without doing its share)
        return NCHMM_E_HIP;
    }
    return NCHMM_OK;
}

}  // namespace nchmm

namespace {

// streaming: the caller queues batch after batch (nchmm_viterbi_dev_enqueue) -- launches run beside each other and the tail of
// one is covered by the next; else the batch is on its own (nchmm_viterbi_dev) and its duration is what counts.
int viterbi_dev_enqueue(nchmm_ctx* c, bool streaming, size_t n_reads, size_t max_events, size_t total_events,
                        const uint64_t* d_off, const float* d_cmean, const float* d_stdv, const float* d_lstdv,
                        const int32_t* d_model_slot, const int32_t* d_trans_slot, const uint32_t* d_order,
                        uint16_t* d_out_state, float* d_out_logp, int32_t* d_out_status)
{
    if (!c) return NCHMM_E_INVALID;
    if (n_reads == 0) return NCHMM_OK;
    if (!d_off || !d_out_logp || n_reads > 0xFFFFFFF0ull || max_events > 0x7FFFFFF0ull) return NCHMM_E_INVALID;
    if (total_events && (!d_cmean || !d_stdv || !d_lstdv || !d_out_state)) return NCHMM_E_INVALID;
    if (pipe_in_flight(c)) return NCHMM_E_INVALID;   // the host-pointer pipeline owns the lanes (nchmm_viterbi_begin)
    HIP_TRY(c, hipSetDevice(c->device));
    // behind what the CALLER has on its stream.  The context's own stream is lane 0: what is on it are this context's earlier
    // launches (every other entry point that uses it returns synchronised), and waiting for those is what the lanes are there
    // to avoid.
    hipEvent_t after = nullptr;
    if (c->external_stream) {
        HIP_TRY(c, hipEventRecord(c->ev_entry, c->stream));
        after = c->ev_entry;
    }
    int rc;
    // ---- the plan, made on the device when the lengths are only there (plan_kernel.hip) ----
    // ragged: the caller states a longest read well above the mean -- a launch lasts as long as its longest read, so the reads
    // are handed out longest first.  tight: reads so long that a full pool of regions of that length does not fit the budget --
    // if only a few reads are that long they get regions of their own and the pool is sized for the rest (what the host-pointer
    // forms do from the host's copy of the offsets, nchmm_plan.hpp: plan_outliers); the counts come back with one small copy.
    size_t budget = 0;
    if ((rc = viterbi_ws_budget(c, &budget))) return rc;
    const size_t pool = (size_t)kXcds * std::min<size_t>(c->slots_per_xcd, n_reads * kVitLanes);
    // (a batch shape -- reads, longest -- whose plan found most reads to be long is not planned again: the plan costs every lane a
    // wait and a read-back, and for uniformly long reads it changes nothing; such batches go the usual way, longest first)
    const bool tight = !d_order && n_reads >= 8 && (size_t)max_events * kBpRowBytes > budget / pool &&
                       !(c->tight_skip_reads == n_reads && c->tight_skip_longest == max_events);
    const bool ragged = !d_order && n_reads > 1 && total_events && (double)max_events * (double)n_reads > 1.25 * (double)total_events;
    auto plan_buffer = [&](int slot) -> int {           // [order n | outliers n] for lane `slot` (kVitLanes: the tight path's own)
        if (c->plan_cap[slot] >= 2 * n_reads) return NCHMM_OK;
        for (int l = 0; l < kVitLanes; ++l) HIP_TRY(c, hipStreamSynchronize(c->lane[l].stream));    // (a sweep may be reading the old one)
        if (c->d_plan[slot]) { HIP_TRY(c, hipFree(c->d_plan[slot])); c->counters[6] -= 4 * c->plan_cap[slot]; c->d_plan[slot] = nullptr; c->plan_cap[slot] = 0; }
        const size_t cap = 2 * n_reads + n_reads / 4;
        void* p = nullptr;
        const int r = dev_alloc(c, &p, 4 * cap);
        if (r != NCHMM_OK) return r;
        c->d_plan[slot] = (uint32_t*)p; c->plan_cap[slot] = cap;
        return NCHMM_OK;
    };
    if (!c->d_plan_counts && (tight || ragged)) {
        void* p = nullptr;
        if ((rc = dev_alloc(c, &p, sizeof(unsigned long long) * 4 * (kVitLanes + 1)))) return rc;
        c->d_plan_counts = (unsigned long long*)p;
        HIP_TRY(c, hipHostMalloc(&p, sizeof(unsigned long long) * 4, hipHostMallocDefault));
        c->h_plan_counts = (unsigned long long*)p;
    }
    if (tight) {
        // every lane idle first: the tight path's buffer is read by whichever lanes the launches below land on
        for (int l = 0; l < kVitLanes; ++l) { HIP_TRY(c, hipStreamSynchronize(c->lane[l].stream)); c->lane[l].pending = false; c->lane[l].joined = true; }
        if ((rc = plan_buffer(kVitLanes))) return rc;
        const uint64_t small_cap = (uint64_t)(budget / 10 * 7 / pool / kBpRowBytes);     // events a pooled region may hold (plan_outliers)
        hipStream_t s0 = c->lane[0].stream;
        if (after) HIP_TRY(c, hipStreamWaitEvent(s0, after, 0));
        uint32_t* const d_in = c->d_plan[kVitLanes];
        uint32_t* const d_out = d_in + n_reads;
        unsigned long long* const d_cnt = c->d_plan_counts + 4 * kVitLanes;
        launch_plan_order(d_off, (unsigned)n_reads, max_events, small_cap, d_in, d_out, d_cnt, s0);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(c->h_plan_counts, d_cnt, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, s0));
        HIP_TRY(c, hipStreamSynchronize(s0));
        const size_t n_in = (size_t)c->h_plan_counts[0], n_out = (size_t)c->h_plan_counts[1];
        const uint64_t longest_in = c->h_plan_counts[2], longest = c->h_plan_counts[3];
        if (n_in + n_out != n_reads || longest > max_events) return NCHMM_E_INVALID;      // (the offsets do not match what the caller stated)
        c->counters[0] += n_reads;
        c->counters[1] += total_events;
        if (!(small_cap >= 256 && n_out * 8 <= n_reads)) { c->tight_skip_reads = n_reads; c->tight_skip_longest = max_events; }
        if (small_cap >= 256 && n_out * 8 <= n_reads && n_out > 0) {
            // a few long reads: the pool for the rest, regions of their own for them, one more launch beside the pooled one
            if ((rc = viterbi_ws_prepare(c, std::max<uint64_t>(longest_in, 1), std::max<size_t>(n_in, 1), budget - budget / 10 * 3))) return rc;
            // (no memory for regions of their own -- an earlier batch may hold the whole budget as the pool: everything the usual way)
            const int rc_big = viterbi_big_prepare(c, longest, n_out, budget / 10 * 3);
            if (rc_big == NCHMM_OK && c->ws_pooled) {
                // (the outliers first: they are the longest reads of the batch and set its duration -- one per CU in the low-latency
                // form while there are no more of them than CUs; every lane is idle here, so their blocks are placed before the
                // pooled launch's)
                if ((rc = launch_viterbi_outliers(c, nullptr, n_out, 0, d_off, d_cmean, d_stdv, d_lstdv, d_model_slot, d_trans_slot, d_out, d_out_state,
                                                  d_out_logp, d_out_status, nullptr, n_out <= (size_t)c->n_cu ? kSweepLl : kSweepWide))) return rc;
                if (n_in)
                    rc = launch_viterbi_range(c, nullptr, 0, n_in, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_model_slot, d_trans_slot, d_in,
                                              d_out_state, d_out_logp, d_out_status, nullptr,
                                              choose_sweep_bounds(n_in, longest_in, total_events, (size_t)c->n_cu, (size_t)c->vit_slots, true, rates_at_clock(c->plan_clock_mhz)));
                return rc;
            }
        }
        // most reads are long (or the budget is tiny): everything the usual way on as many blocks as the budget has regions for --
        // longest first all the same (the two lists one after the other: outliers, then the rest)
        if ((rc = viterbi_ws_prepare(c, max_events, n_reads))) return rc;
        if (n_out && n_in) {
            // [outliers | rest] contiguous: the outliers' list sits behind the order; rotate through the spare quarter is not worth a
            // kernel -- two launches in the order outliers, rest
            if ((rc = launch_viterbi_range(c, nullptr, 0, n_out, 0, d_off, d_cmean, d_stdv, d_lstdv, d_model_slot, d_trans_slot, d_out, d_out_state,
                                           d_out_logp, d_out_status, nullptr, kSweepWide))) return rc;
            return launch_viterbi_range(c, nullptr, 0, n_in, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_model_slot, d_trans_slot, d_in, d_out_state,
                                        d_out_logp, d_out_status, nullptr, kSweepWide);
        }
        return launch_viterbi_range(c, nullptr, 0, n_reads, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_model_slot, d_trans_slot, n_out ? d_out : d_in,
                                    d_out_state, d_out_logp, d_out_status, nullptr,
                                    choose_sweep_bounds(n_reads, max_events, total_events, (size_t)c->n_cu, (size_t)c->vit_slots, streaming, rates_at_clock(c->plan_clock_mhz)));
    }
    if ((rc = viterbi_ws_prepare(c, max_events, n_reads))) return rc;
    c->counters[0] += n_reads;
    c->counters[1] += total_events;
    const uint32_t* order = d_order;
    if (ragged) {
        // longest first, by a kernel on the lane the sweep will run on, in front of it: nothing is waited for
        const int li = c->ws_pooled ? c->next_lane : 0;
        if ((rc = plan_buffer(li))) return rc;
        hipStream_t sl = c->lane[li].stream;
        if (after) HIP_TRY(c, hipStreamWaitEvent(sl, after, 0));
        launch_plan_order(d_off, (unsigned)n_reads, max_events, ~0ull, c->d_plan[li], c->d_plan[li] + n_reads, c->d_plan_counts + 4 * li, sl);
        HIP_TRY(c, hipGetLastError());
        order = c->d_plan[li];
    }
    // the lengths are on the device: the form of the sweep follows from what the caller states (reads, longest, total); emissions
    // ahead = of every read of the batch (row of read r's event i: off[r] + i), when that fits the buffer
    const uint64_t em_rows = viterbi_em_budget_rows(c);
    int sweep = choose_sweep_bounds(n_reads, max_events, total_events, (size_t)c->n_cu, (size_t)c->vit_slots, streaming, rates_at_clock(c->plan_clock_mhz), em_rows);
    if (c->sweep_mode == kSweepAhead && !streaming && total_events <= em_rows && n_reads <= kMaxAheadReads) sweep = kSweepAhead;
    AheadArgs ahead;
    if (sweep == kSweepAhead) { ahead.n = n_reads; ahead.rows = total_events; ahead.longest = max_events; ahead.d_row0 = nullptr; }
    return launch_viterbi_range(c, after, 0, n_reads, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_model_slot, d_trans_slot,
                                order, d_out_state, d_out_logp, d_out_status, nullptr, sweep, ahead.n ? &ahead : nullptr);
}

}  // namespace

extern "C" {

// The batch starts behind whatever is on the context's stream now and may run beside the batch queued before it (the blocks
// of one launch start where the previous launch's blocks run out of reads).  Nothing is waited for here.
int nchmm_viterbi_dev_enqueue(nchmm_ctx* c, size_t n_reads, size_t max_events, size_t total_events,
                              const uint64_t* d_off, const float* d_cmean, const float* d_stdv, const float* d_lstdv,
                              const int32_t* d_model_slot, const int32_t* d_trans_slot, const uint32_t* d_order,
                              uint16_t* d_out_state, float* d_out_logp, int32_t* d_out_status)
{
    return viterbi_dev_enqueue(c, true, n_reads, max_events, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_model_slot, d_trans_slot, d_order,
                               d_out_state, d_out_logp, d_out_status);
}

// Put every batch queued by nchmm_viterbi_dev_enqueue in front of whatever comes next on the context's stream.
int nchmm_viterbi_dev_join(nchmm_ctx* c)
{
    if (!c) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    return viterbi_join(c, c->stream);
}

int nchmm_viterbi_dev(nchmm_ctx* c, size_t n_reads, size_t max_events, size_t total_events,
                      const uint64_t* d_off, const float* d_cmean, const float* d_stdv, const float* d_lstdv,
                      const int32_t* d_model_slot, const int32_t* d_trans_slot, const uint32_t* d_order,
                      uint16_t* d_out_state, float* d_out_logp, int32_t* d_out_status)
{
    const int rc = viterbi_dev_enqueue(c, false, n_reads, max_events, total_events, d_off, d_cmean, d_stdv, d_lstdv, d_model_slot,
                                       d_trans_slot, d_order, d_out_state, d_out_logp, d_out_status);
    return rc == NCHMM_OK && n_reads ? nchmm_viterbi_dev_join(c) : rc;
}

}  // extern "C"

extern "C" {

int nchmm_logf(nchmm_ctx* c, size_t n, const float* in, float* out)
{
    if (!c || (n && (!in || !out))) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t chunk = (size_t)1 << 24;
    int rc = ensure(c, &c->d_stage, &c->stage_bytes, 2 * sizeof(float) * std::min(n ? n : 1, chunk));
    if (rc != NCHMM_OK) return rc;
    float* d_in = (float*)c->d_stage;
    float* d_out = d_in + std::min(n ? n : 1, chunk);
    for (size_t b = 0; b < n; b += chunk) {
        const size_t m = std::min(chunk, n - b);
        HIP_TRY(c, hipMemcpyAsync(d_in, in + b, m * sizeof(float), hipMemcpyHostToDevice, c->stream));
        launch_logf(d_in, d_out, m, c->stream);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(out + b, d_out, m * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return NCHMM_OK;
}

int nchmm_fwbw_dev(nchmm_ctx* c, size_t n_win, size_t max_events, size_t total_events, const uint64_t* d_off,
                   const float* d_cmean, const float* d_stdv, const float* d_lstdv, const int32_t* d_scaled_slot,
                   const float* d_pm_params, const int32_t* d_trans_slot, const float* d_st_params,
                   float* d_out_lpd, float* d_out_pm, float* d_out_st, float* d_out_alpha, float* d_out_beta)
{
    if (!c) return NCHMM_E_INVALID;
    if (n_win == 0) return NCHMM_OK;
    if (!d_off || !d_out_lpd || n_win > 0xFFFFFFF0ull) return NCHMM_E_INVALID;
    if (total_events && (!d_cmean || !d_stdv || !d_lstdv)) return NCHMM_E_INVALID;
    (void)max_events;
    HIP_TRY(c, hipSetDevice(c->device));
    float* alpha = d_out_alpha;
    if (!alpha) {
        void* p = c->d_fb_ws;
        size_t have = c->fb_ws_floats * sizeof(float);
        int rc = ensure(c, &p, &have, std::max<size_t>(total_events, 1) * kStates * sizeof(float));
        c->d_fb_ws = (float*)p; c->fb_ws_floats = have / sizeof(float);
        if (rc != NCHMM_OK) return rc;
        alpha = c->d_fb_ws;
    }
    // the rescaled linear-space kernels cannot hand out log matrices: those calls take the log-space pair
    const bool scaled = !d_out_alpha && !d_out_beta && !c->fb_force_log;
    auto al256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_lpd = 0, o_zf = o_lpd + al256(4 * n_win), o_list = o_zf + al256(4 * n_win), o_flag = o_list + al256(4 * n_win);
    const size_t o_exp = o_flag + al256(n_win), aux_need = o_exp + al256(4 * std::max<size_t>(total_events, 1));
    {
        int rc = ensure(c, &c->d_fb_aux, &c->fb_aux_bytes, aux_need);
        if (rc != NCHMM_OK) return rc;
    }
    char* aux = (char*)c->d_fb_aux;
    FwbwArgs a;
    a.cmean = d_cmean; a.stdv = d_stdv; a.lstdv = d_lstdv; a.off = d_off;
    a.scaled_slot = d_scaled_slot; a.pm_params = d_pm_params; a.trans_slot = d_trans_slot;
    a.st_params = d_st_params; a.models = c->d_models; a.trans_fb = c->d_trans_fb; a.trans = c->d_trans; a.train_mask = c->d_train_mask;
    a.ws_alpha = alpha; a.ws_lpd2 = (float*)(aux + o_lpd); a.alpha_natural = d_out_alpha ? 1 : 0;
    a.ws_zfin = (float*)(aux + o_zf); a.fb_list = (unsigned*)(aux + o_list); a.fb_flag = (uint8_t*)(aux + o_flag);
    a.ws_exp = (int32_t*)(aux + o_exp); a.fb_count = c->d_queue + 5; a.fb_total = c->d_fb_total; a.win_list = nullptr; a.n_list = nullptr;
    a.out_log_pr_data = d_out_lpd; a.out_pm_sums = d_out_pm; a.out_st_sums = d_out_st;
    a.out_beta = d_out_beta; a.queue = c->d_queue + 1; a.n_win = (unsigned)n_win;
    a.prof = c->profile ? c->d_prof : nullptr;
    a.log_n_states = std::log(static_cast<float>(kStates));           // Forward_Backward.hpp:53
    a.log_2pi = static_cast<float>(std::log(2.0 * M_PI));
    const int grid = (int)std::min<size_t>((size_t)c->fb_slots, n_win);
    HIP_TRY(c, hipMemsetAsync(c->d_queue + 1, 0, 5 * sizeof(unsigned), c->stream));   // four work queues + the redo count
    if (scaled) HIP_TRY(c, hipMemsetAsync(aux + o_flag, 0, n_win, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev_fb0, c->stream));
    launch_fwbw(a, grid, c->stream, scaled);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_fb1, c->stream));
    c->fb_timed = true;
    c->counters[3] += 1; c->counters[4] += n_win; c->counters[5] += total_events;
    return NCHMM_OK;
}

}  // extern "C"

namespace {

// Largest number of events a forward-backward launch may cover: its alpha rows are 16 KiB per event (plus, for the
// host-pointer form with matrices, two staged n x S outputs).  NCHMM_FB_BUDGET_MB overrides a quarter of the device memory.
size_t fb_budget_events(nchmm_ctx* c, size_t bytes_per_event)
{
    if (c->fb_budget == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { total_b = (size_t)64 << 30; free_b = total_b; }
        const char* e = std::getenv("NCHMM_FB_BUDGET_MB");
        // (a quarter of the device, but no more than most of what is free now: other contexts and processes may hold the rest)
        c->fb_budget = e ? (size_t)std::strtoull(e, nullptr, 10) << 20 : std::min(total_b / 4, free_b / 10 * 8);
        if (c->fb_budget < ((size_t)16 << 20)) c->fb_budget = (size_t)16 << 20;
    }
    return std::max<size_t>(c->fb_budget / bytes_per_event, 1);
}

int fwbw_host_range(nchmm_ctx* c, size_t n_win, const uint64_t* off, const float* cmean, const float* stdv, const float* lstdv,
                    const int32_t* scaled_slot, const float* pm_params, const int32_t* trans_slot, const float* st_params,
                    float* out_lpd, float* out_pm, float* out_st, float* out_alpha, float* out_beta);

}  // namespace

extern "C" {

// Windows are independent: a batch whose alpha rows exceed the budget is cut into consecutive window ranges that run one
// after the other through the same workspace (a single window larger than the budget still runs, alone).
int nchmm_fwbw(nchmm_ctx* c, size_t n_win, const uint64_t* off, const float* cmean, const float* stdv, const float* lstdv,
               const int32_t* scaled_slot, const float* pm_params, const int32_t* trans_slot, const float* st_params,
               float* out_lpd, float* out_pm, float* out_st, float* out_alpha, float* out_beta)
{
    if (!c) return NCHMM_E_INVALID;
    if (n_win == 0) return NCHMM_OK;
    size_t max_events = 0, total = 0;
    int rc = check_offsets(n_win, off, &max_events, &total);
    if (rc != NCHMM_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t per_event = (size_t)kStates * sizeof(float) * (1 + (out_alpha ? 1 : 0) + (out_beta ? 1 : 0));
    const size_t cap = fb_budget_events(c, per_event);
    if (total <= cap)
        return fwbw_host_range(c, n_win, off, cmean, stdv, lstdv, scaled_slot, pm_params, trans_slot, st_params, out_lpd, out_pm, out_st,
                               out_alpha, out_beta);
    std::vector<uint64_t> sub;
    for (size_t w0 = 0; w0 < n_win;) {
        size_t w1 = w0 + 1;
        while (w1 < n_win && off[w1 + 1] - off[w0] <= cap) ++w1;
        sub.assign(w1 - w0 + 1, 0);
        for (size_t w = w0; w <= w1; ++w) sub[w - w0] = off[w] - off[w0];
        const size_t e0 = (size_t)off[w0];
        rc = fwbw_host_range(c, w1 - w0, sub.data(), cmean ? cmean + e0 : nullptr, stdv ? stdv + e0 : nullptr, lstdv ? lstdv + e0 : nullptr,
                             scaled_slot ? scaled_slot + w0 : nullptr, pm_params ? pm_params + 6 * w0 : nullptr,
                             trans_slot ? trans_slot + w0 : nullptr, st_params ? st_params + 2 * w0 : nullptr, out_lpd + w0,
                             out_pm ? out_pm + 6 * e0 : nullptr, out_st ? out_st + 3 * w0 : nullptr,
                             out_alpha ? out_alpha + e0 * (size_t)kStates : nullptr, out_beta ? out_beta + e0 * (size_t)kStates : nullptr);
        if (rc != NCHMM_OK) return rc;
        w0 = w1;
    }
    return NCHMM_OK;
}

}  // extern "C"

namespace {

int fwbw_host_range(nchmm_ctx* c, size_t n_win, const uint64_t* off, const float* cmean, const float* stdv, const float* lstdv,
                    const int32_t* scaled_slot, const float* pm_params, const int32_t* trans_slot, const float* st_params,
                    float* out_lpd, float* out_pm, float* out_st, float* out_alpha, float* out_beta)
{
    if (!c) return NCHMM_E_INVALID;
    if (n_win == 0) return NCHMM_OK;
    size_t max_events = 0, total = 0;
    int rc = check_offsets(n_win, off, &max_events, &total);
    if (rc != NCHMM_OK) return rc;
    if (!out_lpd || (total && (!cmean || !stdv || !lstdv))) return NCHMM_E_INVALID;
    for (size_t w = 0; w < n_win; ++w) {
        const int ms = scaled_slot ? scaled_slot[w] : 0;
        const int ts = trans_slot ? trans_slot[w] : 0;
        if (ms < 0 || ms >= c->n_slots || ts < 0 || ts >= c->n_slots || !c->model_set[ms] || !c->trans_set[ts])
            return NCHMM_E_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t mat = out_alpha || out_beta ? al(4 * total * kStates) : 0;
    size_t o_off = 0, o_cm = o_off + al(8 * (n_win + 1)), o_sd = o_cm + al(4 * total), o_ls = o_sd + al(4 * total);
    size_t o_ss = o_ls + al(4 * total), o_us = o_ss + al(4 * n_win), o_ts = o_us + al(24 * n_win), o_sp = o_ts + al(4 * n_win);
    size_t o_lp = o_sp + al(8 * n_win), o_pm = o_lp + al(4 * n_win), o_st = o_pm + al(24 * total), o_al = o_st + al(12 * n_win);
    size_t o_be = o_al + (out_alpha ? mat : 0), need = o_be + (out_beta ? mat : 0);
    rc = ensure(c, &c->d_stage, &c->stage_bytes, need);
    if (rc != NCHMM_OK) return rc;
    char* d = (char*)c->d_stage;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(d + o_off, off, 8 * (n_win + 1), hipMemcpyHostToDevice, s));
    if (total) {
        HIP_TRY(c, hipMemcpyAsync(d + o_cm, cmean, 4 * total, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(d + o_sd, stdv, 4 * total, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(d + o_ls, lstdv, 4 * total, hipMemcpyHostToDevice, s));
    }
    if (scaled_slot) HIP_TRY(c, hipMemcpyAsync(d + o_ss, scaled_slot, 4 * n_win, hipMemcpyHostToDevice, s));
    if (pm_params) HIP_TRY(c, hipMemcpyAsync(d + o_us, pm_params, 24 * n_win, hipMemcpyHostToDevice, s));
    if (trans_slot) HIP_TRY(c, hipMemcpyAsync(d + o_ts, trans_slot, 4 * n_win, hipMemcpyHostToDevice, s));
    if (st_params) HIP_TRY(c, hipMemcpyAsync(d + o_sp, st_params, 8 * n_win, hipMemcpyHostToDevice, s));
    rc = nchmm_fwbw_dev(c, n_win, max_events, total, (const uint64_t*)(d + o_off), (const float*)(d + o_cm),
                        (const float*)(d + o_sd), (const float*)(d + o_ls), scaled_slot ? (const int32_t*)(d + o_ss) : nullptr,
                        pm_params ? (const float*)(d + o_us) : nullptr, trans_slot ? (const int32_t*)(d + o_ts) : nullptr,
                        st_params ? (const float*)(d + o_sp) : nullptr, (float*)(d + o_lp), out_pm ? (float*)(d + o_pm) : nullptr,
                        out_st ? (float*)(d + o_st) : nullptr, out_alpha ? (float*)(d + o_al) : nullptr,
                        out_beta ? (float*)(d + o_be) : nullptr);
    if (rc != NCHMM_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_lpd, d + o_lp, 4 * n_win, hipMemcpyDeviceToHost, s));
    if (out_pm && total) HIP_TRY(c, hipMemcpyAsync(out_pm, d + o_pm, 24 * total, hipMemcpyDeviceToHost, s));
    if (out_st) HIP_TRY(c, hipMemcpyAsync(out_st, d + o_st, 12 * n_win, hipMemcpyDeviceToHost, s));
    if (out_alpha && total) HIP_TRY(c, hipMemcpyAsync(out_alpha, d + o_al, 4 * total * kStates, hipMemcpyDeviceToHost, s));
    if (out_beta && total) HIP_TRY(c, hipMemcpyAsync(out_beta, d + o_be, 4 * total * kStates, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return NCHMM_OK;
}

int em_round_range(nchmm_ctx* c, size_t n_win, const uint64_t* win_src, const uint32_t* win_len, const float* win_drift,
                   const float* win_pm, const int32_t* scaled_slot, const int32_t* trans_slot, const float* st_params,
                   size_t n_jobs, const uint32_t* job_first_win, int train_drift, float* out_lpd, float* out_st, double* out_acc);

}  // namespace

extern "C" {

int nchmm_em_load_events(nchmm_ctx* c, size_t n_events, const float* mean, const float* stdv, const float* start,
                         const float* log_stdv)
{
    if (!c || (n_events && (!mean || !stdv || !start || !log_stdv))) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t stride = (n_events + 63) & ~(size_t)63;
    int rc = ensure(c, &c->d_em_events, &c->em_events_bytes, std::max<size_t>(4 * stride, 64) * sizeof(float));
    if (rc != NCHMM_OK) return rc;
    c->em_n_events = n_events;
    float* d = (float*)c->d_em_events;
    const float* src[4] = {mean, stdv, start, log_stdv};
    for (int k = 0; k < 4; ++k)
        if (n_events) HIP_TRY(c, hipMemcpyAsync(d + k * stride, src[k], n_events * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NCHMM_OK;
}

// A round whose alpha rows exceed the forward-backward budget is cut at job boundaries (a job's windows stay together:
// its outer sums are reduced on the device) and the ranges run one after the other.
// Take the alpha-row workspace for batches of up to `events` window events NOW (bounded by the FB budget).  A host that knows a
// long run is coming calls this while it is still reading its input: the first allocation of a multi-GiB workspace on a device
// whose memory is not mapped yet costs ~20 ms per GiB, which otherwise lands in the first EM round (nanocall: 690 ms against
// 200 for the first chunk's rounds, profiles/r05_notes.md).
int nchmm_reserve_fb_workspace(nchmm_ctx* c, size_t events)
{
    if (!c) return NCHMM_E_INVALID;
    if (events == 0) return NCHMM_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t cap = fb_budget_events(c, (size_t)kStates * sizeof(float));
    void* p = c->d_fb_ws;
    size_t have = c->fb_ws_floats * sizeof(float);
    const int rc = ensure(c, &p, &have, std::min(events, cap) * kStates * sizeof(float));
    c->d_fb_ws = (float*)p; c->fb_ws_floats = have / sizeof(float);
    return rc;
}

// The same for the Viterbi back-pointer regions: a full pool (launches of any size, all three lanes) of regions for reads of up
// to longest_events; 0 = the longest a full pool fits in the budget (NCHMM_WS_BUDGET_MB).
int nchmm_reserve_viterbi_workspace(nchmm_ctx* c, size_t longest_events)
{
    if (!c) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t budget = 0;
    int rc = viterbi_ws_budget(c, &budget);
    if (rc != NCHMM_OK) return rc;
    const size_t fit = budget / ((size_t)kXcds * c->slots_per_xcd) / kBpRowBytes;
    if (longest_events == 0 || longest_events > fit) longest_events = fit;
    // (the pool keeps 1/8 head-room on top of what it is asked for: ask for that much less)
    return longest_events ? viterbi_ws_prepare(c, std::max<size_t>(longest_events * 8 / 9, 1), (size_t)c->vit_slots) : NCHMM_OK;
}

int nchmm_em_round(nchmm_ctx* c, size_t n_win, const uint64_t* win_src, const uint32_t* win_len, const float* win_drift,
                   const float* win_pm, const int32_t* scaled_slot, const int32_t* trans_slot, const float* st_params,
                   size_t n_jobs, const uint32_t* job_first_win, int train_drift, float* out_lpd, float* out_st, double* out_acc)
{
    if (!c) return NCHMM_E_INVALID;
    if (n_win == 0) return NCHMM_OK;
    if (!win_src || !win_len || !win_drift || !out_lpd || (n_jobs && (!job_first_win || !out_acc)) || n_win > 0xFFFFFFF0ull)
        return NCHMM_E_INVALID;
    if (n_jobs && (job_first_win[0] != 0 || job_first_win[n_jobs] != n_win)) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t total = 0;
    for (size_t w = 0; w < n_win; ++w) total += win_len[w];
    const size_t cap = fb_budget_events(c, (size_t)kStates * sizeof(float));
    if (total <= cap || n_jobs <= 1)
        return em_round_range(c, n_win, win_src, win_len, win_drift, win_pm, scaled_slot, trans_slot, st_params, n_jobs, job_first_win,
                              train_drift, out_lpd, out_st, out_acc);
    std::vector<uint64_t> ev_before(n_win + 1, 0);
    for (size_t w = 0; w < n_win; ++w) ev_before[w + 1] = ev_before[w] + win_len[w];
    std::vector<uint32_t> jf;
    for (size_t j0 = 0; j0 < n_jobs;) {
        size_t j1 = j0 + 1;
        while (j1 < n_jobs && ev_before[job_first_win[j1 + 1]] - ev_before[job_first_win[j0]] <= cap) ++j1;
        const size_t w0 = job_first_win[j0], w1 = job_first_win[j1];
        jf.assign(j1 - j0 + 1, 0);
        for (size_t j = j0; j <= j1; ++j) jf[j - j0] = job_first_win[j] - (uint32_t)w0;
        const int rc = em_round_range(c, w1 - w0, win_src + w0, win_len + w0, win_drift + w0, win_pm ? win_pm + 6 * w0 : nullptr,
                                      scaled_slot ? scaled_slot + w0 : nullptr, trans_slot ? trans_slot + w0 : nullptr,
                                      st_params ? st_params + 2 * w0 : nullptr, j1 - j0, jf.data(), train_drift, out_lpd + w0,
                                      out_st ? out_st + 3 * w0 : nullptr, out_acc + 13 * j0);
        if (rc != NCHMM_OK) return rc;
        j0 = j1;
    }
    return NCHMM_OK;
}

}  // extern "C"

namespace {

int em_round_range(nchmm_ctx* c, size_t n_win, const uint64_t* win_src, const uint32_t* win_len, const float* win_drift,
                   const float* win_pm, const int32_t* scaled_slot, const int32_t* trans_slot, const float* st_params,
                   size_t n_jobs, const uint32_t* job_first_win, int train_drift, float* out_lpd, float* out_st, double* out_acc)
{
    EmPending pend;
    int rc = em_round_enqueue(c, n_win, win_src, win_len, win_drift, win_pm, scaled_slot, trans_slot, st_params, n_jobs, job_first_win,
                              train_drift, &pend);
    if (rc != NCHMM_OK) return rc;
    return em_round_collect(c, pend, out_lpd, out_st, out_acc);
}

}  // namespace

namespace nchmm {

namespace {
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
// the descriptors of a round, in one block (host pinned and device alike): off | src | drift | pm | scaled slot | trans slot | st params | job_first_win
struct EmIn {
    size_t off, src, dr, pm, ss, ts, sp, jf, bytes;
    EmIn(size_t n_win, size_t n_jobs)
    {
        off = 0; src = off + al256(8 * (n_win + 1)); dr = src + al256(8 * n_win); pm = dr + al256(4 * n_win); ss = pm + al256(24 * n_win);
        ts = ss + al256(4 * n_win); sp = ts + al256(4 * n_win); jf = sp + al256(8 * n_win); bytes = jf + al256(4 * (n_jobs + 1));
    }
};
// ... and what comes back, in one block: log Pr(data) per window | the three transition sums per window | 13 doubles per job
struct EmOut {
    size_t lp, st, ac, bytes;
    EmOut(size_t n_win, size_t n_jobs) { lp = 0; st = lp + al256(4 * n_win); ac = st + al256(12 * n_win); bytes = ac + al256(104 * std::max<size_t>(n_jobs, 1)); }
};
}  // namespace

size_t em_round_pin_bytes(size_t n_win, size_t n_jobs)
{
    // the round's descriptors and results + the table uploads in front of it (nchmm_put_models_scaled: 40 B per slot in three
    // pieces, nchmm_put_transitions_fast: 256 B per slot; two slots per job) + slack for the pieces' alignment
    return EmIn(n_win, n_jobs).bytes + EmOut(n_win, n_jobs).bytes + al256(2 * n_jobs * 296) + 16 * 256;
}

int em_round_enqueue(nchmm_ctx* c, size_t n_win, const uint64_t* win_src, const uint32_t* win_len, const float* win_drift, const float* win_pm,
                     const int32_t* scaled_slot, const int32_t* trans_slot, const float* st_params, size_t n_jobs, const uint32_t* job_first_win,
                     int train_drift, EmPending* pend)
{
    const EmIn I(n_win, n_jobs);
    const EmOut O(n_win, n_jobs);
    HIP_TRY(c, hipSetDevice(c->device));
    void* hp = nullptr;
    int rc = pinned(c, I.bytes + O.bytes, &hp);
    if (rc != NCHMM_OK) return rc;
    char* h = (char*)hp;
    uint64_t* off = (uint64_t*)(h + I.off);
    size_t max_events = 0;
    off[0] = 0;
    for (size_t w = 0; w < n_win; ++w) {
        if (win_src[w] + win_len[w] > c->em_n_events) return NCHMM_E_INVALID;
        off[w + 1] = off[w] + win_len[w];
        max_events = std::max<size_t>(max_events, win_len[w]);
        const int ms = scaled_slot ? scaled_slot[w] : 0, ts = trans_slot ? trans_slot[w] : 0;
        if (ms < 0 || ms >= c->n_slots || ts < 0 || ts >= c->n_slots || !c->model_set[ms] || !c->trans_set[ts]) return NCHMM_E_INVALID;
    }
    const size_t total = (size_t)off[n_win];
    if (c->em_async && total * kStates > c->fb_ws_floats) return NCHMM_E_INVALID;      // (a lane's share of the alpha rows is fixed while rounds are in flight)
    std::memcpy(h + I.src, win_src, 8 * n_win);
    std::memcpy(h + I.dr, win_drift, 4 * n_win);
    if (win_pm) std::memcpy(h + I.pm, win_pm, 24 * n_win);
    if (scaled_slot) std::memcpy(h + I.ss, scaled_slot, 4 * n_win);
    if (trans_slot) std::memcpy(h + I.ts, trans_slot, 4 * n_win);
    if (st_params) std::memcpy(h + I.sp, st_params, 8 * n_win);
    if (n_jobs) std::memcpy(h + I.jf, job_first_win, 4 * (n_jobs + 1));
    // device staging: [descriptors, as above] [results, as above] cmean | stdv | log stdv | six pm sums per event
    const size_t o_out = I.bytes, o_cm = o_out + O.bytes, o_sd = o_cm + al256(4 * total), o_ls = o_sd + al256(4 * total), o_ps = o_ls + al256(4 * total);
    const size_t need = o_ps + al256(24 * total);
    rc = ensure(c, &c->d_stage, &c->stage_bytes, need);
    if (rc != NCHMM_OK) return rc;
    char* d = (char*)c->d_stage;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(d, h, I.bytes, hipMemcpyHostToDevice, s));
    const size_t stride = (c->em_n_events + 63) & ~(size_t)63;
    const float* ev = (const float*)c->d_em_events;
    EmGatherArgs g;
    g.mean = ev; g.stdv = ev + stride; g.start = ev + 2 * stride; g.lstdv = ev + 3 * stride;
    g.win_src = (const uint64_t*)(d + I.src); g.off = (const uint64_t*)(d + I.off); g.win_drift = (const float*)(d + I.dr);
    g.cmean = (float*)(d + o_cm); g.out_stdv = (float*)(d + o_sd); g.out_lstdv = (float*)(d + o_ls);
    launch_em_gather(g, (unsigned)n_win, s, (unsigned)max_events);
    HIP_TRY(c, hipGetLastError());
    rc = nchmm_fwbw_dev(c, n_win, max_events, total, (const uint64_t*)(d + I.off), (const float*)(d + o_cm), (const float*)(d + o_sd),
                        (const float*)(d + o_ls), scaled_slot ? (const int32_t*)(d + I.ss) : nullptr,
                        win_pm ? (const float*)(d + I.pm) : nullptr, trans_slot ? (const int32_t*)(d + I.ts) : nullptr,
                        st_params ? (const float*)(d + I.sp) : nullptr, (float*)(d + o_out + O.lp), (float*)(d + o_ps), (float*)(d + o_out + O.st),
                        nullptr, nullptr);
    if (rc != NCHMM_OK) return rc;
    if (n_jobs) {
        EmReduceArgs r;
        r.mean = g.mean; r.stdv = g.stdv; r.start = g.start; r.win_src = g.win_src; r.off = g.off;
        r.job_first_win = (const uint32_t*)(d + I.jf); r.pm_sums = (const float*)(d + o_ps); r.train_drift = train_drift;
        r.out = (double*)(d + o_out + O.ac);
        launch_em_reduce(r, (unsigned)n_jobs, s);
        HIP_TRY(c, hipGetLastError());
    }
    // (one copy back: the three result blocks are adjacent; a round without jobs leaves the last one unwritten and unread)
    HIP_TRY(c, hipMemcpyAsync(h + I.bytes, d + o_out, O.bytes, hipMemcpyDeviceToHost, s));
    pend->n_win = n_win; pend->n_jobs = n_jobs; pend->stream = (void*)s;
    pend->h_lpd = (const float*)(h + I.bytes + O.lp); pend->h_st = (const float*)(h + I.bytes + O.st); pend->h_acc = (const double*)(h + I.bytes + O.ac);
    return NCHMM_OK;
}

int em_round_collect(nchmm_ctx* c, const EmPending& pend, float* out_lpd, float* out_st, double* out_acc)
{
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)pend.stream));
    std::memcpy(out_lpd, pend.h_lpd, 4 * pend.n_win);
    if (out_st) std::memcpy(out_st, pend.h_st, 12 * pend.n_win);
    if (pend.n_jobs && out_acc) std::memcpy(out_acc, pend.h_acc, 104 * pend.n_jobs);
    return NCHMM_OK;
}

// The second lane computes on the stream of Viterbi lane 1 (idle while a context trains: its calls are one after the other) --
// a stream of its own would be the fifth of the context, and the runtime maps streams onto four hardware queues in turn: it would
// share the queue of the first lane and run behind it, not beside it (see kVitLanes).
int em_lanes_prepare(nchmm_ctx* c, size_t pin_bytes, size_t events_lane0, size_t events_both)
{
    HIP_TRY(c, hipSetDevice(c->device));
    em_lane_select(c, 0);
    EmLaneRes& o = c->em_other;
    {
        // the alpha rows of both lanes in the ONE workspace, lane 1's behind lane 0's share: a second workspace would be mapped on
        // first use, 20 ms per GiB (nchmm_reserve_fb_workspace), which is more than the lanes save on a run of a few rounds
        void* p = c->d_fb_ws;
        size_t have = c->fb_ws_floats * sizeof(float);
        const int rc = ensure(c, &p, &have, std::max<size_t>(events_both, 1) * kStates * sizeof(float));
        c->d_fb_ws = (float*)p; c->fb_ws_floats = have / sizeof(float);
        if (rc != NCHMM_OK) return rc;
        o.d_fb_ws = c->d_fb_ws + events_lane0 * kStates;
        o.fb_ws_floats = c->fb_ws_floats - events_lane0 * kStates;
    }
    if (!c->em_other_made) {
        o.stream = c->lane[1].stream;
        if (hipEventCreate(&o.ev_fb0) != hipSuccess || hipEventCreate(&o.ev_fb1) != hipSuccess) return NCHMM_E_HIP;
        int rc = dev_alloc(c, (void**)&o.d_queue, sizeof(unsigned) * 16);
        if (rc != NCHMM_OK) return rc;
        HIP_TRY(c, hipMemset(o.d_queue, 0, sizeof(unsigned) * 16));
        c->em_other_made = true;
    }
    // (nothing is in flight on either lane here: the arenas may move)
    pin_bytes += pin_bytes / 8;
    if (c->h_pin_bytes < pin_bytes) {
        if (c->h_pin) { HIP_TRY(c, hipHostFree(c->h_pin)); c->h_pin = nullptr; c->h_pin_bytes = 0; }
        HIP_TRY(c, hipHostMalloc(&c->h_pin, pin_bytes, hipHostMallocDefault));
        c->h_pin_bytes = pin_bytes;
    }
    if (o.h_pin_bytes < pin_bytes) {
        if (o.h_pin) { HIP_TRY(c, hipHostFree(o.h_pin)); o.h_pin = nullptr; o.h_pin_bytes = 0; }
        HIP_TRY(c, hipHostMalloc(&o.h_pin, pin_bytes, hipHostMallocDefault));
        o.h_pin_bytes = pin_bytes;
    }
    return NCHMM_OK;
}

void em_lanes_async(nchmm_ctx* c, bool on) { c->em_async = on; }

void em_lanes_end(nchmm_ctx* c)
{
    c->em_async = false;
    (void)em_lanes_wait(c);
    em_lane_select(c, 0);
    c->em_other.d_fb_ws = nullptr; c->em_other.fb_ws_floats = 0;      // (a view into lane 0's workspace, which may move from here on)
}

void em_lane_rewind(nchmm_ctx* c) { c->pin_cursor = 0; }
size_t em_fb_cap_events(nchmm_ctx* c) { return fb_budget_events(c, (size_t)kStates * sizeof(float)); }

int em_lanes_wait(nchmm_ctx* c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->em_other_made && c->em_other.stream) HIP_TRY(c, hipStreamSynchronize(c->em_other.stream));
    return NCHMM_OK;
}

void em_lane_select(nchmm_ctx* c, int lane)
{
    if (c->em_lane == lane) return;
    EmLaneRes& o = c->em_other;
    std::swap(c->stream, o.stream);
    std::swap(c->d_stage, o.d_stage); std::swap(c->stage_bytes, o.stage_bytes);
    std::swap(c->d_fb_ws, o.d_fb_ws); std::swap(c->fb_ws_floats, o.fb_ws_floats);
    std::swap(c->d_fb_aux, o.d_fb_aux); std::swap(c->fb_aux_bytes, o.fb_aux_bytes);
    std::swap(c->d_queue, o.d_queue);
    std::swap(c->d_tab_stage, o.d_tab_stage); std::swap(c->tab_stage_bytes, o.tab_stage_bytes);
    std::swap(c->h_pin, o.h_pin); std::swap(c->h_pin_bytes, o.h_pin_bytes); std::swap(c->pin_cursor, o.pin_cursor);
    std::swap(c->ev_fb0, o.ev_fb0); std::swap(c->ev_fb1, o.ev_fb1); std::swap(c->fb_timed, o.fb_timed);
    c->em_lane = lane;
}

}  // namespace nchmm

extern "C" {

int nchmm_counters(const nchmm_ctx* c, uint64_t out[8])
{
    if (!c || !out) return NCHMM_E_INVALID;
    std::memcpy(out, c->counters, sizeof(c->counters));
    // [7] lives on the device (the kernels count the windows they hand to the log-space redo)
    unsigned long long redo = 0;
    if (c->d_fb_total) {
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess
            || hipMemcpy(&redo, c->d_fb_total, sizeof(redo), hipMemcpyDeviceToHost) != hipSuccess)
            return NCHMM_E_HIP;
    }
    out[7] = redo;
    return NCHMM_OK;
}

int nchmm_last_kernel_ms(nchmm_ctx* c, float out[4])
{
    if (!c || !out) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    out[0] = out[1] = out[2] = out[3] = 0;
    if (c->vit_timed && c->last_lane >= 0) {
        // the most recent launch (sweep + in-block traceback): meaningful as a kernel time when nothing ran beside it
        const VitLaneState& L = c->lane[c->last_lane];
        HIP_TRY(c, hipEventSynchronize(L.ev1));
        HIP_TRY(c, hipEventElapsedTime(&out[0], L.ev0, L.ev1));
    }
    if (c->fb_timed) {
        HIP_TRY(c, hipEventSynchronize(c->ev_fb1));
        HIP_TRY(c, hipEventElapsedTime(&out[2], c->ev_fb0, c->ev_fb1));
    }
    return NCHMM_OK;
}

int nchmm_shader_clock_mhz(nchmm_ctx* c, double* out_mhz)
{
    if (!c || !out_mhz) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int wall_khz = 0;
    HIP_TRY(c, hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->device));
    if (wall_khz <= 0) wall_khz = 100000;
    unsigned long long* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, 3 * sizeof(unsigned long long)));
    unsigned long long h[3] = {0, 0, 0};
    hipError_t e = hipMemsetAsync(d, 0, sizeof(h), c->stream);
    if (e == hipSuccess) {
        launch_clock_probe(d, 4 * c->n_cu, 40000, c->stream);       // 4 waves per SIMD for about 3 ms
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIP_TRY(c, e);
    if (h[1] == 0) return NCHMM_E_HIP;
    *out_mhz = (double)h[0] / (double)h[1] * (double)wall_khz * 1e-3;
    if (!std::getenv("NCHMM_PLAN_CLOCK_MHZ")) c->plan_clock_mhz = *out_mhz;      // (the sweep plan prices with the clock the device was last seen to hold under load)
    return NCHMM_OK;
}

int nchmm_profile_blocks(nchmm_ctx* c, uint64_t* out /* 6144: 2048 x (start, end) ticks, then 2048 x (xcc << 32 | hw_id) */)
{
    if (!c || !out) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->d_prof + 8, sizeof(uint64_t) * 6144, hipMemcpyDeviceToHost));
    return NCHMM_OK;
}

int nchmm_profile_ticks(nchmm_ctx* c, uint64_t out[8], int reset)
{
    if (!c || !out) return NCHMM_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->d_prof, sizeof(uint64_t) * 8, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(c, hipMemset(c->d_prof, 0, sizeof(uint64_t) * 8));
    return NCHMM_OK;
}

int nchmm_mem_stats(const nchmm_ctx* c, uint64_t out[2])
{
    if (!c || !out) return NCHMM_E_INVALID;
    out[0] = c->counters[6];
    out[1] = c->peak_bytes;
    return NCHMM_OK;
}

int nchmm_set_sweep(nchmm_ctx* c, int mode)
{
    if (!c || mode < kSweepAuto || mode > kSweepAhead) return NCHMM_E_INVALID;
    c->sweep_mode = mode;
    return NCHMM_OK;
}

int nchmm_sweep_stats(const nchmm_ctx* c, uint64_t out[4])
{
    if (!c || !out) return NCHMM_E_INVALID;
    for (int i = 0; i < 4; ++i) out[i] = c->sweep_stats[i];
    return NCHMM_OK;
}

int nchmm_ahead_stats(const nchmm_ctx* c, uint64_t out[3])
{
    if (!c || !out) return NCHMM_E_INVALID;
    for (int i = 0; i < 3; ++i) out[i] = c->ahead_stats[i];
    return NCHMM_OK;
}

int nchmm_grid_slots(const nchmm_ctx* c, int* v)
{
    if (!c || !v) return NCHMM_E_INVALID;
    *v = c->vit_slots;
    return NCHMM_OK;
}

}  // extern "C"
