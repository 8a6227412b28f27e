// nchmm_combine.cpp -- nchmm_viterbi_strand: the reference's call shape (one strand per call, from many pfor worker threads at
// once: basecall_strand, nanocall.cpp:645-690 inside :611-621) on top of the batched decode.  Calls made on one context at the
// same time are combined into launches by StrandCombiner (nchmm_combine.hpp); this file is its device side.
#include "nanocall_hip.h"
#include "nchmm_combine.hpp"
#include "nchmm_ctx.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <numeric>
#include <vector>

using namespace nchmm;

namespace {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// carve one pinned block (the batched entry points copy from pinned memory with the SDMA engines at PCIe rate)
struct Carver {
    char* base = nullptr; size_t at = 0;
    template <typename T> T* take(size_t n) { T* p = reinterpret_cast<T*>(base + at); at += al256(sizeof(T) * n); return p; }
};

void debug_line(const char* what, size_t items, size_t events, std::chrono::steady_clock::time_point t0, std::chrono::steady_clock::time_point t1,
                std::chrono::steady_clock::time_point t2, std::chrono::steady_clock::time_point t3, std::chrono::steady_clock::time_point* last_end)
{
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[%s] batch of %zu / %zu events: %.2f ms since the previous batch ended, models %.2f ms, transitions %.2f ms, device call %.2f ms\n",
                 what, items, events, ms(*last_end, t0), ms(t0, t1), ms(t1, t2), ms(t2, t3));
    *last_end = t3;
}

struct StrandRunner {
    nchmm_ctx* c;
    std::vector<int32_t>* slots;   // 0, 1, 2, ... (strand k decodes with model slot k and transition slot k)
    std::vector<const float*>* tables;   // distinct unscaled tables of a run of strands given as (table, parameters)
    std::vector<int32_t>* table_idx;
    std::vector<float>* states;

    void release(StrandBatch& B)
    {
        if (B.images) (void)hipHostFree(B.images);   // one block: images first
        B.images = nullptr; B.cap[0] = B.cap[1] = B.cap[2] = 0;
    }
    int alloc(StrandBatch& B, const size_t cap[kCombineDims])
    {
        if (hipSetDevice(c->device) != hipSuccess) return NCHMM_E_HIP;
        release(B);
        const size_t reads = cap[0], events = cap[1];
        const size_t bytes = al256(sizeof(float) * kImageFloats * reads) + 5 * al256(4 * reads) + al256(8 * reads) + al256(24 * reads) + al256(8 * (reads + 1))
                             + 3 * al256(4 * events) + al256(2 * events);
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return NCHMM_E_NOMEM;
        Carver q{(char*)p, 0};
        B.images = q.take<float>(kImageFloats * reads);
        B.fast = q.take<int32_t>(reads); B.p_skip = q.take<float>(reads); B.p_stay = q.take<float>(reads);
        B.logp = q.take<float>(reads); B.status = q.take<int32_t>(reads);
        B.base = q.take<const float*>(reads); B.scale6 = q.take<float>(6 * reads);
        B.off = q.take<uint64_t>(reads + 1);
        B.cm = q.take<float>(events); B.sd = q.take<float>(events); B.ls = q.take<float>(events);
        B.states = q.take<uint16_t>(events);
        B.off[0] = 0;
        B.cap[0] = reads; B.cap[1] = events; B.cap[2] = reads;
        return NCHMM_OK;
    }
    int run(StrandBatch& B)
    {
        std::lock_guard<std::mutex> device(c->combine_run);   // (a batch of training windows may be running on another thread)
        const size_t n = B.used[0];
        if (slots->size() < n) { slots->resize(n); std::iota(slots->begin(), slots->end(), 0); }
        const bool dbg = std::getenv("NCHMM_DEBUG") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        // models: runs of consecutive strands of one kind -- by image (one copy per run), or scaled on the device from the
        // distinct unscaled tables of the run (a handful)
        int rc = NCHMM_OK;
        for (size_t k0 = 0; k0 < n && rc == NCHMM_OK;) {
            size_t k1 = k0 + 1;
            while (k1 < n && (B.base[k1] != nullptr) == (B.base[k0] != nullptr)) ++k1;
            if (!B.base[k0]) {
                rc = nchmm_put_model_images(c, (int)k0, k1 - k0, B.images + k0 * kImageFloats, B.fast + k0);
            } else {
                tables->clear();
                table_idx->resize(k1 - k0);
                for (size_t k = k0; k < k1; ++k) {
                    size_t t = 0;
                    while (t < tables->size() && (*tables)[t] != B.base[k]) ++t;
                    if (t == tables->size()) tables->push_back(B.base[k]);
                    (*table_idx)[k - k0] = (int32_t)t;
                }
                const size_t per = (size_t)kStates * 10;
                states->resize(per * tables->size());
                for (size_t t = 0; t < tables->size(); ++t) std::memcpy(states->data() + per * t, (*tables)[t], per * sizeof(float));
                rc = nchmm_put_models_scaled(c, (int)k0, k1 - k0, states->data(), table_idx->data(), B.scale6 + 6 * k0);
            }
            k0 = k1;
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == NCHMM_OK) rc = nchmm_put_transitions_fast(c, 0, n, B.p_skip, B.p_stay);
        const auto t2 = std::chrono::steady_clock::now();
        if (rc == NCHMM_OK) rc = nchmm_viterbi(c, n, B.off, B.cm, B.sd, B.ls, slots->data(), slots->data(), B.states, B.logp, B.status);
        if (dbg) {
            static std::chrono::steady_clock::time_point last_end = t0;
            debug_line("nchmm_viterbi_strand", n, B.used[1], t0, t1, t2, std::chrono::steady_clock::now(), &last_end);
        }
        return rc == NCHMM_E_NUMERIC ? NCHMM_OK : rc;   // (per strand: in status[])
    }
};

struct WindowRunner {
    nchmm_ctx* c;
    std::vector<const float*>* tables;   // distinct unscaled tables of the batch being run
    std::vector<int32_t>* table_idx;
    std::vector<float>* states;          // their S x 10 states, concatenated

    void release(WindowBatch& B)
    {
        if (B.base) (void)hipHostFree(B.base);
        B.base = nullptr; B.cap[0] = B.cap[1] = B.cap[2] = 0;
    }
    int alloc(WindowBatch& B, const size_t cap[kCombineDims])
    {
        if (hipSetDevice(c->device) != hipSuccess) return NCHMM_E_HIP;
        release(B);
        const size_t wins = cap[0], events = cap[1], models = cap[2];
        const size_t bytes = al256(8 * models) + al256(24 * models) + 2 * al256(4 * models) + al256(8 * (wins + 1)) + al256(4 * wins) + al256(24 * wins)
                             + al256(8 * wins) + 3 * al256(4 * events) + al256(4 * wins) + al256(24 * events) + al256(12 * wins);
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return NCHMM_E_NOMEM;
        Carver q{(char*)p, 0};
        B.base = q.take<const float*>(models);
        B.scale6 = q.take<float>(6 * models); B.p_skip = q.take<float>(models); B.p_stay = q.take<float>(models);
        B.off = q.take<uint64_t>(wins + 1);
        B.slot = q.take<int32_t>(wins); B.pm_params = q.take<float>(6 * wins); B.st_params = q.take<float>(2 * wins);
        B.cm = q.take<float>(events); B.sd = q.take<float>(events); B.ls = q.take<float>(events);
        B.lpd = q.take<float>(wins); B.pm_sums = q.take<float>(6 * events); B.st_sums = q.take<float>(3 * wins);
        B.off[0] = 0;
        B.cap[0] = wins; B.cap[1] = events; B.cap[2] = models;
        return NCHMM_OK;
    }
    int run(WindowBatch& B)
    {
        std::lock_guard<std::mutex> device(c->combine_run);   // (a batch of strands may be running on another thread)
        const size_t n_win = B.used[0], n_mod = B.used[2];
        const bool dbg = std::getenv("NCHMM_DEBUG") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        // the distinct unscaled tables (a pore's template / complement models: a handful per run)
        tables->clear();
        table_idx->resize(n_mod);
        for (size_t k = 0; k < n_mod; ++k) {
            size_t t = 0;
            while (t < tables->size() && (*tables)[t] != B.base[k]) ++t;
            if (t == tables->size()) tables->push_back(B.base[k]);
            (*table_idx)[k] = (int32_t)t;
        }
        const size_t per = (size_t)kStates * 10;
        states->resize(per * tables->size());
        for (size_t t = 0; t < tables->size(); ++t) std::memcpy(states->data() + per * t, (*tables)[t], per * sizeof(float));
        int rc = nchmm_put_models_scaled(c, 0, n_mod, states->data(), table_idx->data(), B.scale6);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == NCHMM_OK) rc = nchmm_put_transitions_fast(c, 0, n_mod, B.p_skip, B.p_stay);
        const auto t2 = std::chrono::steady_clock::now();
        if (rc == NCHMM_OK) rc = nchmm_fwbw(c, n_win, B.off, B.cm, B.sd, B.ls, B.slot, B.pm_params, B.slot, B.st_params, B.lpd, B.pm_sums, B.st_sums, nullptr, nullptr);
        if (dbg) {
            static std::chrono::steady_clock::time_point last_end = t0;
            debug_line("nchmm_fwbw_windows", n_win, B.used[1], t0, t1, t2, std::chrono::steady_clock::now(), &last_end);
        }
        return rc;
    }
};

struct Combiner {
    std::vector<int32_t> slots;
    std::vector<const float*> tables;
    std::vector<int32_t> table_idx;
    std::vector<float> states;
    StrandCombiner<StrandRunner> sc;
    Combiner(nchmm_ctx* c, size_t reads, size_t events, unsigned linger) : sc(StrandRunner{c, &slots, &tables, &table_idx, &states}, reads, events, linger) {}
};

struct WinCombiner {
    std::vector<const float*> tables;
    std::vector<int32_t> table_idx;
    std::vector<float> states;
    BatchCombiner<WindowBatch, WindowRunner> bc;
    WinCombiner(nchmm_ctx* c, const size_t cap[kCombineDims], unsigned linger) : bc(WindowRunner{c, &tables, &table_idx, &states}, cap, linger) {}
};

std::mutex g_create;

size_t env_or(const char* name, size_t dflt)
{
    const char* e = std::getenv(name);
    return e ? (size_t)std::strtoull(e, nullptr, 10) : dflt;
}

}  // namespace

namespace nchmm {
void combine_destroy(nchmm_ctx* c)
{
    delete static_cast<Combiner*>(c->combiner);
    c->combiner = nullptr;
    delete static_cast<WinCombiner*>(c->win_combiner);
    c->win_combiner = nullptr;
}
}  // namespace nchmm

namespace {

int strand_call(nchmm_ctx* c, const float* table_Sx6, const float* unscaled_Sx10, const float* pm_params6, float p_skip, float p_stay, size_t n_events,
                const float* cmean, const float* stdv, const float* lstdv, uint16_t* out_state, float* out_logp)
{
    if (!c || !out_logp || (n_events && (!cmean || !stdv || !lstdv || !out_state))) return NCHMM_E_INVALID;
    if (n_events > 0x7FFFFFF0ull) return NCHMM_E_INVALID;
    if (n_events == 0) { *out_logp = __builtin_nanf(""); return NCHMM_OK; }
    Combiner* K;
    {
        std::lock_guard<std::mutex> g(g_create);
        if (!c->combiner) {
            // a batch: up to two grid-fulls of strands and 16 M events (staged in pinned memory: 130 KB + 14 B per event a
            // strand); NCHMM_COMBINE_READS / _EVENTS / _LINGER_US override
            c->combiner = new (std::nothrow) Combiner(c, env_or("NCHMM_COMBINE_READS", 2 * (size_t)std::max(c->vit_slots, 1)),
                                                      env_or("NCHMM_COMBINE_EVENTS", (size_t)16 << 20), (unsigned)env_or("NCHMM_COMBINE_LINGER_US", 200));
            if (!c->combiner) return NCHMM_E_NOMEM;
        }
        K = static_cast<Combiner*>(c->combiner);
    }
    return K->sc.submit([&](float* image, int32_t* fast) { (void)nchmm_model_image(table_Sx6, image, fast); }, p_skip, p_stay, n_events,
                        cmean, stdv, lstdv, out_state, out_logp, unscaled_Sx10, pm_params6);
}

}  // namespace

extern "C" int nchmm_viterbi_strand(nchmm_ctx* c, const float* table_Sx6, float p_skip, float p_stay, size_t n_events, const float* cmean,
                                    const float* stdv, const float* lstdv, uint16_t* out_state, float* out_logp)
{
    if (!table_Sx6) return NCHMM_E_INVALID;
    return strand_call(c, table_Sx6, nullptr, nullptr, p_skip, p_stay, n_events, cmean, stdv, lstdv, out_state, out_logp);
}

// The strand's model as the caller got it: an unscaled table and the Pore_Model_Parameters it was scaled by (pm.scale(pm_params) in
// basecall_strand, nanocall.cpp:653-657) -- 32 bytes to the device instead of a 128 KiB image; Pore_Model::scale runs there.
extern "C" int nchmm_viterbi_strand_scaled(nchmm_ctx* c, const float* unscaled_Sx10, const float* pm_params6, float p_skip, float p_stay, size_t n_events,
                                           const float* cmean, const float* stdv, const float* lstdv, uint16_t* out_state, float* out_logp)
{
    if (!unscaled_Sx10 || !pm_params6) return NCHMM_E_INVALID;
    return strand_call(c, nullptr, unscaled_Sx10, pm_params6, p_skip, p_stay, n_events, cmean, stdv, lstdv, out_state, out_logp);
}

// The same for one read's training windows (train_one_round, Parameter_Trainer.hpp:541-579, from every worker of the pfor of
// nanocall.cpp:282-579): forward-backward + the EM sums of up to a few windows over the read's one or two models, each the
// unscaled model of its strand scaled by the call's Pore_Model_Parameters.
extern "C" int nchmm_fwbw_windows(nchmm_ctx* c, size_t n_models, const float* const* unscaled_Sx10, const float* pm_params6, const float* p_skip,
                                  const float* p_stay, size_t n_win, const uint64_t* off, const float* cmean, const float* stdv, const float* lstdv,
                                  const int32_t* win_model, const float* st_params_nx2, float* out_lpd, float* out_pm_sums, float* out_st_sums)
{
    if (!c || !n_models || !unscaled_Sx10 || !pm_params6 || !p_skip || !p_stay || !off || !win_model || !out_lpd) return NCHMM_E_INVALID;
    if (n_win == 0) return NCHMM_OK;
    if (off[0] != 0) return NCHMM_E_INVALID;
    for (size_t w = 0; w < n_win; ++w)
        if (off[w + 1] < off[w] || win_model[w] < 0 || (size_t)win_model[w] >= n_models) return NCHMM_E_INVALID;
    const size_t total = (size_t)off[n_win];
    if (total && (!cmean || !stdv || !lstdv)) return NCHMM_E_INVALID;
    for (size_t m = 0; m < n_models; ++m) if (!unscaled_Sx10[m]) return NCHMM_E_INVALID;
    WinCombiner* K;
    {
        std::lock_guard<std::mutex> g(g_create);
        if (!c->win_combiner) {
            // a batch: up to 8192 windows (two full FB launches' worth of blocks), 1 M events, 4096 models
            const size_t cap[kCombineDims] = {env_or("NCHMM_COMBINE_WINDOWS", 8192), env_or("NCHMM_COMBINE_WINDOW_EVENTS", (size_t)1 << 20),
                                              env_or("NCHMM_COMBINE_WINDOW_MODELS", 4096)};
            c->win_combiner = new (std::nothrow) WinCombiner(c, cap, (unsigned)env_or("NCHMM_COMBINE_LINGER_US", 200));
            if (!c->win_combiner) return NCHMM_E_NOMEM;
        }
        K = static_cast<WinCombiner*>(c->win_combiner);
    }
    const size_t need[kCombineDims] = {n_win, total, n_models};
    return K->bc.submit(need,
        [&](WindowBatch& B, const CombinePos& p) {
            const size_t w0 = p.at[0], e0 = p.at[1], m0 = p.at[2];
            for (size_t m = 0; m < n_models; ++m) {
                B.base[m0 + m] = unscaled_Sx10[m];
                std::memcpy(B.scale6 + 6 * (m0 + m), pm_params6, 6 * sizeof(float));
                B.p_skip[m0 + m] = p_skip[m]; B.p_stay[m0 + m] = p_stay[m];
            }
            for (size_t w = 0; w < n_win; ++w) {
                B.off[w0 + w + 1] = e0 + off[w + 1];
                B.slot[w0 + w] = (int32_t)(m0 + (size_t)win_model[w]);
                std::memcpy(B.pm_params + 6 * (w0 + w), pm_params6, 6 * sizeof(float));   // the parameters behind the window's scaled model
                if (st_params_nx2) std::memcpy(B.st_params + 2 * (w0 + w), st_params_nx2 + 2 * w, 2 * sizeof(float));
                else { B.st_params[2 * (w0 + w)] = p_stay[win_model[w]]; B.st_params[2 * (w0 + w) + 1] = p_skip[win_model[w]]; }
            }
            if (total) {
                std::memcpy(B.cm + e0, cmean, total * sizeof(float));
                std::memcpy(B.sd + e0, stdv, total * sizeof(float));
                std::memcpy(B.ls + e0, lstdv, total * sizeof(float));
            }
        },
        [&](WindowBatch& B, const CombinePos& p) {
            std::memcpy(out_lpd, B.lpd + p.at[0], n_win * sizeof(float));
            if (out_pm_sums && total) std::memcpy(out_pm_sums, B.pm_sums + 6 * p.at[1], 6 * total * sizeof(float));
            if (out_st_sums) std::memcpy(out_st_sums, B.st_sums + 3 * p.at[0], 3 * n_win * sizeof(float));
            return 0;
        });
}
