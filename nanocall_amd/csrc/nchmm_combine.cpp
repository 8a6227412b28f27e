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

struct LibRunner {
    nchmm_ctx* c;
    std::vector<int32_t>* slots;   // 0, 1, 2, ... (strand k decodes with model slot k and transition slot k)

    void release(CombineBatch& B)
    {
        if (B.images) (void)hipHostFree(B.images);   // one block: images first
        B.images = nullptr; B.cap_reads = B.cap_events = 0;
    }
    // pinned: the batched entry points copy from here with the SDMA engines at PCIe rate
    int alloc(CombineBatch& B, size_t reads, size_t events)
    {
        if (hipSetDevice(c->device) != hipSuccess) return NCHMM_E_HIP;
        release(B);
        const size_t b_img = al256(sizeof(float) * kImageFloats * reads), b_r4 = al256(4 * reads), b_off = al256(8 * (reads + 1));
        const size_t b_e4 = al256(4 * events), b_e2 = al256(2 * events);
        void* p = nullptr;
        if (hipHostMalloc(&p, b_img + 5 * b_r4 + b_off + 3 * b_e4 + b_e2, hipHostMallocDefault) != hipSuccess) return NCHMM_E_NOMEM;
        char* q = (char*)p;
        B.images = (float*)q; q += b_img;
        B.fast = (int32_t*)q; q += b_r4;
        B.p_skip = (float*)q; q += b_r4;
        B.p_stay = (float*)q; q += b_r4;
        B.logp = (float*)q; q += b_r4;
        B.status = (int32_t*)q; q += b_r4;
        B.off = (uint64_t*)q; q += b_off;
        B.cm = (float*)q; q += b_e4;
        B.sd = (float*)q; q += b_e4;
        B.ls = (float*)q; q += b_e4;
        B.states = (uint16_t*)q;
        B.off[0] = 0;
        B.cap_reads = reads; B.cap_events = events;
        return NCHMM_OK;
    }
    int run(CombineBatch& B)
    {
        const size_t n = B.n;
        if (slots->size() < n) { slots->resize(n); std::iota(slots->begin(), slots->end(), 0); }
        const bool dbg = std::getenv("NCHMM_DEBUG") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = nchmm_put_model_images(c, 0, n, B.images, B.fast);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == NCHMM_OK) rc = nchmm_put_transitions_fast(c, 0, n, B.p_skip, B.p_stay);
        const auto t2 = std::chrono::steady_clock::now();
        if (rc == NCHMM_OK) rc = nchmm_viterbi(c, n, B.off, B.cm, B.sd, B.ls, slots->data(), slots->data(), B.states, B.logp, B.status);
        if (dbg) {
            const auto t3 = std::chrono::steady_clock::now();
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            static std::chrono::steady_clock::time_point last_end = t0;
            std::fprintf(stderr, "[nchmm_viterbi_strand] batch of %zu strands / %zu events: %.2f ms since the previous batch ended, models %.2f ms, transitions %.2f ms, decode %.2f ms\n",
                         n, (size_t)B.total, ms(last_end, t0), ms(t0, t1), ms(t1, t2), ms(t2, t3));
            last_end = t3;
        }
        return rc == NCHMM_E_NUMERIC ? NCHMM_OK : rc;   // (per strand: in status[])
    }
};

struct Combiner {
    std::vector<int32_t> slots;
    StrandCombiner<LibRunner> sc;
    Combiner(nchmm_ctx* c, size_t reads, size_t events, unsigned linger) : sc(LibRunner{c, &slots}, reads, events, linger) {}
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
}
}  // namespace nchmm

extern "C" int nchmm_viterbi_strand(nchmm_ctx* c, const float* table_Sx6, float p_skip, float p_stay, size_t n_events, const float* cmean,
                                    const float* stdv, const float* lstdv, uint16_t* out_state, float* out_logp)
{
    if (!c || !table_Sx6 || !out_logp || (n_events && (!cmean || !stdv || !lstdv || !out_state))) return NCHMM_E_INVALID;
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
                        cmean, stdv, lstdv, out_state, out_logp);
}
