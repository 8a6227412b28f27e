// nchmm_internal.hpp -- host-side arithmetic shared by nchmm_host.cpp (single-table ABI functions) and
// nchmm_api.cpp (batched uploads).  Inline so both translation units execute the same float operations.
#ifndef NCHMM_INTERNAL_HPP
#define NCHMM_INTERNAL_HPP

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <thread>
#include <type_traits>
#include <vector>

#include "nchmm_kmer.hpp"

namespace nchmm {

// Host worker pool (nchmm_host.cpp): created on first use -- after the command line has forked its reader processes --
// and kept: a parallel_for used to spawn and join up to 32 std::threads per call (~1-2 ms), which was most of the host
// share of an EM round and of the decode prologue.  Size = the CPUs this process may run on (sched_getaffinity), at most 32;
// NCHMM_HOST_THREADS overrides.  run_chunks calls fn(arg, i) for i in [0, n_chunks), chunk 0 on the calling thread; if
// another thread is using the pool the caller runs every chunk itself.
unsigned host_threads();
void run_chunks(unsigned n_chunks, void (*fn)(void*, unsigned), void* arg);

// f(begin, end) over [0, n) on the host cores (the work items are independent)
template <typename F>
void parallel_for(size_t n, F&& f)
{
    unsigned nt = host_threads();
    if (n < 4 || nt < 2) { f(0, n); return; }
    nt = (unsigned)std::min<size_t>(nt, n);
    using Fn = typename std::remove_reference<F>::type;
    struct Job { Fn* f; size_t n; unsigned nt; } job{&f, n, nt};
    run_chunks(nt, [](void* a, unsigned i) {
        Job* j = static_cast<Job*>(a);
        (*j->f)(j->n * i / j->nt, j->n * (i + 1) / j->nt);
    }, &job);
}


// State_Transitions::get_trans_prob, State_Transitions.hpp:125-144: float accumulator, double pow terms
inline float trans_prob(unsigned i, unsigned j, float p_stay, float p_step, float p_skip_1)
{
    float p = 0;
    if (i == j) p += p_stay;
    if (Kmer6::suffix(i, 5) == Kmer6::prefix(j, 5)) p += p_step / 4;
    for (unsigned l = 2; l < 6; ++l)
        if (Kmer6::suffix(i, 6 - l) == Kmer6::prefix(j, 6 - l))
            p += std::pow(static_cast<double>(p_skip_1), static_cast<double>(l - 1)) / (1u << (2 * l));
    p += (std::pow(static_cast<double>(p_skip_1), 5.0) / (1.0f - p_skip_1)) / 4096u;
    return p;
}

// The same sum with the pow() terms of one p_skip_1 taken from a table: pw[l] = pow(p_skip_1, l) for l = 1..4 (double),
// tail = (pow(p_skip_1, 5.0) / (1.0f - p_skip_1)) / 4096u.  pow is a pure function, so every addend -- and the order they
// are added in -- is the one trans_prob forms; evaluating a transition table's 18 distinct overlap masks costs 5 pow calls
// instead of ~60.
struct TransPow {
    double pw[5];
    double tail;
    explicit TransPow(float p_skip_1)
    {
        pw[0] = 1.0;
        for (unsigned l = 1; l < 5; ++l) pw[l] = std::pow(static_cast<double>(p_skip_1), static_cast<double>(l));
        tail = (std::pow(static_cast<double>(p_skip_1), 5.0) / (1.0f - p_skip_1)) / 4096u;
    }
};
inline float trans_prob(unsigned i, unsigned j, float p_stay, float p_step, const TransPow& T)
{
    float p = 0;
    if (i == j) p += p_stay;
    if (Kmer6::suffix(i, 5) == Kmer6::prefix(j, 5)) p += p_step / 4;
    for (unsigned l = 2; l < 6; ++l)
        if (Kmer6::suffix(i, 6 - l) == Kmer6::prefix(j, 6 - l)) p += T.pw[l - 1] / (1u << (2 * l));
    p += T.tail;
    return p;
}

// compute_transitions_fast :198-202
inline void step_params(float p_skip, float p_stay, float& p_step, float& p_skip_1)
{
    p_step = static_cast<float>(1.0 - p_stay - p_skip);
    p_skip_1 = static_cast<float>(p_skip / (p_skip + 1.0));
}

}  // namespace nchmm

// ---- EM rounds on two lanes (nchmm_api.cpp; used by nchmm_train.cpp, which is compiled without HIP: no runtime types here) ----
// A context holds two sets of what an EM round uses on the device (nchmm_ctx.hpp: EmLaneRes); every function that queues
// forward-backward work runs on the selected one.  With `async` on, table uploads and rounds queue their work on the selected
// lane and return; the lane's pinned arena is handed out piecewise until em_lane_rewind (after the lane has been waited for).
struct nchmm_ctx;
namespace nchmm {
struct EmPending { size_t n_win = 0, n_jobs = 0; const float* h_lpd = nullptr; const float* h_st = nullptr; const double* h_acc = nullptr; void* stream = nullptr; };
// make the second lane (first use): `pin_bytes` of arena on both, the alpha rows of `events_both` events with lane 1's share behind
// the first `events_lane0`; lane 0 selected, nothing in flight
int em_lanes_prepare(nchmm_ctx* c, size_t pin_bytes, size_t events_lane0, size_t events_both);
void em_lanes_end(nchmm_ctx* c);                             // async off, both lanes waited for, lane 0 selected
void em_lane_select(nchmm_ctx* c, int lane);                 // the context's forward-backward fields become those of `lane` (0 / 1)
void em_lanes_async(nchmm_ctx* c, bool on);
void em_lane_rewind(nchmm_ctx* c);                           // the selected lane's arena from its start again
int em_lanes_wait(nchmm_ctx* c);                             // everything queued on either lane has finished
size_t em_round_pin_bytes(size_t n_win, size_t n_jobs);      // arena one round with its table uploads takes
size_t em_fb_cap_events(nchmm_ctx* c);                       // events whose alpha rows one launch may hold (the forward-backward budget)
// One round of nchmm_em_round queued on the selected lane: inputs copied to its arena, results land there ...
int em_round_enqueue(nchmm_ctx* c, size_t n_win, const uint64_t* win_src, const uint32_t* win_len, const float* win_drift, const float* win_pm,
                     const int32_t* scaled_slot, const int32_t* trans_slot, const float* st_params, size_t n_jobs, const uint32_t* job_first_win,
                     int train_drift, EmPending* pend);
// ... and copied out once the round's stream has been waited for.
int em_round_collect(nchmm_ctx* c, const EmPending& pend, float* out_lpd, float* out_st, double* out_acc);
}  // namespace nchmm
#endif
