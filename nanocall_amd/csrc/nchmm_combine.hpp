// nchmm_combine.hpp -- many host threads, one strand per call, combined into batched launches.
//
// The reference decodes one strand per call from a pfor worker thread and hides the latency of a call behind the other
// workers (basecall_strand, nanocall.cpp:645-690, inside the pfor of :611-621).  A GPU wants hundreds of reads per launch.
// StrandCombiner lets the call site stay as it is: every caller puts its strand (model image, transition parameters,
// events) into the staging arrays of the batch that is currently open and sleeps; the first caller of a batch leads it:
// it waits for the batch before it to finish on the device -- that is how long a batch collects callers -- closes it,
// runs it as ONE batched decode, and wakes the others, who copy their own results out.  Two batches alternate: one
// collects while the other runs.  With T calling threads about T/2 strands go into every launch.
//
// The device work is a callback (Runner) so that the synchronisation can be exercised without a device, under the
// sanitizers (tools/asan_host.cpp).  Staging memory comes from the Runner too (pinned host memory in the library).
#ifndef NCHMM_COMBINE_HPP
#define NCHMM_COMBINE_HPP

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>

namespace nchmm {

constexpr size_t kImageFloats = 8 * 4096;   // device image of one scaled pore model (nchmm_device.h: kModelFloats)

struct CombineBatch {
    // staging (Runner::alloc): capacity in strands / events
    size_t cap_reads = 0, cap_events = 0;
    float* images = nullptr;      // [cap_reads][kImageFloats]
    int32_t* fast = nullptr;      // [cap_reads]
    float* p_skip = nullptr;      // [cap_reads]
    float* p_stay = nullptr;      // [cap_reads]
    uint64_t* off = nullptr;      // [cap_reads + 1]
    float* cm = nullptr;          // [cap_events]
    float* sd = nullptr;
    float* ls = nullptr;
    uint16_t* states = nullptr;   // [cap_events]
    float* logp = nullptr;        // [cap_reads]
    int32_t* status = nullptr;    // [cap_reads]
    // state, under StrandCombiner::m_ (ready / consumed: atomic, written outside it; done / rc: under dm)
    size_t n = 0, total = 0;
    std::atomic<size_t> ready{0}, consumed{0};
    bool has_leader = false, closed = false, busy = false;   // busy: closed and not yet given back by its last caller
    // completion has its own lock: the hundreds of callers that wake up when a batch is done do not queue on m_, where the
    // next batch is being joined and led
    std::mutex dm;
    std::condition_variable dcv;
    bool done = false;
    int rc = 0;
};

// Runner: int alloc(CombineBatch&, size_t reads, size_t events)   (re)allocate the staging arrays of an EMPTY batch
//         void release(CombineBatch&)
//         int run(CombineBatch&)                                   decode strands [0, n): fills states / logp / status
template <typename Runner>
class StrandCombiner {
public:
    StrandCombiner(Runner r, size_t max_reads, size_t max_events, unsigned linger_us)
        : runner_(r), max_reads_(max_reads ? max_reads : 1), max_events_(max_events ? max_events : 1), linger_us_(linger_us) {}
    ~StrandCombiner() { for (CombineBatch& B : b_) runner_.release(B); }
    StrandCombiner(const StrandCombiner&) = delete;
    StrandCombiner& operator=(const StrandCombiner&) = delete;

    // One strand.  fill_image(dst, &fast) writes the kImageFloats of the strand's model.  Returns the batch's error code, or the
    // strand's own status (0 / negative) when the batch ran.  Thread-safe; blocks until the strand is decoded.
    template <typename FillImage>
    int submit(FillImage&& fill_image, float p_skip, float p_stay, size_t n_events, const float* cm, const float* sd, const float* ls,
               uint16_t* out_state, float* out_logp)
    {
        std::unique_lock<std::mutex> lk(m_);
        CombineBatch* B = nullptr;
        for (;;) {
            B = &b_[open_];
            if (!B->closed && !B->busy) {
                if (B->n == 0 && (B->cap_reads < max_reads_ || B->cap_events < std::max(max_events_, n_events))) {
                    // an empty batch: nobody reads its staging, (re)size it here (a strand longer than the batch still goes: alone)
                    const int rc = runner_.alloc(*B, max_reads_, std::max(max_events_, n_events));
                    if (rc != 0) return rc;
                }
                if (B->n < B->cap_reads && B->total + n_events <= B->cap_events) break;
                if (B->n == 0) return -1;   // (cannot happen: an empty batch was just sized for this strand)
                if (B->has_leader) full_.notify_all();   // full: its leader need not linger any longer
            }
            cv_.wait(lk);    // until the open batch changes or is free again
        }
        const size_t idx = B->n++, at = B->total;
        B->total += n_events;
        B->off[idx] = at; B->off[idx + 1] = B->total;
        const bool lead = !B->has_leader;
        B->has_leader = true;
        lk.unlock();

        // every caller stages its own strand (in parallel with the others)
        B->fast[idx] = 1;
        fill_image(B->images + idx * kImageFloats, &B->fast[idx]);
        B->p_skip[idx] = p_skip; B->p_stay[idx] = p_stay;
        if (n_events) {
            std::memcpy(B->cm + at, cm, n_events * sizeof(float));
            std::memcpy(B->sd + at, sd, n_events * sizeof(float));
            std::memcpy(B->ls + at, ls, n_events * sizeof(float));
        }
        B->ready.fetch_add(1, std::memory_order_release);

        if (lead) {
            std::unique_lock<std::mutex> run(run_m_);      // the batch before this one is on the device: meanwhile callers join
            lk.lock();
            if (B->n < B->cap_reads && linger_us_)          // nothing to wait behind (or it was quick): give concurrent callers a moment
                // (system_clock: pthread_cond_timedwait, which ThreadSanitizer can follow -- steady_clock waits go through
                // pthread_cond_clockwait, which gcc 11's does not intercept; a clock step only lengthens or shortens one linger)
                full_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(linger_us_), [&] { return B->n >= B->cap_reads; });
            B->closed = B->busy = true;
            open_ ^= 1;
            const size_t n = B->n;
            lk.unlock();
            cv_.notify_all();                               // callers waiting for an open batch
            while (B->ready.load(std::memory_order_acquire) < n) std::this_thread::yield();
            const int rc = runner_.run(*B);
            run.unlock();                                   // the next batch's leader first: it has a launch to get going
            { std::lock_guard<std::mutex> g(B->dm); B->rc = rc; B->done = true; }
            B->dcv.notify_all();
        } else {
            std::unique_lock<std::mutex> dl(B->dm);
            B->dcv.wait(dl, [&] { return B->done; });
        }
        // every caller takes its own results
        int rc = B->rc;
        if (rc == 0) {
            if (n_events) std::memcpy(out_state, B->states + at, n_events * sizeof(uint16_t));
            *out_logp = B->logp[idx];
            rc = B->status[idx];
        }
        if (B->consumed.fetch_add(1, std::memory_order_acq_rel) + 1 == B->n) {   // (n is final: the batch is closed)
            // the last caller out gives the batch back
            { std::lock_guard<std::mutex> g(B->dm); B->done = false; B->rc = 0; }
            lk.lock();
            B->n = B->total = 0;
            B->ready.store(0, std::memory_order_relaxed);
            B->consumed.store(0, std::memory_order_relaxed);
            B->has_leader = B->closed = B->busy = false;
            lk.unlock();
            cv_.notify_all();
        }
        return rc;
    }

private:
    Runner runner_;
    const size_t max_reads_, max_events_;
    const unsigned linger_us_;
    std::mutex m_, run_m_;
    std::condition_variable cv_, full_;
    CombineBatch b_[2];
    unsigned open_ = 0;
};

}  // namespace nchmm
#endif
