// nchmm_combine.hpp -- many host threads, one small piece of work per call, combined into batched launches.
//
// The reference decodes one strand per call from a pfor worker thread (basecall_strand, nanocall.cpp:645-690, inside the pfor
// of :611-621) and trains one read per call (train_one_round, Parameter_Trainer.hpp:541-579, inside the pfor of
// nanocall.cpp:282-579), hiding the latency of a call behind the other workers.  A GPU wants hundreds of reads per launch.
// BatchCombiner lets the call sites stay as they are: every caller puts its piece (model images, transition parameters,
// events) into the staging arrays of the batch that is currently open and sleeps; the first caller of a batch leads it:
// it waits for the batch before it to come off the device -- that is how long a batch collects callers -- closes it,
// runs it as ONE batched call, and wakes the others, who copy their own results out.  Two batches alternate: one
// collects while the other runs.  With T calling threads about T/2 pieces go into every launch.
//
// The device work is a callback (Runner) so that the synchronisation can be exercised without a device, under the
// sanitizers (tools/asan_host.cpp).  Staging memory comes from the Runner too (pinned host memory in the library).
#ifndef NCHMM_COMBINE_HPP
#define NCHMM_COMBINE_HPP

#include <algorithm>
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
constexpr int kCombineDims = 3;             // what a batch has room for: items (strands / windows), events, model images

// What every batch carries whatever its payload.  State under BatchCombiner::m_ unless stated.
struct CombineCore {
    size_t cap[kCombineDims] = {0, 0, 0}, used[kCombineDims] = {0, 0, 0};
    size_t callers = 0;
    std::atomic<size_t> ready{0}, consumed{0};                // written outside m_
    bool has_leader = false, closed = false;                  // closed: from the leader's close until its last caller has left
    bool resizing = false;                                    // its staging is being re-allocated by one caller, m_ released
    // completion has its own lock: the hundreds of callers that wake up when a batch is done do not queue on m_, where the
    // next batch is being joined and led
    std::mutex dm;
    std::condition_variable dcv;
    bool done = false;
    int rc = 0;
};

struct CombinePos { size_t at[kCombineDims]; };

// Batch: derives from CombineCore and holds the staging arrays.
// Runner: int alloc(Batch&, const size_t cap[3])   (re)allocate the staging arrays of an EMPTY batch and set its cap[]
//         void release(Batch&)
//         int run(Batch&)                           do the batch: used[] says how much of it is filled
template <typename Batch, typename Runner>
class BatchCombiner {
public:
    BatchCombiner(Runner r, const size_t max_cap[kCombineDims], unsigned linger_us) : runner_(r), linger_us_(linger_us)
    {
        for (int d = 0; d < kCombineDims; ++d) { max_[d] = max_cap[d] ? max_cap[d] : 1; want_[d] = std::max<size_t>(max_[d] / 64, 1); }
    }
    ~BatchCombiner() { for (Batch& B : b_) runner_.release(B); }
    BatchCombiner(const BatchCombiner&) = delete;
    BatchCombiner& operator=(const BatchCombiner&) = delete;

    // One caller's piece: `need` of each dimension.  stage(batch, pos) writes it into the batch at pos (called without the
    // lock, in parallel with the other callers); collect(batch, pos) -> rc copies the caller's results out after the batch
    // has run.  Returns the batch's error code if it failed, else what collect returned.  Thread-safe; blocks until done.
    template <typename Stage, typename Collect>
    int submit(const size_t need[kCombineDims], Stage&& stage, Collect&& collect)
    {
        std::unique_lock<std::mutex> lk(m_);
        Batch* B = nullptr;
        for (;;) {
            B = &b_[open_];
            if (B->resizing) { cv_.wait(lk); continue; }     // (another caller is re-allocating its staging, outside the lock)
            if (!B->closed) {
                // the staging of a batch follows demand: it starts at 1/64 of the capacity (one caller decoding one strand at a
                // time never needs more) and doubles whenever a batch was found full, up to max_
                bool small = false;
                for (int d = 0; d < kCombineDims; ++d) small = small || B->cap[d] < std::max(want_[d], need[d]);
                if (B->callers == 0 && small) {
                    // an empty batch: nobody reads its staging, (re)size it -- with the lock released (pinned allocations of this
                    // size take milliseconds; callers that arrive meanwhile wait above).  A piece larger than a batch still goes: alone
                    size_t cap[kCombineDims];
                    for (int d = 0; d < kCombineDims; ++d) cap[d] = std::max(want_[d], need[d]);
                    B->resizing = true;
                    lk.unlock();
                    const int rc = runner_.alloc(*B, cap);
                    lk.lock();
                    B->resizing = false;
                    cv_.notify_all();
                    if (rc != 0) return rc;
                    continue;
                }
                bool fits = true;
                for (int d = 0; d < kCombineDims; ++d) fits = fits && B->used[d] + need[d] <= B->cap[d];
                if (fits) break;
                if (B->callers == 0) return -1;          // (cannot happen: an empty batch was just sized for this piece)
                for (int d = 0; d < kCombineDims; ++d)   // the next empty batch is sized for more
                    if (B->used[d] + need[d] > B->cap[d]) want_[d] = std::min(max_[d], std::max(want_[d] * 2, B->used[d] + need[d]));
                full_batch_ = B;                         // no room: its leader need not linger any longer
                full_.notify_all();
            }
            cv_.wait(lk);    // until the open batch changes or is free again
        }
        CombinePos pos;
        for (int d = 0; d < kCombineDims; ++d) { pos.at[d] = B->used[d]; B->used[d] += need[d]; }
        B->callers += 1;
        const bool lead = !B->has_leader;
        B->has_leader = true;
        lk.unlock();

        stage(*B, pos);      // every caller stages its own piece
        B->ready.fetch_add(1, std::memory_order_release);

        if (lead) {
            std::unique_lock<std::mutex> run(run_m_);      // the batch before this one is on the device: meanwhile callers join
            lk.lock();
            // nothing to wait behind (or it was quick): give concurrent callers a moment -- unless the batch before this one had a
            // single caller too (one thread decoding strand after strand, the reference's default -t 1: nobody else will come)
            if (full_batch_ != B && linger_us_ && (last_callers_ > 1 || B->callers > 1))
                // (system_clock: pthread_cond_timedwait, which ThreadSanitizer can follow -- steady_clock waits go through
                // pthread_cond_clockwait, which gcc 11's does not intercept; a clock step only lengthens or shortens one linger)
                full_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(linger_us_), [&] { return full_batch_ == B; });
            if (full_batch_ == B) full_batch_ = nullptr;
            B->closed = true;
            open_ ^= 1;
            const size_t n = B->callers;
            last_callers_ = n;
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
        int rc = B->rc;
        if (rc == 0) rc = collect(*B, pos);                 // every caller takes its own results
        if (B->consumed.fetch_add(1, std::memory_order_acq_rel) + 1 == B->callers) {   // (callers is final: the batch is closed)
            // the last caller out gives the batch back
            { std::lock_guard<std::mutex> g(B->dm); B->done = false; B->rc = 0; }
            lk.lock();
            for (int d = 0; d < kCombineDims; ++d) B->used[d] = 0;
            B->callers = 0;
            B->ready.store(0, std::memory_order_relaxed);
            B->consumed.store(0, std::memory_order_relaxed);
            B->has_leader = B->closed = false;
            lk.unlock();
            cv_.notify_all();
        }
        return rc;
    }

private:
    Runner runner_;
    size_t max_[kCombineDims];
    size_t want_[kCombineDims];      // what an empty batch is sized for now (<= max_; under m_)
    const unsigned linger_us_;
    std::mutex m_, run_m_;
    std::condition_variable cv_, full_;
    Batch b_[2];
    Batch* full_batch_ = nullptr;    // the open batch somebody found no room in (its leader stops lingering)
    unsigned open_ = 0;
    size_t last_callers_ = 2;        // callers of the batch closed last (the first batch lingers)
};

// ---- payload of nchmm_viterbi_strand: one strand = one item, its events, one model image ----
// A strand's model comes either as its device image (built by the caller from the scaled S x 6 table) or, when the caller knows
// where it came from, as (unscaled S x 10 states, scaling parameters): 32 bytes instead of 128 KiB, scaled on the device.
struct StrandBatch : CombineCore {
    float* images = nullptr;      // [cap[2]][kImageFloats]   (strands with base[k] == nullptr)
    int32_t* fast = nullptr;      // [cap[2]]
    const float** base = nullptr; // [cap[0]]  unscaled states of strand k's model, or nullptr
    float* scale6 = nullptr;      // [cap[0]][6]
    float* p_skip = nullptr;      // [cap[0]]
    float* p_stay = nullptr;
    uint64_t* off = nullptr;      // [cap[0] + 1]
    float* cm = nullptr;          // [cap[1]]
    float* sd = nullptr;
    float* ls = nullptr;
    uint16_t* states = nullptr;   // [cap[1]]
    float* logp = nullptr;        // [cap[0]]
    int32_t* status = nullptr;    // [cap[0]]
};

template <typename Runner>
class StrandCombiner {
public:
    StrandCombiner(Runner r, size_t max_reads, size_t max_events, unsigned linger_us) : bc_(r, caps(max_reads, max_events).c, linger_us) {}
    // fill_image(dst, &fast) writes the kImageFloats of the strand's model.  Returns the batch's error code, or the strand's own
    // status (0 / negative) when the batch ran.
    // base / scale6: the model as (unscaled states, parameters) instead (fill_image is then not called); nullptr: by image.
    template <typename FillImage>
    int submit(FillImage&& fill_image, float p_skip, float p_stay, size_t n_events, const float* cm, const float* sd, const float* ls,
               uint16_t* out_state, float* out_logp, const float* base = nullptr, const float* scale6 = nullptr)
    {
        const size_t need[kCombineDims] = {1, n_events, 1};
        return bc_.submit(need,
            [&](StrandBatch& B, const CombinePos& p) {
                const size_t idx = p.at[0], at = p.at[1];
                B.off[idx + 1] = at + n_events;      // (off[0] = 0 from alloc; strand idx starts where strand idx-1 ends)
                B.fast[idx] = 1;
                B.base[idx] = base;
                if (base) std::memcpy(B.scale6 + 6 * idx, scale6, 6 * sizeof(float));
                else fill_image(B.images + idx * kImageFloats, &B.fast[idx]);
                B.p_skip[idx] = p_skip; B.p_stay[idx] = p_stay;
                if (n_events) {
                    std::memcpy(B.cm + at, cm, n_events * sizeof(float));
                    std::memcpy(B.sd + at, sd, n_events * sizeof(float));
                    std::memcpy(B.ls + at, ls, n_events * sizeof(float));
                }
            },
            [&](StrandBatch& B, const CombinePos& p) {
                if (n_events) std::memcpy(out_state, B.states + p.at[1], n_events * sizeof(uint16_t));
                *out_logp = B.logp[p.at[0]];
                return (int)B.status[p.at[0]];
            });
    }
private:
    struct Caps { size_t c[kCombineDims]; };
    static Caps caps(size_t reads, size_t events) { return Caps{{reads ? reads : 1, events ? events : 1, reads ? reads : 1}}; }
    BatchCombiner<StrandBatch, Runner> bc_;
};

// ---- payload of nchmm_fwbw_windows: a read's training windows = items, their events, one model per strand ----
// A model is (unscaled table, scaling parameters): 32 bytes where its image would be 128 KiB against the 400 events a read
// trains on -- the images are built on the device (nchmm_put_models_scaled), the distinct unscaled tables of a batch (a handful)
// uploaded once.
struct WindowBatch : CombineCore {
    const float** base = nullptr; // [cap[2]]  unscaled S x 10 states of model slot k (the caller's, valid while it is in the call)
    float* scale6 = nullptr;      // [cap[2]][6]  Pore_Model_Parameters the slot's model is scaled by
    float* p_skip = nullptr;      // [cap[2]]  transitions of model slot k = compute_transitions_fast(p_skip[k], p_stay[k])
    float* p_stay = nullptr;
    uint64_t* off = nullptr;      // [cap[0] + 1]
    int32_t* slot = nullptr;      // [cap[0]]  model (= transition) slot of a window
    float* pm_params = nullptr;   // [cap[0]][6]
    float* st_params = nullptr;   // [cap[0]][2]  {p_stay, p_skip}
    float* cm = nullptr;          // [cap[1]]
    float* sd = nullptr;
    float* ls = nullptr;
    float* lpd = nullptr;         // [cap[0]]
    float* pm_sums = nullptr;     // [cap[1]][6]
    float* st_sums = nullptr;     // [cap[0]][3]
};

}  // namespace nchmm
#endif
