// nchmm_hostpool.cpp -- the host worker pool behind nchmm::parallel_for (declared in nchmm_internal.hpp).
//
// The reference spreads its host work over `-t` pfor threads that live for the whole run (nanocall.cpp:282,611).  The
// library's host loops (event prep, per-job EM finish, per-slot transition weights, winner copies) are short -- tens of
// microseconds to a millisecond -- so the workers must already exist when a loop starts.
#include "nchmm_internal.hpp"

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>

#include <sched.h>
#include <unistd.h>

namespace nchmm {

namespace {

struct Pool {
    std::mutex submit;                 // one parallel loop at a time; a second caller runs its loop itself
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    void (*fn)(void*, unsigned) = nullptr;
    void* arg = nullptr;
    unsigned n_chunks = 0;
    std::atomic<unsigned> next{0};
    unsigned busy = 0;                 // workers that have not yet finished the current generation
    unsigned long long gen = 0;
    unsigned n_workers = 0;
    pid_t owner = 0;
};

Pool* g_pool = nullptr;
std::once_flag g_once;

void worker(Pool* P)
{
    unsigned long long seen = 0;
    for (;;) {
        std::unique_lock<std::mutex> lk(P->m);
        P->cv_work.wait(lk, [&] { return P->gen != seen; });
        seen = P->gen;
        void (*fn)(void*, unsigned) = P->fn;
        void* arg = P->arg;
        const unsigned n = P->n_chunks;
        lk.unlock();
        for (unsigned i; (i = P->next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(arg, i);
        lk.lock();
        if (--P->busy == 0) P->cv_done.notify_one();
    }
}

unsigned usable_cpus()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int n = CPU_COUNT(&set);
        if (n > 0) return (unsigned)n;
    }
    const unsigned hc = std::thread::hardware_concurrency();
    return hc ? hc : 4;
}

}  // namespace

unsigned host_threads()
{
    static const unsigned n = [] {
        if (const char* e = std::getenv("NCHMM_HOST_THREADS")) {
            const long v = std::atol(e);
            if (v >= 1) return (unsigned)std::min<long>(v, 256);
        }
        return std::min<unsigned>(usable_cpus(), 32);
    }();
    return n;
}

void run_chunks(unsigned n_chunks, void (*fn)(void*, unsigned), void* arg)
{
    if (n_chunks == 0) return;
    const unsigned nt = host_threads();
    if (n_chunks == 1 || nt < 2) {
        for (unsigned i = 0; i < n_chunks; ++i) fn(arg, i);
        return;
    }
    std::call_once(g_once, [&] {
        // never destroyed: the workers are detached and sleep in cv_work for the life of the process
        Pool* P = new Pool();
        P->owner = getpid();
        P->n_workers = nt - 1;
        for (unsigned i = 0; i < P->n_workers; ++i) std::thread(worker, P).detach();
        g_pool = P;
    });
    Pool* P = g_pool;
    // a forked child has no workers; a second concurrent loop (one host thread per device) does not queue behind the first
    if (P->owner != getpid() || !P->submit.try_lock()) {
        for (unsigned i = 0; i < n_chunks; ++i) fn(arg, i);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(P->m);
        P->fn = fn; P->arg = arg; P->n_chunks = n_chunks;
        P->next.store(0, std::memory_order_relaxed);
        P->busy = P->n_workers;
        ++P->gen;
    }
    P->cv_work.notify_all();
    for (unsigned i; (i = P->next.fetch_add(1, std::memory_order_relaxed)) < n_chunks;) fn(arg, i);
    {
        std::unique_lock<std::mutex> lk(P->m);
        P->cv_done.wait(lk, [&] { return P->busy == 0; });
    }
    P->submit.unlock();
}

}  // namespace nchmm
