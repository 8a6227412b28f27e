// nchmm_hostpool.cpp -- the host worker pool behind nchmm::parallel_for (declared in nchmm_internal.hpp).
//
// The reference spreads its host work over `-t` pfor threads that live for the whole run (nanocall.cpp:282,611).  The
// library's host loops (event prep, per-job EM finish, per-slot transition weights, winner copies) are short -- tens of
// microseconds to a millisecond -- so the workers must already exist when a loop starts; and on an 8-GPU node eight device
// threads (nchmm_pool.cpp) run such loops at the same time, so the pool serves several loops at once: a loop is a JOB in one
// of a few slots, workers take chunks from whichever jobs are open, the thread that posted a job works on it too and waits
// only for its own chunks.
#include "nchmm_internal.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include <sched.h>
#include <unistd.h>

namespace nchmm {

namespace {

constexpr unsigned kJobSlots = 16;

struct Job {
    void (*fn)(void*, unsigned) = nullptr;
    void* arg = nullptr;
    unsigned n_chunks = 0;
    std::atomic<unsigned> next{0};     // next chunk to hand out
    std::atomic<unsigned> done{0};     // chunks finished
    std::atomic<bool> open{false};     // chunks may still be taken
    std::atomic<unsigned> inside{0};   // workers currently looking at this slot (it is not reused while any is)
};

struct Pool {
    std::mutex m;                      // guards slot allocation and the sleep / wake-up of workers
    std::condition_variable cv_work;
    Job job[kJobSlots];
    bool used[kJobSlots] = {};
    unsigned long long posted = 0;     // bumped under m whenever a job opens
    unsigned n_workers = 0;
    pid_t owner = 0;
};

Pool* g_pool = nullptr;
std::once_flag g_once;

// take chunks of job J until none are left; returns how many this thread ran
unsigned drain(Job& J)
{
    unsigned ran = 0;
    const unsigned n = J.n_chunks;
    for (unsigned i; (i = J.next.fetch_add(1, std::memory_order_acq_rel)) < n;) {
        J.fn(J.arg, i);
        J.done.fetch_add(1, std::memory_order_acq_rel);
        ++ran;
    }
    return ran;
}

void worker(Pool* P)
{
    unsigned long long seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(P->m);
            P->cv_work.wait(lk, [&] { return P->posted != seen; });
            seen = P->posted;
        }
        // serve every open job; re-scan while any is open (a job may be posted while this thread is busy)
        for (bool any = true; any;) {
            any = false;
            for (unsigned s = 0; s < kJobSlots; ++s) {
                Job& J = P->job[s];
                if (!J.open.load()) continue;
                // announce, THEN look again: the poster closes the job and then waits for `inside` to drop to zero, so either
                // this thread sees the job closed or the poster sees this thread (both sequentially consistent)
                J.inside.fetch_add(1);
                if (J.open.load() && J.next.load(std::memory_order_relaxed) < J.n_chunks) any = drain(J) != 0 || any;
                J.inside.fetch_sub(1);
            }
        }
    }
}

// CPUs' worth of time the process's cgroup allows per period (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us); 0 = no limit.
// A container may show 256 CPUs in its affinity mask and be allowed the time of 16: threads beyond the quota only get throttled
// (the test boxes of this repository are such containers: 128 spinning processes do LESS work per second there than 32).
unsigned cgroup_cpu_quota()
{
    auto read2 = [](const char* path, long long* a, long long* b) {
        FILE* f = std::fopen(path, "r");
        if (!f) return 0;
        char tok[64] = {0};
        int n = 0;
        if (std::fscanf(f, "%63s", tok) == 1) {
            if (std::strcmp(tok, "max") == 0) { *a = -1; n = 1; }
            else { *a = std::atoll(tok); n = 1; }
            if (b && std::fscanf(f, "%lld", b) == 1) n = 2;
        }
        std::fclose(f);
        return n;
    };
    long long quota = -1, period = 100000;
    if (read2("/sys/fs/cgroup/cpu.max", &quota, &period) >= 1) {
        if (quota > 0 && period > 0) return (unsigned)((quota + period - 1) / period);
        return 0;
    }
    if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &quota, nullptr) >= 1 && quota > 0
        && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &period, nullptr) >= 1 && period > 0)
        return (unsigned)((quota + period - 1) / period);
    return 0;
}

unsigned usable_cpus()
{
    unsigned n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
    if (!n) n = std::thread::hardware_concurrency();
    if (!n) n = 4;
    const unsigned q = cgroup_cpu_quota();
    return q ? std::min(n, q) : n;
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
    int slot = -1;
    if (P->owner == getpid()) {         // (a forked child has no workers)
        std::lock_guard<std::mutex> lk(P->m);
        for (unsigned s = 0; s < kJobSlots; ++s)
            if (!P->used[s]) { slot = (int)s; P->used[s] = true; break; }
        if (slot >= 0) {
            Job& J = P->job[slot];
            J.fn = fn; J.arg = arg; J.n_chunks = n_chunks;
            J.next.store(0, std::memory_order_relaxed);
            J.done.store(0, std::memory_order_relaxed);
            J.open.store(true);
            ++P->posted;
        }
    }
    if (slot < 0) {                     // every slot taken (or a forked child): run the loop here
        for (unsigned i = 0; i < n_chunks; ++i) fn(arg, i);
        return;
    }
    P->cv_work.notify_all();
    Job& J = P->job[slot];
    drain(J);
    // chunks handed to workers are short: spin politely until the last one has finished
    while (J.done.load(std::memory_order_acquire) < n_chunks) std::this_thread::yield();
    J.open.store(false);
    while (J.inside.load() != 0) std::this_thread::yield();     // nobody may still hold the old n_chunks when the slot is reused
    {
        std::lock_guard<std::mutex> lk(P->m);
        P->used[slot] = false;
    }
}

}  // namespace nchmm
