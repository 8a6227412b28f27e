// nchmm_pool.cpp -- one context + one host thread per GPU; reads sharded by event count (LPT), no data-path
// collective; the per-device counters are summed with one RCCL all-reduce (include/nanocall_hip.h "Device pool").
#include "nanocall_hip.h"

#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <limits>
#include <numeric>
#include <set>
#include <thread>
#include <vector>

// One host thread per device, alive as long as the pool: the batched entry points keep "only grows" host staging with
// their calling thread (thread_local), so a thread spawned per call would allocate and page-fault it again for every chunk
// of a run -- the cost that buffer exists to avoid.  run_per_device hands each worker its closure and waits for all.
struct Device_Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> task;     // set by the caller, cleared by the worker
    int rc = 0;
    bool has_task = false, done = false, quit = false;

    void loop()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_task || quit; });
            if (quit) return;
            std::function<int()> f = std::move(task);
            lk.unlock();
            int r;
            try { r = f(); } catch (...) { r = NCHMM_E_NOMEM; }     // std::bad_alloc is what can come out of the staging code
            lk.lock();
            rc = r; has_task = false; done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> f)
    {
        { std::lock_guard<std::mutex> g(m); task = std::move(f); has_task = true; done = false; }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    void stop()
    {
        { std::lock_guard<std::mutex> g(m); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct nchmm_pool {
    std::vector<int> device;
    std::vector<nchmm_ctx*> ctx;
    std::vector<std::unique_ptr<Device_Worker>> worker;     // one per context when there are two or more
    std::mutex call_mutex;                                  // one batched call at a time per pool
};

namespace {

// librccl, loaded on first use: the library must not depend on it at link time (a 1-GPU run never needs it, and a
// host process such as Python may carry its own copy)
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;      // one process per GPU (nchmm_counters_allreduce)
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    bool ok = false, ok_ranks = false;
    Rccl()
    {
        // RCCL writes its version banner / warnings to stdout unless told otherwise; a host program may be streaming
        // FASTA there (nanocall without -o), so send them to stderr -- unless the user chose a file
        setenv("NCCL_DEBUG_FILE", "/dev/stderr", 0);
        // NCHMM_RCCL_LIB names the library to load instead of the usual places (a site's own build; the tests point it at a
        // missing file and at a library whose ncclCommInitAll fails, to walk the fall-back below)
        if (const char* e = std::getenv("NCHMM_RCCL_LIB")) {
            handle = dlopen(e, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (handle) break;
            }
        }
        if (!handle) return;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(handle, "ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(handle, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(handle, "ncclCommInitRank"));
        ok = CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd;
        ok_ranks = GetUniqueId && CommInitRank && CommDestroy && AllReduce;
    }
};

Rccl& rccl()
{
    static Rccl r;
    return r;
}

// RCCL 2.27 prints a version banner on stdout from ncclCommInitAll whatever NCCL_DEBUG says.  A host program may be
// streaming FASTA on stdout (nanocall without -o): while the communicators are created, fd 1 points at stderr.
struct Stdout_To_Stderr {
    int saved;
    Stdout_To_Stderr() { std::fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
    ~Stdout_To_Stderr() { std::fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
};

// sum of eight uint64 per device over the devices, through RCCL; false on any failure (the caller falls back)
bool rccl_sum(const std::vector<int>& dev, std::vector<uint64_t>& per_dev /* 8 per device, in/out: every slot gets the sum */)
{
    Rccl& R = rccl();
    if (!R.ok) return false;
    const int n = (int)dev.size();
    std::vector<ncclComm_t> comm((size_t)n);
    {
        Stdout_To_Stderr guard;
        if (R.CommInitAll(comm.data(), n, dev.data()) != ncclSuccess) return false;
    }
    std::vector<uint64_t*> buf((size_t)n, nullptr);
    std::vector<hipStream_t> st((size_t)n, nullptr);
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
        ok = hipSetDevice(dev[i]) == hipSuccess && hipMalloc((void**)&buf[i], 8 * sizeof(uint64_t)) == hipSuccess
             && hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) == hipSuccess
             && hipMemcpyAsync(buf[i], &per_dev[8 * (size_t)i], 8 * sizeof(uint64_t), hipMemcpyHostToDevice, st[i]) == hipSuccess;
    }
    if (ok) {
        ok = R.GroupStart() == ncclSuccess;
        for (int i = 0; i < n && ok; ++i) ok = R.AllReduce(buf[i], buf[i], 8, ncclUint64, ncclSum, comm[i], st[i]) == ncclSuccess;
        ok = (R.GroupEnd() == ncclSuccess) && ok;
    }
    for (int i = 0; i < n && ok; ++i)
        ok = hipSetDevice(dev[i]) == hipSuccess
             && hipMemcpyAsync(&per_dev[8 * (size_t)i], buf[i], 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, st[i]) == hipSuccess
             && hipStreamSynchronize(st[i]) == hipSuccess;
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(dev[i]);
        if (st[i]) (void)hipStreamDestroy(st[i]);
        if (buf[i]) (void)hipFree(buf[i]);
        R.CommDestroy(comm[i]);
    }
    return ok;
}

// The shard of one device: its reads (ascending caller index), their events gathered into contiguous arrays, and the
// jobs of those reads with read indices renumbered.
struct Shard {
    std::vector<size_t> reads;
    std::vector<uint64_t> strand_off;            // 2 * reads + 1
    std::vector<float> mean, stdv, start;
    std::vector<size_t> jobs;                    // caller job indices, ascending
    std::vector<int32_t> job_read, job_m0, job_m1;
};

int build_shards(int n_dev, size_t n_reads, const uint64_t* strand_off, const float* mean, const float* stdv, const float* start,
                 size_t n_jobs, const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1, std::vector<Shard>& sh,
                 std::vector<int32_t>& local_read)
{
    std::vector<uint64_t> w(n_reads);
    for (size_t r = 0; r < n_reads; ++r) {
        if (strand_off[2 * r + 2] < strand_off[2 * r]) return NCHMM_E_INVALID;
        w[r] = strand_off[2 * r + 2] - strand_off[2 * r];
    }
    std::vector<int32_t> of(n_reads);
    int rc = nchmm_lpt_partition(n_reads, w.data(), n_dev, of.data());
    if (rc != NCHMM_OK) return rc;
    sh.assign((size_t)n_dev, Shard());
    local_read.assign(n_reads, -1);
    for (size_t r = 0; r < n_reads; ++r) {
        Shard& s = sh[(size_t)of[r]];
        local_read[r] = (int32_t)s.reads.size();
        s.reads.push_back(r);
    }
    for (size_t k = 0; k < n_jobs; ++k) {
        if (job_read[k] < 0 || (size_t)job_read[k] >= n_reads) return NCHMM_E_INVALID;
        Shard& s = sh[(size_t)of[(size_t)job_read[k]]];
        s.jobs.push_back(k);
        s.job_read.push_back(local_read[(size_t)job_read[k]]);
        s.job_m0.push_back(job_m0[k]);
        s.job_m1.push_back(job_m1[k]);
    }
    for (Shard& s : sh) {
        s.strand_off.assign(2 * s.reads.size() + 1, 0);
        for (size_t i = 0; i < s.reads.size(); ++i) {
            const size_t r = s.reads[i];
            s.strand_off[2 * i + 1] = s.strand_off[2 * i] + (strand_off[2 * r + 1] - strand_off[2 * r]);
            s.strand_off[2 * i + 2] = s.strand_off[2 * i + 1] + (strand_off[2 * r + 2] - strand_off[2 * r + 1]);
        }
        const size_t tot = (size_t)s.strand_off.back();
        s.mean.resize(tot); s.stdv.resize(tot); s.start.resize(tot);
        for (size_t i = 0; i < s.reads.size(); ++i) {
            const size_t r = s.reads[i];
            const size_t src = (size_t)strand_off[2 * r], len = (size_t)(strand_off[2 * r + 2] - strand_off[2 * r]);
            const size_t dst = (size_t)s.strand_off[2 * i];
            std::memcpy(&s.mean[dst], mean + src, len * sizeof(float));
            std::memcpy(&s.stdv[dst], stdv + src, len * sizeof(float));
            std::memcpy(&s.start[dst], start + src, len * sizeof(float));
        }
    }
    return NCHMM_OK;
}

template <typename F>
int run_per_device(nchmm_pool* pool, F&& f)
{
    const size_t n = pool->ctx.size();
    std::vector<int> rc(n, NCHMM_OK);
    if (n == 1) return f(0);
    {
        std::lock_guard<std::mutex> one_call(pool->call_mutex);
        for (size_t d = 0; d < n; ++d) pool->worker[d]->post([&f, d] { return f(d); });
        for (size_t d = 0; d < n; ++d) rc[d] = pool->worker[d]->wait();
    }
    int worst = NCHMM_OK;
    for (int v : rc)
        if (v != NCHMM_OK && (worst == NCHMM_OK || worst == NCHMM_E_NUMERIC)) worst = v;
    return worst;
}

}  // namespace

extern "C" {

int nchmm_device_count(int* n)
{
    if (!n) return NCHMM_E_INVALID;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) { *n = 0; return NCHMM_E_NO_DEVICE; }
    *n = c;
    return NCHMM_OK;
}

int nchmm_device_mem_info(int device_id, uint64_t* free_bytes, uint64_t* total_bytes)
{
    if (!free_bytes || !total_bytes) return NCHMM_E_INVALID;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return NCHMM_E_NO_DEVICE;
    if (device_id < 0 || device_id >= c) return NCHMM_E_INVALID;
    int prev = 0;
    (void)hipGetDevice(&prev);
    size_t f = 0, t = 0;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemGetInfo(&f, &t);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return NCHMM_E_HIP;
    *free_bytes = (uint64_t)f;
    *total_bytes = (uint64_t)t;
    return NCHMM_OK;
}

int nchmm_lpt_partition(size_t n_items, const uint64_t* weight, int n_shards, int32_t* shard_of_item)
{
    if (n_shards < 1 || (n_items && (!weight || !shard_of_item))) return NCHMM_E_INVALID;
    bool equal = true;
    for (size_t i = 1; i < n_items && equal; ++i) equal = weight[i] == weight[0];
    if (equal) {   // contiguous slices with boundaries at floor(k * n / shards), as numpy.linspace(...).astype(int)
        for (int k = 0; k < n_shards; ++k) {
            const size_t a = (size_t)((double)n_items * k / n_shards), b = (size_t)((double)n_items * (k + 1) / n_shards);
            for (size_t i = a; i < b; ++i) shard_of_item[i] = k;
        }
        return NCHMM_OK;
    }
    std::vector<size_t> order(n_items);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight[a] > weight[b]; });
    std::vector<uint64_t> load((size_t)n_shards, 0);
    for (size_t i : order) {
        const size_t k = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
        shard_of_item[i] = (int32_t)k;
        load[k] += weight[i];
    }
    return NCHMM_OK;
}

int nchmm_pool_create(nchmm_pool** out, int n_devices, const int* device_ids)
{
    if (!out || n_devices < 1 || n_devices > 64) return NCHMM_E_INVALID;
    *out = nullptr;
    nchmm_pool* p = new (std::nothrow) nchmm_pool();
    if (!p) return NCHMM_E_NOMEM;
    for (int i = 0; i < n_devices; ++i) {
        const int id = device_ids ? device_ids[i] : i;
        nchmm_ctx* c = nullptr;
        const int rc = nchmm_create(&c, id);
        if (rc != NCHMM_OK) { nchmm_pool_destroy(p); return rc; }
        p->device.push_back(id);
        p->ctx.push_back(c);
    }
    if (p->ctx.size() > 1)
        for (size_t d = 0; d < p->ctx.size(); ++d) {
            p->worker.emplace_back(new Device_Worker());
            Device_Worker* w = p->worker.back().get();
            w->th = std::thread([w] { w->loop(); });
        }
    *out = p;
    return NCHMM_OK;
}

int nchmm_pool_destroy(nchmm_pool* p)
{
    if (!p) return NCHMM_E_INVALID;
    for (auto& w : p->worker) w->stop();
    for (nchmm_ctx* c : p->ctx) nchmm_destroy(c);
    delete p;
    return NCHMM_OK;
}

int nchmm_pool_size(const nchmm_pool* p) { return p ? (int)p->ctx.size() : 0; }

nchmm_ctx* nchmm_pool_ctx(nchmm_pool* p, int i) { return (p && i >= 0 && (size_t)i < p->ctx.size()) ? p->ctx[(size_t)i] : nullptr; }

int nchmm_pool_train_reads(nchmm_pool* pool, const nchmm_train_opts* o, size_t n_models, const float* model_states, size_t n_reads,
                           const uint64_t* strand_off, const float* mean, const float* stdv, const float* start, size_t n_jobs,
                           const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1, float* job_pm, float* job_st,
                           float* job_fit, uint32_t* job_rounds, int32_t* read_preferred)
{
    if (!pool || pool->ctx.empty()) return NCHMM_E_INVALID;
    if (pool->ctx.size() == 1)
        return nchmm_train_reads(pool->ctx[0], o, n_models, model_states, n_reads, strand_off, mean, stdv, start, n_jobs, job_read,
                                 job_m0, job_m1, job_pm, job_st, job_fit, job_rounds, read_preferred);
    if (!strand_off || !mean || !stdv || !start || !job_read || !job_m0 || !job_m1 || !job_pm || !job_st || !job_fit || !job_rounds)
        return NCHMM_E_INVALID;
    std::vector<Shard> sh;
    std::vector<int32_t> local_read;
    int rc = build_shards((int)pool->ctx.size(), n_reads, strand_off, mean, stdv, start, n_jobs, job_read, job_m0, job_m1, sh, local_read);
    if (rc != NCHMM_OK) return rc;
    return run_per_device(pool, [&](size_t d) -> int {
        Shard& s = sh[d];
        const size_t nj = s.jobs.size(), nr = s.reads.size();
        if (nr == 0) return NCHMM_OK;
        std::vector<float> pm(6 * nj), st(4 * nj), fit(nj);
        std::vector<uint32_t> rounds(nj);
        std::vector<int32_t> pref(3 * nr, -1);
        for (size_t k = 0; k < nj; ++k) {
            std::memcpy(&pm[6 * k], job_pm + 6 * s.jobs[k], 6 * sizeof(float));
            std::memcpy(&st[4 * k], job_st + 4 * s.jobs[k], 4 * sizeof(float));
        }
        int r = NCHMM_OK;
        if (nj)
            r = nchmm_train_reads(pool->ctx[d], o, n_models, model_states, nr, s.strand_off.data(), s.mean.data(), s.stdv.data(),
                                  s.start.data(), nj, s.job_read.data(), s.job_m0.data(), s.job_m1.data(), pm.data(), st.data(),
                                  fit.data(), rounds.data(), read_preferred ? pref.data() : nullptr);
        if (r != NCHMM_OK) return r;
        for (size_t k = 0; k < nj; ++k) {      // every job belongs to exactly one shard: disjoint writes
            std::memcpy(job_pm + 6 * s.jobs[k], &pm[6 * k], 6 * sizeof(float));
            std::memcpy(job_st + 4 * s.jobs[k], &st[4 * k], 4 * sizeof(float));
            job_fit[s.jobs[k]] = fit[k];
            job_rounds[s.jobs[k]] = rounds[k];
        }
        if (read_preferred)
            for (size_t i = 0; i < nr; ++i)
                for (int q = 0; q < 3; ++q) {
                    const int32_t v = pref[3 * i + q];
                    read_preferred[3 * s.reads[i] + q] = v < 0 ? -1 : (int32_t)s.jobs[(size_t)v];
                }
        return NCHMM_OK;
    });
}

int nchmm_pool_basecall_reads(nchmm_pool* pool, const nchmm_train_opts* o, size_t n_models, const float* model_states, size_t n_reads,
                              const uint64_t* strand_off, const float* mean, const float* stdv, const float* start, size_t n_jobs,
                              const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1, const float* job_pm,
                              const float* job_st, const int32_t* read_preferred, uint16_t* out_state, int32_t* out_best_job,
                              float* out_best_logp)
{
    if (!pool || pool->ctx.empty()) return NCHMM_E_INVALID;
    if (pool->ctx.size() == 1)
        return nchmm_basecall_reads(pool->ctx[0], o, n_models, model_states, n_reads, strand_off, mean, stdv, start, n_jobs, job_read,
                                    job_m0, job_m1, job_pm, job_st, read_preferred, out_state, out_best_job, out_best_logp);
    if (!strand_off || !mean || !stdv || !start || !job_read || !job_m0 || !job_m1 || !job_pm || !job_st || !out_state || !out_best_job
        || !out_best_logp)
        return NCHMM_E_INVALID;
    std::vector<Shard> sh;
    std::vector<int32_t> local_read;
    int rc = build_shards((int)pool->ctx.size(), n_reads, strand_off, mean, stdv, start, n_jobs, job_read, job_m0, job_m1, sh, local_read);
    if (rc != NCHMM_OK) return rc;
    return run_per_device(pool, [&](size_t d) -> int {
        Shard& s = sh[d];
        const size_t nj = s.jobs.size(), nr = s.reads.size();
        if (nr == 0) return NCHMM_OK;
        std::vector<float> pm(6 * nj), st(4 * nj), logp(2 * nr);
        std::vector<int32_t> pref, best(2 * nr);
        std::vector<uint16_t> states((size_t)s.strand_off.back() + 1);
        for (size_t k = 0; k < nj; ++k) {
            std::memcpy(&pm[6 * k], job_pm + 6 * s.jobs[k], 6 * sizeof(float));
            std::memcpy(&st[4 * k], job_st + 4 * s.jobs[k], 4 * sizeof(float));
        }
        if (read_preferred) {
            // caller job index -> shard job index (the jobs of a shard are in ascending caller order)
            pref.assign(3 * nr, -1);
            for (size_t i = 0; i < nr; ++i)
                for (int q = 0; q < 3; ++q) {
                    const int32_t v = read_preferred[3 * s.reads[i] + q];
                    if (v < 0) continue;
                    const auto it = std::lower_bound(s.jobs.begin(), s.jobs.end(), (size_t)v);
                    if (it == s.jobs.end() || *it != (size_t)v) return NCHMM_E_INVALID;
                    pref[3 * i + q] = (int32_t)(it - s.jobs.begin());
                }
        }
        int r = NCHMM_OK;
        if (nj) {
            r = nchmm_basecall_reads(pool->ctx[d], o, n_models, model_states, nr, s.strand_off.data(), s.mean.data(), s.stdv.data(),
                                     s.start.data(), nj, s.job_read.data(), s.job_m0.data(), s.job_m1.data(), pm.data(), st.data(),
                                     read_preferred ? pref.data() : nullptr, states.data(), best.data(), logp.data());
        } else {
            // a shard whose reads all fell under min_ed_events has reads but no jobs (and then empty, i.e. null, job arrays):
            // nothing to decode -- report what the single-context call reports for a read without candidates
            std::fill(best.begin(), best.end(), -1);
            std::fill(logp.begin(), logp.end(), std::numeric_limits<float>::quiet_NaN());
        }
        if (r != NCHMM_OK && r != NCHMM_E_NUMERIC) return r;
        for (size_t i = 0; i < nr; ++i) {
            const size_t g = s.reads[i];
            for (int q = 0; q < 2; ++q) {
                const int32_t b = best[2 * i + q];
                out_best_job[2 * g + q] = b < 0 ? -1 : (int32_t)s.jobs[(size_t)b];
                out_best_logp[2 * g + q] = logp[2 * i + q];
                if (b >= 0) {   // (undecoded strands keep whatever the caller had there, as the single-context call does)
                    const size_t src = (size_t)s.strand_off[2 * i + q], len = (size_t)(s.strand_off[2 * i + q + 1] - s.strand_off[2 * i + q]);
                    std::memcpy(out_state + strand_off[2 * g + q], &states[src], len * sizeof(uint16_t));
                }
            }
        }
        return r;
    });
}

int nchmm_pool_reserve_fb_workspace(nchmm_pool* pool, size_t events_per_device)
{
    if (!pool || pool->ctx.empty()) return NCHMM_E_INVALID;
    for (nchmm_ctx* c : pool->ctx) {
        const int rc = nchmm_reserve_fb_workspace(c, events_per_device);
        if (rc != NCHMM_OK) return rc;
    }
    return NCHMM_OK;
}

int nchmm_pool_reserve_viterbi_workspace(nchmm_pool* pool, size_t longest_events)
{
    if (!pool || pool->ctx.empty()) return NCHMM_E_INVALID;
    for (nchmm_ctx* c : pool->ctx) {
        const int rc = nchmm_reserve_viterbi_workspace(c, longest_events);
        if (rc != NCHMM_OK) return rc;
    }
    return NCHMM_OK;
}

int nchmm_pool_counters(nchmm_pool* pool, uint64_t out[8], int* used_rccl)
{
    if (!pool || !out) return NCHMM_E_INVALID;
    const size_t n = pool->ctx.size();
    std::vector<uint64_t> per(8 * n);
    for (size_t d = 0; d < n; ++d) {
        const int rc = nchmm_counters(pool->ctx[d], &per[8 * d]);
        if (rc != NCHMM_OK) return rc;
    }
    if (used_rccl) *used_rccl = 0;
    const bool distinct = std::set<int>(pool->device.begin(), pool->device.end()).size() == n;
    const char* force = std::getenv("NCHMM_POOL_FORCE_RCCL");
    if (distinct && (n > 1 || (force && force[0] == '1'))) {
        std::vector<uint64_t> red = per;
        if (rccl_sum(pool->device, red)) {
            std::memcpy(out, red.data(), 8 * sizeof(uint64_t));
            if (used_rccl) *used_rccl = 1;
            return NCHMM_OK;
        }
    }
    std::memset(out, 0, 8 * sizeof(uint64_t));
    for (size_t d = 0; d < n; ++d)
        for (int k = 0; k < 8; ++k) out[k] += per[8 * d + (size_t)k];
    return NCHMM_OK;
}

static_assert(sizeof(ncclUniqueId) == NCHMM_RCCL_ID_BYTES, "the id travels as NCHMM_RCCL_ID_BYTES bytes");

int nchmm_rccl_unique_id(uint8_t id[NCHMM_RCCL_ID_BYTES])
{
    if (!id) return NCHMM_E_INVALID;
    Rccl& R = rccl();
    if (!R.ok_ranks) return NCHMM_E_NO_DEVICE;
    ncclUniqueId u;
    {
        Stdout_To_Stderr guard;
        if (R.GetUniqueId(&u) != ncclSuccess) return NCHMM_E_HIP;
    }
    std::memcpy(id, &u, sizeof(u));
    return NCHMM_OK;
}

int nchmm_counters_allreduce(int device_id, int n_ranks, int rank, const uint8_t id[NCHMM_RCCL_ID_BYTES], uint64_t inout[8])
{
    if (!id || !inout || n_ranks < 1 || rank < 0 || rank >= n_ranks || device_id < 0) return NCHMM_E_INVALID;
    Rccl& R = rccl();
    if (!R.ok_ranks) return NCHMM_E_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return NCHMM_E_HIP;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    {
        Stdout_To_Stderr guard;
        if (R.CommInitRank(&comm, n_ranks, u, rank) != ncclSuccess) return NCHMM_E_HIP;
    }
    uint64_t* buf = nullptr;
    hipStream_t st = nullptr;
    bool ok = hipMalloc((void**)&buf, 8 * sizeof(uint64_t)) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess
              && hipMemcpyAsync(buf, inout, 8 * sizeof(uint64_t), hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && R.AllReduce(buf, buf, 8, ncclUint64, ncclSum, comm, st) == ncclSuccess;
    uint64_t got[8];
    ok = ok && hipMemcpyAsync(got, buf, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    if (st) (void)hipStreamDestroy(st);
    if (buf) (void)hipFree(buf);
    R.CommDestroy(comm);
    if (!ok) return NCHMM_E_HIP;
    std::memcpy(inout, got, sizeof(got));
    return NCHMM_OK;
}

}  // extern "C"
