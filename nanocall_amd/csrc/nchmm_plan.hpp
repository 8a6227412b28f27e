// nchmm_plan.hpp -- how a host-pointer batch is cut up (nchmm_pipeline.cpp): the ranges it goes up in, the processing order
// inside them, and which of its reads are outliers that get back-pointer regions of their own.  Plain host arithmetic over the
// batch's offsets, kept apart from the HIP code so that it runs under the sanitizers on every shape (tools/asan_host.cpp).
#ifndef NCHMM_PLAN_HPP
#define NCHMM_PLAN_HPP

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <numeric>
#include <vector>

namespace nchmm {

struct PipeRange {
    size_t r0, r1;       // reads [r0, r1)
    uint64_t e0, e1;     // their events [e0, e1) of the packed arrays
    uint64_t raw_hi;     // raw form: raw events [0, raw_hi) must be on the device before this range is gathered
    size_t max_events;
};

// Contiguous read ranges.  The first range's copy-in is overlapped with nothing, so a batch that goes up alone starts with a
// short head -- one grid-full of reads -- and everything else follows as ONE range whose copy-in runs under the head's kernel:
// its launch rolls into the head's on the next lane, and its longest reads, wherever they sit in the batch, are handed out
// first (a launch lasts as long as its longest read: 4096 log-normally long reads in one call 296 -> 309 Mevents/s, config-3
// decode 305 -> 333 against ranges that double, profiles/r04b_range_policy_ab.txt).
//   * streaming form, up to two grid-fulls of reads (the 1024-read batches a streaming caller sends): one range -- its
//     copy-in overlaps the previous BATCH instead
//   * one-call form (`alone`) from 1.5 grid-fulls up, and every larger batch: head + rest
// forced > 0 (NCHMM_PIPE_READS, test hook): ranges of that many reads.
inline void cut_ranges(const uint64_t* off, size_t n, size_t slots, bool alone, size_t forced, std::vector<PipeRange>* out)
{
    slots = std::max<size_t>(slots, 1);
    auto range_of = [&](size_t r0, size_t r1) {
        size_t mx = 0;
        for (size_t r = r0; r < r1; ++r) mx = std::max<size_t>(mx, (size_t)(off[r + 1] - off[r]));
        return PipeRange{r0, r1, off[r0], off[r1], 0, mx};
    };
    out->clear();
    if (forced) {
        for (size_t r0 = 0; r0 < n; r0 += forced) out->push_back(range_of(r0, std::min(n, r0 + forced)));
        return;
    }
    if (n <= (alone ? slots + slots / 2 : 2 * slots)) {
        out->push_back(range_of(0, n));
        return;
    }
    // the head: one grid-full of reads, and at least 1 M events (short reads: too little work to cover the copy-in of the rest)
    size_t r1 = slots;
    while (r1 < n && off[r1] - off[0] < ((uint64_t)1 << 20)) ++r1;
    if (n - r1 < slots / 2) r1 = n;      // (a remainder not worth a launch)
    out->push_back(range_of(0, r1));
    if (r1 < n) out->push_back(range_of(r1, n));
}

// Longest-first processing order inside each range (the device work queue hands reads out in this order; equal lengths keep
// their input order).
inline void order_ranges(const uint64_t* off, size_t n, const std::vector<PipeRange>& ranges, std::vector<uint32_t>* order)
{
    order->resize(n);
    std::iota(order->begin(), order->end(), 0u);
    for (const PipeRange& g : ranges)
        std::stable_sort(order->begin() + g.r0, order->begin() + g.r1,
                         [&](uint32_t a, uint32_t b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
}

// Outliers.  pool = back-pointer regions the pooled launches share, budget = bytes the workspace may take, row = bytes per event.
// When the longest read is too long for `pool` regions within the budget but only a few reads are that long (at most one in
// eight), those few -- every read longer than what 70 % of the budget gives a pooled region -- go through regions of their own
// (the other 30 %) as one more launch, and the pool is sized for the rest.  Otherwise: no outliers, pool_longest = longest.
//   n_out[k]   how many reads of range k are outliers: a PREFIX of its (longest-first) order
//   outliers   their read indices, longest first
struct OutlierPlan {
    uint64_t pool_longest = 1;
    std::vector<uint32_t> outliers;
    std::vector<size_t> n_out;
    size_t budget_big = 0;
};
inline OutlierPlan plan_outliers(const uint64_t* off, size_t n, const std::vector<PipeRange>& ranges, const std::vector<uint32_t>& order,
                                 size_t pool, size_t budget, size_t row)
{
    OutlierPlan P;
    P.n_out.assign(ranges.size(), 0);
    uint64_t longest = 1;
    for (size_t r = 0; r < n; ++r) longest = std::max<uint64_t>(longest, off[r + 1] - off[r]);
    P.pool_longest = longest;
    pool = std::max<size_t>(pool, 1);
    if ((size_t)longest * row <= budget / pool) return P;
    const uint64_t small_cap = (uint64_t)(budget / 10 * 7 / pool / row);   // events a pooled region may hold
    size_t n_long = 0;
    uint64_t longest_short = 1;
    for (size_t r = 0; r < n; ++r) {
        const uint64_t len = off[r + 1] - off[r];
        if (len > small_cap) ++n_long; else longest_short = std::max(longest_short, len);
    }
    if (small_cap < 256 || n_long * 8 > n) return P;      // (most reads are long: everything the usual way, on fewer blocks)
    P.pool_longest = longest_short;
    for (size_t k = 0; k < ranges.size(); ++k) {
        const PipeRange& g = ranges[k];
        while (g.r0 + P.n_out[k] < g.r1 && off[order[g.r0 + P.n_out[k]] + 1] - off[order[g.r0 + P.n_out[k]]] > small_cap) ++P.n_out[k];
        P.outliers.insert(P.outliers.end(), order.begin() + g.r0, order.begin() + g.r0 + P.n_out[k]);
    }
    std::stable_sort(P.outliers.begin(), P.outliers.end(), [&](uint32_t a, uint32_t b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
    P.budget_big = budget / 10 * 3;
    return P;
}

// ---- which form of the sweep a launch takes ----
// viterbi_kernel.hip (`wide`: 8 waves per read, two reads per CU) decodes the most events per second and CU; viterbi_ll_kernel.hip
// (`ll`: 16 waves per read, one read per CU) halves the time of a read.  A launch hands its reads out longest first to
// persistent blocks, so it lasts about as long as a longest-processing-time schedule of its reads on its block slots: that is
// simulated for both forms (exactly up to kSimReads reads, by its two lower bounds beyond) and the shorter one wins.
// Microseconds per event, measured (profiles/r05_ll_sweep.md): a wide block with a busy neighbour on its CU, a wide block
// alone on its CU, an ll block.
enum Sweep : int { kSweepAuto = 0, kSweepWide = 1, kSweepLl = 2, kSweepAhead = 3 };
constexpr double kRatesClockMHz = 2100.0;      // the shader clock the event rates below were measured at (boxes at 2.0-2.15 GHz)
// kSweepAhead: the low-latency form with the emissions of the launch's longest reads computed ahead by the whole device
// (emission_kernel.hip): their columns carry the recurrence only.  us per event of a read: ahead; CU-us per event of the
// emission kernel: em_cu (it runs on every CU at once, before the sweep: rows * em_cu / n_cu of wall time).
// em_launch_us: what a second kernel in front of the sweep costs (launch + its tail).
// The event rates follow the shader clock (the kernels are VALU-issue bound) and the boxes of the pool sustain 1.9-2.35 GHz; the
// decision is a comparison of durations, so a common factor cancels, and what does not scale with the clock -- em_cu on rows
// from HBM, the per-read and per-launch constants -- only matters near the break-even: tools/plan_props.cpp walks 20 000 batch
// shapes with every group of rates scaled by 0.8-1.25 on its own and finds no decision that costs more than 5 % of the best
// form's duration under the scaled rates (tests/test_host_prep.py).
struct SweepRates { double wide_shared = 1.55, wide_alone = 1.15, ll = 0.85, ahead = 0.60, em_cu = 0.85; double per_read_us = 60.0; double em_launch_us = 30.0; };
// A read whose emissions are ahead streams 16 KiB per event back in (27 GB/s per block at 0.6 us per event; 256 such streams are the device's whole bandwidth).  Beyond ~100 such
// blocks at once the sweep is HBM-bound and slower than computing the emissions in place (256 x 5000 events: 4.85 ms against 4.23).
constexpr size_t kMaxAheadReads = 96;
// The rates on a device whose sustained shader clock is known (nchmm_shader_clock_mhz has measured it under load, or
// NCHMM_PLAN_CLOCK_MHZ states it): the sweeps are VALU-issue bound and follow the clock; rows from HBM (em_cu) and the fixed costs do not.
inline SweepRates rates_at_clock(double mhz)
{
    SweepRates r;
    if (mhz > 0.0) {
        const double f = kRatesClockMHz / std::min(std::max(mhz, 800.0), 3000.0);
        r.wide_shared *= f; r.wide_alone *= f; r.ll *= f; r.ahead *= f;
    }
    return r;
}

// makespan (us) of `lens` (events per read, any order) handed out longest first to `slots` blocks at `us_per_event`
inline double lpt_makespan_us(std::vector<uint64_t> lens, size_t slots, double us_per_event, double per_read_us)
{
    constexpr size_t kSimReads = 16384;
    slots = std::max<size_t>(slots, 1);
    if (lens.empty()) return 0.0;
    uint64_t longest = 0, total = 0;
    for (uint64_t l : lens) { longest = std::max(longest, l); total += l; }
    if (lens.size() <= slots) return (double)longest * us_per_event + per_read_us;
    if (lens.size() > kSimReads)
        return std::max((double)longest * us_per_event + per_read_us,
                        ((double)total * us_per_event + (double)lens.size() * per_read_us) / (double)slots);
    std::sort(lens.begin(), lens.end(), std::greater<uint64_t>());
    // a heap of block finish times (min first)
    std::vector<double> heap(slots, 0.0);
    auto cmp = [](double a, double b) { return a > b; };
    double last = 0.0;
    for (uint64_t l : lens) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        heap.back() += (double)l * us_per_event + per_read_us;
        last = std::max(last, heap.back());
        std::push_heap(heap.begin(), heap.end(), cmp);
    }
    return last;
}

// ---- which reads of a low-latency launch get their emissions ahead ----
// lens: the launch's reads LONGEST FIRST.  Candidates: the first K reads, K = 0, 1, 2, 4, ... (and all), as far as `budget_rows`
// (16 KiB per event) reaches; each is priced as the emission kernel's wall time plus the longest-processing-time schedule of
// the sweep on n_cu blocks, and the cheapest wins.  Returns K; *t_us = its duration.
// duration (us) of a low-latency launch of `lens_desc` (longest first) whose first K reads have their emissions ahead
inline double price_ahead_us(const std::vector<uint64_t>& lens_desc, size_t K, size_t n_cu, const SweepRates& R)
{
    n_cu = std::max<size_t>(n_cu, 1);
    uint64_t rows = 0;
    for (size_t k = 0; k < K; ++k) rows += lens_desc[k];
    // (times, not lengths, go through the scheduler: scale the ahead reads' lengths by ahead / ll)
    std::vector<uint64_t> eff(lens_desc);
    for (size_t k = 0; k < K; ++k) eff[k] = (uint64_t)((double)eff[k] * (R.ahead / R.ll));
    return (double)rows * R.em_cu / (double)n_cu + (K ? R.em_launch_us : 0.0) + lpt_makespan_us(std::move(eff), n_cu, R.ll, R.per_read_us);
}

inline size_t plan_ahead(const std::vector<uint64_t>& lens_desc, size_t n_cu, uint64_t budget_rows, double* t_us, const SweepRates& R = SweepRates())
{
    const size_t n = lens_desc.size();
    n_cu = std::max<size_t>(n_cu, 1);
    auto price = [&](size_t K) { return price_ahead_us(lens_desc, K, n_cu, R); };
    size_t best_k = 0;
    double best = price(0);
    uint64_t rows = 0;
    size_t reach = 0;                                    // how many leading reads fit the budget
    while (reach < n && reach < kMaxAheadReads && rows + lens_desc[reach] <= budget_rows) rows += lens_desc[reach++];
    for (size_t K = 1; K <= reach; K = (K < reach && 2 * K > reach) ? reach : 2 * K) {
        const double t = price(K);
        if (t < best * 0.97) { best = t; best_k = K; }   // (3 %: not worth a second kernel for less)
        if (K == reach) break;
    }
    if (t_us) *t_us = best;
    return best_k;
}

// lens = the launch's reads.  busy: other launches run beside this one (a streaming caller's batches in flight, three lanes):
// the tail of a launch is then covered by its neighbours and only throughput counts -- unless the launches are so small that
// three of them do not fill the wide sweep's block slots.  (Three launches of 256 x 50 000-event reads in turn: 316 Mevents/s
// wide, 288 low-latency, same box -- profiles/r05_ll_sweep.md.)
inline Sweep choose_sweep(const std::vector<uint64_t>& lens, size_t n_cu, size_t wide_slots, bool busy, const SweepRates& R = SweepRates(),
                          uint64_t ahead_budget_rows = 0, size_t* n_ahead = nullptr)
{
    if (n_ahead) *n_ahead = 0;
    if (lens.empty() || n_cu == 0) return kSweepWide;
    if (busy && lens.size() * 3 >= wide_slots) return kSweepWide;
    double t_ll = lpt_makespan_us(lens, n_cu, R.ll, R.per_read_us);
    size_t k_ahead = 0;
    if (ahead_budget_rows && !busy) {
        std::vector<uint64_t> desc(lens);
        std::sort(desc.begin(), desc.end(), std::greater<uint64_t>());
        double t = t_ll;
        k_ahead = plan_ahead(desc, n_cu, ahead_budget_rows, &t, R);
        if (k_ahead) t_ll = t;
    }
    const double t_wide = lens.size() <= n_cu ? lpt_makespan_us(lens, n_cu, R.wide_alone, R.per_read_us)
                                              : lpt_makespan_us(lens, wide_slots, R.wide_shared, R.per_read_us);
    if (t_ll >= t_wide) return kSweepWide;
    if (n_ahead) *n_ahead = k_ahead;
    return k_ahead ? kSweepAhead : kSweepLl;
}

// the same decision from what a device-pointer caller states: number of reads, longest read, total events
inline Sweep choose_sweep_bounds(size_t n_reads, uint64_t longest, uint64_t total, size_t n_cu, size_t wide_slots, bool busy, const SweepRates& R = SweepRates(),
                                 uint64_t ahead_budget_rows = 0)
{
    if (n_reads == 0 || n_cu == 0) return kSweepWide;
    if (busy && n_reads * 3 >= wide_slots) return kSweepWide;
    if (longest == 0 || longest > total) longest = total / n_reads + 1;
    // (the two lower bounds of the longest-first schedule, every read with its fixed cost -- as lpt_makespan_us beyond kSimReads;
    // round 5 charged the fixed cost once per launch, which put batches of many very short reads on the low-latency form:
    // 455 reads of 64 events 229 us there against 159 us wide, tools/plan_props.cpp)
    // Reads of (nearly) one length -- the longest within a quarter of the mean -- go through in whole rounds: 264 reads on 256
    // blocks take two rounds, not 1.03 (the averaged bound had such batches on the low-latency form at 219 us against 150 wide).
    const bool uniform = (double)longest * (double)n_reads <= 1.25 * (double)total;
    auto bound = [&](size_t slots, double us) {
        double par = 0.0;
        if (n_reads > slots)
            par = uniform ? (double)((n_reads + slots - 1) / slots) * ((double)total / (double)n_reads * us + R.per_read_us)
                          : ((double)total * us + (double)n_reads * R.per_read_us) / (double)slots;
        return std::max((double)longest * us + R.per_read_us, par);
    };
    const double t_ll = bound(n_cu, R.ll);
    const double t_wide = n_reads <= n_cu ? bound(n_cu, R.wide_alone) : bound(wide_slots, R.wide_shared);
    // every read ahead (the lengths are not known here): worth it when the batch is a few reads -- idle CUs compute the emissions
    const double t_ahead = !busy && ahead_budget_rows && total <= ahead_budget_rows && n_reads <= kMaxAheadReads
                               ? (double)total * R.em_cu / (double)n_cu + R.em_launch_us + bound(n_cu, R.ahead) : 1e300;
    if (t_ahead < 0.97 * std::min(t_ll, t_wide)) return kSweepAhead;
    return t_ll < t_wide ? kSweepLl : kSweepWide;
}

}  // namespace nchmm
#endif
