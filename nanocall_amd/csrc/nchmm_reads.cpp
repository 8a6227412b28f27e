// nchmm_reads.cpp -- what Fast5_Summary (src/nanocall/Fast5_Summary.hpp) computes from a read's EventDetection
// table before the HMM sees it: event cap, abasic level, hairpin / strand detection, event filter, Event fields,
// initial scaling.  Pure host code over plain arrays; where the table comes from (FAST5, text) is the caller's
// business (nanocall_fast5.h, tools/nanocall.cpp).
#include "nanocall_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

namespace {

typedef std::pair<unsigned, unsigned> Island;   // [first, second) run of events at or above the abasic level

// find_islands_5_consec, Fast5_Summary.hpp:545-571
std::vector<Island> islands_5_consec(const nchmm_ed_event* ed, unsigned n, float level)
{
    std::vector<Island> v;
    for (unsigned i = 0; i < n;) {
        if (!(ed[i].mean >= level)) { ++i; continue; }
        unsigned j = i + 1;
        while (j < n && ed[j].mean >= level) ++j;
        if (j - i >= 5) v.emplace_back(i, j);
        i = j + 1;   // (the event that ended the run is not looked at again, as in the reference)
    }
    return v;
}

// detect_strands, Fast5_Summary.hpp:653-731
void detect_strands(const nchmm_segment_opts& o, const nchmm_ed_event* ed, unsigned n, float level, uint32_t sb[4])
{
    std::vector<Island> isl = islands_5_consec(ed, n, level);
    const unsigned reach = std::max(o.trim_margins[2], o.trim_margins[3]);
    for (size_t i = 1; i < isl.size(); ++i) {           // :665-676 -- merge neighbours, start over after each merge
        if (isl[i - 1].second + reach >= isl[i].first) {
            isl[i - 1].second = isl[i].second;
            isl.erase(isl.begin() + (long)i);
            i = 0;
        }
    }
    if (isl.empty()) return;                            // template only, :685-690
    const long mid = (long)n / 2;
    auto dist = [&](const Island& p) {                  // :694-697
        return std::min((unsigned)std::labs((long)p.first - mid), (unsigned)std::labs((long)p.second - mid));
    };
    size_t pick = 0;                                    // alg::min_of: the first minimum
    for (size_t i = 1; i < isl.size(); ++i)
        if (dist(isl[i]) < dist(isl[pick])) pick = i;
    if (dist(isl[pick]) > n / 6) return;                // hairpin not in the middle third: template only, :700-713
    sb[0] = o.trim_margins[0];
    if (isl.front().first < o.trim_margins[0] + o.trim_margins[2]) sb[0] = std::max(sb[0], isl.front().second);
    sb[1] = isl[pick].first - o.trim_margins[2];
    sb[2] = isl[pick].first + o.trim_margins[3];
    sb[3] = n - o.trim_margins[1];
    if ((uint64_t)isl.back().second > (uint64_t)n - ((uint64_t)o.trim_margins[3] + o.trim_margins[1]))   // size_t arithmetic in the reference
        sb[3] = std::min(sb[3], isl.back().first);
}

inline bool keep_event(const nchmm_ed_event& e, float level)   // filter_ed_event, :734-745
{
    return !(e.mean >= level) && !(e.stdv > 4.0);
}

}  // namespace

extern "C" {

int nchmm_segment_opts_default(nchmm_segment_opts* o, const char* pore)
{
    if (!o || !pore) return NCHMM_E_INVALID;
    o->min_ed_events = 10;
    o->max_ed_events = 100000;
    o->template_only = 0;
    for (int k = 0; k < 4; ++k) o->trim_margins[k] = 50;
    o->abasic_level_top_percent = 1.0;
    if (std::strcmp(pore, "r9") == 0) o->abasic_level_top_offset = 0.0;          // nanocall.cpp:943-953
    else if (std::strcmp(pore, "r73") == 0) o->abasic_level_top_offset = 5.0;    // :954-964
    else return NCHMM_E_INVALID;
    return NCHMM_OK;
}

int nchmm_mean_stdv(size_t n, const float* v, float* mean, float* stdv)
{
    if ((n && !v) || !mean || !stdv) return NCHMM_E_INVALID;
    float sum = 0, sum_sq = 0;
    for (size_t i = 0; i < n; ++i) {
        sum += v[i];
        sum_sq += v[i] * v[i];
    }
    const float cnt = static_cast<float>(n);
    const float m = n ? sum / cnt : 0.0f;
    const float var = n > 1 ? (sum_sq - sum * m) / (cnt - 1) : 0.0f;
    *mean = m;
    *stdv = var > 0 ? std::sqrt(var) : 0.0f;
    return NCHMM_OK;
}

int nchmm_read_load_events(const nchmm_read_summary* s, const nchmm_ed_event* ed, float sampling_rate, int st, float* mean,
                           float* stdv, float* start, float* length, size_t* n_out)
{
    if (!s || !n_out || st < 0 || st > 1) return NCHMM_E_INVALID;
    *n_out = 0;
    if (s->num_ed_events == 0) return NCHMM_OK;          // Fast5_Summary.hpp:325-328
    const unsigned lo = s->strand_bounds[2 * st], hi = s->strand_bounds[2 * st + 1];
    if (lo >= hi) return NCHMM_OK;
    if (!ed || !mean || !stdv || !start || !length || hi > s->num_ed_events) return NCHMM_E_INVALID;
    const int64_t t0 = ed[s->strand_bounds[s->scale_strands_together ? 0 : 2 * st]].start;   // :359
    size_t k = 0;
    for (unsigned j = lo; j < hi; ++j) {
        if (!keep_event(ed[j], s->abasic_level)) continue;
        mean[k] = static_cast<float>(ed[j].mean);
        float sd = static_cast<float>(ed[j].stdv);
        if (sd == 0.0) sd = static_cast<float>(0.01);    // Event::update_logs, Event.hpp:39-42
        stdv[k] = sd;
        start[k] = static_cast<float>(ed[j].start - t0) / sampling_rate;
        length[k] = static_cast<float>(ed[j].length) / sampling_rate;
        ++k;
    }
    *n_out = k;
    return NCHMM_OK;
}

int nchmm_read_summarize(const nchmm_segment_opts* o, size_t n_ed, const nchmm_ed_event* ed, float sampling_rate,
                         int sst, nchmm_read_summary* out)
{
    if (!o || !out || (n_ed && !ed)) return NCHMM_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (sampling_rate < 1000.0 || sampling_rate > 10000.0) return NCHMM_OK;      // :168-172
    const unsigned n = (unsigned)std::min<size_t>(n_ed, o->max_ed_events);       // load_ed_events, :505-525
    if (n < o->trim_margins[0] + o->trim_margins[1] + o->min_ed_events) return NCHMM_OK;   // :185-191
    {
        // detect_abasic_level, :528-543: the (1 - top_percent/100) quantile of the level means, plus the offset
        std::vector<float> lv(n);
        for (unsigned i = 0; i < n; ++i) lv[i] = static_cast<float>(ed[i].mean);
        const size_t q = static_cast<size_t>(static_cast<double>(n) * (1.0 - o->abasic_level_top_percent / 100.0));
        if (q >= n) return NCHMM_E_INVALID;
        std::nth_element(lv.begin(), lv.begin() + (long)q, lv.end());            // == sorted[q]
        out->abasic_level = static_cast<float>(lv[q] + o->abasic_level_top_offset);
    }
    if (out->abasic_level <= 1.0) return NCHMM_OK;                               // :194-200
    uint32_t* sb = out->strand_bounds;
    sb[0] = o->trim_margins[0]; sb[1] = n - o->trim_margins[1]; sb[2] = 0; sb[3] = 0;      // :202
    if (!o->template_only) detect_strands(*o, ed, n, out->abasic_level, sb);
    if (sb[1] <= sb[0]) return NCHMM_OK;                                         // no template strand, :204-209
    out->num_ed_events = n;
    out->scale_strands_together = (sst && sb[1] - sb[0] >= o->min_ed_events && sb[3] - sb[2] >= o->min_ed_events) ? 1 : 0;
    for (int st = 0; st < 2; ++st) {                                             // time lengths, :214-219
        if (sb[2 * st + 1] <= sb[2 * st]) continue;
        const size_t cap = sb[2 * st + 1] - sb[2 * st];
        std::vector<float> buf(4 * cap);
        size_t m = 0;
        const int rc = nchmm_read_load_events(out, ed, sampling_rate, st, buf.data(), buf.data() + cap, buf.data() + 2 * cap,
                                              buf.data() + 3 * cap, &m);
        if (rc != NCHMM_OK) return rc;
        if (m >= o->min_ed_events) out->time_length[st] = buf[2 * cap + m - 1] + buf[3 * cap + m - 1];
    }
    return NCHMM_OK;
}

int nchmm_initial_scaling(int together, const float r0[2], const float r1[2], const float m0[2], const float m1[2],
                          float* scale, float* shift)
{
    if (!r0 || !m0 || !scale || !shift || (together && (!r1 || !m1))) return NCHMM_E_INVALID;
    if (together) {   // Fast5_Summary.hpp:237-241
        const float sc = (r0[1] / m0[1] + r1[1] / m1[1]) / 2;
        *scale = sc;
        *shift = (r0[0] - sc * m0[0] + r1[0] - sc * m1[0]) / 2;
    } else {          // :265-267
        const float sc = r0[1] / m0[1];
        *scale = sc;
        *shift = r0[0] - sc * m0[0];
    }
    return NCHMM_OK;
}

}  // extern "C"
