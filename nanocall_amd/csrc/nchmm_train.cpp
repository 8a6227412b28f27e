// nchmm_train.cpp -- the EM driver loop of nanocall's train_reads (src/nanocall/nanocall.cpp:275-582),
// batched across reads: every round, ALL jobs that are still training contribute their windows to one
// nchmm_fwbw launch, then each job finishes its round on the host exactly as
// Parameter_Trainer::train_one_round does (Parameter_Trainer.hpp:541-579) and applies the reference's
// stop / roll-back rules (nanocall.cpp:394-426, :510-542) and model selection (:437-459, :552-570).
//
// A "job" is one (read, model) in per-strand mode or one (read, template model, complement model) in
// scale_strands_together mode -- one iteration of the reference's `for m_name...` loops.
// Pure host C++ over the C ABI (no HIP here).
#include "nanocall_hip.h"
#include "nchmm_internal.hpp"
#include "nchmm_pipe.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <vector>

using nchmm::parallel_for;

namespace {

struct Window { uint64_t begin; uint32_t len; uint32_t strand; };

struct Job {
    int read, m[2];
    float pm[6], st[4], fit;
    float old_pm[6], old_st[4], old_fit;
    unsigned round = 0;
    bool active = true;
    std::vector<Window> win;   // training windows, strand 0 first (nanocall.cpp:344-352)
    bool have[2] = {false, false};
};

}  // namespace

extern "C" {

int nchmm_train_opts_default(nchmm_train_opts* o)
{
    if (!o) return NCHMM_E_INVALID;
    o->scaling_num_events = 200;      // nanocall.cpp:72
    o->scaling_max_rounds = 10;       // :71
    o->scaling_min_progress = 1.0f;   // :70
    o->scaling_select_threshold = 20.0f;   // :69
    o->min_ed_events = 10;            // :66
    o->train_scaling = 1; o->train_transitions = 1; o->train_drift = 1;
    o->default_p_stay = .1f;          // :85
    o->default_p_skip = .3f;          // :84
    return NCHMM_OK;
}

// the model_list / pair loops of train_reads (nanocall.cpp:300-323, :356-358, :474)
int nchmm_train_enumerate(const nchmm_train_opts* o, size_t n_models, const int32_t* model_strand, size_t n_reads,
                          const uint64_t* strand_off, const uint8_t* together, size_t* n_jobs, int32_t* job_read,
                          int32_t* job_m0, int32_t* job_m1)
{
    if (!o || !model_strand || !strand_off || !n_jobs) return NCHMM_E_INVALID;
    const size_t cap = *n_jobs;
    size_t n = 0;
    auto push = [&](size_t r, int a, int b) {
        if (job_read && n < cap) { job_read[n] = (int32_t)r; job_m0[n] = a; job_m1[n] = b; }
        ++n;
    };
    for (size_t r = 0; r < n_reads; ++r) {
        const bool ok0 = strand_off[2 * r + 1] - strand_off[2 * r] >= o->min_ed_events;
        const bool ok1 = strand_off[2 * r + 2] - strand_off[2 * r + 1] >= o->min_ed_events;
        if (together && together[r]) {
            if (!ok0 || !ok1) continue;
            for (size_t a = 0; a < n_models; ++a) {
                if (model_strand[a] == 1) continue;
                for (size_t b = 0; b < n_models; ++b)
                    if (model_strand[b] != 0) push(r, (int)a, (int)b);
            }
        } else {
            for (int st = 0; st < 2; ++st) {
                if (!(st ? ok1 : ok0)) continue;
                for (size_t a = 0; a < n_models; ++a)
                    if (model_strand[a] == st || model_strand[a] == 2) push(r, st == 0 ? (int)a : -1, st == 1 ? (int)a : -1);
            }
        }
    }
    *n_jobs = n;
    return (job_read && n > cap) ? NCHMM_E_NOMEM : NCHMM_OK;
}

int nchmm_train_reads(nchmm_ctx* ctx, const nchmm_train_opts* o, size_t n_models, const float* model_states_Sx10,
                      size_t n_reads, const uint64_t* strand_off, const float* mean, const float* stdv, const float* start,
                      size_t n_jobs, const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1, float* job_pm,
                      float* job_st, float* job_fit, uint32_t* job_rounds, int32_t* read_preferred)
{
    if (!ctx || !o || !model_states_Sx10 || !strand_off || !mean || !stdv || !start || !job_read || !job_m0 || !job_m1
        || !job_pm || !job_st || !job_fit || !job_rounds)
        return NCHMM_E_INVALID;
    const float NEG_INF = -std::numeric_limits<float>::infinity();
    const uint64_t total_events = strand_off[2 * n_reads];
    const auto t_call = std::chrono::steady_clock::now();
    // Only the training windows (first and last scaling_num_events / 2 events of a strand, nanocall.cpp:333-337) are
    // ever read: they are compacted strand by strand -- [head window | tail window] at position 2 * half * k of the
    // compact arrays for strand k -- and only those go to the device.  log_stdv: Event::update_logs happened at load
    // time in the reference (float libm).
    (void)total_events;
    std::vector<uint64_t> cbase(2 * n_reads + 1, 0);
    for (size_t k = 0; k < 2 * n_reads; ++k)
        cbase[k + 1] = cbase[k] + 2 * (std::min<uint64_t>(o->scaling_num_events, strand_off[k + 1] - strand_off[k]) / 2);
    const size_t n_compact = (size_t)cbase[2 * n_reads];
    std::unique_ptr<float[]> c_mean(new float[n_compact + 1]), c_stdv(new float[n_compact + 1]), c_start(new float[n_compact + 1]),
        c_lstdv(new float[n_compact + 1]);
    parallel_for(2 * n_reads, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) {
            const uint64_t b = strand_off[k], e = strand_off[k + 1];
            const uint64_t half = (cbase[k + 1] - cbase[k]) / 2;
            uint64_t d = cbase[k];
            for (int part = 0; part < 2; ++part) {
                const uint64_t from = part == 0 ? b : e - half;
                for (uint64_t i = from; i < from + half; ++i, ++d) {
                    c_mean[d] = mean[i]; c_stdv[d] = stdv[i]; c_start[d] = start[i]; c_lstdv[d] = std::log(stdv[i]);
                }
            }
        }
    });

    std::vector<Job> jobs(n_jobs);
    for (size_t k = 0; k < n_jobs; ++k) {
        Job& j = jobs[k];
        j.read = job_read[k]; j.m[0] = job_m0[k]; j.m[1] = job_m1[k];
        if (j.read < 0 || (size_t)j.read >= n_reads || (j.m[0] < 0 && j.m[1] < 0)) return NCHMM_E_INVALID;
        for (int s = 0; s < 2; ++s)
            if (j.m[s] >= (int)n_models) return NCHMM_E_INVALID;
        std::memcpy(j.pm, job_pm + 6 * k, sizeof(j.pm));
        std::memcpy(j.st, job_st + 4 * k, sizeof(j.st));
        j.fit = NEG_INF;   // nanocall.cpp:365
        for (int s = 0; s < 2; ++s) {
            if (j.m[s] < 0) continue;
            const uint64_t b = strand_off[2 * j.read + s], e = strand_off[2 * j.read + s + 1];
            const uint64_t n_ev = e - b;
            if (n_ev < o->min_ed_events) { if (j.m[1 - s] < 0) j.active = false; continue; }
            // nanocall.cpp:333-337: first and last num_train_events / 2 events
            const uint64_t num = std::min<uint64_t>(o->scaling_num_events, n_ev);
            const uint32_t half = (uint32_t)(num / 2);
            (void)b; (void)e;
            const uint64_t cb = cbase[2 * j.read + s];                       // the strand's two windows in the compact arrays
            j.win.push_back(Window{cb, half, (uint32_t)s});
            j.win.push_back(Window{cb + half, half, (uint32_t)s});
            j.have[s] = true;
        }
        if (j.win.empty()) j.active = false;
    }

    // model slots [0, n_models) stay free (the statistics are taken from the scaled states + pm_params); transition slot 0
    // holds the default weights
    int rc;
    if ((rc = nchmm_put_transitions_fast(ctx, 0, 1, &o->default_p_skip, &o->default_p_stay))) return rc;
    // the events of every read go to the device once; each round only sends per-window descriptors
    if ((rc = nchmm_em_load_events(ctx, n_compact, c_mean.get(), c_stdv.get(), c_start.get(), c_lstdv.get()))) return rc;

    const bool dbg_time = std::getenv("NCHMM_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_rounds = now();

    // The jobs train in one part or in several.  Several (a call with enough jobs): parts of at most half the forward-backward
    // budget's alpha rows, taking turns on the two EM lanes of the context (nchmm_internal.hpp); a part's round is queued on a lane
    // without waiting -- while the device runs it, the host finishes the round of the part on the other lane (the 3 x 3 solves,
    // the stop rules) and queues the next part there, so the device does not wait for the host between rounds (a seventh of a
    // round's time at config-3 size, profiles/r06_notes.md section 13) and the workspace is two parts' rows however many jobs
    // the call brings.  A job's rounds are the same in any arrangement: windows are independent in the kernels, jobs on the host.
    struct Part {
        std::vector<size_t> act;                     // its jobs that still train
        size_t na = 0, n_win = 0;                    // the round in flight
        std::vector<uint32_t> first_win;
        std::vector<size_t> job_events;
        std::vector<float> lpd, st_sums;
        std::vector<double> acc;
        nchmm::EmPending pend;
        bool in_flight = false;
        double t_prep = 0, t_tables = 0, t_queue = 0;
    };
    std::vector<Part> part;
    size_t lane_slot0[2] = {0, 0};                   // the (job, strand) tables of the part on lane h: model slots n_models + lane_slot0[h] + ..., transition slots 1 + lane_slot0[h] + ...
    size_t n_act = 0, ev_act = 0;
    for (size_t k = 0; k < n_jobs; ++k)
        if (jobs[k].active) { ++n_act; for (const Window& w : jobs[k].win) ev_act += w.len; }
    if (n_act == 0) goto rounds_done;
    {
    const char* lanes_env = std::getenv("NCHMM_EM_LANES");
    const size_t cap = nchmm::em_fb_cap_events(ctx);
    const bool two = (lanes_env ? std::atoi(lanes_env) >= 2 : true) && n_act >= 64;
    // one part (few jobs: a round over the budget is cut by nchmm_em_round), or as many as it takes to keep each within half the
    // budget, at least two, of about equal events, in job order
    const size_t n_parts = two ? std::max<size_t>(2, (ev_act + cap / 2 - 1) / std::max<size_t>(cap / 2, 1)) : 1;
    part.resize(n_parts);
    size_t w_max = 0, j_max = 0, ev_max = 0;
    {
        size_t ev = 0;
        std::vector<size_t> part_ev(n_parts, 0), part_win(n_parts, 0);
        for (size_t k = 0; k < n_jobs; ++k) {
            if (!jobs[k].active) continue;
            size_t mine = 0;
            for (const Window& w : jobs[k].win) mine += w.len;
            size_t q = std::min(n_parts - 1, (size_t)((unsigned __int128)ev * n_parts / std::max<size_t>(ev_act, 1)));
            // (a job's windows stay together: one that would take its part over the lane's share opens the next part -- the parts were counted with room for that)
            if (two && part_ev[q] && part_ev[q] + mine > cap / 2 && q + 1 < n_parts) ++q;
            part[q].act.push_back(k);
            part_ev[q] += mine; part_win[q] += jobs[k].win.size();
            ev += mine;
        }
        for (size_t q = 0; q < n_parts; ++q) {
            w_max = std::max(w_max, part_win[q]); j_max = std::max(j_max, part[q].act.size()); ev_max = std::max(ev_max, part_ev[q]);
        }
    }
    lane_slot0[1] = two ? 2 * j_max : 0;
    if ((rc = nchmm_reserve_slots(ctx, (int)std::max<size_t>(n_models, 1) + (int)(two ? 4 * j_max : 2 * n_act) + 2))) return rc;
    struct Lanes_Guard {       // whatever way this function is left: nothing in flight, lane 0 selected, calls wait again
        nchmm_ctx* c; bool on;
        ~Lanes_Guard() { if (on) nchmm::em_lanes_end(c); }
    } guard{ctx, two};
    if (two) {
        // (a part over the lane's share -- one job's windows alone can be -- still runs: the workspace grows to twice the largest part)
        if ((rc = nchmm::em_lanes_prepare(ctx, nchmm::em_round_pin_bytes(w_max, j_max), ev_max, 2 * ev_max))) return rc;
        nchmm::em_lanes_async(ctx, true);
    }

    // ---- Parameter_Trainer::fill_train_data (Parameter_Trainer.hpp:99-155), all the part's active jobs at once: queue one round ----
    auto queue_round = [&](Part& P, int lane) -> int {
        const auto t_0 = now();
        const size_t na = P.act.size();
        P.na = na;
        const size_t slot0 = lane_slot0[lane];
        std::vector<int32_t> m_idx(2 * na, 0); std::vector<float> m_par(12 * na, 0.f);   // scaled models: slot n_models + slot0 + 2p + s
        std::vector<float> t_skip(2 * na, o->default_p_skip), t_stay(2 * na, o->default_p_stay);
        // per-window descriptors (the drift correction and SoA packing of :130-140 happen on the device)
        P.first_win.assign(na + 1, 0);
        for (size_t p = 0; p < na; ++p) P.first_win[p + 1] = P.first_win[p] + (uint32_t)jobs[P.act[p]].win.size();
        const size_t n_win = P.first_win[na];
        P.n_win = n_win;
        std::vector<uint64_t> w_src(n_win);
        std::vector<uint32_t> w_len(n_win);
        std::vector<float> w_drift(n_win), stp(2 * n_win), w_pm(6 * n_win);   // w_pm: the parameters behind the window's scaled model
        std::vector<int32_t> s_slot(n_win), t_slot(n_win);
        P.job_events.assign(na, 0);
        for (size_t p = 0; p < na; ++p) {
            Job& j = jobs[P.act[p]];
            std::memcpy(j.old_pm, j.pm, sizeof(j.pm)); std::memcpy(j.old_st, j.st, sizeof(j.st)); j.old_fit = j.fit;
            for (int s = 0; s < 2; ++s) {
                m_idx[2 * p + s] = j.m[s] >= 0 ? j.m[s] : std::max(j.m[0], j.m[1]);
                std::memcpy(&m_par[6 * (2 * p + s)], j.old_pm, sizeof(j.old_pm));
                t_stay[2 * p + s] = j.old_st[2 * s]; t_skip[2 * p + s] = j.old_st[2 * s + 1];
            }
            size_t wi = P.first_win[p];
            for (const Window& w : j.win) {
                w_src[wi] = w.begin; w_len[wi] = w.len; w_drift[wi] = j.old_pm[2];   // apply_drift_correction, Event.hpp:77-84
                P.job_events[p] += w.len;
                s_slot[wi] = (int32_t)(n_models + slot0 + 2 * p + w.strand);
                std::memcpy(&w_pm[6 * wi], j.old_pm, sizeof(j.old_pm));
                // is_default(): compares against the CLI defaults (State_Transitions.hpp:34-37)
                const bool dflt = j.old_st[2 * w.strand] == o->default_p_stay && j.old_st[2 * w.strand + 1] == o->default_p_skip;
                t_slot[wi] = dflt ? 0 : (int32_t)(1 + slot0 + 2 * p + w.strand);
                stp[2 * wi] = j.old_st[2 * w.strand]; stp[2 * wi + 1] = j.old_st[2 * w.strand + 1];
                ++wi;
            }
        }
        const auto t_1 = now();
        int qrc;
        if (two) { nchmm::em_lane_select(ctx, lane); nchmm::em_lane_rewind(ctx); }
        if ((qrc = nchmm_put_models_scaled(ctx, (int)(n_models + slot0), 2 * na, model_states_Sx10, m_idx.data(), m_par.data()))) return qrc;
        if ((qrc = nchmm_put_transitions_fast(ctx, (int)(1 + slot0), 2 * na, t_skip.data(), t_stay.data()))) return qrc;
        const auto t_2 = now();
        // forward-backward, inner and outer sums on the device: 13 doubles per job come back
        P.lpd.resize(n_win); P.st_sums.resize(3 * n_win); P.acc.resize(13 * na);
        if (two)
            qrc = nchmm::em_round_enqueue(ctx, n_win, w_src.data(), w_len.data(), w_drift.data(), w_pm.data(), s_slot.data(), t_slot.data(), stp.data(),
                                         na, P.first_win.data(), o->train_drift, &P.pend);
        else
            qrc = nchmm_em_round(ctx, n_win, w_src.data(), w_len.data(), w_drift.data(), w_pm.data(), s_slot.data(), t_slot.data(), stp.data(),
                                na, P.first_win.data(), o->train_drift, P.lpd.data(), P.st_sums.data(), P.acc.data());
        if (qrc != NCHMM_OK) return qrc;
        P.in_flight = true;
        P.t_prep = ms(t_0, t_1); P.t_tables = ms(t_1, t_2); P.t_queue = ms(t_2, now());
        return NCHMM_OK;
    };

    // ---- wait for the part's round, finish it per job (Parameter_Trainer.hpp:557-578), apply the stop rules ----
    auto finish_round = [&](Part& P, int lane) -> int {
        const auto t_0 = now();
        if (two) {
            nchmm::em_lane_select(ctx, lane);
            const int crc = nchmm::em_round_collect(ctx, P.pend, P.lpd.data(), P.st_sums.data(), P.acc.data());
            if (crc != NCHMM_OK) return crc;
        }
        P.in_flight = false;
        const auto t_1 = now();
        const size_t na = P.na;
        const std::vector<float>& lpd = P.lpd; const std::vector<float>& st_sums = P.st_sums; const std::vector<double>& acc = P.acc;
        std::atomic<int> first_err{NCHMM_OK};
        parallel_for(na, [&](size_t p_lo, size_t p_hi) {
        for (size_t p = p_lo; p < p_hi; ++p) {
            Job& j = jobs[P.act[p]];
            const size_t w0 = P.first_win[p], w1 = P.first_win[p + 1];
            float fit = 0;
            for (size_t w = w0; w < w1; ++w) fit += lpd[w];   // data.fit += log_pr_data, :154
            j.fit = fit;
            bool done = false;
            if (o->train_scaling) {
                int d = 0;
                const int src = nchmm_train_pm_solve(P.job_events[p], &acc[13 * p], o->train_drift, j.old_pm, j.pm, &d);
                if (src != NCHMM_OK) { first_err = src; return; }
                done = d != 0;
            }
            if (done) {
                std::memcpy(j.st, j.old_st, sizeof(j.st));   // new_st_params = crt_st_params, :568-572
            } else if (o->train_transitions) {
                for (int s = 0; s < 2; ++s) {
                    if (!j.have[s]) continue;   // (the reference leaves NaN in the absent strand's slot; it is never read)
                    std::vector<float> mine;
                    for (size_t w = w0; w < w1; ++w)
                        if (j.win[w - w0].strand == (uint32_t)s) mine.insert(mine.end(), &st_sums[3 * w], &st_sums[3 * w] + 3);
                    const int frc = nchmm_train_st_finish(mine.size() / 3, mine.data(), &j.st[2 * s], &j.st[2 * s + 1]);
                    if (frc != NCHMM_OK) { first_err = frc; return; }
                }
            }
            // nanocall.cpp:394-426 (2D) / :510-542 (1D)
            if (done) { j.active = false; continue; }
            if (j.fit < j.old_fit) {   // regression: roll back
                std::memcpy(j.pm, j.old_pm, sizeof(j.pm)); std::memcpy(j.st, j.old_st, sizeof(j.st)); j.fit = j.old_fit;
                j.active = false;
                continue;
            }
            ++j.round;
            const bool two_d = j.m[0] >= 0 && j.m[1] >= 0;
            const unsigned limit = two_d ? 2u * o->scaling_max_rounds : o->scaling_max_rounds;
            if (j.round >= limit || (j.round > 1 && j.fit < j.old_fit + o->scaling_min_progress)) j.active = false;
        }
        });
        if (first_err != NCHMM_OK) return first_err;
        // (the part's jobs keep their places relative to each other; its tables are rewritten from slot0 every round)
        size_t kept = 0;
        for (size_t p = 0; p < na; ++p) if (jobs[P.act[p]].active) P.act[kept++] = P.act[p];
        P.act.resize(kept);
        if (dbg_time)
            std::fprintf(stderr, "[nchmm_train_reads] round%s: %zu jobs, gather %.2f ms, tables %.2f ms, %s %.2f ms, finish %.2f ms\n", two ? (lane ? " (lane 1)" : " (lane 0)") : "", na,
                         P.t_prep, P.t_tables, two ? "queued in" : "fwbw", P.t_queue, ms(t_1, now()));
        (void)t_0;
        return NCHMM_OK;
    };

    // the parts take turns: whichever lane comes free takes the part that has waited longest for its next round
    std::deque<size_t> waiting;
    for (size_t q = 0; q < n_parts; ++q) if (!part[q].act.empty()) waiting.push_back(q);
    long on_lane[2] = {-1, -1};
    auto fill = [&](int lane) -> int {
        if (waiting.empty()) return NCHMM_OK;
        const size_t q = waiting.front();
        waiting.pop_front();
        on_lane[lane] = (long)q;
        return queue_round(part[q], lane);
    };
    const int n_lanes = two ? 2 : 1;
    for (int h = 0; h < n_lanes; ++h) if ((rc = fill(h))) return rc;
    for (int h = 0; on_lane[0] >= 0 || on_lane[1] >= 0; h = (h + 1) % n_lanes) {
        if (on_lane[h] < 0) continue;
        Part& P = part[(size_t)on_lane[h]];
        if ((rc = finish_round(P, h))) return rc;
        if (!P.act.empty()) waiting.push_back((size_t)on_lane[h]);
        on_lane[h] = -1;
        if ((rc = fill(h))) return rc;
    }
    }
rounds_done:
    if (dbg_time)
        std::fprintf(stderr, "[nchmm_train_reads] %zu jobs of %zu reads: setup (windows, job table, events up) %.2f ms, rounds %.2f ms\n", n_jobs, n_reads,
                     ms(t_call, t_rounds), ms(t_rounds, now()));
    for (size_t k = 0; k < n_jobs; ++k) {
        std::memcpy(job_pm + 6 * k, jobs[k].pm, sizeof(jobs[k].pm));
        std::memcpy(job_st + 4 * k, jobs[k].st, sizeof(jobs[k].st));
        job_fit[k] = jobs[k].fit; job_rounds[k] = jobs[k].round;
    }
    // ---- model selection, nanocall.cpp:437-459 / :552-570 ----
    if (read_preferred) {
        for (size_t i = 0; i < 3 * n_reads; ++i) read_preferred[i] = -1;
        if (o->scaling_select_threshold < std::numeric_limits<float>::infinity()) {
            // the jobs of each read, in job order (a scan of every job for every read is quadratic: 8000 reads x 16 000 jobs)
            std::vector<uint32_t> first(n_reads + 1, 0), by_read(n_jobs);
            for (size_t k = 0; k < n_jobs; ++k) if ((size_t)jobs[k].read < n_reads) ++first[(size_t)jobs[k].read + 1];
            for (size_t r = 0; r < n_reads; ++r) first[r + 1] += first[r];
            {
                std::vector<uint32_t> fill(first.begin(), first.end() - 1);
                for (size_t k = 0; k < n_jobs; ++k) if ((size_t)jobs[k].read < n_reads) by_read[fill[(size_t)jobs[k].read]++] = (uint32_t)k;
            }
            for (size_t r = 0; r < n_reads; ++r)
                for (int kind = 0; kind < 3; ++kind) {
                    long best = -1;
                    for (uint32_t q = first[r]; q < first[r + 1]; ++q) {
                        const size_t k = by_read[q];
                        const Job& j = jobs[k];
                        const int jk = (j.m[0] >= 0 && j.m[1] >= 0) ? 2 : (j.m[0] >= 0 ? 0 : 1);
                        if (jk != kind || j.win.empty()) continue;
                        if (best < 0 || j.fit > jobs[best].fit) best = (long)k;   // first maximum
                    }
                    if (best < 0) continue;
                    bool unique = true;
                    for (uint32_t q = first[r]; q < first[r + 1] && unique; ++q) {
                        const size_t k = by_read[q];
                        const Job& j = jobs[k];
                        const int jk = (j.m[0] >= 0 && j.m[1] >= 0) ? 2 : (j.m[0] >= 0 ? 0 : 1);
                        if (jk != kind || j.win.empty() || (long)k == best) continue;
                        if (!(j.fit + o->scaling_select_threshold < jobs[best].fit)) unique = false;
                    }
                    if (unique) read_preferred[3 * r + kind] = (int32_t)best;
                }
        }
    }
    return NCHMM_OK;
}

// basecall_reads, nanocall.cpp:593-868: for every read, Viterbi-decode each candidate model (pair) with
// its trained parameters and keep the one with the highest (summed) path log-probability.
int nchmm_basecall_reads(nchmm_ctx* ctx, const nchmm_train_opts* o, size_t n_models, const float* model_states_Sx10,
                         size_t n_reads, const uint64_t* strand_off, const float* mean, const float* stdv, const float* start,
                         size_t n_jobs, const int32_t* job_read, const int32_t* job_m0, const int32_t* job_m1,
                         const float* job_pm, const float* job_st, const int32_t* read_preferred, uint16_t* out_state,
                         int32_t* out_best_job, float* out_best_logp)
{
    if (!ctx || !o || !model_states_Sx10 || !strand_off || !mean || !stdv || !start || !job_read || !job_m0 || !job_m1
        || !job_pm || !job_st || !out_state || !out_best_job || !out_best_logp)
        return NCHMM_E_INVALID;
    const auto t_fn = std::chrono::steady_clock::now();
    // candidate list (nanocall.cpp:696-709 / :790-806): the preferred job if one was selected, else every job
    struct Cand { size_t job; int strand; size_t vread; };
    std::vector<Cand> cands;
    std::vector<uint64_t> off{0};
    std::vector<int32_t> m_idx, slot_m, slot_t;
    std::vector<float> m_par, t_skip, t_stay;
    for (size_t k = 0; k < n_jobs; ++k) {
        const int r = job_read[k];
        if (r < 0 || (size_t)r >= n_reads) return NCHMM_E_INVALID;
        const int kind = (job_m0[k] >= 0 && job_m1[k] >= 0) ? 2 : (job_m0[k] >= 0 ? 0 : 1);
        if (read_preferred && read_preferred[3 * r + kind] >= 0 && read_preferred[3 * r + kind] != (int32_t)k) continue;
        const int m[2] = {job_m0[k], job_m1[k]};
        for (int s = 0; s < 2; ++s) {
            if (m[s] < 0) continue;
            if (m[s] >= (int)n_models) return NCHMM_E_INVALID;
            const uint64_t b = strand_off[2 * r + s], e = strand_off[2 * r + s + 1];
            // single-strand candidates skip short strands (nanocall.cpp:788); a 2D pair decodes both of its strands
            // whatever their length (:715-732) -- only an empty one is left out (the reference would read ev[0])
            if (kind == 2 ? e == b : e - b < o->min_ed_events) continue;
            off.push_back(off.back() + (e - b));
            const size_t v = cands.size();
            cands.push_back(Cand{k, s, v});
            m_idx.push_back(m[s]);
            m_par.insert(m_par.end(), job_pm + 6 * k, job_pm + 6 * k + 6);
            const float p_stay = job_st[4 * k + 2 * s], p_skip = job_st[4 * k + 2 * s + 1];
            t_stay.push_back(p_stay); t_skip.push_back(p_skip);
            slot_m.push_back((int32_t)v);
            slot_t.push_back((int32_t)v);
        }
    }
    // Every candidate's events are drift-corrected with ITS parameters and get their log_stdv ON THE DEVICE, and its scaled
    // model / transitions are built there too, range by range in front of each range's kernels (nchmm_pipeline.cpp): the raw
    // events go up once, range k+1's share while range k computes; only the candidate table is built here.
    const bool dbg_time = std::getenv("NCHMM_DEBUG") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    const size_t tot = (size_t)off.back();
    const size_t nc = cands.size();
    std::vector<uint64_t> c_src(nc);
    std::vector<uint32_t> c_len(nc);
    std::vector<float> c_drift(nc);
    std::vector<long> last_cand(2 * n_reads, -1);        // per (read, strand-or-pair key): its last candidate
    for (size_t v = 0; v < nc; ++v) {
        const size_t k = cands[v].job;
        const int r = job_read[k];
        c_src[v] = strand_off[2 * r + cands[v].strand];
        c_len[v] = (uint32_t)(off[v + 1] - off[v]);
        c_drift[v] = job_pm[6 * k + 2];   // corrected_events.apply_drift_correction(pm_params.drift), nanocall.cpp:685-686
        last_cand[2 * r] = last_cand[2 * r + 1] = (long)v;   // (a read's candidates decide together: both keys wait for the last)
    }
    for (size_t i = 0; i < 2 * n_reads; ++i) { out_best_job[i] = -1; out_best_logp[i] = std::numeric_limits<float>::quiet_NaN(); }
    if (cands.empty()) return NCHMM_OK;
    (void)slot_m; (void)slot_t;
    const nchmm::PipeTables tab{model_states_Sx10, n_models, m_idx.data(), m_par.data(), t_skip.data(), t_stay.data()};
    int rc = nchmm::pipe_raw_tables_begin(ctx, (size_t)strand_off[2 * n_reads], mean, stdv, start, nc, c_src.data(), c_len.data(),
                                          c_drift.data(), tab);
    if (rc != NCHMM_OK) return rc;
    const auto t_1 = std::chrono::steady_clock::now();
    const uint16_t* states = nullptr; const float* logp = nullptr; const int32_t* status = nullptr;
    nchmm::pipe_results(ctx, &states, &logp, &status);
    // choose per read: 2D jobs by the float sum of both strands (:725-739), 1D jobs per strand (:807-825);
    // `sort ... back()` = the highest value, the later candidate among exact ties
    std::vector<float> best_total(2 * n_reads, -std::numeric_limits<float>::infinity());
    std::vector<long> best_c0(2 * n_reads, -1), best_c1(2 * n_reads, -1);
    // reads in the order in which their last candidate completes
    std::vector<uint32_t> by_last;
    for (size_t r = 0; r < n_reads; ++r) if (last_cand[2 * r] >= 0) by_last.push_back((uint32_t)r);
    std::stable_sort(by_last.begin(), by_last.end(), [&](uint32_t a, uint32_t b) { return last_cand[2 * a] < last_cand[2 * b]; });
    size_t scan = 0, emit_pos = 0;
    bool any_numeric = false;
    auto emit = [&](long v) {
        const Cand& c = cands[(size_t)v];
        const int r = job_read[c.job];
        const uint64_t b = strand_off[2 * r + c.strand], n = strand_off[2 * r + c.strand + 1] - b;
        if (status[(size_t)v] == 0) std::memcpy(out_state + b, states + off[(size_t)v], n * sizeof(uint16_t));
        out_best_job[2 * r + c.strand] = (int32_t)c.job;
        out_best_logp[2 * r + c.strand] = logp[(size_t)v];
    };
    double ms_wait = 0, ms_host = 0;
    const size_t n_ranges = nchmm::pipe_n_ranges(ctx);
    for (size_t g = 0; g < n_ranges; ++g) {
        size_t r0 = 0, r1 = 0;
        const auto t_a = std::chrono::steady_clock::now();
        if ((rc = nchmm::pipe_wait_range(ctx, g, &r0, &r1)) != NCHMM_OK) { (void)nchmm::pipe_release(ctx); return rc; }
        const auto t_b = std::chrono::steady_clock::now();
        // the candidates that have landed: [scan, r1), except a 2D pair whose second strand is in the next range
        while (scan < r1) {
            const size_t v = scan;
            const size_t k = cands[v].job;
            const int r = job_read[k];
            const bool two_d = job_m0[k] >= 0 && job_m1[k] >= 0;
            // A candidate that failed to decode (status != 0, NaN log-probability) must not shadow a later valid one: the
            // reference's `sort` has no defined order with a NaN key, here a NaN ranks below everything (-INF) and is kept
            // only while nothing else has been seen.
            auto key = [](float x) { return std::isnan(x) ? -std::numeric_limits<float>::infinity() : x; };
            if (two_d && v + 1 < nc && cands[v + 1].job == k) {
                if (v + 1 >= r1) break;
                const float tot2 = key(logp[v] + logp[v + 1]);
                if (tot2 >= best_total[2 * r] || best_c0[2 * r] < 0) { best_total[2 * r] = tot2; best_c0[2 * r] = (long)v; best_c1[2 * r] = (long)v + 1; }
                any_numeric |= status[v] != 0 || status[v + 1] != 0;
                scan += 2;
            } else {
                const int s = cands[v].strand;
                if (!two_d && (key(logp[v]) >= best_total[2 * r + s] || best_c0[2 * r + s] < 0)) {
                    // a 1D job on a read that also has a 2D winner is a different mode; the caller passes one mode per read
                    best_total[2 * r + s] = key(logp[v]); best_c0[2 * r + s] = (long)v; best_c1[2 * r + s] = -1;
                }
                any_numeric |= status[v] != 0;
                scan += 1;
            }
        }
        // the reads all of whose candidates have been seen: winners' states -> the caller's array, under the next range's kernels
        const size_t e0 = emit_pos;
        while (emit_pos < by_last.size() && (size_t)last_cand[2 * by_last[emit_pos]] < scan) ++emit_pos;
        parallel_for(emit_pos - e0, [&](size_t lo, size_t hi) {     // (per read: its slices belong to no other read)
            for (size_t q = e0 + lo; q < e0 + hi; ++q)
                for (size_t i = 2 * (size_t)by_last[q]; i < 2 * (size_t)by_last[q] + 2; ++i) {
                    if (best_c0[i] >= 0) emit(best_c0[i]);
                    if (best_c1[i] >= 0) emit(best_c1[i]);
                }
        });
        const auto t_c = std::chrono::steady_clock::now();
        ms_wait += std::chrono::duration<double, std::milli>(t_b - t_a).count();
        ms_host += std::chrono::duration<double, std::milli>(t_c - t_b).count();
    }
    rc = nchmm::pipe_release(ctx);
    if (rc != NCHMM_OK) return rc;
    if (dbg_time) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[nchmm_basecall_reads] %zu candidates, %zu events, %zu ranges: candidate table %.2f ms, begin (tables + copy-in + launches) %.2f ms, "
                             "waiting for ranges %.2f ms, choose+copy %.2f ms\n",
                     nc, tot, n_ranges, ms(t_fn, t_0), ms(t_0, t_1), ms_wait, ms_host);
    }
    return any_numeric ? NCHMM_E_NUMERIC : NCHMM_OK;
}

}  // extern "C"
