// bench_cpp_layer.cpp -- what a nanocall maintainer gets after the header swap: Viterbi::fill_batch over many strands
// (AoS Event_Sequence in, model_state_idx / model_state / move written back), wall time including the host loops
// on both sides of nchmm_viterbi.   bench_cpp_layer [reads] [events] [threads]   (synthetic events: a noisy walk over the model)
// With [threads] > 0 also the reference's own call shape: that many worker threads, each calling Viterbi::fill on one strand at
// a time with its own copy of the model (basecall_strand inside the pfor, nanocall.cpp:611-621,645-690) -- every result compared
// with what fill_batch decoded.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <atomic>
#include <thread>

#include "nanocall_amd/nanocall_amd.hpp"

using namespace nanocall_amd;
typedef Pore_Model<float, 6> Pore_Model_Type;
typedef State_Transitions<float, 6> State_Transitions_Type;
typedef Event<float, 6> Event_Type;
typedef Event_Sequence<float, 6> Event_Sequence_Type;
typedef Viterbi<float, 6> Viterbi_Type;

int main(int argc, char* argv[])
{
    const size_t n_reads = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 1024, n_ev = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 5000;
    const size_t n_threads = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 0;
    try {
        // a smooth synthetic model: levels 45..95 pA over the k-mer index, sd columns typical of r7.3
        std::vector<float> table(4096 * 4);
        std::mt19937 rng(7);
        std::uniform_real_distribution<float> u(0.f, 1.f);
        for (unsigned j = 0; j < 4096; ++j) {
            table[4 * j] = 45.f + 50.f * u(rng); table[4 * j + 1] = 1.0f + u(rng);
            table[4 * j + 2] = 0.9f + 0.6f * u(rng); table[4 * j + 3] = 0.3f + 0.2f * u(rng);
        }
        Pore_Model_Type pm;
        pm.load_from_vector(table);
        State_Transitions_Type st;
        st.compute_transitions_fast(.3f, .1f);
        std::vector<Event_Sequence_Type> reads(n_reads);
        std::normal_distribution<float> g(0.f, 1.f);
        for (auto& ev : reads) {
            unsigned k = rng() & 4095u;
            float t = 0;
            for (size_t i = 0; i < n_ev; ++i) {
                const float r = u(rng);
                if (r >= .1f) k = r < .7f ? ((k << 2) | (rng() & 3u)) & 4095u : ((k << 4) | (rng() & 15u)) & 4095u;
                Event_Type e;
                e.mean = table[4 * k] + table[4 * k + 1] * g(rng); e.stdv = std::max(0.05f, table[4 * k + 2] + 0.3f * table[4 * k + 3] * g(rng));
                e.start = t; e.length = 0.01f; t += e.length; e.corrected_mean = e.mean;
                e.update_logs();
                ev.push_back(e);
            }
        }
        std::vector<Event_Sequence_Type*> ptrs;
        for (auto& ev : reads) ptrs.push_back(&ev);
        Viterbi_Type::fill_batch(pm, st, ptrs);   // warm-up: context, workspace
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<float> pp = Viterbi_Type::fill_batch(pm, st, ptrs);
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = std::min(best, s);
        }
        float km[4];
        nchmm_last_kernel_ms(Device::instance().ctx(), km);
        std::string seq = reads[0].get_base_seq();
        std::printf("{\"what\": \"Viterbi::fill_batch (C++ layer, host AoS in / annotated events out)\", \"reads\": %zu, \"events_per_read\": %zu, "
                    "\"wall_s\": %.4f, \"Mevents_per_s\": %.1f, \"kernel_ms\": [%.2f, %.2f], \"read0_bases\": %zu}\n",
                    n_reads, n_ev, best, n_reads * n_ev / best / 1e6, km[0], km[1], seq.size());
        if (n_threads) {
            // what fill_batch wrote into the events is the expectation
            std::vector<std::vector<unsigned>> want(n_reads);
            for (size_t r = 0; r < n_reads; ++r) { want[r].reserve(n_ev); for (const auto& e : reads[r]) want[r].push_back(e.model_state_idx); }
            std::vector<float> want_pp = Viterbi_Type::fill_batch(pm, st, ptrs);
            double best_t = 1e30;
            std::atomic<long> bad{0};
            for (int rep = 0; rep < 3; ++rep) {
                for (auto& ev : reads) for (auto& e : ev) e.model_state_idx = 0xFFFFFFFFu;
                std::atomic<size_t> next{0};
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> workers;
                for (size_t w = 0; w < n_threads; ++w)
                    workers.emplace_back([&] {
                        for (;;) {
                            const size_t r = next.fetch_add(1);
                            if (r >= n_reads) break;
                            Pore_Model_Type my_pm = pm;            // basecall_strand scales its own copy (nanocall.cpp:653-657)
                            Viterbi_Type vit;
                            vit.fill(my_pm, st, reads[r]);
                            if (vit.path_probability() != want_pp[r]) bad++;
                        }
                    });
                for (auto& t : workers) t.join();
                best_t = std::min(best_t, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                for (size_t r = 0; r < n_reads; ++r)
                    for (size_t i = 0; i < n_ev; ++i) if (reads[r][i].model_state_idx != want[r][i]) { bad++; break; }
            }
            std::printf("{\"what\": \"Viterbi::fill, one strand per call from %zu worker threads (combined by nchmm_viterbi_strand)\", \"reads\": %zu, "
                        "\"events_per_read\": %zu, \"wall_s\": %.4f, \"Mevents_per_s\": %.1f, \"mismatches_vs_fill_batch\": %ld}\n",
                        n_threads, n_reads, n_ev, best_t, n_reads * n_ev / best_t / 1e6, bad.load());
            if (bad.load()) return 2;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
