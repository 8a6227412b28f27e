// bench_train_threads.cpp -- Parameter_Trainer::train_one_round, one read per call, from T worker threads: the reference's own call
// shape (the body of the pfor in train_reads, nanocall.cpp:282-579, calling Parameter_Trainer.hpp:541-579) through the header swap.
// Every call brings a read's two scaled models and its four 100-event windows; nchmm_fwbw_windows combines the calls in progress.
//   bench_train_threads [reads] [threads...]      one JSON line per thread count; fits compared with the first run
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>

#include "nanocall_amd/nanocall_amd.hpp"

using namespace nanocall_amd;
typedef Pore_Model<float, 6> PM;
typedef State_Transitions<float, 6> ST;
typedef Event<float, 6> EV;
typedef Event_Sequence<float, 6> ES;
typedef Parameter_Trainer<float, 6> PT;

int main(int argc, char* argv[])
{
    const size_t n_reads = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 2048;
    std::vector<size_t> thread_counts;
    for (int a = 2; a < argc; ++a) thread_counts.push_back(std::strtoul(argv[a], nullptr, 10));
    if (thread_counts.empty()) thread_counts = {1, 16, 64, 256, 1024};
    try {
        std::mt19937 rng(7);
        std::uniform_real_distribution<float> u(0.f, 1.f);
        std::normal_distribution<float> g(0.f, 1.f);
        PM pm[2];
        std::vector<float> table[2];
        for (int s = 0; s < 2; ++s) {
            table[s].resize(4096 * 4);
            for (unsigned j = 0; j < 4096; ++j) {
                table[s][4 * j] = 45.f + 50.f * u(rng); table[s][4 * j + 1] = 1.0f + u(rng);
                table[s][4 * j + 2] = 0.9f + 0.6f * u(rng); table[s][4 * j + 3] = 0.3f + 0.2f * u(rng);
            }
            pm[s].load_from_vector(table[s]);
        }
        ST st;
        st.compute_transitions_fast(State_Transition_Parameters<float>());   // default_transitions (nanocall.cpp:1010-1013)
        PT::init();
        // per read: 2 strands x 2 windows x 100 events
        std::vector<std::array<ES, 4>> win(n_reads);
        for (auto& W : win)
            for (int w = 0; w < 4; ++w) {
                const std::vector<float>& T = table[w / 2];
                unsigned k = rng() & 4095u; float t = 0;
                for (int i = 0; i < 100; ++i) {
                    const float r = u(rng);
                    if (r >= .1f) k = r < .7f ? ((k << 2) | (rng() & 3u)) & 4095u : ((k << 4) | (rng() & 15u)) & 4095u;
                    EV e; e.mean = T[4 * k] + T[4 * k + 1] * g(rng); e.stdv = std::max(0.05f, T[4 * k + 2] + 0.3f * T[4 * k + 3] * g(rng));
                    e.start = t; e.length = 0.01f; t += e.length; e.corrected_mean = e.mean; e.update_logs();
                    W[w].push_back(e);
                }
            }
        std::vector<float> want_fit(n_reads, 0.f);
        bool have_ref = false;
        for (size_t T : thread_counts) {
            double best = 1e30;
            std::atomic<long> bad{0};
            std::vector<float> fits(n_reads);
            for (int rep = 0; rep < 3; ++rep) {
                std::atomic<size_t> next{0};
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> workers;
                for (size_t w = 0; w < T; ++w)
                    workers.emplace_back([&] {
                        for (;;) {
                            const size_t r = next.fetch_add(1);
                            if (r >= n_reads) break;
                            std::vector<std::pair<const ES*, unsigned>> ptrs;
                            for (int q = 0; q < 4; ++q) ptrs.emplace_back(&win[r][q], (unsigned)(q / 2));
                            const std::array<const PM*, 2> models{{&pm[0], &pm[1]}};
                            Pore_Model_Parameters<float> crt, nw;
                            crt.scale = 1.0f + 0.0001f * (float)(r % 7); crt.shift = 0.1f * (float)(r % 5);
                            std::array<State_Transition_Parameters<float>, 2> cst, nst;
                            float fit = 0; bool done = false;
                            PT::train_one_round(ptrs, models, st, crt, cst, nw, nst, fit, done, true, true);
                            fits[r] = fit;
                        }
                    });
                for (auto& t : workers) t.join();
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                if (have_ref) { for (size_t r = 0; r < n_reads; ++r) if (fits[r] != want_fit[r]) bad++; }
                else { want_fit = fits; have_ref = true; }
            }
            std::printf("{\"what\": \"Parameter_Trainer::train_one_round, one read per call (2 models, 4 windows x 100 events) from %zu worker threads\", "
                        "\"reads\": %zu, \"event_rounds\": %zu, \"wall_s\": %.4f, \"Mevent_rounds_per_s\": %.2f, \"fits_differing_from_first_run\": %ld}\n",
                        T, n_reads, 400 * n_reads, best, 400.0 * n_reads / best / 1e6, bad.load());
            std::fflush(stdout);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
