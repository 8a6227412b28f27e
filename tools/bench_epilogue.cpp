// bench_epilogue.cpp -- does the HOST keep up with the GPUs on the epilogue of the Viterbi path?  (second half)
//
// After the traceback the reference computes, per decoded strand, the moves (Viterbi::fill_move_seq, Viterbi.hpp:144-150, with
// Kmer::min_skip, Kmer.hpp:51-68), the base string (Event_Sequence::get_base_seq, Event.hpp:85-99) and the FASTA record
// (nanocall.cpp:584-591).  Here that is nchmm_base_seq + nchmm_write_fasta on the host.  This tool times exactly those two
// calls through the C ABI on T host threads (one strand of 5000 events per call, as the command line's worker threads make
// them) and compares the rate with what eight GPUs decode.  No GPU is touched.  One JSON line.
//     ./bench_epilogue [strands=16384] [threads...=1 8 32 64 128]      GPU_MEVENTS_PER_S=316 (one MI355X, config 2)
#include "nanocall_hip.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv)
{
    const size_t n_events = 5000;
    const size_t n_strands = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 16384;
    std::vector<int> threads;
    for (int i = 2; i < argc; ++i) threads.push_back(std::atoi(argv[i]));
    if (threads.empty()) threads = {1, 8, 32, 64, 128};
    const double gpu_rate = std::getenv("GPU_MEVENTS_PER_S") ? std::atof(std::getenv("GPU_MEVENTS_PER_S")) : 316.0;
    const unsigned hw = std::thread::hardware_concurrency();
    // decoded paths with the generator's stay / step / skip mix: .10 / .60 / .30
    std::vector<uint16_t> st(n_strands * n_events);
    for (size_t r = 0; r < n_strands; ++r) {
        uint64_t seed = 0x6E616E6Full ^ r;
        unsigned k = (unsigned)(splitmix(seed) & 4095u);
        for (size_t i = 0; i < n_events; ++i) {
            const uint64_t z = splitmix(seed);
            const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
            const unsigned b = (unsigned)(z & 15u);
            if (u >= 0.10) k = u < 0.70 ? ((k << 2) | (b & 3u)) & 4095u : ((k << 4) | b) & 4095u;
            st[r * n_events + i] = (uint16_t)k;
        }
    }
    std::string json = "{\"what\": \"nchmm_base_seq (moves + base string) + nchmm_write_fasta (80-column record) per strand of 5000 events, C ABI, "
                       "std::thread workers\", \"strands\": " + std::to_string(n_strands) + ", \"host_logical_cpus\": " + std::to_string(hw) +
                       ", \"gpu_rate_Mevents_per_s_one_gpu\": " + std::to_string(gpu_rate) + ", \"needed_Mevents_per_s_at_8_gpus\": " +
                       std::to_string(8 * gpu_rate) + ", \"by_threads\": {";
    double best = 0;
    size_t bases = 0;
    bool first = true;
    for (int T : threads) {
        if (T < 1 || (hw && (unsigned)T > hw)) continue;
        std::vector<size_t> nb((size_t)T, 0);
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int tid = 0; tid < T; ++tid)
            th.emplace_back([&, tid] {
                std::vector<int32_t> mv(n_events);
                std::vector<char> seq(6 * n_events + 1), rec(8 * n_events + 256);
                for (size_t r = (size_t)tid; r < n_strands; r += (size_t)T) {
                    size_t len = 0, wr = 0;
                    if (nchmm_base_seq(n_events, &st[r * n_events], mv.data(), seq.data(), &len) != 0) std::abort();
                    seq[len] = 0;
                    if (nchmm_write_fasta("read:file:0", seq.data(), 80, rec.data(), rec.size(), &wr) != 0) std::abort();
                    nb[(size_t)tid] += len;
                }
            });
        for (auto& t : th) t.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double rate = (double)(n_strands * n_events) / dt / 1e6;
        bases = 0;
        for (size_t v : nb) bases += v;
        if (rate > best) best = rate;
        char buf[256];
        std::snprintf(buf, sizeof buf, "%s\"%d\": {\"Mevents_per_s\": %.1f, \"seconds\": %.3f, \"x_of_8_gpus\": %.2f}", first ? "" : ", ", T, rate, dt,
                      rate / (8 * gpu_rate));
        json += buf;
        first = false;
    }
    char tail[256];
    std::snprintf(tail, sizeof tail, "}, \"bases_per_event\": %.4f, \"host_keeps_up_with_8_gpus\": %s}", (double)bases / (double)(n_strands * n_events),
                  best >= 8 * gpu_rate ? "true" : "false");
    json += tail;
    std::puts(json.c_str());
    return 0;
}
