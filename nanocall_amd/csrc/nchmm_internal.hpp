// nchmm_internal.hpp -- host-side arithmetic shared by nchmm_host.cpp (single-table ABI functions) and
// nchmm_api.cpp (batched uploads).  Inline so both translation units execute the same float operations.
#ifndef NCHMM_INTERNAL_HPP
#define NCHMM_INTERNAL_HPP

#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#include "nchmm_kmer.hpp"

namespace nchmm {

// f(begin, end) over [0, n) on the host cores (the work items are independent)
template <typename F>
void parallel_for(size_t n, F&& f)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt ? std::min<unsigned>(nt, 32) : 4;
    if (n < 4 || nt < 2) { f(0, n); return; }
    nt = (unsigned)std::min<size_t>(nt, n);
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; ++i) th.emplace_back([&, i] { f(n * i / nt, n * (i + 1) / nt); });
    for (auto& t : th) t.join();
}


// State_Transitions::get_trans_prob, State_Transitions.hpp:125-144: float accumulator, double pow terms
inline float trans_prob(unsigned i, unsigned j, float p_stay, float p_step, float p_skip_1)
{
    float p = 0;
    if (i == j) p += p_stay;
    if (Kmer6::suffix(i, 5) == Kmer6::prefix(j, 5)) p += p_step / 4;
    for (unsigned l = 2; l < 6; ++l)
        if (Kmer6::suffix(i, 6 - l) == Kmer6::prefix(j, 6 - l))
            p += std::pow(static_cast<double>(p_skip_1), static_cast<double>(l - 1)) / (1u << (2 * l));
    p += (std::pow(static_cast<double>(p_skip_1), 5.0) / (1.0f - p_skip_1)) / 4096u;
    return p;
}

// compute_transitions_fast :198-202
inline void step_params(float p_skip, float p_stay, float& p_step, float& p_skip_1)
{
    p_step = static_cast<float>(1.0 - p_stay - p_skip);
    p_skip_1 = static_cast<float>(p_skip / (p_skip + 1.0));
}

}  // namespace nchmm
#endif
