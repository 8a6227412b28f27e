// nchmm_internal.hpp -- host-side arithmetic shared by nchmm_host.cpp (single-table ABI functions) and
// nchmm_api.cpp (batched uploads).  Inline so both translation units execute the same float operations.
#ifndef NCHMM_INTERNAL_HPP
#define NCHMM_INTERNAL_HPP

#include <cmath>

#include "nchmm_kmer.hpp"

namespace nchmm {

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
