// nchmm_pipe.hpp -- the host-pointer Viterbi pipeline (nchmm_pipeline.cpp) as the library's own callers use it:
// nchmm_basecall_reads decodes thousands of candidates whose scaled models and transitions are built on the device, range
// by range in front of each range's kernels, and picks winners range by range as the results land in host memory.
#ifndef NCHMM_PIPE_HPP
#define NCHMM_PIPE_HPP

#include <cstddef>
#include <cstdint>

#include "nanocall_hip.h"

namespace nchmm {

// Candidate v gets model slot v = table table_idx[v] scaled by params_nx6[v] (Pore_Model::scale, as
// nchmm_put_models_scaled) and transition slot v = compute_transitions_fast(p_skip[v], p_stay[v]).
struct PipeTables {
    const float* states_Sx10;
    size_t n_tables;
    const int32_t* table_idx;
    const float* params_nx6;
    const float* p_skip;
    const float* p_stay;
};

// nchmm_viterbi_raw_begin with the tables of every candidate built by the pipeline itself.  One batch at a time.
int pipe_raw_tables_begin(nchmm_ctx* c, size_t n_raw, const float* mean, const float* stdv, const float* start, size_t n_cand,
                          const uint64_t* src, const uint32_t* len, const float* drift, const PipeTables& tab);
size_t pipe_n_ranges(const nchmm_ctx* c);
// Block until range k of the oldest batch has finished; candidates [*r0, *r1) are then valid in the arrays below.
int pipe_wait_range(nchmm_ctx* c, size_t k, size_t* r0, size_t* r1);
// The oldest batch's results where the kernels wrote them (pinned host memory, states packed by candidate).
void pipe_results(const nchmm_ctx* c, const uint16_t** states, const float** logp, const int32_t** status);
// End the oldest batch without copying anything anywhere.
int pipe_release(nchmm_ctx* c);

}  // namespace nchmm
#endif
