// emission_kernel.hip -- the emission log-densities of whole reads, computed AHEAD of their Viterbi sweep.
//
// A read's sweep is a recurrence over its events: one column after the other, however many CUs are idle.  But 60 % of the
// arithmetic of a column -- the emission log_pr_corrected_emission(state, event) (Pore_Model.hpp:145-149, 19 float operations
// and three divisions per cell) -- does not depend on the recurrence at all.  For the reads that set the duration of a launch
// (the longest reads of a ragged batch, a strand decoded on its own: nanocall.cpp:687-689) it is computed here, by every CU of
// the device at once and in no particular order, into rows of 4096 floats per event; the sweep of such a read
// (viterbi_ll_kernel.hip, column_ahead) then carries only the max-plus recurrence and adds the row: about half the time per
// event.  16 KiB per event written and read back once -- affordable for the few reads that need it, not for a whole batch
// (nchmm_plan.hpp decides which).
//
// Bit-exactness: the same emission<>() as the sweeps (viterbi_common.hpp), the reciprocal-division form inside its validated
// range and true division outside, chosen per event here (the two are bit-identical where both apply).
// Layout: row-major by event, thread-major inside a row for the low-latency sweep's ownership map: thread tau = 4t + y of that
// sweep owns the states t + 256 (4x + y), x = 0..3, and finds them at floats [4 tau, 4 tau + 4) of the row -- one 16-byte load.
#include "nchmm_device.h"

#pragma clang fp contract(off)

namespace nchmm {

namespace {

#include "viterbi_common.hpp"

constexpr unsigned kEmChunk = 32;      // events a block computes per work item

}  // namespace

__global__ __launch_bounds__(kLlThreads) void emission_kernel(ViterbiArgs P, unsigned n_ahead, float* __restrict__ em)
{
    const unsigned tau = threadIdx.x, t = tau >> 2, yy = tau & 3u;
    for (unsigned k = blockIdx.y; k < n_ahead; k += gridDim.y) {
        const unsigned r = P.order ? P.order[k] : P.first_read + k;
        const uint64_t e0 = P.off[r];
        const unsigned n = (unsigned)(P.off[r + 1] - e0);
        if (blockIdx.x * kEmChunk >= n) continue;
        const uint64_t row0 = P.em_row0 ? P.em_row0[r] : e0;
        if (row0 == kNoEmRow || row0 > P.em_rows || n > P.em_rows - row0) continue;   // (not ahead, or not inside the buffer: the sweep computes in place)
        const int ms = P.model_slot ? P.model_slot[r] : 0;
        const float* __restrict__ M = P.models + (size_t)ms * kModelFloats;
        const bool model_fast = P.model_fast[ms] != 0;
        float mu[4], sg[4], rsg[4], eta[4], reta[4], lam[4], nls[4], cc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned j = t + 256u * (4u * (unsigned)i + yy);
            mu[i] = M[MF_MU * kStates + j]; sg[i] = M[MF_SIGMA * kStates + j]; rsg[i] = M[MF_RSIGMA * kStates + j];
            eta[i] = M[MF_ETA * kStates + j]; reta[i] = M[MF_RETA * kStates + j]; lam[i] = M[MF_LAMBDA * kStates + j];
            nls[i] = M[MF_NEG_LOG_SIGMA * kStates + j]; cc[i] = M[MF_C * kStates + j];
        }
        const float* __restrict__ ex = P.cmean + e0;
        const float* __restrict__ ey = P.stdv + e0;
        const float* __restrict__ el = P.lstdv + e0;
        for (unsigned base = blockIdx.x * kEmChunk; base < n; base += gridDim.x * kEmChunk) {
            const unsigned hi = base + kEmChunk < n ? base + kEmChunk : n;
            for (unsigned i = base; i < hi; ++i) {
                const float x = ex[i], y = ey[i], ly3 = 3.0f * el[i], ry = 1.0f / y;     // (uniform: scalar loads)
                float4 e;
                if (model_fast && event_in_fast_range(x, y)) {
                    e.x = emission<true>(x, y, ry, ly3, P.log_2pi, mu[0], sg[0], rsg[0], nls[0], eta[0], reta[0], lam[0], cc[0]);
                    e.y = emission<true>(x, y, ry, ly3, P.log_2pi, mu[1], sg[1], rsg[1], nls[1], eta[1], reta[1], lam[1], cc[1]);
                    e.z = emission<true>(x, y, ry, ly3, P.log_2pi, mu[2], sg[2], rsg[2], nls[2], eta[2], reta[2], lam[2], cc[2]);
                    e.w = emission<true>(x, y, ry, ly3, P.log_2pi, mu[3], sg[3], rsg[3], nls[3], eta[3], reta[3], lam[3], cc[3]);
                } else {
                    e.x = emission<false>(x, y, ry, ly3, P.log_2pi, mu[0], sg[0], rsg[0], nls[0], eta[0], reta[0], lam[0], cc[0]);
                    e.y = emission<false>(x, y, ry, ly3, P.log_2pi, mu[1], sg[1], rsg[1], nls[1], eta[1], reta[1], lam[1], cc[1]);
                    e.z = emission<false>(x, y, ry, ly3, P.log_2pi, mu[2], sg[2], rsg[2], nls[2], eta[2], reta[2], lam[2], cc[2]);
                    e.w = emission<false>(x, y, ry, ly3, P.log_2pi, mu[3], sg[3], rsg[3], nls[3], eta[3], reta[3], lam[3], cc[3]);
                }
                reinterpret_cast<float4*>(em + (row0 + i) * (uint64_t)kStates)[tau] = e;
            }
        }
    }
}

void launch_emissions(const ViterbiArgs& a, unsigned n_ahead, uint64_t max_events, float* em, hipStream_t stream)
{
    if (n_ahead == 0 || max_events == 0) return;
    // x: chunks of a read (as many as the longest has, at most 512 per read: the rest by stride); y: reads (at most 1024 rows of
    // blocks: the rest by stride)
    const unsigned gx = (unsigned)std::min<uint64_t>((max_events + kEmChunk - 1) / kEmChunk, 512);
    const unsigned gy = std::min(n_ahead, 1024u);
    hipLaunchKernelGGL(emission_kernel, dim3(gx, gy), dim3(kLlThreads), 0, stream, a, n_ahead, em);
}

}  // namespace nchmm
