// em_kernel.hip -- the two small kernels that keep an EM round on the device (nchmm_em_round):
//
//   em_gather_kernel  Parameter_Trainer::fill_train_data's per-window copy + drift correction
//                     (Parameter_Trainer.hpp:130-140, Event.hpp:77-84) from the resident raw event arrays into the
//                     packed SoA the forward-backward kernels read: corrected_mean = mean - drift * start (two
//                     roundings, as the reference), stdv and log_stdv copied.
//   em_reduce_kernel  the outer sums of train_pm_params (Parameter_Trainer.hpp:297-312) over one job's windows:
//                     the six per-event inner sums times the UNCORRECTED event's mean / stdv / start, products in
//                     float exactly as the reference writes them, accumulated in double.  13 doubles per job go back
//                     to the host, which solves the 3x3 system (nchmm_train_pm_solve).
#include "nchmm_device.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace nchmm {

constexpr unsigned kGatherChunk = 4096;   // events per block along a window / strand (blockIdx.y)

__global__ __launch_bounds__(256) void em_gather_kernel(EmGatherArgs P)
{
    const unsigned w = blockIdx.x;
    const uint64_t src = P.win_src[w], dst = P.off[w];
    const unsigned n = (unsigned)(P.off[w + 1] - dst);
    const float drift = P.win_drift[w];
    const unsigned lo = blockIdx.y * kGatherChunk, hi = lo + kGatherChunk < n ? lo + kGatherChunk : n;
    for (unsigned i = lo + threadIdx.x; i < hi; i += 256) {
        float c = P.mean[src + i];
        c -= drift * P.start[src + i];          // apply_drift_correction
        P.cmean[dst + i] = c;
        float sd = P.stdv[src + i];
        if (P.lstdv) {
            P.out_lstdv[dst + i] = P.lstdv[src + i];
        } else {
            if (sd == 0.0f) sd = 0.01f;         // Event::update_logs, Event.hpp:39-42
            P.out_lstdv[dst + i] = glibc_logf(sd);
        }
        P.out_stdv[dst + i] = sd;
    }
}

__global__ __launch_bounds__(256) void logf_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = glibc_logf(in[i]);
}

__global__ __launch_bounds__(256) void em_reduce_kernel(EmReduceArgs P)
{
    __shared__ double sAcc[256 / 64][13];
    const unsigned job = blockIdx.x;
    double acc[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) acc[k] = 0.0;
    for (unsigned w = P.job_first_win[job]; w < P.job_first_win[job + 1]; ++w) {
        const uint64_t src = P.win_src[w], dst = P.off[w];
        const unsigned n = (unsigned)(P.off[w + 1] - dst);
        for (unsigned i = threadIdx.x; i < n; i += 256) {
            const float* s = P.pm_sums + 6 * (dst + i);
            const float x = P.mean[src + i], y = P.stdv[src + i], t = P.start[src + i];
            acc[0] += s[0];                    // A00
            acc[1] += s[1];                    // A01
            acc[2] += s[2];                    // A11
            acc[3] += s[0] * x;                // B0   (float product, as the reference's `s[0] * x_i`)
            acc[4] += s[1] * x;                // B1
            if (P.train_drift) {
                acc[5] += s[0] * t;            // A02
                acc[6] += s[1] * t;            // A12
                acc[7] += s[0] * t * t;        // A22
                acc[8] += s[0] * x * t;        // B2
            }
            acc[9] += s[0] * x * x;            // D
            acc[10] += s[5] * y;               // V_numer
            acc[11] += s[4];                   // V_denom
            acc[12] += s[3] / y;               // U_pos
        }
    }
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        double v = acc[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) sAcc[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 13) P.out[13 * (size_t)job + threadIdx.x] = (sAcc[0][threadIdx.x] + sAcc[1][threadIdx.x]) + (sAcc[2][threadIdx.x] + sAcc[3][threadIdx.x]);
}

void launch_em_gather(const EmGatherArgs& a, unsigned n_win, hipStream_t stream, unsigned max_events)
{
    const unsigned chunks = max_events > kGatherChunk ? (max_events + kGatherChunk - 1) / kGatherChunk : 1;
    if (n_win) hipLaunchKernelGGL(em_gather_kernel, dim3(n_win, chunks), dim3(256), 0, stream, a);
}

void launch_logf(const float* in, float* out, size_t n, hipStream_t stream)
{
    if (n) hipLaunchKernelGGL(logf_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0, stream, in, out, n);
}

void launch_em_reduce(const EmReduceArgs& a, unsigned n_jobs, hipStream_t stream)
{
    if (n_jobs) hipLaunchKernelGGL(em_reduce_kernel, dim3(n_jobs), dim3(256), 0, stream, a);
}

}  // namespace nchmm
