// plan_kernel.hip -- the processing order of a batch whose read lengths are only on the device (nchmm_viterbi_dev[_enqueue]).
//
// The host-pointer forms sort their reads longest first on the host and give the few reads that are too long for the pooled
// back-pointer regions regions of their own (nchmm_plan.hpp: order_ranges, plan_outliers).  A device-pointer caller's offsets
// are in device memory -- possibly written by a kernel that has not run yet -- so the same plan is made here, by one block on
// the launch's own stream, in front of the sweep:
//   * order[]     the reads that are NOT outliers, longest first up to the resolution of 2048 length classes (a launch hands
//                 reads out in this order; within a class the order is whatever the atomics give -- reads are independent,
//                 the results do not depend on it)
//   * outlier[]   the reads longer than `outlier_above` events, same ordering (empty when outlier_above = ~0)
//   * counts[]    [0] = reads in order[], [1] = reads in outlier[], [2] = longest read in order[], [3] = longest read at all
// One block of 1024 threads: two passes over the offsets (histogram, scatter); 100 000 reads take ~20 us.
#include "nchmm_device.h"

namespace nchmm {

namespace {
constexpr unsigned kBins = 2048;
constexpr unsigned kPlanThreads = 1024;
}

__global__ __launch_bounds__(kPlanThreads) void plan_order_kernel(const uint64_t* __restrict__ off, unsigned n, unsigned long long span,
                                                                  unsigned long long outlier_above, uint32_t* __restrict__ order,
                                                                  uint32_t* __restrict__ outlier, unsigned long long* __restrict__ counts)
{
    __shared__ unsigned bin_in[kBins], bin_out[kBins];     // counts, then (exclusive, longest class first) start positions
    __shared__ unsigned long long s_longest_in, s_longest;
    const unsigned tid = threadIdx.x;
    for (unsigned b = tid; b < kBins; b += kPlanThreads) { bin_in[b] = 0; bin_out[b] = 0; }
    if (tid == 0) { s_longest_in = 0; s_longest = 0; }
    __syncthreads();
    // class of a length: kBins - 1 for the longest reads (span = the longest length the caller stated; longer ones share the
    // top class)
    auto klass = [&](unsigned long long len) -> unsigned {
        const unsigned long long k = span ? len * (kBins - 1) / span : 0;
        return (unsigned)(k < kBins ? k : kBins - 1);
    };
    unsigned long long longest_in = 0, longest = 0;
    for (unsigned r = tid; r < n; r += kPlanThreads) {
        const unsigned long long len = off[r + 1] - off[r];
        longest = len > longest ? len : longest;
        if (len > outlier_above) atomicAdd(&bin_out[klass(len)], 1u);
        else { atomicAdd(&bin_in[klass(len)], 1u); longest_in = len > longest_in ? len : longest_in; }
    }
    atomicMax(&s_longest_in, longest_in);
    atomicMax(&s_longest, longest);
    __syncthreads();
    if (tid < 2) {     // (2048 classes: a serial prefix by one lane per array is a few microseconds)
        unsigned* bin = tid == 0 ? bin_in : bin_out;
        unsigned at = 0;
        for (int b = (int)kBins - 1; b >= 0; --b) { const unsigned c = bin[b]; bin[b] = at; at += c; }
        counts[tid] = at;
    }
    if (tid == 2) { counts[2] = s_longest_in; counts[3] = s_longest; }
    __syncthreads();
    for (unsigned r = tid; r < n; r += kPlanThreads) {
        const unsigned long long len = off[r + 1] - off[r];
        if (len > outlier_above) outlier[atomicAdd(&bin_out[klass(len)], 1u)] = r;
        else order[atomicAdd(&bin_in[klass(len)], 1u)] = r;
    }
}

void launch_plan_order(const uint64_t* d_off, unsigned n, uint64_t span, uint64_t outlier_above, uint32_t* d_order, uint32_t* d_outlier,
                       unsigned long long* d_counts, hipStream_t stream)
{
    hipLaunchKernelGGL(plan_order_kernel, dim3(1), dim3(kPlanThreads), 0, stream, d_off, n, (unsigned long long)span,
                       (unsigned long long)outlier_above, d_order, d_outlier, d_counts);
}

}  // namespace nchmm
