// probe_kernel.hip -- what shader clock does this device run at under a full-chip VALU load?
//
// The Viterbi and forward-backward kernels are bound by VALU issue, so their duration scales
// with 1 / shader clock, and the boxes of one pool do not all run at the same clock: the same binary has measured
// 15.8 ms and 23.2 ms per config-2 launch on two MI355X boxes.  bench.py launches this probe right after its timed
// region and reports the clock next to the throughput, so a number from a slow box can be read for what it is.
#include <hip/hip_runtime.h>

#include "nchmm_probe.h"

namespace nchmm {

// Every wave runs `iters` x 16 dependent v_fma_f32; lane 0 of block 0 reports the shader-clock ticks (s_memtime) and the
// constant-rate wall-clock ticks (s_memrealtime, hipDeviceAttributeWallClockRate kHz) the loop took.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters)
{
    float x = (float)threadIdx.x * 1e-3f, y = 1.0f + 1e-7f * (float)blockIdx.x;
    const long long c0 = clock64();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = __builtin_fmaf(x, y, 1e-6f);
    }
    const long long c1 = clock64();
    const unsigned long long w1 = wall_clock64();
    if (x == 123456.0f) out[2] = 1;                        // keeps the loop alive
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (unsigned long long)(c1 - c0); out[1] = w1 - w0; }
}

void launch_clock_probe(unsigned long long* d_out, int grid, int iters, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)grid), dim3(256), 0, stream, d_out, iters);
}

}  // namespace nchmm
