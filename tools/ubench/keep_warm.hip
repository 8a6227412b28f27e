// keep_warm.hip -- does a cheap resident kernel keep the shader clock up across a gap in the real work?
//
// Finding that prompted it (profiles/r04_hostpath_gap.json): a 15 ms VALU-bound kernel runs 9 % slower after an idle gap of
// 2 ms and 17 % slower after 20 ms, and needs ~50 ms of uninterrupted load to come back.  This measures the shader clock
// (s_memtime cycles / wall ticks over a ~3 ms full-chip FMA loop, then ~12 ms more) after a gap of G ms spent:
//   idle         nothing on the GPU
//   sleep1       one wave per CU looping on s_sleep            (the GPU is "busy", the VALUs are not)
//   sleepfull    every wave slot looping on s_sleep
//   valu1        one wave per SIMD running FMAs
// hipcc --offload-arch=gfx950 -O3 -o keep_warm keep_warm.hip && ./keep_warm
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters)
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
    if (x == 123456.0f) out[2] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (unsigned long long)(c1 - c0); out[1] = w1 - w0; }
}

// mode 0: s_sleep loop; mode 1: FMA loop.  Runs until `ticks` wall-clock ticks have passed.
__global__ __launch_bounds__(64) void keeper(unsigned long long ticks, int mode, float* sink)
{
    const unsigned long long w0 = wall_clock64();
    float x = (float)threadIdx.x;
    while (wall_clock64() - w0 < ticks) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(64);
        } else {
#pragma unroll
            for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-6f);
        }
    }
    if (x == 123456.0f) *sink = x;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long* d = nullptr;
    float* sink = nullptr;
    CK(hipMalloc(&d, 64));
    CK(hipMalloc(&sink, 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto mhz = [&](int iters, double* out) -> int {
        unsigned long long h[3] = {0, 0, 0};
        probe<<<4 * n_cu, 256, 0, s>>>(d, iters);
        CK(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        *out = (double)h[0] / (double)h[1] * wall_khz * 1e-3;
        return 0;
    };
    auto warm = [&]() -> int {           // ~150 ms of load
        double x;
        for (int i = 0; i < 12; ++i) if (mhz(160000, &x)) return 1;
        return 0;
    };
    const char* names[] = {"idle", "sleep1", "sleepfull", "valu1"};
    printf("device: %s, %d CUs, wall clock %d kHz\n", prop.name, n_cu, wall_khz);
    double base;
    if (warm() || mhz(40000, &base)) return 1;
    printf("warm clock: %.0f MHz\n", base);
    for (double gap_ms : {0.5, 2.0, 20.0}) {
        for (int mode = 0; mode < 4; ++mode) {
            double first = 0, later = 0;
            for (int rep = 0; rep < 3; ++rep) {
                if (warm()) return 1;
                const unsigned long long ticks = (unsigned long long)(gap_ms * wall_khz);
                if (mode == 0) {
                    std::this_thread::sleep_for(std::chrono::microseconds((long)(gap_ms * 1000)));
                } else if (mode == 1) {
                    keeper<<<n_cu, 64, 0, s>>>(ticks, 0, sink);
                } else if (mode == 2) {
                    keeper<<<n_cu * 32, 64, 0, s>>>(ticks, 0, sink);
                } else {
                    keeper<<<n_cu * 4, 64, 0, s>>>(ticks, 1, sink);
                }
                double a, b;
                if (mhz(40000, &a) || mhz(160000, &b)) return 1;   // the first ~3 ms after the gap, then the next ~12 ms
                first += a / 3; later += b / 3;
            }
            printf("gap %5.1f ms  %-9s  clock in the first 3 ms after: %.0f MHz (%.3f of warm), next 12 ms: %.0f MHz (%.3f)\n", gap_ms,
                   names[mode], first, first / base, later, later / base);
        }
    }
    return 0;
}
