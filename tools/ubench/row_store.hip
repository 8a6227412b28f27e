// How fast can 512 persistent blocks stream 16 KiB rows to HBM, by store shape?
//   A  the forward sweep's shape: thread (t, h) stores 8 dwords at j = t + 256 (h + 2 q): every instruction of a wave
//      writes two full 128-byte lines 1 KiB apart
//   B  two global_store_dwordx4 per thread, a wave writes 1 KiB contiguous per instruction
//   C  like A but with a little arithmetic between the stores (the real kernel spreads them over ~150 VALU ops)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int kStates = 4096, kThreads = 512;
template <int MODE, int SYNC>
__global__ __launch_bounds__(kThreads, 4) void k(float* ws, int rows_per_block, int work)
{
    __shared__ float sX[2][1024];
    const unsigned tau = threadIdx.x, t = tau >> 1, h = tau & 1u;
    float* rowp = ws + (size_t)blockIdx.x * rows_per_block * kStates;
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = (float)(tau + q);
    for (int i = 0; i < rows_per_block; ++i) {
        if (SYNC) {       // the sweep's per-event exchange: group sums through LDS behind one barrier
            sX[i & 1][tau] = v[0] + v[1]; sX[i & 1][512 + tau] = v[2] + v[3];
            __syncthreads();
            v[4] += sX[i & 1][(tau * 7u) & 1023u]; v[5] += sX[i & 1][(tau * 3u + 1u) & 1023u];
        }
        for (int w = 0; w < work; ++w)
            for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
        if (MODE == 1) {
            float4* p = reinterpret_cast<float4*>(rowp) + ((tau >> 6) * 128 + (tau & 63u));
            p[0] = make_float4(v[0], v[1], v[2], v[3]);
            p[64] = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            for (int q = 0; q < 8; ++q) rowp[t + 256u * (h + 2u * q)] = v[q];
        }
        rowp += kStates;
    }
}
int main()
{
    const int blocks = 512, rows = 800;
    float* ws;
    const size_t bytes = (size_t)blocks * rows * kStates * 4;
    if (hipMalloc(&ws, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int work : {0, 4, 16}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) k<0, 0><<<blocks, kThreads>>>(ws, rows, work);
                else if (mode == 1) k<1, 0><<<blocks, kThreads>>>(ws, rows, work);
                else if (mode == 2) k<0, 1><<<blocks, kThreads>>>(ws, rows, work);
                else k<1, 1><<<blocks, kThreads>>>(ws, rows, work);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("work=%2d fma/state  %-24s %s  %.3f ms  %.2f TB/s\n", work, (mode & 1) ? "dwordx4 contiguous" : "8 x dword, 1 KiB stride",
                   mode >= 2 ? "barrier + LDS exchange per row" : "no barrier                    ", best, bytes / best * 1e-9);
        }
    }
    return 0;
}
