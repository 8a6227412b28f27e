// gfx950 lane-swap instructions: semantics of v_permlane32_swap / v_permlane16_swap, and a check of the clang builtins
// (two results, added -- the pattern of wave_sum7_rows in fwbw_common.hpp) against the folds they should compute.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_asm(float* out, const float* in)
{
    float a = in[threadIdx.x], b = in[64 + threadIdx.x];
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float c = in[128 + threadIdx.x], d = in[192 + threadIdx.x];
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
}
__global__ void k_builtin(float* out, const float* in)
{
    // values computed by VALU ops right before the swap, results added
    const float a = in[threadIdx.x] * 2.0f, b = in[64 + threadIdx.x] * 3.0f;
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    out[threadIdx.x] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    const float c = in[128 + threadIdx.x] * 2.0f, d = in[192 + threadIdx.x] * 3.0f;
    const auto q = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d), false, false);
    out[64 + threadIdx.x] = __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1]);
}
int main()
{
    float h[256], o[256];
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float *di, *dout;
    (void)hipMalloc(&di, 1024); (void)hipMalloc(&dout, 1024);
    (void)hipMemcpy(di, h, 1024, hipMemcpyHostToDevice);
    k_asm<<<1, 64>>>(dout, di);
    (void)hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) { for (int i = 0; i < 64; i += 8) printf("%g ", o[64 * r + i]); printf("\n"); }
    k_builtin<<<1, 64>>>(dout, di);
    (void)hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        // 32-swap fold: lanes < 32: a[l] + a[l + 32]; lanes >= 32: b[l - 32] + b[l]
        const float e32 = l < 32 ? 2.0f * h[l] + 2.0f * h[l + 32] : 3.0f * h[64 + l - 32] + 3.0f * h[64 + l];
        // 16-swap fold: rows { c.0 + c.1, d.0 + d.1, c.2 + c.3, d.2 + d.3 }
        const int row = l >> 4, x = l & 15;
        const float e16 = row == 0 ? 2.0f * h[128 + x] + 2.0f * h[128 + 16 + x] : row == 1 ? 3.0f * h[192 + x] + 3.0f * h[192 + 16 + x]
                        : row == 2 ? 2.0f * h[128 + 32 + x] + 2.0f * h[128 + 48 + x] : 3.0f * h[192 + 32 + x] + 3.0f * h[192 + 48 + x];
        if (o[l] != e32 || o[64 + l] != e16) { if (bad < 4) printf("lane %d: fold32 %g (want %g) fold16 %g (want %g)\n", l, o[l], e32, o[64 + l], e16); ++bad; }
    }
    printf("builtin folds: %d lanes wrong\n", bad);
    return bad != 0;
}
