// sstore_rate.hip -- can the Viterbi forward sweep export its decisions as the SGPR masks the compares already
// produce (bit-planes, one bit per lane) with scalar stores, instead of turning them into per-lane codes with
// v_cndmask / v_lshl_or and storing bytes?  Three questions, answered on the device:
//   A  semantics: s_store_dwordx4 of v_cmp results lands in memory (after s_dcache_wb), also when the data SGPRs
//      are overwritten by the next v_cmp right after the store was issued;
//   B  cost: a Viterbi-shaped loop (512 threads, 2 blocks per CU, one barrier per iteration, ~280 VALU ops per
//      thread-iteration in the kernel's double-rate / single-rate mix) with 0 / 4 / 12 / 24 scalar stores per
//      wave-iteration, and with extra SALU mask logic;
//   C  ns per wave-instruction per SIMD at this occupancy for the VALU ops a redesign would lean on.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize sstore_rate.hip -o sstore_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned long long u64;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- A: semantics
__global__ __launch_bounds__(64) void sstore_semantics(u64* out, int iters)
{
    const unsigned lane = threadIdx.x;
    u64* base = out + (size_t)blockIdx.x * iters * 2;
    for (int it = 0; it < iters; ++it) {
        const float v0 = (float)((lane * 7u + (unsigned)it * 13u + blockIdx.x) & 63u);
        const float v1 = (float)((lane * 11u + (unsigned)it * 5u + 3u * blockIdx.x) & 63u);
        const float w0 = (float)((lane * 3u + (unsigned)it) & 63u);
        u64* p = base + 2 * it;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
        const u64 addr = ((u64)hi << 32) | lo;
        // two masks into one aligned SGPR quad, stored, and the quad overwritten at once by other compares
        asm volatile(
            "v_cmp_gt_f32_e64 s[20:21], %1, %3\n\t"
            "v_cmp_gt_f32_e64 s[22:23], %2, %3\n\t"
            "s_store_dwordx4 s[20:23], %0, 0x0\n\t"
            "v_cmp_gt_f32_e64 s[20:21], %4, %3\n\t"
            "v_cmp_lt_f32_e64 s[22:23], %4, %3\n\t"
            "s_mov_b64 s[20:21], -1\n\t"
            "s_mov_b64 s[22:23], 0\n\t"
            :: "s"(addr), "v"(v0), "v"(v1), "v"(31.5f), "v"(w0)
            : "s20", "s21", "s22", "s23", "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------- B: cost inside a Viterbi-shaped loop
// per thread-iteration: 8 cells x (19 "emission" double-rate ops: 8 fma + 11 add/mul; 3 compares; 1 max3; SELS selects;
// 2 adds) + ~40 scan-like single-rate ops.  STORES scalar stores (dwordx4) per wave-iteration; SALU extra s_and/s_or pairs.
template <int STORES, int SELS, int SALU, bool X2>
__global__ __launch_bounds__(512, 4) void vit_shape(float* out, u64* planes, int iters)
{
    __shared__ float sX[2][1024];
    const unsigned tau = threadIdx.x, wave = tau >> 6;
    float a[8], mu[8], sg[8];
    for (int i = 0; i < 8; ++i) { a[i] = -1.0f - 0.001f * tau - i; mu[i] = 50.0f + i + 0.01f * tau; sg[i] = 1.0f + 0.001f * (tau + i); }
    float x = 55.0f + 0.001f * blockIdx.x;
    unsigned bp = 0;
    u64 acc = 0;
    u64* row = planes + ((size_t)blockIdx.x * 8 + wave) * 32;   // 256 B per wave, rewritten every iteration (hot in cache)
    for (int it = 0; it < iters; ++it) {
        // scan-like phase: 40 single-rate ops on alpha
        float m = a[0];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
#pragma unroll
            for (int i = 1; i < 8; ++i) {
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(m) : "v"(a[i]));
            }
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(m) : "v"(x));
        }
        sX[it & 1][tau] = m; sX[it & 1][512 + (tau ^ 37u)] = m;
        __syncthreads();
        const float g1 = sX[it & 1][(tau * 5u + 3u) & 1023u], g2 = sX[it & 1][(tau * 9u + 1u) & 1023u];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)row);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)row >> 32));
        const u64 addr = ((u64)hi << 32) | lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // emission-like: 8 fma + 11 add/mul
            float d = x - mu[i];
            float q = d * sg[i];
            float e = __builtin_fmaf(-q, sg[i], d);
            q = __builtin_fmaf(e, sg[i], q);
            float t = q * q + 1.8378f;
            float d2 = x - sg[i];
            float q2 = d2 * mu[i];
            float e2 = __builtin_fmaf(-q2, mu[i], d2);
            q2 = __builtin_fmaf(e2, sg[i], q2);
            float l = mu[i] * q2; l = l * q2;
            float q3 = l * x;
            float e3 = __builtin_fmaf(-q3, x, l);
            q3 = __builtin_fmaf(e3, x, q3);
            float u = sg[i] - x; u = u - q3;
            float n = __builtin_fmaf(-0.5f, t, mu[i]);
            float em = __builtin_fmaf(0.5f, u, n);
            // combine-like: add, max3, 3 compares, SELS selects (+ lshl_or when selecting), add
            const float s0 = a[i] + sg[i];
            float best;
            asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(best) : "v"(s0), "v"(g1), "v"(g2));
            if (STORES > 0 || SALU > 0) {
                // masks straight into a fixed SGPR quad + pair; tie logic on the scalar unit
                asm volatile(
                    "v_cmp_eq_f32_e64 s[20:21], %1, %4\n\t"
                    "v_cmp_eq_f32_e64 s[22:23], %2, %4\n\t"
                    "v_cmp_eq_f32_e64 s[24:25], %3, %4\n\t"
                    "s_and_b64 s[26:27], s[20:21], s[22:23]\n\t"
                    "s_or_b64 s[28:29], s[20:21], s[22:23]\n\t"
                    "s_and_b64 s[28:29], s[28:29], s[24:25]\n\t"
                    "s_or_b64 s[26:27], s[26:27], s[28:29]\n\t"
                    "s_or_b64 %0, %0, s[26:27]\n\t"
                    : "+s"(acc) : "v"(s0), "v"(g1), "v"(g2), "v"(best)
                    : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
                if (STORES >= 8 || (STORES == 4 && (i & 1))) {
                    if (X2) {
                        asm volatile("s_store_dwordx2 s[20:21], %0, %1\n\ts_store_dwordx2 s[22:23], %0, %2" :: "s"(addr), "n"(i * 16), "n"(i * 16 + 8) : "memory");
                    } else {
                        asm volatile("s_store_dwordx4 s[20:23], %0, %1" :: "s"(addr), "n"(i * 16) : "memory");
                    }
                }
                if (STORES >= 12 && i < 4) {
                    asm volatile("s_store_dwordx4 s[24:27], %0, %1" :: "s"(addr), "n"(128 + i * 16) : "memory");
                }
#pragma unroll
                for (int k = 0; k < SALU; ++k)
                    asm volatile("s_andn2_b64 s[28:29], s[28:29], s[24:25]\n\ts_or_b64 %0, %0, s[28:29]" : "+s"(acc) :: "s28", "s29");
            } else {
                u64 e0, e1, e2;
                asm volatile("v_cmp_eq_f32_e64 %0, %1, %2" : "=s"(e0) : "v"(s0), "v"(best));
                asm volatile("v_cmp_eq_f32_e64 %0, %1, %2" : "=s"(e1) : "v"(g1), "v"(best));
                asm volatile("v_cmp_eq_f32_e64 %0, %1, %2" : "=s"(e2) : "v"(g2), "v"(best));
                acc |= (e0 & e1) | ((e0 | e1) & e2);
                if (SELS == 2) {
                    unsigned slot;
                    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(slot) : "v"(tau), "v"(bp), "s"(e1));
                    asm volatile("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(slot) : "v"(slot), "s"(e0));
                    asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(bp) : "v"(slot));
                }
            }
            a[i] = best + em;
        }
        x += 0.25f;
        if (STORES > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // what the next iteration's LDS wait amounts to
    }
    if (STORES > 0) asm volatile("s_dcache_wb" ::: "memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 512 + tau] = s + (float)bp + (float)(acc & 1);
}

// ---------------------------------------------------------------- C: VALU op rates at 16 waves per CU
template <int MODE>
__global__ __launch_bounds__(512, 4) void rate(float* out, int iters)
{
    float a[8], g[8], h[8], k[8], b = 1.0001f, c = 0.5f, d2 = 0.25f, t5 = 0.f, z6 = 0.f, f7 = 1.f, f8 = 2.f;
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; g[i] = a[i] + 1.f; h[i] = a[i] + 2.f; k[i] = a[i] + 3.f; }
    u64 msk = 0x5555555555555555ull ^ (u64)iters;
    unsigned sv = (unsigned)iters;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
                if (MODE == 4) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 6) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 7) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
                if (MODE == 8) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]));
                if (MODE == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 10) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 12) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 13) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 14) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"(sv));
                if (MODE == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(sv));
                if (MODE == 16) asm volatile("v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (MODE == 17) asm volatile("v_cmp_eq_u32_e64 %1, %0, %2" : : "v"(a[i]), "s"(msk), "v"(b));
                if (MODE == 18) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 19) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 20) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 21) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 22) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 23) asm volatile("v_cmp_gt_f32_e64 %1, %0, %2" : : "v"(a[i]), "s"(msk), "v"(b));
                if (MODE == 24) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 25) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 26) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (MODE == 27) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 28) asm volatile("v_cmp_class_f32_e64 %1, %0, %2" : : "v"(a[i]), "s"(msk), "v"(b));
                if (MODE == 29) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 30) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 31) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sv) : "v"(a[i]));
                // round-3 addendum: do the VCC / VOP2 / VOPC encodings of compare and select cost the same as the SGPR-pair ones?
                if (MODE == 32) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if (MODE == 33) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (MODE == 34) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (MODE == 35) asm volatile("v_cmp_gt_f32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
                if (MODE == 37) asm volatile("v_cndmask_b32_e64 %0, 1.0, 2.0, %1" : "=v"(a[i]) : "s"(msk));
                if (MODE == 38) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 39) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 40) asm volatile("v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
                // sequences as the Viterbi kernel has them (time per SEQUENCE): one compare feeding two selects (a scan step), and a
                // 3-way combine cell (max3, three equality compares, tie logic on the scalar unit, two selects)
                if (MODE == 43) asm volatile("v_cmp_gt_f32_e64 %3, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %3\n\tv_cndmask_b32_e64 %1, %1, %2, %3"
                                             : "+v"(a[i]), "+v"(c) : "v"(b), "s"(msk));
                if (MODE == 44) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %1, %1, %2, vcc"
                                             : "+v"(a[i]), "+v"(c) : "v"(b) : "vcc");
                if (MODE == 45) asm volatile("v_max3_f32 %1, %0, %2, %3\n\t"
                                             "v_cmp_eq_f32_e64 s[20:21], %2, %1\n\tv_cmp_eq_f32_e64 s[22:23], %3, %1\n\tv_cmp_eq_f32_e64 s[24:25], %0, %1\n\t"
                                             "v_cndmask_b32_e64 %0, %3, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, 0, s[24:25]\n\t"
                                             "s_and_b64 s[26:27], s[24:25], s[20:21]\n\ts_or_b64 s[24:25], s[24:25], s[20:21]\n\t"
                                             "s_and_b64 s[24:25], s[24:25], s[22:23]\n\ts_or_b64 s[26:27], s[26:27], s[24:25]\n\ts_or_b64 %4, %4, s[26:27]"
                                             : "+v"(a[i]), "+v"(c), "+v"(b), "+v"(d2), "+s"(msk) : : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
                if (MODE == 46) asm volatile("v_max3_f32 %1, %0, %2, %3\n\t"
                                             "v_cmp_eq_f32_e64 s[22:23], %3, %1\n\t"
                                             "v_cmp_eq_f32_e32 vcc, %2, %1\n\ts_mov_b64 s[20:21], vcc\n\tv_cndmask_b32_e32 %5, %3, %2, vcc\n\t"
                                             "v_cmp_eq_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %5, %6, vcc\n\t"
                                             "s_and_b64 s[26:27], vcc, s[20:21]\n\ts_or_b64 s[24:25], vcc, s[20:21]\n\t"
                                             "s_and_b64 s[24:25], s[24:25], s[22:23]\n\ts_or_b64 s[26:27], s[26:27], s[24:25]\n\ts_or_b64 %4, %4, s[26:27]"
                                             : "+v"(a[i]), "+v"(c), "+v"(b), "+v"(d2), "+s"(msk), "=&v"(t5) : "v"(z6) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
                if (MODE == 50) asm volatile("v_max3_f32 %1, %0, %2, %3\n\t"
                                             "v_cmp_eq_f32_e32 vcc, %2, %1\n\tv_cndmask_b32_e32 %5, %3, %2, vcc\n\ts_mov_b64 s[20:21], vcc\n\t"
                                             "v_cmp_eq_f32_e64 s[22:23], %3, %1\n\t"
                                             "v_cmp_eq_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %5, %6, vcc\n\t"
                                             "s_and_b64 s[26:27], vcc, s[20:21]\n\ts_or_b64 s[24:25], vcc, s[20:21]\n\t"
                                             "s_and_b64 s[24:25], s[24:25], s[22:23]\n\ts_or_b64 s[26:27], s[26:27], s[24:25]\n\ts_or_b64 %4, %4, s[26:27]"
                                             : "+v"(a[i]), "+v"(c), "+v"(b), "+v"(d2), "+s"(msk), "=&v"(t5) : "v"(z6) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
                if (MODE == 51) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\ts_nop 0\n\tv_cndmask_b32_e32 %1, %1, %2, vcc"
                                             : "+v"(a[i]), "+v"(c) : "v"(b) : "vcc");
                if (MODE == 52) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %1, %2, %0, vcc"
                                             : "+v"(a[i]), "=v"(t5) : "v"(b) : "vcc");
                if (MODE == 53) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %1, %1, %2, vcc\n\tv_cndmask_b32_e32 %0, %0, %2, vcc"
                                             : "+v"(a[i]), "+v"(c) : "v"(b) : "vcc");
                // the combine cell with two independent full-rate ops (as the emission right after it provides) between the selects
                if (MODE == 54) asm volatile("v_max3_f32 %1, %0, %2, %3\n\t"
                                             "v_cmp_eq_f32_e64 s[20:21], %2, %1\n\tv_cmp_eq_f32_e64 s[22:23], %3, %1\n\tv_cmp_eq_f32_e64 s[28:29], %0, %1\n\t"
                                             "v_cndmask_b32_e64 %5, %3, %2, s[20:21]\n\tv_add_f32 %6, %6, %8\n\tv_cndmask_b32_e64 %0, %5, 0, s[28:29]\n\tv_add_f32 %7, %7, %8\n\t"
                                             "s_and_b64 s[26:27], s[28:29], s[20:21]\n\ts_or_b64 s[24:25], s[28:29], s[20:21]\n\ts_and_b64 s[24:25], s[24:25], s[22:23]\n\ts_or_b64 s[26:27], s[26:27], s[24:25]\n\ts_or_b64 %4, %4, s[26:27]"
                                             : "+v"(a[i]), "+v"(c), "+v"(b), "+v"(d2), "+s"(msk), "=&v"(t5), "+v"(f7), "+v"(f8) : "v"(z6) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
                if (MODE == 55) asm volatile("v_max3_f32 %1, %0, %2, %3\n\t"
                                             "v_cmp_eq_f32_e64 s[22:23], %3, %1\n\t"
                                             "v_cmp_eq_f32_e32 vcc, %2, %1\n\tv_cndmask_b32_e32 %5, %3, %2, vcc\n\tv_add_f32 %6, %6, %8\n\ts_mov_b64 s[20:21], vcc\n\t"
                                             "v_cmp_eq_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %5, %8, vcc\n\tv_add_f32 %7, %7, %8\n\t"
                                             "s_and_b64 s[26:27], vcc, s[20:21]\n\ts_or_b64 s[24:25], vcc, s[20:21]\n\ts_and_b64 s[24:25], s[24:25], s[22:23]\n\ts_or_b64 s[26:27], s[26:27], s[24:25]\n\ts_or_b64 %4, %4, s[26:27]"
                                             : "+v"(a[i]), "+v"(c), "+v"(b), "+v"(d2), "+s"(msk), "=&v"(t5), "+v"(f7), "+v"(f8) : "v"(z6) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
                if (MODE == 56) asm volatile("v_max3_f32 %1, %0, %2, %3\n\t"
                                             "v_cmp_eq_f32_e64 s[22:23], %3, %1\n\tv_cmp_eq_f32_e64 s[20:21], %2, %1\n\t"
                                             "v_cmp_eq_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e64 %5, %3, %2, s[20:21]\n\tv_add_f32 %6, %6, %8\n\tv_cndmask_b32_e32 %0, %5, %8, vcc\n\tv_add_f32 %7, %7, %8\n\t"
                                             "s_and_b64 s[26:27], vcc, s[20:21]\n\ts_or_b64 s[24:25], vcc, s[20:21]\n\ts_and_b64 s[24:25], s[24:25], s[22:23]\n\ts_or_b64 s[26:27], s[26:27], s[24:25]\n\ts_or_b64 %4, %4, s[26:27]"
                                             : "+v"(a[i]), "+v"(c), "+v"(b), "+v"(d2), "+s"(msk), "=&v"(t5), "+v"(f7), "+v"(f8) : "v"(z6) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
                if (MODE == 57) asm volatile("v_cmp_gt_f32_e64 %3, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %3\n\tv_add_f32 %4, %4, %2\n\tv_cndmask_b32_e64 %1, %1, %2, %3"
                                             : "+v"(a[i]), "+v"(c) : "v"(b), "s"(msk), "v"(f7));
                if (MODE == 58) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_nop\n\tv_cndmask_b32_e32 %1, %1, %2, vcc"
                                             : "+v"(a[i]), "+v"(c) : "v"(b) : "vcc");
                if (MODE == 59) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_fma_f32 %3, %3, %2, %2\n\tv_cndmask_b32_e32 %1, %1, %2, vcc"
                                             : "+v"(a[i]), "+v"(c) : "v"(b), "v"(f7) : "vcc");
                // half-rate and full-rate ops interleaved, every chain independent (a[i] for the half-rate op, g[i] / h[i] / k[i] for the adds)
                if (MODE == 60) asm volatile("v_max_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b));
                if (MODE == 61) asm volatile("v_max_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %3" : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]) : "v"(b));
                if (MODE == 62) asm volatile("v_max_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]), "+v"(k[i]) : "v"(b));
                if (MODE == 63) asm volatile("v_cndmask_b32_e64 %0, %0, %2, %3\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b), "s"(msk));
                if (MODE == 64) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b) : "s20", "s21");
                if (MODE == 65) asm volatile("v_max_f32 %0, %0, %2\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b));
                if (MODE == 66) asm volatile("v_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]), "+v"(k[i]) : "v"(b));
                if (MODE == 67) asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b));
                if (MODE == 68) asm volatile("v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b));
                if (MODE == 69) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %4\n\tv_add_f32 %2, %2, %4\n\tv_cndmask_b32_e64 %0, %0, %4, s[20:21]\n\tv_add_f32 %3, %3, %4\n\tv_cndmask_b32_e64 %1, %1, %4, s[20:21]"
                                             : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]), "+v"(k[i]) : "v"(b) : "s20", "s21");
                if (MODE == 70) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %2, s[20:21]"
                                             : "+v"(a[i]), "+v"(g[i]) : "v"(b) : "s20", "s21");
                if (MODE == 71) asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b), "v"(c));
                if (MODE == 72) asm volatile("v_lshl_or_b32 %0, %0, 8, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b));
                if (MODE == 73) asm volatile("v_lshl_add_u32 %0, %0, 2, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b));
                if (MODE == 74) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b));
                if (MODE == 75) asm volatile("v_cndmask_b32_e64 %0, %0, %2, %3\n\tv_mul_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b), "s"(msk));
                if (MODE == 76) asm volatile("v_cndmask_b32_e64 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b), "s"(msk));
                if (MODE == 77) asm volatile("v_add_f32 %1, %1, %2\n\tv_cmp_gt_f32_e64 s[20:21], %0, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b) : "s20", "s21");
                if (MODE == 78) asm volatile("v_cmp_eq_f32_e64 s[20:21], %0, %2\n\tv_cmp_eq_f32_e64 s[22:23], %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b) : "s20", "s21", "s22", "s23");
                if (MODE == 79) asm volatile("v_cndmask_b32_e64 %0, %0, %3, %4\n\tv_sub_f32 %1, %1, %3\n\tv_cndmask_b32_e64 %2, %2, %3, %4\n\tv_mul_f32 %5, %5, %3"
                                             : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]) : "v"(b), "s"(msk), "v"(k[i]));
                // the combine cell with SIX independent full-rate riders (sub / mul / add), one after each half-rate op
                if (MODE == 80) asm volatile("v_max3_f32 %1, %0, %2, %3\n\tv_sub_f32 %6, %6, %8\n\t"
                                             "v_cmp_eq_f32_e64 s[20:21], %2, %1\n\tv_mul_f32 %7, %7, %8\n\t"
                                             "v_cmp_eq_f32_e64 s[22:23], %3, %1\n\tv_sub_f32 %9, %9, %8\n\t"
                                             "v_cmp_eq_f32_e64 s[28:29], %0, %1\n\tv_mul_f32 %10, %10, %8\n\t"
                                             "v_cndmask_b32_e64 %5, %3, %2, s[20:21]\n\tv_add_f32 %6, %6, %8\n\tv_cndmask_b32_e64 %0, %5, 0, s[28:29]\n\tv_add_f32 %7, %7, %8\n\t"
                                             "s_and_b64 s[26:27], s[28:29], s[20:21]\n\ts_or_b64 s[24:25], s[28:29], s[20:21]\n\ts_and_b64 s[24:25], s[24:25], s[22:23]\n\ts_or_b64 s[26:27], s[26:27], s[24:25]\n\ts_or_b64 %4, %4, s[26:27]"
                                             : "+v"(a[i]), "+v"(c), "+v"(b), "+v"(d2), "+s"(msk), "=&v"(t5), "+v"(g[i]), "+v"(h[i]) : "v"(z6), "v"(k[i]), "v"(f7)
                                             : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
                // transcendental + riders
                if (MODE == 90) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 91) asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %2" : "+v"(a[i]), "+v"(g[i]) : "v"(b));
                if (MODE == 92) asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %3" : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]) : "v"(b));
                if (MODE == 93) asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]), "+v"(k[i]) : "v"(b));
                if (MODE == 94) asm volatile("v_exp_f32 %0, %0\n\tv_max_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %3" : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]) : "v"(b));
                if (MODE == 47) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc"
                                             : "+v"(a[i]), "+v"(c) : "v"(b) : "vcc");
                if (MODE == 48) asm volatile("s_mov_b64 vcc, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "s"(msk) : "vcc");
                if (MODE == 49) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_add_f32 %1, %1, %2\n\tv_cndmask_b32_e32 %1, %1, %2, vcc"
                                             : "+v"(a[i]), "+v"(c) : "v"(b) : "vcc");
                if (MODE == 41) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
                if (MODE == 42) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            }
        }
    }
    // clustered half-rate / full-rate blocks (8 + 8, the shape the compiler gives the Viterbi column) in step on all waves (83),
    // or out of phase between the two waves a block has on each SIMD (84): can ANOTHER wave's full-rate op use the idle pass?
    if (MODE == 83 || MODE == 84) {
        const bool flip = MODE == 84 && ((threadIdx.x >> 8) & 1u);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!flip) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(g[i]) : "v"(b));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(g[i]) : "v"(b));
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                }
            }
        }
    }
    // the Viterbi column's shape: 56 half-rate ops and 168 full-rate ops per iteration, clustered (85), clustered with a
    // barrier per iteration (86), or interleaved H F F F (87), interleaved + barrier (88).  Time per 224 instructions.
    if (MODE >= 85 && MODE <= 88) {
        for (int it = 0; it < iters / 8; ++it) {
            if (MODE == 85 || MODE == 86) {
#pragma unroll
                for (int r = 0; r < 7; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#pragma unroll
                for (int r = 0; r < 7; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %3" : "+v"(g[i]), "+v"(h[i]), "+v"(k[i]) : "v"(b));
            } else {
#pragma unroll
                for (int r = 0; r < 7; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        asm volatile("v_max_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(a[i]), "+v"(g[i]), "+v"(h[i]), "+v"(k[i]) : "v"(b));
            }
            if (MODE == 86 || MODE == 88) __syncthreads();
        }
    }
    // eight transcendentals then twenty-four adds, clustered (95) -- against MODE 93's exp add add add, same instruction mix
    if (MODE == 95) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %3" : "+v"(g[i]), "+v"(h[i]), "+v"(k[i]) : "v"(b));
            }
        }
    }
    float s = (float)sv + c + d2 + t5 + b + f7 + f8 + (float)(msk & 1);
    for (int i = 0; i < 8; ++i) s += a[i] + g[i] + h[i] + k[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename F> static float time_ms(F launch)
{
    launch(); CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

template <int MODE> static void run_rate(const char* name, float* d)
{
    const int iters = 20000, grid = 512;
    const float ms = time_ms([&] { hipLaunchKernelGGL(rate<MODE>, dim3(grid), dim3(512), 0, 0, d, iters); });
    // 512 blocks x 8 waves over 1024 SIMDs = 4 waves per SIMD, each 32 * iters instructions
    printf("rate  %-22s %8.3f ms  -> %.3f ns per wave-instruction per SIMD (4 waves/SIMD)\n", name, ms, ms * 1e6 / (4.0 * 32.0 * iters));
}

template <int MODE> static void run_shape224(const char* name, float* d)
{
    const int iters = 20000, grid = 512;
    const float ms = time_ms([&] { hipLaunchKernelGGL(rate<MODE>, dim3(grid), dim3(512), 0, 0, d, iters); });
    printf("col   %-52s %8.3f ms  -> %.1f ns per 224 instructions per wave and SIMD share (additive: 56 x 1.80 + 168 x 1.07 = 280.6)\n", name, ms, ms * 1e6 / (4.0 * (iters / 8)));
}

template <int STORES, int SELS, int SALU, bool X2> static double run_shape(const char* name, float* d, u64* planes)
{
    const int iters = 4000, grid = 512;
    const float ms = time_ms([&] { hipLaunchKernelGGL((vit_shape<STORES, SELS, SALU, X2>), dim3(grid), dim3(512), 0, 0, d, planes, iters); });
    const double ns = ms * 1e6 / iters;      // per block-iteration pair (two blocks per CU run concurrently)
    printf("shape %-52s %8.3f ms  -> %.1f ns per event per block-pair\n", name, ms, ns);
    return ns;
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);   // a kernel that hangs must not take the lines before it along
    float* d; CHECK(hipMalloc(&d, 64 << 20));
    u64* planes; CHECK(hipMalloc(&planes, 64 << 20));
    CHECK(hipMemset(planes, 0xAB, 64 << 20));
    // ---- A
    {
        const int iters = 1000, grid = 512;
        hipLaunchKernelGGL(sstore_semantics, dim3(grid), dim3(64), 0, 0, planes, iters);
        CHECK(hipDeviceSynchronize());
        std::vector<u64> h((size_t)grid * iters * 2);
        CHECK(hipMemcpy(h.data(), planes, h.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int b = 0; b < grid; ++b)
            for (int it = 0; it < iters; ++it) {
                u64 m0 = 0, m1 = 0;
                for (unsigned lane = 0; lane < 64; ++lane) {
                    if ((float)((lane * 7u + (unsigned)it * 13u + b) & 63u) > 31.5f) m0 |= 1ull << lane;
                    if ((float)((lane * 11u + (unsigned)it * 5u + 3u * b) & 63u) > 31.5f) m1 |= 1ull << lane;
                }
                const u64* p = &h[((size_t)b * iters + it) * 2];
                if (p[0] != m0 || p[1] != m1) { if (bad < 5) printf("  mismatch block %d it %d: %016llx %016llx vs %016llx %016llx\n", b, it, p[0], p[1], m0, m1); ++bad; }
            }
        printf("A  s_store_dwordx4 of v_cmp masks, data SGPRs overwritten right after issue: %zu mismatches of %d stores\n", bad, grid * iters);
    }
    // ---- B
    const double base = run_shape<0, 2, 0, false>("selects + lshl_or (today's shape), no scalar stores", d, planes);
    const double nosel = run_shape<0, 0, 0, false>("no selects, no stores (the VALU floor of the plane form)", d, planes);
    const double salu0 = run_shape<0, 0, 1, false>("masks in fixed SGPRs + tie logic on SALU, no stores", d, planes);
    const double st4 = run_shape<4, 0, 0, false>("4 s_store_dwordx4 per wave-event", d, planes);
    const double st8 = run_shape<8, 0, 0, false>("8 s_store_dwordx4 per wave-event", d, planes);
    const double st12 = run_shape<12, 0, 0, false>("12 s_store_dwordx4 per wave-event", d, planes);
    const double st16x2 = run_shape<8, 0, 0, true>("16 s_store_dwordx2 per wave-event", d, planes);
    const double salu4 = run_shape<12, 0, 4, false>("12 stores + 64 extra SALU ops per wave-event", d, planes);
    printf("B  relative to today's shape: no-select %.3f, SALU-masks %.3f, 4 stores %.3f, 8 stores %.3f, 12 stores %.3f, 16 x2 %.3f, 12 + SALU %.3f\n",
           nosel / base, salu0 / base, st4 / base, st8 / base, st12 / base, st16x2 / base, salu4 / base);
    // ---- C
    run_rate<0>("v_add_f32", d); run_rate<21>("v_sub_f32", d); run_rate<22>("v_mul_f32", d); run_rate<1>("v_fma_f32", d);
    run_rate<2>("v_max_f32", d); run_rate<24>("v_max3_f32", d); run_rate<13>("v_med3_f32", d);
    run_rate<3>("v_cndmask_b32_e64", d); run_rate<23>("v_cmp_gt_f32 -> sgpr", d); run_rate<17>("v_cmp_eq_u32 -> sgpr", d);
    run_rate<28>("v_cmp_class_f32 -> sgpr", d);
    run_rate<27>("v_and_b32", d); run_rate<4>("v_or_b32", d); run_rate<5>("v_xor_b32", d); run_rate<30>("v_add_u32", d); run_rate<6>("v_sub_u32", d);
    run_rate<7>("v_lshlrev_b32", d); run_rate<8>("v_ashrrev_i32", d); run_rate<25>("v_lshl_or_b32", d); run_rate<18>("v_lshl_add_u32", d);
    run_rate<19>("v_add3_u32", d); run_rate<20>("v_mad_u32_u24", d); run_rate<9>("v_perm_b32", d); run_rate<10>("v_bfi_b32", d);
    run_rate<11>("v_and_or_b32", d); run_rate<12>("v_min_u32", d); run_rate<29>("v_min3_u32", d);
    run_rate<26>("v_mov_b32_dpp", d); run_rate<16>("v_max_f32_dpp", d);
    run_rate<15>("v_mov_b32 from sgpr", d); run_rate<14>("v_writelane_b32", d); run_rate<31>("v_readlane_b32", d);
    run_rate<32>("v_cndmask_b32_e32 (vcc)", d); run_rate<33>("v_cmp_gt_f32_e32 -> vcc", d); run_rate<37>("v_cndmask_e64 consts", d);
    run_rate<34>("cmp_e32 + cndmask_e32 (2 instr)", d); run_rate<35>("cmp_e64 + cndmask_e64 (2 instr)", d);
    run_rate<38>("v_max_i32", d); run_rate<39>("v_pk_max_f16", d); run_rate<40>("v_cndmask_b32_sdwa", d);
    run_rate<43>("SEQ cmp_e64 + 2 cndmask_e64", d); run_rate<44>("SEQ cmp_e32 + 2 cndmask_e32", d); run_rate<49>("SEQ cmp_e32 cnd add cnd", d);
    run_rate<47>("SEQ cmp_e32 + add + cndmask_e32", d); run_rate<48>("SEQ s_mov vcc + cndmask_e32", d);
    run_rate<51>("SEQ cmp_e32 cnd s_nop cnd", d); run_rate<52>("SEQ cmp_e32 cnd cnd (2nd into a fresh reg)", d); run_rate<53>("SEQ cmp_e32 cnd(shared) cnd(own)", d);
    run_rate<54>("SEQ combine cell + 2 adds, e64 masks", d); run_rate<55>("SEQ combine cell + 2 adds, vcc selects", d); run_rate<56>("SEQ combine cell + 2 adds, e0 only in vcc", d);
    run_rate<57>("SEQ cmp_e64 cnd add cnd", d); run_rate<58>("SEQ cmp_e32 cnd v_nop cnd", d); run_rate<59>("SEQ cmp_e32 cnd fma cnd", d);
    run_rate<67>("SEQ add add", d); run_rate<60>("SEQ max add", d); run_rate<61>("SEQ max add add", d); run_rate<62>("SEQ max add add add", d); run_rate<66>("SEQ max max add add", d);
    run_rate<71>("SEQ max3 add", d); run_rate<72>("SEQ lshl_or add", d); run_rate<73>("SEQ lshl_add add", d); run_rate<74>("SEQ mov_dpp add", d);
    run_rate<75>("SEQ cndmask mul", d); run_rate<76>("SEQ cndmask fma", d); run_rate<77>("SEQ add cmp", d); run_rate<78>("SEQ cmp cmp", d); run_rate<79>("SEQ cnd sub cnd mul", d);
    run_rate<80>("SEQ combine cell + 6 riders", d);
    run_rate<83>("SEQ 8 max then 8 add, x2 (per 32 instr)", d); run_rate<84>("SEQ same, waves out of phase", d);
    run_shape224<85>("224-instr column shape, clustered 56 H then 168 F", d); run_shape224<86>("  same + barrier per iteration", d);
    run_shape224<87>("224-instr column shape, interleaved H F F F", d); run_shape224<88>("  same + barrier per iteration", d);
    run_rate<90>("v_exp_f32", d); run_rate<91>("SEQ exp add", d); run_rate<92>("SEQ exp add add", d); run_rate<93>("SEQ exp add add add", d); run_rate<94>("SEQ exp max add", d); run_rate<95>("SEQ 8 exp then 24 add (per exp+3 add)", d);
    run_rate<65>("SEQ max fma", d); run_rate<68>("SEQ max max", d); run_rate<70>("SEQ cmp cnd cnd (own regs)", d); run_rate<69>("SEQ cmp add cnd add cnd (own regs)", d); run_rate<63>("SEQ cndmask_e64 add", d); run_rate<64>("SEQ cmp_e64 add", d);
    run_rate<50>("SEQ combine cell, vcc selects, s_mov after", d);
    run_rate<45>("SEQ combine cell, e64 masks", d); run_rate<46>("SEQ combine cell, vcc selects", d);
    run_rate<41>("v_sub_co_u32", d); run_rate<42>("v_subb_co_u32", d);
    return 0;
}
