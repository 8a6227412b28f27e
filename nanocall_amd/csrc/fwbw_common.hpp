// fwbw_common.hpp -- device helpers shared by the two forward-backward implementations
// (fwbw_kernel.hip: base-2 log space, exact for every cell; fwbw_scaled_kernel.hip: rescaled linear
// space, the fast path of the EM rounds).
#pragma once
#include "nchmm_device.h"

#pragma clang fp contract(off)

namespace nchmm {
namespace fb {


constexpr unsigned kFbChunk = 128;   // events staged in LDS at a time
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -3.0e38f;
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32 is log2

__device__ __forceinline__ float swap1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// Pore_Model_State::log_pr_corrected_emission (Pore_Model.hpp:145-149) in base 2, regrouped as
//   k0 - [ (x - mu)^2 r2 + (y - eta)^2 lq / y + l3 ]
// with the per-state constants  r2 = log2e / (2 sigma^2),  lq = log2e lambda / (2 eta^2),
// k0 = log2e (-log sigma + (log lambda - 2 log 2pi) / 2)  and the per-event  l3 = log2e * 3 log(y) / 2.
// (Same real function; the regrouping moves each cell by a few 1e-7 relative, far inside the 1e-4 FB tolerance.)
struct StateK { float mu, r2, eta, lq, k0; };

__device__ __forceinline__ StateK make_state(const float* __restrict__ M, unsigned j, float log_2pi)
{
    StateK s;
    const float rsg = M[MF_RSIGMA * kStates + j], reta = M[MF_RETA * kStates + j];
    s.mu = M[MF_MU * kStates + j];
    s.eta = M[MF_ETA * kStates + j];
    s.r2 = (0.5f * kLog2e) * rsg * rsg;
    s.lq = (0.5f * kLog2e) * M[MF_LAMBDA * kStates + j] * reta * reta;
    s.k0 = kLog2e * (M[MF_NEG_LOG_SIGMA * kStates + j] + 0.5f * (M[MF_C * kStates + j] - log_2pi));
    return s;
}

__device__ __forceinline__ float emission2(float x, float y, float ry, float l3, float mu, float r2, float eta, float lq, float k0)
{
    const float dx = x - mu, dy = y - eta;
    return k0 - __builtin_fmaf(dx * dx, r2, __builtin_fmaf(dy * dy * lq, ry, l3));
}

struct MaxSum { float m, s; };   // running base-2 log-sum-exp: value = m + log2(s)

__device__ __forceinline__ MaxSum lse_merge(MaxSum a, MaxSum b)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(a.m, b.m), kNegBig);
    return MaxSum{m, a.s * ex2(a.m - m) + b.s * ex2(b.m - m)};
}
__device__ __forceinline__ MaxSum lse4(float a, float b, float c, float d)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(c, d)), kNegBig);
    return MaxSum{m, ex2(a - m) + ex2(b - m) + ex2(c - m) + ex2(d - m)};
}
__device__ __forceinline__ float lse3(float a, float b, float c)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a, b), c), kNegBig);
    return m + lg2(ex2(a - m) + ex2(b - m) + ex2(c - m));
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// Sum over the 64 lanes of a wave on the DPP path (no LDS traffic): quad swaps, row mirrors, then the two
// row broadcasts.  Only lane 63 holds the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF));
}
__device__ __forceinline__ float wave_sum_lane63(float v)
{
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);   // row_mirror: every lane of a row holds the row's sum
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// Seven wave sums at once, hand-scheduled: the seven chains are interleaved so that every DPP read of a
// register comes six instructions after its last write (the 2-wait-state VALU-write -> DPP-read hazard never
// needs a nop) and each step is ONE v_add_f32_dpp.  Totals land in lane 63 only.
__device__ __forceinline__ void wave_sum7_lane63(float& a, float& b, float& c, float& d, float& e, float& f, float& g)
{
#define NCHMM_DPP7(CTRL)                                   \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"                \
    "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"                \
    "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"                \
    "v_add_f32_dpp %3, %3, %3 " CTRL "\n\t"                \
    "v_add_f32_dpp %4, %4, %4 " CTRL "\n\t"                \
    "v_add_f32_dpp %5, %5, %5 " CTRL "\n\t"                \
    "v_add_f32_dpp %6, %6, %6 " CTRL "\n\t"
    asm volatile("s_nop 1\n\t"
                 NCHMM_DPP7("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 NCHMM_DPP7("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 NCHMM_DPP7("row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 NCHMM_DPP7("row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 NCHMM_DPP7("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 NCHMM_DPP7("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g));
#undef NCHMM_DPP7
}

// Seven wave sums in 21 instructions instead of 42 (gfx950's lane-swap instructions).  v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, so one swap + one add folds TWO values from 64 to 32 lanes
// (each then lives in one half); v_permlane16_swap does the same between 16-lane rows.  After the two folds the seven
// values sit one per row in two registers and four DPP steps sum each row.  Layout on return, row r = lanes 16 r .. 16 r + 15,
// every lane of a row holding the row's total:   q0 = rows { a, b, c, d },   q1 = rows { e, f, g, 0 }.
// (Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: hipcc 7.2 miscompiles the builtins' second result -- the add of
// the two halves comes out as `v_add_f32 v1, v1, v1`; tools/ubench/permlane_swap.hip shows it on the device.  The
// s_nop covers the two wait states between a VALU write and a lane-swap read, which the compiler cannot see in here.)
__device__ __forceinline__ void wave_sum7_rows(float a, float b, float c, float d, float e, float f, float g, float& q0, float& q1)
{
    float z = 0.0f;
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\t"      // a = [ a.lo | c.lo ], c = [ a.hi | c.hi ]
                 "v_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\t"
                 "v_permlane32_swap_b32 %6, %7"
                 : "+v"(a), "+v"(c), "+v"(b), "+v"(d), "+v"(e), "+v"(g), "+v"(f), "+v"(z));
    float ac = a + c, bd = b + d, eg = e + g, f0 = f + z;             // halves { a | c }, { b | d }, { e | g }, { f | 0 }
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\t"      // ac = rows { ac.0, bd.0, ac.2, bd.2 }, bd = rows { ac.1, bd.1, ac.3, bd.3 }
                 "v_permlane16_swap_b32 %2, %3"
                 : "+v"(ac), "+v"(bd), "+v"(eg), "+v"(f0));
    float u = ac + bd, v = eg + f0;                                    // rows { a, b, c, d } and { e, f, g, 0 }
    u = dpp_add<0xB1, 0xF>(u); v = dpp_add<0xB1, 0xF>(v);
    u = dpp_add<0x4E, 0xF>(u); v = dpp_add<0x4E, 0xF>(v);
    u = dpp_add<0x141, 0xF>(u); v = dpp_add<0x141, 0xF>(v);
    u = dpp_add<0x140, 0xF>(u); v = dpp_add<0x140, 0xF>(v);
    q0 = u; q1 = v;
}

// LDS tables are pair-major: the two floats of thread tau's cell pair `pair` of a table sit at
// sTab[f][(pair * 512 + tau) * 2 ..]: consecutive lanes read consecutive 8-byte words (conflict-free
// ds_read_b64) and every access of a thread is ONE base register plus an immediate offset.
__device__ __forceinline__ unsigned tab_off(unsigned tau, unsigned pair) { return (pair * (unsigned)kThreads + tau) * 2u; }


// resident blocks per CU from the kernel's own resource figures (the occupancy API prices LDS against 64 KiB)
inline int fb_blocks_per_cu(const void* fn)
{
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return 1;
    const int by_lds = fa.sharedSizeBytes > 0 ? (int)(163840 / fa.sharedSizeBytes) : 8;
    const int regs = ((fa.numRegs + 7) / 8) * 8;
    const int waves_per_simd = regs > 0 ? 512 / regs : 8;
    int nb = waves_per_simd * 4 / (kThreads / 64);
    if (by_lds < nb) nb = by_lds;
    if (nb > 32 / (kThreads / 64)) nb = 32 / (kThreads / 64);
    return nb < 1 ? 1 : nb;
}

}  // namespace fb
}  // namespace nchmm
