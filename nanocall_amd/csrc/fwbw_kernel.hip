// fwbw_kernel.hip -- forward-backward over the 4096-state pore HMM fused with the EM sufficient
// statistics of one training round, gfx950.
//
// Replaces Forward_Backward::fill (src/nanocall/Forward_Backward.hpp:46-135) as driven by
// Parameter_Trainer::fill_train_data (Parameter_Trainer.hpp:141-155), plus the inner state sums of
// train_pm_params (:273-296) and the per-kmer sums of train_st_params (:451-515).  One training
// window per thread-block (persistent blocks pull windows from a queue).
//
// Numerics: log space, fp32, max-shifted log-sum-exp (the reference's `logsumset` is an un-vendored
// hpptools class; any exact log-sum-exp is inside the 1e-4 relative tolerance north_star sets for
// forward log-likelihoods -- tests hold log_pr_data, alpha/beta cells and the trained parameters to
// that).  Transcendentals are the hardware exp2/log2 (v_exp_f32 / v_log_f32).
//
// Structure (same k-mer algebra as viterbi_kernel.hip, sums instead of maxima):
//   forward   alpha_i[j] = e_j(i) + LSE( c0[j] + alpha[j], c1[j] + G1[j>>2], c2[j] + G2[j>>4] )
//             G1[r] = LSE over the 4 states with low 10 bits r, G2[q] = LSE over the 16 states with low
//             8 bits q of the previous column (raw, unweighted).  Thread 2t+h owns 8 of the 16 states
//             with low 8 bits t, so both group sums are in-register + one DPP swap; consumers read
//             them from LDS (one barrier per event).  alpha rows go to HBM (fp32, 16 KiB per event).
//   backward  beta_i[j] = LSE( c0b[j] + g[j], c1b[j] + H1[j&1023], c2b[j] + H2[j&255] ),
//             g[q] = e_q(i+1) + beta_{i+1}[q]; H1/H2 = LSE over the 4 / 16 CONSECUTIVE successor
//             states, so here thread tau owns the 8 consecutive states 8*tau..8*tau+7.
//             beta never leaves the chip unless the caller asks for it.
//   The per-state weights c0/c1/c2 (forward) and c0b/c1b/c2b (backward) are the factorised arc
//   weights with the double-counted arcs of the 28 low-complexity k-mers folded in on the host
//   (nchmm_api.cpp: fb_weights), so the kernel has no special cases.
//   statistics, in the backward sweep: p_ij = exp(alpha + beta - log_pr_data);
//             per event {s0,s1,s2,l0,l1,l2} = sum_j p_ij {1, mu, mu^2}/sigma^2, p_ij lambda {1, 1/eta,
//             1/eta^2} over the UNSCALED model (Parameter_Trainer.hpp:273-296);
//             per window, over the "clean" k-mers (Parameter_Trainer.hpp:30-57) and events i < n-1:
//             sum p, sum min(p_stay_joint, p), sum (p - min(p_stay_joint + p_step_joint, p))
//             (:451-515; linear-space sums of probabilities, returned as logs).
#include "nanocall_hip.h"
#include "nchmm_device.h"

#pragma clang fp contract(off)

namespace nchmm {

namespace {

constexpr unsigned kFbChunk = 128;   // events staged in LDS at a time
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -3.0e38f;

__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * kLn2; }   // v_log_f32 is log2

__device__ __forceinline__ float swap1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// n / d through the precomputed reciprocal r = RN(1/d) with one residual correction: within 1 ulp of
// the IEEE quotient (FB is tolerance-checked at 1e-4, not bit-checked), 3 ops instead of ~10 + v_rcp.
__device__ __forceinline__ float quot(float n, float d, float r)
{
    const float q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-q, d, n), r, q);
}

// Pore_Model_State::log_pr_corrected_emission, Pore_Model.hpp:145-149 (same expression as the
// Viterbi kernel)
__device__ __forceinline__ float emission(float x, float y, float ry, float ly3, float log_2pi, float mu, float sg, float rsg,
                                          float nls, float eta, float reta, float lam, float c)
{
    const float a = quot(x - mu, sg, rsg);
    const float n = nls - (log_2pi + a * a) / 2.0f;
    const float b = quot(y - eta, eta, reta);
    const float ig = (c - ly3 - quot(lam * b * b, y, ry)) / 2.0f;
    return n + ig;
}

struct MaxSum { float m, s; };   // running log-sum-exp: value = m + log(s)

__device__ __forceinline__ MaxSum lse_merge(MaxSum a, MaxSum b)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(a.m, b.m), kNegBig);
    return MaxSum{m, a.s * fexp(a.m - m) + b.s * fexp(b.m - m)};
}

__device__ __forceinline__ MaxSum lse4(float a, float b, float c, float d)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(c, d)), kNegBig);
    return MaxSum{m, fexp(a - m) + fexp(b - m) + fexp(c - m) + fexp(d - m)};
}

__device__ __forceinline__ float lse3(float a, float b, float c)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a, b), c), kNegBig);
    return m + flog(fexp(a - m) + fexp(b - m) + fexp(c - m));
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

}  // namespace

__global__ __launch_bounds__(kThreads, 2) void fwbw_kernel(FwbwArgs P)
{
    __shared__ __attribute__((aligned(16))) float sG1[2][1024];
    __shared__ __attribute__((aligned(16))) float sG2[2][256];
    __shared__ __attribute__((aligned(16))) float4 sEv[kFbChunk];   // x, y, 3 log y, start-of-chunk pad
    __shared__ float sRed[16];
    __shared__ float sAcc[2][8];
    __shared__ unsigned sWork;

    const unsigned tau = threadIdx.x;
    const unsigned t = tau >> 1, h = tau & 1u;
    const unsigned wave = tau >> 6, lane = tau & 63u;
    float* const ws = P.ws_alpha;

    for (;;) {
        if (tau == 0) sWork = atomicAdd(P.queue, 1u);
        __syncthreads();
        const unsigned w = sWork;
        __syncthreads();
        if (w >= P.n_win) break;
        const uint64_t e0 = P.off[w];
        const unsigned n = (unsigned)(P.off[w + 1] - e0);
        if (n == 0) {
            if (tau == 0) {
                P.out_log_pr_data[w] = __builtin_nanf("");
                if (P.out_st_sums) { P.out_st_sums[3 * w] = P.out_st_sums[3 * w + 1] = P.out_st_sums[3 * w + 2] = -__builtin_inff(); }
            }
            continue;
        }
        const int ms = P.scaled_slot ? P.scaled_slot[w] : 0;
        const int us = P.unscaled_slot ? P.unscaled_slot[w] : ms;
        const int ts = P.trans_slot ? P.trans_slot[w] : 0;
        const float* __restrict__ M = P.models + (size_t)ms * kModelFloats;
        const float* __restrict__ U = P.models + (size_t)us * kModelFloats;
        const float* __restrict__ C = P.trans_fb + (size_t)ts * kFbTransFloats;
        const float* __restrict__ ex = P.cmean + e0;
        const float* __restrict__ ey = P.stdv + e0;
        const float* __restrict__ el = P.lstdv + e0;
        float* const arow = ws + e0 * (uint64_t)kStates;   // alpha row i at arow + i*4096

        // =========================== forward ===========================
        float lpd;
        {
            float mu[8], sg[8], rsg[8], nls[8], eta[8], reta[8], lam[8], cc[8], c0[8], c1[8], c2[8], alpha[8];
            unsigned jj[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned k = 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1) + h;
                const unsigned j = t + 256u * k;
                jj[i] = j;
                mu[i] = M[MF_MU * kStates + j]; sg[i] = M[MF_SIGMA * kStates + j]; rsg[i] = M[MF_RSIGMA * kStates + j];
                nls[i] = M[MF_NEG_LOG_SIGMA * kStates + j]; eta[i] = M[MF_ETA * kStates + j]; reta[i] = M[MF_RETA * kStates + j];
                lam[i] = M[MF_LAMBDA * kStates + j]; cc[i] = M[MF_C * kStates + j];
                c0[i] = C[0 * kStates + j]; c1[i] = C[1 * kStates + j]; c2[i] = C[2 * kStates + j];
            }
            const unsigned r1_base = (h << 6) + (t >> 2), q_base = (h << 4) + (t >> 4);
            for (unsigned base = 0; base < n; base += kFbChunk) {
                const unsigned ie = base + tau;
                if (tau < kFbChunk && ie < n) sEv[tau] = make_float4(ex[ie], ey[ie], 3.0f * el[ie], 1.0f / ey[ie]);
                __syncthreads();
                const unsigned hi = (n - base < kFbChunk) ? n - base : kFbChunk;
                for (unsigned c = 0; c < hi; ++c) {
                    const float4 ev = sEv[c];
                    const unsigned i = base + c;
                    if (i == 0) {
                        // Forward_Backward.hpp:58-68
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            alpha[u] = emission(ev.x, ev.y, ev.w, ev.z, P.log_2pi, mu[u], sg[u], rsg[u], nls[u], eta[u], reta[u],
                                                lam[u], cc[u]) - P.log_n_states;
                    } else {
                        // Forward_Backward.hpp:72-89
                        const unsigned buf = i & 1u;
                        const MaxSum a = lse4(alpha[0], alpha[2], alpha[4], alpha[6]);   // y = h
                        const MaxSum b = lse4(alpha[1], alpha[3], alpha[5], alpha[7]);   // y = h + 2
                        MaxSum s8 = lse_merge(a, b);
                        s8 = lse_merge(s8, MaxSum{swap1(s8.m), swap1(s8.s)});
                        sG1[buf][(h << 8) | t] = a.m + flog(a.s);
                        sG1[buf][((2u + h) << 8) | t] = b.m + flog(b.s);
                        if (h == 0) sG2[buf][t] = s8.m + flog(s8.s);
                        __syncthreads();
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const unsigned kc = 4u * (unsigned)(u >> 1) + 2u * (unsigned)(u & 1);
                            const float g1 = sG1[buf][r1_base + (kc << 6)], g2 = sG2[buf][q_base + (kc << 4)];
                            const float e = emission(ev.x, ev.y, ev.w, ev.z, P.log_2pi, mu[u], sg[u], rsg[u], nls[u], eta[u], reta[u],
                                                     lam[u], cc[u]);
                            alpha[u] = e + lse3(c0[u] + alpha[u], c1[u] + g1, c2[u] + g2);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) arow[(uint64_t)i * kStates + jj[u]] = alpha[u];
                }
                __syncthreads();
            }
            // log_pr_data = LSE_j alpha[n-1][j]  (Forward_Backward.hpp:129-134)
            float m = kNegBig;
#pragma unroll
            for (int u = 0; u < 8; ++u) m = __builtin_fmaxf(m, alpha[u]);
            m = wave_max(m);
            if (lane == 0) sRed[wave] = m;
            __syncthreads();
            float bm = sRed[0];
#pragma unroll
            for (int q = 1; q < kThreads / 64; ++q) bm = __builtin_fmaxf(bm, sRed[q]);
            float s = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += fexp(alpha[u] - bm);
            s = wave_sum(s);
            if (lane == 0) sRed[8 + wave] = s;
            __syncthreads();
            float bs = 0;
#pragma unroll
            for (int q = 0; q < kThreads / 64; ++q) bs += sRed[8 + q];
            lpd = bm + flog(bs);
            if (tau == 0) P.out_log_pr_data[w] = lpd;
            __syncthreads();   // alpha rows visible to the whole block (different ownership below); sRed reusable
        }

        // =========================== backward + statistics ===========================
        {
            const unsigned j0 = tau * 8u;
            float mu[8], sg[8], rsg[8], nls[8], eta[8], reta[8], lam[8], cc[8], c0[8], c1[8], c2[8];
            float u0[8], u1[8], u2[8], v0[8], v1[8], v2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned j = j0 + u;
                mu[u] = M[MF_MU * kStates + j]; sg[u] = M[MF_SIGMA * kStates + j]; rsg[u] = M[MF_RSIGMA * kStates + j];
                nls[u] = M[MF_NEG_LOG_SIGMA * kStates + j]; eta[u] = M[MF_ETA * kStates + j]; reta[u] = M[MF_RETA * kStates + j];
                lam[u] = M[MF_LAMBDA * kStates + j]; cc[u] = M[MF_C * kStates + j];
                c0[u] = C[3 * kStates + j]; c1[u] = C[4 * kStates + j]; c2[u] = C[5 * kStates + j];
                // Parameter_Trainer.hpp:284-289 on the UNSCALED model
                const float usg = U[MF_SIGMA * kStates + j], umu = U[MF_MU * kStates + j];
                const float ulam = U[MF_LAMBDA * kStates + j], ueta = U[MF_ETA * kStates + j];
                u0[u] = 1.0f / (usg * usg); u1[u] = u0[u] * umu; u2[u] = u1[u] * umu;
                v0[u] = ulam; v1[u] = v0[u] / ueta; v2[u] = v1[u] / ueta;
            }
            const unsigned train = P.train_mask[tau];   // bit u: state j0+u is a transition-training k-mer
            float lps = 0.0f, lps4 = 0.0f;
            if (P.st_params) {
                const float p_stay = P.st_params[2 * w], p_skip = P.st_params[2 * w + 1];
                lps = flog(p_stay);                                   // Parameter_Trainer.hpp:444
                lps4 = flog(1.0f - p_stay - p_skip) - flog(4.0f);     // :445
            }
            float beta[8], gprev[8], h1prev[8];
            float acc_p = 0, acc_stay = 0, acc_skip = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) { beta[u] = 0.0f; gprev[u] = 0.0f; h1prev[u] = 0.0f; }   // Forward_Backward.hpp:93-103
            if (tau < 16) { sAcc[0][tau & 7] = 0.0f; sAcc[1][tau & 7] = 0.0f; }
            __syncthreads();
            // alpha rows and event values are fetched one event ahead so their latency hides behind the
            // previous event's arithmetic
            float4 nx_lo = *reinterpret_cast<const float4*>(arow + (uint64_t)(n - 1) * kStates + j0);
            float4 nx_hi = *reinterpret_cast<const float4*>(arow + (uint64_t)(n - 1) * kStates + j0 + 4);
            float nx_x = ex[n - 1], nx_y = ey[n - 1], nx_l = el[n - 1];
            for (int i = (int)n - 1; i >= 0; --i) {
                const unsigned buf = (unsigned)i & 1u;
                // alpha_i of my 8 consecutive states
                const float al[8] = {nx_lo.x, nx_lo.y, nx_lo.z, nx_lo.w, nx_hi.x, nx_hi.y, nx_hi.z, nx_hi.w};
                const float x = nx_x, y = nx_y, ly3 = 3.0f * nx_l, ry = 1.0f / y;
                if (i > 0) {
                    nx_lo = *reinterpret_cast<const float4*>(arow + (uint64_t)(i - 1) * kStates + j0);
                    nx_hi = *reinterpret_cast<const float4*>(arow + (uint64_t)(i - 1) * kStates + j0 + 4);
                    nx_x = ex[i - 1]; nx_y = ey[i - 1]; nx_l = el[i - 1];
                }
                float ps[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float lp = al[u] + beta[u] - lpd;          // Forward_Backward::log_posterior
                    const float p = fexp(lp);
                    ps[0] += p * u0[u]; ps[1] += p * u1[u]; ps[2] += p * u2[u];
                    ps[3] += p * v0[u]; ps[4] += p * v1[u]; ps[5] += p * v2[u];
                    if ((i + 1 < (int)n) && ((train >> u) & 1u)) {
                        // Parameter_Trainer.hpp:470-512 for the pair (i, i+1); gprev/h1prev belong to event i+1
                        const float pst = __builtin_fminf(fexp(al[u] + lps + gprev[u] - lpd), p);
                        const float pstep = fexp(al[u] + lps4 + h1prev[u] - lpd);
                        const float p01 = __builtin_fminf(pst + pstep, p);
                        acc_p += p; acc_stay += pst; acc_skip += p - p01;
                    }
                }
                if (P.out_beta) {
                    float* brow = P.out_beta + (e0 + (uint64_t)i) * kStates + j0;
                    *reinterpret_cast<float4*>(brow) = make_float4(beta[0], beta[1], beta[2], beta[3]);
                    *reinterpret_cast<float4*>(brow + 4) = make_float4(beta[4], beta[5], beta[6], beta[7]);
                }
                // per-event block sums -> out_pm_sums[e0 + i][0..5]
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const float v = wave_sum(ps[q]);
                    if (lane == 0) atomicAdd(&sAcc[buf][q], v);
                }
                if (i > 0) {
                    // g = emission(event i) + beta_i; H1/H2 over consecutive successor groups (Forward_Backward.hpp:107-125)
                    float g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        g[u] = emission(x, y, ry, ly3, P.log_2pi, mu[u], sg[u], rsg[u], nls[u], eta[u], reta[u], lam[u], cc[u]) + beta[u];
                    const MaxSum a = lse4(g[0], g[1], g[2], g[3]);
                    const MaxSum b = lse4(g[4], g[5], g[6], g[7]);
                    MaxSum s8 = lse_merge(a, b);
                    s8 = lse_merge(s8, MaxSum{swap1(s8.m), swap1(s8.s)});
                    sG1[buf][2 * tau] = a.m + flog(a.s);
                    sG1[buf][2 * tau + 1] = b.m + flog(b.s);
                    if (h == 0) sG2[buf][t] = s8.m + flog(s8.s);
                    __syncthreads();
                    const unsigned rb = j0 & 1023u, qb = j0 & 255u;
                    const float4 h1a = *reinterpret_cast<const float4*>(&sG1[buf][rb]);
                    const float4 h1b = *reinterpret_cast<const float4*>(&sG1[buf][rb + 4]);
                    const float4 h2a = *reinterpret_cast<const float4*>(&sG2[buf][qb]);
                    const float4 h2b = *reinterpret_cast<const float4*>(&sG2[buf][qb + 4]);
                    const float H1[8] = {h1a.x, h1a.y, h1a.z, h1a.w, h1b.x, h1b.y, h1b.z, h1b.w};
                    const float H2[8] = {h2a.x, h2a.y, h2a.z, h2a.w, h2b.x, h2b.y, h2b.z, h2b.w};
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        beta[u] = lse3(c0[u] + g[u], c1[u] + H1[u], c2[u] + H2[u]);
                        gprev[u] = g[u]; h1prev[u] = H1[u];
                    }
                } else {
                    __syncthreads();
                }
                // the barrier above also completed every wave's atomicAdd for event i
                if (tau < 6 && P.out_pm_sums) P.out_pm_sums[(e0 + (uint64_t)i) * 6 + tau] = sAcc[buf][tau];
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the read above precedes the reset below
                if (tau < 6) sAcc[buf][tau] = 0.0f;    // reused two events later, after another barrier
            }
            // window totals of the transition statistics
            acc_p = wave_sum(acc_p); acc_stay = wave_sum(acc_stay); acc_skip = wave_sum(acc_skip);
            __syncthreads();
            if (lane == 0) { sRed[wave] = acc_p; sRed[8 + wave] = acc_stay; }
            __syncthreads();
            float tp = 0, tst = 0;
            if (tau == 0) {
                for (int q = 0; q < kThreads / 64; ++q) { tp += sRed[q]; tst += sRed[8 + q]; }
            }
            __syncthreads();
            if (lane == 0) sRed[wave] = acc_skip;
            __syncthreads();
            if (tau == 0 && P.out_st_sums) {
                float tsk = 0;
                for (int q = 0; q < kThreads / 64; ++q) tsk += sRed[q];
                P.out_st_sums[3 * w + 0] = flog(tp);
                P.out_st_sums[3 * w + 1] = flog(tst);
                P.out_st_sums[3 * w + 2] = flog(tsk);
            }
        }
    }
}

void launch_fwbw(const FwbwArgs& a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(fwbw_kernel, dim3(grid), dim3(kThreads), 0, stream, a);
}

int fwbw_blocks_per_cu()
{
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fwbw_kernel)) != hipSuccess) return 1;
    const int by_lds = fa.sharedSizeBytes > 0 ? (int)(163840 / fa.sharedSizeBytes) : 8;
    const int regs = ((fa.numRegs + 7) / 8) * 8;
    const int waves_per_simd = regs > 0 ? 512 / regs : 8;
    int nb = waves_per_simd * 4 / (kThreads / 64);
    if (by_lds < nb) nb = by_lds;
    if (nb > 32 / (kThreads / 64)) nb = 32 / (kThreads / 64);
    return nb < 1 ? 1 : nb;
}

}  // namespace nchmm
