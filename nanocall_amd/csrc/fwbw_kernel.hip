// fwbw_kernel.hip -- forward-backward over the 4096-state pore HMM fused with the EM sufficient
// statistics of one training round, gfx950: the LOG-SPACE implementation.  It is exact for every cell, however
// improbable, as the reference is; it serves callers that want the alpha / beta matrices, the windows the
// rescaled linear-space kernels (fwbw_scaled_kernel.hip, the path of the EM rounds) flag as out of their
// range, and NCHMM_FB_FORCE_LOG=1.
//
// Replaces Forward_Backward::fill (src/nanocall/Forward_Backward.hpp:46-135) as driven by
// Parameter_Trainer::fill_train_data (Parameter_Trainer.hpp:141-155), plus the inner state sums of
// train_pm_params (:273-296) and the per-kmer sums of train_st_params (:451-515).  One training
// window per thread-block (persistent blocks pull windows from a queue).
//
// Numerics: log space, fp32, max-shifted log-sum-exp (the reference's `logsumset` is an un-vendored
// hpptools class; any exact log-sum-exp is inside the 1e-4 relative tolerance north_star sets for
// forward log-likelihoods -- tests hold log_pr_data, alpha/beta cells and the trained parameters to
// that).  Inside the kernels every log-quantity is kept in BASE 2 (multiplied by log2 e once), so the
// hardware exp2/log2 (v_exp_f32 / v_log_f32) are used without per-term scaling; rows written to HBM
// and all outputs are natural logs again.
//
// NORM (round 6): plain fp32 log space loses digits with the MAGNITUDE of alpha and beta -- a window with an abasic stretch has
// log Pr(data) ~ -2e4, where one ulp is 2e-3: the posteriors exp(alpha + beta - log Pr) came out 2 % off (the reference's own
// fp32 log space: 0.8 %; tools/ubench/fb_log_noise.py), and those are exactly the windows the rescaled kernels hand over.  When the
// caller does not ask for the matrices, every column is kept RELATIVE to an integer offset (in base-2 units: the floor of the
// previous column's maximum, accumulated; exact to apply, exact to undo): the registers, the rows in HBM and the exchange values
// stay within one event's emission of zero, the offsets travel as int32 per event (FwbwArgs::ws_exp) and are combined in integer
// arithmetic, and log2 posterior = a + b + (A_i + B_i - A_last) - l with every term small.  With the matrices requested
// (alpha_natural) the rows ARE the caller's output and stay absolute: the arithmetic of rounds 1-5.
//
// Two kernels, one window per 512-thread block each (persistent blocks + work queue), both within
// 128 VGPRs so that two blocks share a CU (4 waves/SIMD): fwbw_forward_kernel writes the alpha rows
// and log_pr_data, fwbw_backward_kernel reads them back one event ahead of use.  The emission is regrouped
// into five per-state constants (fwbw_common.hpp: make_state / emission2); the backward sweep keeps the
// stay / step coefficients and the emission constant k0 in LDS (48 KiB, pair-major); the skip coefficient
// equals the group weight w2 and is added once by the producer.
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
//             1/eta^2} over the UNSCALED model (Parameter_Trainer.hpp:273-296), accumulated on the scaled
//             constants and mapped back through pm_params (see the backward kernel);
//             per window, over the "clean" k-mers (Parameter_Trainer.hpp:30-57) and events i < n-1:
//             sum p, sum min(p_stay_joint, p), sum (p - min(p_stay_joint + p_step_joint, p))
//             (:451-515; linear-space sums of probabilities, returned as logs).
#include "nanocall_hip.h"
#include "nchmm_device.h"
#include "fwbw_common.hpp"

#include <algorithm>

#pragma clang fp contract(off)

namespace nchmm {

using namespace fb;

// ================================================ forward ================================================
namespace {
// floor of the largest of the eight wave maxima at `p` (block-uniform); 0 when there is no finite maximum to take out
__device__ __forceinline__ float block_floor_max(const float* p)
{
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a.x, a.y), __builtin_fmaxf(a.z, a.w)),
                                    __builtin_fmaxf(__builtin_fmaxf(b.x, b.y), __builtin_fmaxf(b.z, b.w)));
    return (m > -1.0e9f && m < 1.0e9f) ? __builtin_floorf(m) : 0.0f;
}
}  // namespace

template <bool NORM>
__global__ __launch_bounds__(kThreads, 4) void fwbw_forward_kernel(FwbwArgs P)
{
    __shared__ __attribute__((aligned(16))) float sG1[2][1024];
    __shared__ __attribute__((aligned(16))) float sG2[2][256];
    __shared__ __attribute__((aligned(16))) float4 sEv[kFbChunk];     // x, y, log2e * 3 log(y) / 2, 1/y
    __shared__ __attribute__((aligned(16))) float sMax[2][kThreads / 64];   // NORM: the waves' maxima of the previous column
    __shared__ float sRed[16];
    __shared__ unsigned sWork;

    const unsigned tau = threadIdx.x;
    const unsigned t = tau >> 1, h = tau & 1u;
    const unsigned wave = tau >> 6, lane = tau & 63u;

    for (;;) {
        // barrier first, then thread 0's fetch: see fwbw_scaled_kernel.hip (keeps the previous window's closing
        // `if (tau == 0)` and this one from being threaded together across the loop edge)
        __syncthreads();
        if (tau == 0) {
            const unsigned k = atomicAdd(P.queue, 1u);
            sWork = P.win_list ? (k < *P.n_list ? P.win_list[k] : 0xFFFFFFFFu) : k;
        }
        __syncthreads();
        const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)sWork);   // block-uniform: keep it (and all it indexes) scalar
        if (w >= P.n_win) break;
        do {    // (early outs `break` out of this block: one loop back edge)
        const uint64_t e0 = P.off[w];
        const unsigned n = (unsigned)(P.off[w + 1] - e0);
        if (n == 0) {
            if (tau == 0) P.out_log_pr_data[w] = __builtin_nanf("");
            break;
        }
        const int ms = P.scaled_slot ? P.scaled_slot[w] : 0;
        const int ts = P.trans_slot ? P.trans_slot[w] : 0;
        const float* __restrict__ M = P.models + (size_t)ms * kModelFloats;
        const float* __restrict__ C = P.trans_fb + (size_t)ts * kFbTransFloats;
        const float* __restrict__ ex = P.cmean + e0;
        const float* __restrict__ ey = P.stdv + e0;
        const float* __restrict__ el = P.lstdv + e0;
        float* const arow = P.ws_alpha + e0 * (uint64_t)kStates;   // alpha row i at arow + i*4096
        const float store_scale = P.alpha_natural ? kLn2 : 1.0f;

        float mu[8], r2[8], eta[8], lq[8], k0[8], c0[8], c1[8], alpha[8];   // alpha in base 2
        unsigned jj[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned k = 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1) + h;
            const unsigned j = t + 256u * k;
            jj[i] = j;
            const StateK s = make_state(M, j, P.log_2pi);
            mu[i] = s.mu; r2[i] = s.r2; eta[i] = s.eta; lq[i] = s.lq; k0[i] = s.k0;
            c0[i] = C[0 * kStates + j] * kLog2e;
            c1[i] = C[1 * kStates + j] * kLog2e;
        }
        // the skip coefficient c2[j] is the group weight w2[j >> 4] for every state: the producer of group t adds it
        const float w2 = P.trans[(size_t)ts * kTransFloats + kStates + 1024 + t] * kLog2e;
        const unsigned r1_base = (h << 6) + (t >> 2), q_base = (h << 4) + (t >> 4);
        int A = 0;      // NORM: alpha_i[j] (base 2) = alpha[q] + A -- the offset of the column in the registers

        for (unsigned base = 0; base < n; base += kFbChunk) {
            const unsigned ie = base + tau;
            if (tau < kFbChunk && ie < n) {
                const float y = ey[ie];
                sEv[tau] = make_float4(ex[ie], y, (1.5f * kLog2e) * el[ie], 1.0f / y);
            }
            __syncthreads();
            const unsigned hi = (n - base < kFbChunk) ? n - base : kFbChunk;
            for (unsigned c = 0; c < hi; ++c) {
                const float4 ev = sEv[c];
                const unsigned i = base + c;
                const unsigned buf = i & 1u;
                if (i > 0) {
                    // Forward_Backward.hpp:72-89: raw group sums of the previous column
                    const MaxSum a = lse4(alpha[0], alpha[2], alpha[4], alpha[6]);   // y = h
                    const MaxSum b = lse4(alpha[1], alpha[3], alpha[5], alpha[7]);   // y = h + 2
                    MaxSum s8 = lse_merge(a, b);
                    s8 = lse_merge(s8, MaxSum{swap1(s8.m), swap1(s8.s)});
                    sG1[buf][(h << 8) | t] = a.m + lg2(a.s);
                    sG1[buf][((2u + h) << 8) | t] = b.m + lg2(b.s);
                    if (h == 0) sG2[buf][t] = s8.m + lg2(s8.s) + w2;
                    if (NORM) {
                        const float wm = wave_max(__builtin_fmaxf(a.m, b.m));     // (a.m, b.m: the maxima of this thread's eight cells)
                        if (lane == 0) sMax[buf][wave] = wm;
                    }
                    __syncthreads();
                }
                const float* pa = &sG1[buf][r1_base];
                const float* pb = &sG2[buf][q_base];
                // NORM: the previous column's maximum comes out of this column (the same for every state: an exact shift)
                float shift = 0.0f;
                if (NORM && i > 0) {
                    shift = block_floor_max(&sMax[buf][0]);
                    A += (int)shift;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned kc = 4u * (unsigned)(q >> 1) + 2u * (unsigned)(q & 1);
                    const float e = emission2(ev.x, ev.y, ev.w, ev.z, mu[q], r2[q], eta[q], lq[q], k0[q]);
                    if (i == 0) alpha[q] = e - P.log_n_states * kLog2e;                   // Forward_Backward.hpp:58-68
                    else alpha[q] = e + (lse3(c0[q] + alpha[q], c1[q] + pa[kc << 6], pb[kc << 4]) - shift);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) arow[(uint64_t)i * kStates + jj[q]] = alpha[q] * store_scale;
                if (NORM && tau == 0) P.ws_exp[e0 + i] = A;
            }
            __syncthreads();
        }
        // log_pr_data = LSE_j alpha[n-1][j]  (Forward_Backward.hpp:129-134)
        float m = kNegBig;
#pragma unroll
        for (int q = 0; q < 8; ++q) m = __builtin_fmaxf(m, alpha[q]);
        m = wave_max(m);
        if (lane == 0) sRed[wave] = m;
        __syncthreads();
        float bm = sRed[0];
#pragma unroll
        for (int q = 1; q < kThreads / 64; ++q) bm = __builtin_fmaxf(bm, sRed[q]);
        float s = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += ex2(alpha[q] - bm);
        s = wave_sum(s);
        if (lane == 0) sRed[8 + wave] = s;
        __syncthreads();
        float bs = 0;
#pragma unroll
        for (int q = 0; q < kThreads / 64; ++q) bs += sRed[8 + q];
        if (tau == 0) {
            const float l2 = bm + lg2(bs);
            P.ws_lpd2[w] = l2;                       // NORM: relative to the last column's offset A (= ws_exp[e0 + n - 1])
            P.out_log_pr_data[w] = NORM ? (float)(((double)A + (double)l2) * 0.69314718055994530942) : l2 * kLn2;
        }
        } while (0);
    }
}

// ================================================ backward + statistics ================================================
// The emission statistics of Parameter_Trainer.hpp:284-296 are sums over the UNSCALED model:
//   s0,s1,s2 = sum_j p / su^2 {1, mu_u, mu_u^2}     l0,l1,l2 = sum_j p lambda_u {1, 1/eta_u, 1/eta_u^2}
// The kernel holds only the scaled states (mu = mu_u scale + shift, sigma = su var, eta = eta_u scale_sd,
// lambda = lambda_u var_sd; Pore_Model.hpp:126-138) and accumulates, per event,
//   S0,S1,S2 = sum_j p r2 {1, mu, mu^2}            S3,S4,S5 = sum_j p lq {eta^2, eta, 1}
// (r2, lq are the emission constants already in registers).  The six block sums are mapped back by one thread
// each, in double:  s0 = cU S0, s1 = cU (S1 - shift S0) / scale, s2 = cU (S2 - 2 shift S1 + shift^2 S0) / scale^2,
// l0 = cL S3, l1 = cL scale_sd S4, l2 = cL scale_sd^2 S5,  cU = 2 ln2 var^2, cL = 2 ln2 / var_sd.
template <bool NORM>
__global__ __launch_bounds__(kThreads, 4) void fwbw_backward_kernel(FwbwArgs P)
{
    __shared__ __attribute__((aligned(16))) float sTab[3][kStates];   // c0b log2 | c1b log2 | emission constant k0
    __shared__ __attribute__((aligned(16))) float sG1[2][1024];
    __shared__ __attribute__((aligned(16))) float sG2[2][256];
    __shared__ __attribute__((aligned(16))) float sMax[2][kThreads / 64];   // NORM: the waves' maxima of g
    __shared__ float sRed[16];
    __shared__ __attribute__((aligned(16))) float sAcc[2][kThreads / 64][8];   // per-event sums of each wave, by event parity
    __shared__ float sCoef[6][4];   // per window: how output q is formed from the block sums {k_a, k_b, k_c}
    __shared__ unsigned sWork;

    const unsigned tau = threadIdx.x;
    const unsigned t = tau >> 1, h = tau & 1u;
    const unsigned wave = tau >> 6, lane = tau & 63u;
    const unsigned j0 = tau * 8u;

    for (;;) {
        // barrier first, then thread 0's fetch: see fwbw_scaled_kernel.hip (keeps the previous window's closing
        // `if (tau == 0)` and this one from being threaded together across the loop edge)
        __syncthreads();
        if (tau == 0) {
            const unsigned k = atomicAdd(P.queue, 1u);
            sWork = P.win_list ? (k < *P.n_list ? P.win_list[k] : 0xFFFFFFFFu) : k;
        }
        __syncthreads();
        const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)sWork);   // block-uniform: keep it (and all it indexes) scalar
        if (w >= P.n_win) break;
        do {    // (early outs `break` out of this block: one loop back edge)
        const uint64_t e0 = P.off[w];
        const unsigned n = (unsigned)(P.off[w + 1] - e0);
        if (n == 0) {
            if (tau == 0 && P.out_st_sums) { P.out_st_sums[3 * w] = P.out_st_sums[3 * w + 1] = P.out_st_sums[3 * w + 2] = -__builtin_inff(); }
            break;
        }
        const int ms = P.scaled_slot ? P.scaled_slot[w] : 0;
        const int ts = P.trans_slot ? P.trans_slot[w] : 0;
        const float* __restrict__ M = P.models + (size_t)ms * kModelFloats;
        const float* __restrict__ C = P.trans_fb + (size_t)ts * kFbTransFloats;
        const float* __restrict__ ex = P.cmean + e0;
        const float* __restrict__ ey = P.stdv + e0;
        const float* __restrict__ el = P.lstdv + e0;
        const float* const arow = P.ws_alpha + e0 * (uint64_t)kStates;
        const float lpd = P.ws_lpd2[w];   // base 2; NORM: relative to the last column's offset
        const float load_scale = P.alpha_natural ? kLog2e : 1.0f;
        const int32_t* __restrict__ aexp = P.ws_exp + e0;        // NORM: the offset A_i of every alpha row
        const int A_last = NORM ? aexp[n - 1] : 0;
        int B = 0;                                               // NORM: beta_i[j] (base 2) = beta[u] + B

        float mu[8], r2[8], eta[8], lq[8], beta[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned j = j0 + u;
            const StateK s = make_state(M, j, P.log_2pi);
            mu[u] = s.mu; r2[u] = s.r2; eta[u] = s.eta; lq[u] = s.lq;
            const unsigned o = tab_off(tau, (unsigned)u >> 1) + ((unsigned)u & 1u);
            sTab[2][o] = s.k0;
            sTab[0][o] = C[3 * kStates + j] * kLog2e;
            sTab[1][o] = C[4 * kStates + j] * kLog2e;   // (each thread reads back only what it wrote)
            beta[u] = 0.0f;                        // Forward_Backward.hpp:93-103
        }
        // c2b[j] = w2[j & 255] for every state: the producer of successor group q = tau >> 1 adds it
        const float w2 = P.trans[(size_t)ts * kTransFloats + kStates + 1024 + t] * kLog2e;
        const unsigned train = P.train_mask[tau];   // bit u: state j0+u is a transition-training k-mer
        float lps = 0.0f, lps4 = 0.0f;
        if (P.st_params) {
            const float p_stay = P.st_params[2 * w], p_skip = P.st_params[2 * w + 1];
            lps = lg2(p_stay);                              // Parameter_Trainer.hpp:444
            lps4 = lg2(1.0f - p_stay - p_skip) - 2.0f;      // :445  (log2 of 1/4)
        }
        // how thread q < 6 turns the block sums {S_q, S_ib, S_ic} into its output (see the kernel comment)
        if (tau < 6) {
            float k_a = 0, k_b = 0, k_c = 0;
            float scale = 1, shift = 0, var = 1, scale_sd = 1, var_sd = 1;
            if (P.pm_params) {
                const float* q = P.pm_params + 6 * (size_t)w;
                scale = q[0]; shift = q[1]; var = q[3]; scale_sd = q[4]; var_sd = q[5];
            }
            const double two_ln2 = 2.0 * 0.69314718055994530942;
            const double cU = two_ln2 * (double)var * (double)var, cL = two_ln2 / (double)var_sd;
            const double sh = shift, sc = scale, ssd = scale_sd;
            switch (tau) {
            case 0: k_a = (float)cU; break;
            case 1: k_a = (float)(cU / sc); k_b = (float)(-cU * sh / sc); break;                   // with S0
            case 2:
                k_a = (float)(cU / (sc * sc)); k_b = (float)(-2.0 * cU * sh / (sc * sc));          // with S1
                k_c = (float)(cU * sh * sh / (sc * sc));                                             // with S0
                break;
            case 3: k_a = (float)cL; break;
            case 4: k_a = (float)(cL * ssd); break;
            default: k_a = (float)(cL * ssd * ssd); break;
            }
            sCoef[tau][0] = k_a; sCoef[tau][1] = k_b; sCoef[tau][2] = k_c;   // read back by the same thread only
        }
        float acc_p = 0, acc_stay = 0, acc_skip = 0;

        const float* rowp = arow + (uint64_t)(n - 1) * kStates;    // uniform row pointer + 32-bit thread offset
        float4 nx_lo = *reinterpret_cast<const float4*>(rowp + j0);
        float4 nx_hi = *reinterpret_cast<const float4*>(rowp + j0 + 4);
        float nx_x = ex[n - 1], nx_y = ey[n - 1], nx_l = el[n - 1];
        float ps[6];
        // the six emission sums of one state (Parameter_Trainer.hpp:284-296, on the scaled constants)
        auto pm_add = [&](int u, float p) {
            const float t0 = p * r2[u], l0 = p * lq[u];
            const float t1 = t0 * mu[u], l1 = l0 * eta[u];
            ps[0] += t0; ps[1] += t1; ps[2] += t1 * mu[u];
            ps[5] += l0; ps[4] += l1; ps[3] += l1 * eta[u];
        };
        // end of one event's statistics: beta row out, the wave's six sums to LDS.  The block total is formed after
        // the NEXT barrier (publish) -> out_pm_sums[e0 + ei][0..5]
        auto stats_end = [&](unsigned ei) {
            if (P.out_beta) {
                float* brow = P.out_beta + (e0 + (uint64_t)ei) * kStates;
                brow += j0;
                *reinterpret_cast<float4*>(brow) = make_float4(beta[0] * kLn2, beta[1] * kLn2, beta[2] * kLn2, beta[3] * kLn2);
                *reinterpret_cast<float4*>(brow + 4) = make_float4(beta[4] * kLn2, beta[5] * kLn2, beta[6] * kLn2, beta[7] * kLn2);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) ps[q] = wave_sum_lane63(ps[q]);
            if (lane == 63) {
                float* dst = &sAcc[ei & 1u][wave][0];
                *reinterpret_cast<float4*>(dst) = make_float4(ps[0], ps[1], ps[2], ps[3]);
                *reinterpret_cast<f2*>(dst + 4) = f2{ps[4], ps[5]};
            }
        };
        auto publish = [&](unsigned ei) {
            if (tau < 6 && P.out_pm_sums) {
                const unsigned i_b = tau == 2 ? 1u : 0u;    // second term: S1 for s2, S0 for s1; third term: S0
                float va = 0.0f, vb = 0.0f, vc = 0.0f;
#pragma unroll
                for (int wv = 0; wv < kThreads / 64; ++wv) {
                    va += sAcc[ei & 1u][wv][tau]; vb += sAcc[ei & 1u][wv][i_b]; vc += sAcc[ei & 1u][wv][0];
                }
                P.out_pm_sums[(e0 + (uint64_t)ei) * 6 + tau] =
                    __builtin_fmaf(sCoef[tau][0], va, __builtin_fmaf(sCoef[tau][1], vb, sCoef[tau][2] * vc));
            }
        };

        {   // event n-1: beta = 0, no following event
            const float al[8] = {nx_lo.x, nx_lo.y, nx_lo.z, nx_lo.w, nx_hi.x, nx_hi.y, nx_hi.z, nx_hi.w};
#pragma unroll
            for (int q = 0; q < 6; ++q) ps[q] = 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) pm_add(u, ex2(al[u] * load_scale - lpd));
            stats_end(n - 1);
        }
        for (int i = (int)n - 1; i >= 1; --i) {
            // on entry: beta = beta_i; nx_x/y/l = event i.  Leaves beta = beta_{i-1} and the statistics of event i-1.
            const unsigned buf = (unsigned)i & 1u;
            const float x = nx_x, y = nx_y, l3 = (1.5f * kLog2e) * nx_l, ry = 1.0f / y;
            // alpha_{i-1} for the statistics at the end of this iteration, event i-1 for the next one
            rowp -= kStates;
            nx_lo = *reinterpret_cast<const float4*>(rowp + j0);
            nx_hi = *reinterpret_cast<const float4*>(rowp + j0 + 4);
            nx_x = ex[i - 1]; nx_y = ey[i - 1]; nx_l = el[i - 1];
            const int A_prev = NORM ? aexp[i - 1] : 0;
            // g = emission(event i) + beta_i; H1/H2 over consecutive successor groups (Forward_Backward.hpp:107-125)
            float g[8];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const f2 k02 = *reinterpret_cast<const f2*>(&sTab[2][tab_off(tau, pr)]);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int q = 2 * pr + v;
                    g[q] = emission2(x, y, ry, l3, mu[q], r2[q], eta[q], lq[q], k02[v]) + beta[q];
                }
            }
            const MaxSum a = lse4(g[0], g[1], g[2], g[3]);
            const MaxSum b = lse4(g[4], g[5], g[6], g[7]);
            MaxSum s8 = lse_merge(a, b);
            s8 = lse_merge(s8, MaxSum{swap1(s8.m), swap1(s8.s)});
            sG1[buf][2 * tau] = a.m + lg2(a.s);
            sG1[buf][2 * tau + 1] = b.m + lg2(b.s);
            if (h == 0) sG2[buf][t] = s8.m + lg2(s8.s) + w2;
            if (NORM) {
                const float wm = wave_max(__builtin_fmaxf(a.m, b.m));
                if (lane == 0) sMax[buf][wave] = wm;
            }
            __syncthreads();
            publish((unsigned)i);    // the barrier made every wave's sums of event i visible
            // log2 of the posterior of (i-1, u) = alpha + beta + kp; of the joint terms with g / H1 of event i = ... + kq.
            // NORM: g's maximum comes out of beta_{i-1}; the integer offsets of the two columns and of the total combine exactly
            float shift = 0.0f, kp = -lpd, kq = -lpd;
            if (NORM) {
                shift = block_floor_max(&sMax[buf][0]);
                kq = (float)(A_prev + B - A_last) - lpd;
                B += (int)shift;
                kp = (float)(A_prev + B - A_last) - lpd;
            }
            const float al[8] = {nx_lo.x, nx_lo.y, nx_lo.z, nx_lo.w, nx_hi.x, nx_hi.y, nx_hi.z, nx_hi.w};
            const float* ph1 = &sG1[buf][j0 & 1023u];
            const float* ph2 = &sG2[buf][j0 & 255u];
#pragma unroll
            for (int q = 0; q < 6; ++q) ps[q] = 0.0f;
            // per state: beta_{i-1}, then at once everything that needs (g, H1) of event i -- the posterior of event
            // i-1 and the transition sums of the pair (i-1, i), Parameter_Trainer.hpp:470-512 -- so both die here
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const f2 c02 = *reinterpret_cast<const f2*>(&sTab[0][tab_off(tau, pr)]);
                const f2 c12 = *reinterpret_cast<const f2*>(&sTab[1][tab_off(tau, pr)]);
                const f2 h1 = *reinterpret_cast<const f2*>(ph1 + 2 * pr);
                const f2 h2 = *reinterpret_cast<const f2*>(ph2 + 2 * pr);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int u = 2 * pr + v;
                    beta[u] = lse3(c02[v] + g[u], c12[v] + h1[v], h2[v]) - shift;
                    const float a2 = al[u] * load_scale;
                    const float p = ex2(a2 + beta[u] + kp);               // exp(Forward_Backward::log_posterior)
                    pm_add(u, p);
                    const float pm = ((train >> u) & 1u) ? p : 0.0f;      // only transition-training k-mers count
                    const float pst = __builtin_fminf(ex2(a2 + lps + g[u] + kq), pm);
                    const float pstep = ex2(a2 + lps4 + h1[v] + kq);
                    const float p01 = __builtin_fminf(pst + pstep, pm);
                    acc_p += pm; acc_stay += pst; acc_skip += pm - p01;
                }
                __builtin_amdgcn_sched_barrier(0);   // one state pair in flight: bounds the live temporaries
            }
            stats_end((unsigned)(i - 1));
        }
        __syncthreads();
        publish(0u);
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
            P.out_st_sums[3 * w + 0] = lg2(tp) * kLn2;
            P.out_st_sums[3 * w + 1] = lg2(tst) * kLn2;
            P.out_st_sums[3 * w + 2] = lg2(tsk) * kLn2;
        }
        } while (0);
    }
}

void launch_fwbw(const FwbwArgs& a, int grid, hipStream_t stream, bool scaled)
{
    // forward then backward on one stream: the backward sweep reads the alpha rows and log_pr_data of the forward one
    FwbwArgs f = a, b = a;
    f.win_list = b.win_list = nullptr; f.n_list = b.n_list = nullptr;
    if (scaled) {
        launch_fwbw_scaled(a, grid, stream);                  // queue words 0, 1
        // exact redo, in log space, of whatever windows the scaled kernels flagged (normally none: the blocks
        // find an empty list and leave)
        f.win_list = b.win_list = a.fb_list; f.n_list = b.n_list = a.fb_count;
        f.alpha_natural = b.alpha_natural = 0;
        f.queue = a.queue + 2; b.queue = a.queue + 3;
    } else {
        f.queue = a.queue; b.queue = a.queue + 1;
    }
    // the matrices requested: absolute rows in the caller's buffers; else columns relative to integer offsets (NORM, see the top)
    if (f.alpha_natural || f.out_beta) {
        hipLaunchKernelGGL(fwbw_forward_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, f);
        hipLaunchKernelGGL(fwbw_backward_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, b);
    } else {
        hipLaunchKernelGGL(fwbw_forward_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, f);
        hipLaunchKernelGGL(fwbw_backward_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, b);
    }
}

int fwbw_blocks_per_cu()
{
    const int f = std::min(fb_blocks_per_cu(reinterpret_cast<const void*>(fwbw_forward_kernel<false>)), fb_blocks_per_cu(reinterpret_cast<const void*>(fwbw_forward_kernel<true>)));
    const int b = std::min(fb_blocks_per_cu(reinterpret_cast<const void*>(fwbw_backward_kernel<false>)), fb_blocks_per_cu(reinterpret_cast<const void*>(fwbw_backward_kernel<true>)));
    const int s = fwbw_scaled_blocks_per_cu();
    return f < b ? (f < s ? f : s) : (b < s ? b : s);
}

}  // namespace nchmm
