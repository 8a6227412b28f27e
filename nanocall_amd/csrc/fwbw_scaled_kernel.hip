// fwbw_scaled_kernel.hip -- forward-backward + EM statistics in RESCALED LINEAR space, gfx950: the
// fast path of the EM rounds (no alpha / beta matrices requested).
//
// Same recursions, ownership and LDS exchange as fwbw_kernel.hip (Forward_Backward.hpp:46-135,
// Parameter_Trainer.hpp:273-296 and :451-515), but probabilities instead of log-probabilities:
//
//   ahat_i[j] = E_ij ( T0[j] ahat_{i-1}[j] + W1[j] S1[j>>2] + W2 S2[j>>4] ) 2^-e_i
//   E_ij      = 2^( (k0_j - K0max) - (x_i - mu_j)^2 r2_j - (y_i - eta_j)^2 lq_j / y_i )   <= 1
//
// S1 / S2 are plain sums over the 4 / 16 group members, e_i is the binary exponent of the previous
// column's total (exchanged through LDS in the same barrier as the group sums), so every rescaling is an
// exact power of two and the scale of a column is an INTEGER:  alpha_i[j] = ahat_i[j] 2^(R_i + Ia_i - 12),
// Ia_i = sum_{k<=i} e_k,  R_i = sum_{k<=i} (K0max - 3/2 log2 y_k).  The backward sweep does the same with
// g = E bhat over successor groups.  In the posterior the real-valued parts R cancel:
//   p_ij = ahat_i[j] btilde_i[j] 2^(Ia_i + Ib_{i+1} - Ia_{n-1}) / Z,      Z = sum_j ahat_{n-1}[j]
// so one exp per cell (the emission) is the only transcendental left; everything else is FMA work.
//
// Range.  A state whose share of its column is below 2^-126 / (column total) flushes to zero.  Columns are
// renormalised one event late, so the total can sag by the emission of one event (a stdv ten times the model's,
// a level ~10 sigma off every state: 40-50 bits); while it stays above 2^-100 the flushed share is below 2^-26 =
// 1.5e-8 of the column, four orders under the tolerance of anything computed from the posteriors.  A window
// where a column total falls below 2^-100 (an event no state explains), or where forward and backward mass
// barely overlap, is FLAGGED and redone by the exact log-space kernels of fwbw_kernel.hip (launch_fwbw does
// that in the same stream; tests/test_fwbw_gpu.py drives outliers through it).
#include "nanocall_hip.h"
#include "nchmm_device.h"
#include "fwbw_common.hpp"

#pragma clang fp contract(off)

namespace nchmm {

using namespace fb;

namespace {

constexpr float kMinTotal = 0x1p-100f;
constexpr unsigned kFwdG1Pitch = 72;                  // row pitch of the forward sweep's step-group sums (words)
constexpr unsigned kG1Pitch = 544, kG2Pitch = 144;   // plane pitches of the backward sweep's group-sum arrays (words)
constexpr int kMaxKappaExp = 100;     // |Ia + Ib - Ia_final| beyond this: forward and backward mass barely overlap

// binary exponent e of a positive normal float z (2^e <= z < 2^(e+1)) and the exact scale 2^-e
__device__ __forceinline__ int exponent_of(float z) { return (int)((__builtin_bit_cast(unsigned, z) >> 23) & 255u) - 127; }
__device__ __forceinline__ float pow2i(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }   // -126 <= e <= 127

// (x - mu)^2 r2 + (y - eta)^2 lq / y  -- the state-dependent part of -log2 emission.  The division by y is folded into
// the difference: (y - eta) / sqrt(y) = y rsq(y) - eta rsq(y) is one FMA on the two per-event values ysry = y rsq(y) and
// sry = rsq(y) (v_rsq_f32, one per thread and event), which takes a multiply per cell out of both sweeps: 6 VALU ops.
__device__ __forceinline__ float xs(float x, float ysry, float sry, float mu, float r2, float eta, float lq)
{
    const float dx = x - mu, dyp = __builtin_fmaf(-eta, sry, ysry);
    return __builtin_fmaf(dx * dx, r2, dyp * dyp * lq);
}

template <int BYTE_OFFSET>
__device__ __forceinline__ void store_row(float* p, float v)
{
    asm volatile("global_store_dword %0, %1, off offset:%2 sc1 nt" :: "v"(p), "v"(v), "n"(BYTE_OFFSET) : "memory");
}

template <typename T>
__device__ __forceinline__ T uniform_load(const T* p)
{
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ float block_max(float v, float* sRed, unsigned wave, unsigned lane)
{
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sRed[wave] = v;
    __syncthreads();
    float m = sRed[0];
#pragma unroll
    for (int q = 1; q < kThreads / 64; ++q) m = __builtin_fmaxf(m, sRed[q]);
    return m;
}

__device__ __forceinline__ void flag_window(const FwbwArgs& P, unsigned w)
{
    P.fb_flag[w] = 1;
    P.fb_list[atomicAdd(P.fb_count, 1u)] = w;
    atomicAdd(P.fb_total, 1ull);
}

}  // namespace

// ================================================ forward ================================================
__global__ __launch_bounds__(kThreads, 4) void fwbw_forward_scaled_kernel(FwbwArgs P)
{
    // step-group sums, 16 rows of 64 at a pitch of 72 words: the two halves of a wave (h = 0 / 1) read rows r and r + 1 at the
    // same column and write rows 4 apart -- at a pitch of 64 both land on the same banks (SQ_LDS_BANK_CONFLICT was above
    // SQ_ACTIVE_INST_LDS for this kernel); 72 = 64 + 8 and 4 * 72 = 288 = 4 * 64 + 32 separate them
    __shared__ __attribute__((aligned(16))) float sG1[2][16 * kFwdG1Pitch];
    __shared__ __attribute__((aligned(16))) float sG2[2][256];
    __shared__ __attribute__((aligned(16))) float sZ[2][kThreads / 64];
    __shared__ __attribute__((aligned(16))) float4 sEv[kFbChunk];     // x, y rsq(y), rsq(y), log2e * 3 log(y) / 2
    __shared__ float sRed[16];
    __shared__ unsigned sWork;

    const unsigned tau = threadIdx.x;
    const unsigned t = tau >> 1, h = tau & 1u;
    const unsigned wave = tau >> 6, lane = tau & 63u;

    for (;;) {
        // The barrier comes BEFORE thread 0's fetch: it ends the previous window, and it keeps that window's closing
        // `if (tau == 0)` apart from this one.  Back to back (only the loop edge between them) LLVM threads the two
        // tests together and sends every lane but thread 0 straight to the next barrier -- the waves then pass it
        // without a new sWork and spin on the old window forever (seen on gfx950 with ROCm 7.2).
        __syncthreads();
        if (tau == 0) sWork = atomicAdd(P.queue, 1u);
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
        float* rowp = P.ws_alpha + e0 * (uint64_t)kStates;

        float mu[8], r2[8], eta[8], lq[8], k0[8], T0[8], W1[8], ah[8];
        unsigned jj[8];
        float kmax = kNegBig;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned k = 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1) + h;
            const unsigned j = t + 256u * k;
            jj[i] = j;
            const StateK s = make_state(M, j, P.log_2pi);
            mu[i] = s.mu; r2[i] = s.r2; eta[i] = s.eta; lq[i] = s.lq; k0[i] = s.k0;
            kmax = __builtin_fmaxf(kmax, s.k0);
            T0[i] = ex2(C[0 * kStates + j] * kLog2e);
            W1[i] = ex2(C[1 * kStates + j] * kLog2e);
        }
        kmax = block_max(kmax, sRed, wave, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) k0[i] -= kmax;
        // the skip weight c2[j] is the group weight w2[j >> 4] for every state: the producer of group t applies it
        const float W2 = ex2(P.trans[(size_t)ts * kTransFloats + kStates + 1024 + t] * kLog2e);
        const unsigned r1_base = kFwdG1Pitch * h + (t >> 2), q_base = (h << 4) + (t >> 4);
        int Ia = 0;
        bool bad = false;
        double ref_sum = 0.0;   // R_i of the file comment

        for (unsigned base = 0; base < n; base += kFbChunk) {
            const unsigned ie = base + tau;
            if (tau < kFbChunk && ie < n) {
                const float y = ey[ie];
                const float sry = __builtin_amdgcn_rsqf(y);
                sEv[tau] = make_float4(ex[ie], y * sry, sry, (1.5f * kLog2e) * el[ie]);
            }
            __syncthreads();
            const unsigned hi = (n - base < kFbChunk) ? n - base : kFbChunk;
            for (unsigned c = 0; c < hi; ++c) {
                const float4 ev = sEv[c];
                const unsigned i = base + c;
                const unsigned buf = i & 1u;
                // ONE uniform branch per event.  (Written as `i == 0 ? E : E * (...)` per state, hipcc kept eight branches in
                // the loop body: eight serial emission chains, each LDS read waited for on its own.)
                if (i == 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)                                             // Forward_Backward.hpp:58-68
                        ah[q] = ex2(k0[q] - xs(ev.x, ev.y, ev.z, mu[q], r2[q], eta[q], lq[q]));
                } else {
                    // Forward_Backward.hpp:72-89: group sums of the previous column
                    // (producer phase at raised priority, as in the backward sweep: 1.35 -> 1.29 ms; profiles/r06_fb_backward_session.md)
                    __builtin_amdgcn_s_setprio(2);
                    const float a = (ah[0] + ah[2]) + (ah[4] + ah[6]);   // y = h
                    const float b = (ah[1] + ah[3]) + (ah[5] + ah[7]);   // y = h + 2
                    const float s8 = a + b;
                    const float s16 = s8 + swap1(s8);     // (DPP: both lanes of the pair must be active -- keep it outside the branch)
                    sG1[buf][kFwdG1Pitch * (4u * h + (t >> 6)) + (t & 63u)] = a;            // entry 256 h + t
                    sG1[buf][kFwdG1Pitch * (4u * (2u + h) + (t >> 6)) + (t & 63u)] = b;     // entry 256 (2 + h) + t
                    if (h == 0) sG2[buf][t] = s16 * W2;
                    const float z = wave_sum_lane63(s8);
                    if (lane == 63) sZ[buf][wave] = z;
                    __builtin_amdgcn_s_setprio(0);
                    __syncthreads();
                    const float4 z0 = *reinterpret_cast<const float4*>(&sZ[buf][0]);
                    const float4 z1 = *reinterpret_cast<const float4*>(&sZ[buf][4]);
                    const float* pa = &sG1[buf][r1_base];
                    const float* pb = &sG2[buf][q_base];
                    float in1[8], in2[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned kc = 4u * (unsigned)(q >> 1) + 2u * (unsigned)(q & 1);
                        in1[q] = pa[kc * kFwdG1Pitch]; in2[q] = pb[kc << 4];     // entries 64 (h + kc) + (t >> 2), 16 (h + kc) + (t >> 4)
                    }
                    const float Z = ((z0.x + z0.y) + (z0.z + z0.w)) + ((z1.x + z1.y) + (z1.z + z1.w));
                    float sc = 1.0f;
                    if (Z >= kMinTotal) {
                        const int e = exponent_of(Z);
                        Ia += e;
                        sc = pow2i(-e);
                    } else {
                        bad = true;    // also NaN
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float dx = ev.x - mu[q], dyp = __builtin_fmaf(-eta[q], ev.z, ev.y);
                        const float E = ex2(__builtin_fmaf(-(dx * dx), r2[q], __builtin_fmaf(-(dyp * dyp), lq[q], k0[q])));
                        ah[q] = E * __builtin_fmaf(W1[q], in1[q], __builtin_fmaf(T0[q], ah[q], in2[q])) * sc;
                    }
                }
                // The row is read again only by the backward sweep, a launch and 6.9 GB later: written through (sc1) and marked
                // streaming (nt) it does not wait behind L2 write-back -- 3.41 -> 3.24 ms for the two sweeps, any of nt / sc1 /
                // sc0 sc1 gets most of it (profiles/r04_fb_isa_budget.md).  Cells 2p and 2p + 1 of a thread are 512 states apart:
                // one 64-bit address and an immediate offset per pair, as the compiler had it.
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    store_row<0>(&rowp[jj[2 * p]], ah[2 * p]);
                    store_row<2048>(&rowp[jj[2 * p]], ah[2 * p + 1]);
                }
                rowp += kStates;
                if (tau == 0) P.ws_exp[e0 + i] = Ia;
                ref_sum += (double)(kmax - ev.w);
            }
            __syncthreads();
        }
        // log_pr_data = log sum_j alpha[n-1][j]  (Forward_Backward.hpp:129-134)
        float s = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += ah[q];
        s = wave_sum(s);
        if (lane == 0) sRed[8 + wave] = s;
        __syncthreads();
        float Z = 0;
#pragma unroll
        for (int q = 0; q < kThreads / 64; ++q) Z += sRed[8 + q];
        if (tau == 0) {
            if (!(Z >= kMinTotal)) bad = true;
            const double l2 = ref_sum + (double)Ia + (double)lg2(Z) - (double)(P.log_n_states * kLog2e);
            P.ws_zfin[w] = Z;
            P.ws_lpd2[w] = (float)l2;
            P.out_log_pr_data[w] = (float)(l2 * 0.69314718055994530942);
            if (bad) flag_window(P, w);
        }
        } while (0);
    }
}

// ================================================ backward + statistics ================================================
// Statistics as in fwbw_backward_kernel (block sums over the scaled constants, mapped back by sCoef), with
// ptilde = ahat_{i-1} btilde in place of p: the per-event factor kappa = 2^(Ia_{i-1} + Ib_i - Ia_{n-1}) / Z is
// applied once to the six block sums (publish) and once to the per-thread transition partial sums.
__global__ __launch_bounds__(kThreads, 4) void fwbw_backward_scaled_kernel(FwbwArgs P)
{
    __shared__ __attribute__((aligned(16))) float sTab[3][kStates];   // T0b | W1b | k0 - K0max
    // Successor-group sums, laid out for their READERS: thread tau needs the eight consecutive groups 8 tau .. 8 tau + 7
    // (mod 1024 / 256).  As one array that is a 32-byte lane stride -- a 4-way bank conflict on every read, half of the
    // kernel's LDS cycles (SQ_LDS_BANK_CONFLICT, profiles/r02_rocprof_fwbw_scaled.txt).  So members 0-3 and 4-7 of every
    // octet live in two planes of float4: a reader takes two ds_read_b128 at a 16-byte lane stride, and the plane pitch
    // (544 = 512 + 32, 144 = 128 + 16 words) puts the two planes of the writers' b64 / b32 stores on disjoint banks.
    __shared__ __attribute__((aligned(16))) float sG1[2][kG1Pitch + 512];
    __shared__ __attribute__((aligned(16))) float sG2[2][kG2Pitch + 128];
    __shared__ __attribute__((aligned(16))) float sZ[2][kThreads / 64];
    __shared__ float sRed[16];
    __shared__ __attribute__((aligned(16))) float sAcc[2][kThreads / 64][8];   // per-event sums of each wave (+ kappa), by event parity
    __shared__ float sCoef[6][4];
    __shared__ unsigned sWork;

    const unsigned tau = threadIdx.x;
    const unsigned t = tau >> 1;
    const unsigned wave = tau >> 6, lane = tau & 63u;
    const unsigned j0 = tau * 8u;

    for (;;) {
        // The barrier comes BEFORE thread 0's fetch: it ends the previous window, and it keeps that window's closing
        // `if (tau == 0)` apart from this one.  Back to back (only the loop edge between them) LLVM threads the two
        // tests together and sends every lane but thread 0 straight to the next barrier -- the waves then pass it
        // without a new sWork and spin on the old window forever (seen on gfx950 with ROCm 7.2).
        __syncthreads();
        if (tau == 0) sWork = atomicAdd(P.queue, 1u);
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
        if (P.fb_flag[w]) break;    // the forward sweep already handed this window to the log-space kernels
        const int ms = P.scaled_slot ? P.scaled_slot[w] : 0;
        const int ts = P.trans_slot ? P.trans_slot[w] : 0;
        const float* __restrict__ M = P.models + (size_t)ms * kModelFloats;
        const float* __restrict__ C = P.trans_fb + (size_t)ts * kFbTransFloats;
        const float* __restrict__ ex = P.cmean + e0;
        const float* __restrict__ ey = P.stdv + e0;
        const int32_t* __restrict__ iexp = P.ws_exp + e0;
        const float* rowp = P.ws_alpha + (e0 + (uint64_t)(n - 1)) * kStates;   // uniform row pointer + 32-bit thread offset

        float mu[8], r2[8], eta[8], lq[8], bh[8];
        float kmax = kNegBig;
        float k0[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned j = j0 + u;
            const StateK s = make_state(M, j, P.log_2pi);
            mu[u] = s.mu; r2[u] = s.r2; eta[u] = s.eta; lq[u] = s.lq; k0[u] = s.k0;
            kmax = __builtin_fmaxf(kmax, s.k0);
            bh[u] = 1.0f;                          // Forward_Backward.hpp:93-103
        }
        kmax = block_max(kmax, sRed, wave, lane);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned j = j0 + u;
            const unsigned o = tab_off(tau, (unsigned)u >> 1) + ((unsigned)u & 1u);
            sTab[0][o] = ex2(C[3 * kStates + j] * kLog2e);
            sTab[1][o] = ex2(C[4 * kStates + j] * kLog2e);
            sTab[2][o] = k0[u] - kmax;                    // (each thread reads back only what it wrote)
        }
        // c2b[j] = w2[j & 255] for every state: the producer of successor group q = tau >> 1 applies it
        const float W2 = ex2(P.trans[(size_t)ts * kTransFloats + kStates + 1024 + t] * kLog2e);
        const unsigned train = P.train_mask[tau];   // bit u: state j0+u is a transition-training k-mer
        unsigned tmask[8];                          // all-ones / zero per state (bfe_i32: sign-extended bit u)
#pragma unroll
        for (int u = 0; u < 8; ++u) tmask[u] = (unsigned)__builtin_amdgcn_sbfe((int)train, u, 1);
        float p_stay = 0.0f, p_step4 = 0.0f;
        if (P.st_params) {
            p_stay = P.st_params[2 * w];                                         // Parameter_Trainer.hpp:444
            p_step4 = (1.0f - p_stay - P.st_params[2 * w + 1]) * 0.25f;          // :445
        }
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
        const int Iaf = iexp[n - 1];
        // 1 / Z as an exact power of two times a factor in (1/2, 1]: Z may be as small as 2^-100 (below that the forward sweep has
        // flagged the window), and kappa = 2^(Ia + Ib - Ia_final) / Z must not be formed as a product of two factors of up to 2^100
        // each -- round 5 did, and a window with an abasic stretch came back with infinite sums (tools/fb_sweep.py, round 6)
        const float Zfin = P.ws_zfin[w];
        const int ez = exponent_of(Zfin);
        const float rZm = pow2i(ez) / Zfin;
        const float rZ = 1.0f / Zfin;
        int Ib = 0;
        bool bad = false;
        float acc_p = 0, acc_stay = 0, acc_p01 = 0;

        float4 nx_lo = *reinterpret_cast<const float4*>(rowp + j0);
        float4 nx_hi = *reinterpret_cast<const float4*>(rowp + j0 + 4);
        float nx_x = ex[n - 1], nx_y = ey[n - 1];
        int nx_ia = Iaf;
        float ps[6];
        auto pm_add = [&](int u, float p) {
            const float t0 = p * r2[u], l0 = p * lq[u];
            const float t1 = t0 * mu[u], l1 = l0 * eta[u];
            ps[0] += t0; ps[1] += t1; ps[2] = __builtin_fmaf(t1, mu[u], ps[2]);
            ps[5] += l0; ps[4] += l1; ps[3] = __builtin_fmaf(l1, eta[u], ps[3]);
        };
        // the six sums of an event wait in `ps` until the next producer phase, where they share one hand-scheduled
        // 7-way wave reduction with that phase's column total; lane 63 then files them under their event's parity
        unsigned pend_ei = n - 1;
        float pend_kappa = rZ;
        auto publish = [&](unsigned ei, unsigned th) {
            if (th < 6 && P.out_pm_sums) {
                const unsigned i_b = th == 2 ? 1u : 0u;    // second term: S1 for s2, S0 for s1; third term: S0
                float va = 0.0f, vb = 0.0f, vc = 0.0f;
#pragma unroll
                for (int wv = 0; wv < kThreads / 64; ++wv) {
                    va += sAcc[ei & 1u][wv][th]; vb += sAcc[ei & 1u][wv][i_b]; vc += sAcc[ei & 1u][wv][0];
                }
                const float kappa = sAcc[ei & 1u][0][7];
                P.out_pm_sums[(e0 + (uint64_t)ei) * 6 + th] =
                    kappa * __builtin_fmaf(sCoef[th][0], va, __builtin_fmaf(sCoef[th][1], vb, sCoef[th][2] * vc));
            }
        };

        {   // event n-1: btilde = 1, kappa = 1 / Z, no following event
            const float al[8] = {nx_lo.x, nx_lo.y, nx_lo.z, nx_lo.w, nx_hi.x, nx_hi.y, nx_hi.z, nx_hi.w};
#pragma unroll
            for (int q = 0; q < 6; ++q) ps[q] = 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) pm_add(u, al[u]);
        }
        for (int i = (int)n - 1; i >= 1; --i) {
            // on entry: bh = bhat_i (scale exponent Ib); nx_x/y = event i.  Leaves bhat_{i-1} and the statistics of event i-1.
            // Every LDS / row address of the loop is re-derived from this copy of the thread index, which the compiler
            // cannot hoist: a handful of integer ops per event instead of ~10 loop-invariant registers that it
            // otherwise spills and reloads (each reload also waits for the alpha-row prefetch in flight).
            unsigned tl = tau;
            asm volatile("" : "+v"(tl));
            // producer phase at raised priority, back to 0 at the barrier: the waves of a block reach their barrier sooner when the
            // co-resident block's consumers do not take their issue slots (backward sweep 1.86 -> 1.80 ms, profiles/r06_fb_backward_session.md)
            __builtin_amdgcn_s_setprio(2);
            const unsigned buf = (unsigned)i & 1u;
            const float x = nx_x, sry = __builtin_amdgcn_rsqf(nx_y), ysry = nx_y * sry;
            rowp -= kStates;
            nx_lo = *reinterpret_cast<const float4*>(rowp + tl * 8u);
            nx_hi = *reinterpret_cast<const float4*>(rowp + tl * 8u + 4);
            // wave-uniform addresses in memory no kernel writes while this one runs: read through the scalar cache, so that the
            // vector-memory counter tracks the alpha rows alone (a vector load issued after them would make its first use wait
            // for the rows as well)
            nx_x = uniform_load(ex + (i - 1)); nx_y = uniform_load(ey + (i - 1)); nx_ia = uniform_load(iexp + (i - 1));
            // g = emission(event i) * beta_i; H1/H2 sums over consecutive successor groups (Forward_Backward.hpp:107-125)
            float g[8];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const f2 k02 = *reinterpret_cast<const f2*>(&sTab[2][tab_off(tl, pr)]);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int q = 2 * pr + v;
                    {   // k0 - xs folded into the two FMAs of xs: one VALU op less per cell
                        const float dx = x - mu[q], dyp = __builtin_fmaf(-eta[q], sry, ysry);
                        g[q] = ex2(__builtin_fmaf(-(dx * dx), r2[q], __builtin_fmaf(-(dyp * dyp), lq[q], k02[v]))) * bh[q];
                    }
                }
            }
            const float a = (g[0] + g[1]) + (g[2] + g[3]);
            const float b = (g[4] + g[5]) + (g[6] + g[7]);
            const float s8 = a + b;
            const float s16 = s8 + swap1(s8);     // (DPP: both lanes of the pair must be active -- keep it outside the branch)
            // groups 2 tl, 2 tl + 1 = octet tl >> 2, members 2 (tl & 3) + {0, 1}; group tl >> 1 = octet tl >> 4, member (tl >> 1) & 7
            *reinterpret_cast<f2*>(&sG1[buf][((tl >> 1) & 1u) * kG1Pitch + 4u * (tl >> 2) + 2u * (tl & 1u)]) = f2{a, b};
            if ((tl & 1u) == 0) sG2[buf][((tl >> 3) & 1u) * kG2Pitch + 4u * (tl >> 4) + ((tl >> 1) & 3u)] = s16 * W2;
            // the six sums of the previous phase and this phase's column total: rows { ps0..ps3 } and { ps4, ps5, z, - }
            float q0 = 0.0f, q1 = 0.0f;
            wave_sum7_rows(ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], s8, q0, q1);
            if ((tl & 15u) == 0u) {      // lane 16 r files row r: sAcc[..][r] and [4 + r] (kappa takes the empty row's place)
                const unsigned r = (tl >> 4) & 3u;
                float* dst = &sAcc[pend_ei & 1u][tl >> 6][0];
                dst[r] = q0;
                dst[4u + r] = r == 3u ? pend_kappa : q1;
                if (r == 2u) sZ[buf][tl >> 6] = q1;
            }
            asm volatile("" :: "s"(nx_ia));     // the scalar load must have landed by here (hipcc otherwise sinks it to its use
                                                // right after the barrier, where its latency is exposed)
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            publish((unsigned)i, tl);    // the barrier made every wave's sums of event i visible
            const float4 z0 = *reinterpret_cast<const float4*>(&sZ[buf][0]);
            const float4 z1 = *reinterpret_cast<const float4*>(&sZ[buf][4]);
            const float Zg = ((z0.x + z0.y) + (z0.z + z0.w)) + ((z1.x + z1.y) + (z1.z + z1.w));
            // statistics of event i-1 are in units of (ahat_{i-1} g): kappa = 2^(Ia_{i-1} + Ib_i - Ia_{n-1}) / Z
            const int kx = nx_ia + Ib - Iaf - ez;
            float sc = 1.0f, kappa = 0.0f;
            if (Zg >= kMinTotal && kx >= -kMaxKappaExp && kx <= kMaxKappaExp) {
                const int eg = exponent_of(Zg);
                sc = pow2i(-eg);
                Ib += eg;
                kappa = pow2i(kx) * rZm;
            } else {
                bad = true;
            }
            const float al[8] = {nx_lo.x, nx_lo.y, nx_lo.z, nx_lo.w, nx_hi.x, nx_hi.y, nx_hi.z, nx_hi.w};
            const float* ph1 = &sG1[buf][4u * (tl & 127u)];
            const float* ph2 = &sG2[buf][4u * (tl & 31u)];
            const float4 h1a = *reinterpret_cast<const float4*>(ph1), h1b = *reinterpret_cast<const float4*>(ph1 + kG1Pitch);
            const float4 h2a = *reinterpret_cast<const float4*>(ph2), h2b = *reinterpret_cast<const float4*>(ph2 + kG2Pitch);
            const float h1s[8] = {h1a.x, h1a.y, h1a.z, h1a.w, h1b.x, h1b.y, h1b.z, h1b.w};
            const float h2s[8] = {h2a.x, h2a.y, h2a.z, h2a.w, h2b.x, h2b.y, h2b.z, h2b.w};
#pragma unroll
            for (int q = 0; q < 6; ++q) ps[q] = 0.0f;
            float part_p = 0, part_stay = 0, part_p01 = 0;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const f2 t02 = *reinterpret_cast<const f2*>(&sTab[0][tab_off(tl, pr)]);
                const f2 w12 = *reinterpret_cast<const f2*>(&sTab[1][tab_off(tl, pr)]);
                const f2 h1 = f2{h1s[2 * pr], h1s[2 * pr + 1]};
                const f2 h2 = f2{h2s[2 * pr], h2s[2 * pr + 1]};
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int u = 2 * pr + v;
                    const float bt = __builtin_fmaf(w12[v], h1[v], __builtin_fmaf(t02[v], g[u], h2[v]));
                    const float p = al[u] * bt;                              // posterior of (i-1, u) up to kappa
                    pm_add(u, p);
                    // Parameter_Trainer.hpp:470-512 for the pair (i-1, i), same units.  Every term carries the factor
                    // alpha_{i-1}[u] >= 0, and min(a x, a y) = a min(x, y): the clamps (:480-488, :502-510) are taken on
                    // the beta side and alpha multiplies the three contributions once (as the FMAs of the partial sums).
                    const float mbt = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, bt) & tmask[u]);   // bt on training k-mers, else +0
                    const float st_b = __builtin_fminf(g[u] * p_stay, mbt);
                    const float p01_b = __builtin_fminf(__builtin_fmaf(h1[v], p_step4, st_b), mbt);
                    part_p = __builtin_fmaf(al[u], mbt, part_p);
                    part_stay = __builtin_fmaf(al[u], st_b, part_stay);
                    part_p01 = __builtin_fmaf(al[u], p01_b, part_p01);      // skip share = p - p01 (:502-510), taken once per window
                    bh[u] = bt * sc;
                }
            }
            acc_p = __builtin_fmaf(kappa, part_p, acc_p);
            acc_stay = __builtin_fmaf(kappa, part_stay, acc_stay);
            acc_p01 = __builtin_fmaf(kappa, part_p01, acc_p01);
            pend_ei = (unsigned)(i - 1);
            pend_kappa = kappa;
        }
        {
            float q0, q1;
            wave_sum7_rows(ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], 0.0f, q0, q1);
            if ((lane & 15u) == 0u) {
                const unsigned r = lane >> 4;
                float* dst = &sAcc[pend_ei & 1u][wave][0];
                dst[r] = q0;
                dst[4u + r] = r == 3u ? pend_kappa : q1;
            }
        }
        __syncthreads();
        publish(0u, tau);
        // window totals of the transition statistics
        // every term of p - p01 is >= 0 (p01 is clamped to p cell by cell); as a difference of two window sums it can come
        // out a rounding error below zero when nothing skips, hence the max
        float acc_skip = __builtin_fmaxf(acc_p - acc_p01, 0.0f);
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
        if (tau == 0) {
            if (P.out_st_sums) {
                float tsk = 0;
                for (int q = 0; q < kThreads / 64; ++q) tsk += sRed[q];
                P.out_st_sums[3 * w + 0] = lg2(tp) * kLn2;
                P.out_st_sums[3 * w + 1] = lg2(tst) * kLn2;
                P.out_st_sums[3 * w + 2] = lg2(tsk) * kLn2;
            }
            if (bad) flag_window(P, w);
        }
        } while (0);
    }
}

void launch_fwbw_scaled(const FwbwArgs& a, int grid, hipStream_t stream)
{
    // two launches on one stream: the backward sweep reads the rows, exponents and totals of the forward one
    FwbwArgs f = a, b = a;
    f.queue = a.queue; b.queue = a.queue + 1;
    hipLaunchKernelGGL(fwbw_forward_scaled_kernel, dim3(grid), dim3(kThreads), 0, stream, f);
    hipLaunchKernelGGL(fwbw_backward_scaled_kernel, dim3(grid), dim3(kThreads), 0, stream, b);
}

int fwbw_scaled_blocks_per_cu()
{
    const int f = fb_blocks_per_cu(reinterpret_cast<const void*>(fwbw_forward_scaled_kernel));
    const int b = fb_blocks_per_cu(reinterpret_cast<const void*>(fwbw_backward_scaled_kernel));
    return f < b ? f : b;
}

}  // namespace nchmm
