// viterbi_ll_kernel.hip -- the low-latency instantiation of the Viterbi sweep (Viterbi::fill + fill_state_seq,
// src/nanocall/Viterbi.hpp:44-99,120-142), gfx950: ONE read on a whole CU.
//
// Why a second form.  A read is sequential: viterbi_kernel.hip puts a read on 8 waves (8 states per thread) and two such blocks
// on a CU, which is how a CU decodes the most events per second -- but every read then advances at 1.5-1.6 us per event with a
// neighbour on its CU and at 1.12-1.18 without, whatever else the GPU has to do.  A launch lasts as long as its longest read, a batch with fewer reads than block slots leaves
// issue slots empty, and the reference's own call shape is one strand per call (nanocall.cpp:687-689, default -t 1, :93).  Here
// the same column is spread over 16 waves, 4 states per thread, one block per CU: about half the time per event for one read, at
// ~8 % fewer events per second and CU when every CU has two reads to work on.  Which form a launch takes is decided per launch by
// the host (nchmm_plan.hpp: choose_sweep).
//
// Same arithmetic, other ownership map:
//   * 1024 threads.  Thread tau = 4t + y owns the 4 states j = t + 256*(4x + y), x = 0..3: exactly one step group (the four
//     states that share their low 10 bits (y<<8)|t, Kmer.hpp:128-142 inverted) and a quarter of the skip group of low 8 bits t,
//     whose other quarters sit in the other three lanes of the quad: v_max_f32 / v_min_u32 with a DPP operand merge them.
//   * every per-state constant lives in VGPRs (40 per thread); LDS carries only the group winners.  Producer tau's step-group
//     winner is stored at index tau (+ a skew of 8 entries per 256 so that the readers' 16-byte lane groups fall on distinct
//     banks): the four winners a thread needs for its four cells are then CONSECUTIVE entries -- two ds_read_b128 -- and the
//     skip-group winners, stored x-minor, likewise.  5 LDS instructions per wave and event against 25 in the 8-wave form.
//   * the event's four values are broadcast from LDS into VGPRs (no v_readfirstlane: registers are plentiful here).
//   * back-pointers: one byte per state, state j at byte (t<<4) | (y<<2) | x of its row (BpQuad): a thread stores its four as
//     one dword, a wave 256 contiguous bytes; the traceback reads the same 16-byte groups as in the 8-wave form.
//   * group scans on raw alpha with the next-float probe, the exact rescan, ties to the lowest predecessor index, Markstein
//     quotients inside the validated range, half-rate / full-rate pairs: all as in viterbi_kernel.hip (viterbi_common.hpp).
//   * traceback by the same block, 256 four-lane segments per round (20 480 events).
//
// Float contract: -ffp-contract=off, denormals on, no device log/exp.
#include "nchmm_device.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace nchmm {

namespace {

#include "viterbi_common.hpp"

constexpr unsigned kLlChunk = 1024;                 // events staged in LDS at a time: one per thread
constexpr unsigned kV1Pitch = 1024 + 24;            // step-group winners, one per thread, skewed by 8 entries per 256
constexpr unsigned kV2Pitch = 256 + 24;             // skip-group winners, x-minor, skewed by 8 entries per 64

// v_cndmask / v_max3 / v_lshl_or with a full-rate rider, as in viterbi_common.hpp, with the event's value in a VGPR
__device__ __forceinline__ float max3_sub_v(float a, float b, float c, float p, float q, float& r)
{
    float m;
    asm("v_max3_f32 %0, %2, %3, %4\n\tv_sub_f32 %1, %5, %6" : "=&v"(m), "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(p), "v"(q));
    return m;
}
__device__ __forceinline__ unsigned selm_subrev_v(mask_t mk, unsigned if_set, unsigned if_clear, float p, float q, float& r)
{
    unsigned d;
    asm("v_cndmask_b32_e64 %0, %2, %3, %4\n\tv_subrev_f32 %1, %5, %6" : "=&v"(d), "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mk), "v"(p), "v"(q));
    return d;
}

// maximum / minimum over the four lanes of a quad, every lane gets the result: DPP as an operand modifier -- max(lane ^ 1, own),
// then max(lane ^ 2, own) -- two instructions where swap + compare + select would be six.  A DPP read needs two wait states
// after the VALU write of its source (the value comes out of the statements before, and out of the first step).
__device__ __forceinline__ float quad_max(float v)
{
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ unsigned quad_min(unsigned v)
{
    unsigned r;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r) : "v"(v));
    return r;
}

struct StateLL {
    // index x: state j = t + 256*(4x + y)
    float mu[4], sg[4], rsg[4], eta[4], reta[4], lam[4], nls[4], cc[4], w0[4], alpha[4];
    float w1;      // step group r = (y<<8)|t
    float w2;      // skip group q = t
    unsigned n_rescan, n_tie;
};

__device__ __forceinline__ unsigned v1_index(unsigned producer_tau) { return producer_tau + 8u * (producer_tau >> 8); }
__device__ __forceinline__ unsigned v2_index(unsigned q) { return 4u * (q & 63u) + 8u * ((q & 63u) >> 4) + (q >> 6); }

// where a thread writes and reads the exchange arrays (buffer 0; buffer 1 is a compile-time offset away)
struct ExchLL {
    ValSlot* w1;          // own step-group winner
    ValSlot* w2;          // the quad's skip-group winner (lane y == 0 writes)
    const ValSlot* r1;    // four consecutive step-group winners, cells x = 0..3
    const ValSlot* r2;    // four consecutive skip-group winners
};

// ---------------- group scans over the previous column, winners into the exchange arrays, the barrier ----------------
template <int PAR>
__device__ __forceinline__ void scan_and_publish(StateLL& S, const ExchLL& X, unsigned yy)
{
    const float NEG_INF = -__builtin_inff();
    // the step group is this thread's own four states: raw maximum, strict >, ascending x => first maximum; the winner is
    // carried as its back-pointer code 1 + x straight away
    // (maximum first -- v_max3 + v_max -- then the FIRST member that equals it: the same winner as the ascending strict-> scan)
    const float m4 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(S.alpha[0], S.alpha[1]), S.alpha[2]), S.alpha[3]);
    const unsigned x4 = selm(ballot(S.alpha[0] == m4), 1u, selm(ballot(S.alpha[1] == m4), 2u, selm(ballot(S.alpha[2] == m4), 3u, 4u)));
    // skip group k = 4x + y: the own quarter's winner is the step group's; the other quarters sit in the other three lanes of the
    // quad.  Value: the quad's maximum.  Winner: the lowest code 5 + 4x + y among the lanes that hold the maximum -- lower code
    // = lower k, which is the reference's rule for equal values (ascending strict >, Viterbi.hpp:84)
    const float m16 = quad_max(m4);
    const unsigned k16 = quad_min(selm(ballot(m4 == m16), 4u * x4 + (yy + 1u), 255u));     // 5 + 4x + y with x = x4 - 1
    float s1 = S.w1 + m4, s2 = S.w2 + m16;
    unsigned sl1 = x4, sl2 = k16;
    // Is any smaller alpha rounded to the same sum?  probe the next float below the maximum (viterbi_kernel.hip)
    {
        const float c = 0x1.8p-24f;
        const float p0 = __builtin_fmaf(m4, c, m4), p2 = __builtin_fmaf(m16, c, m16);
        const mask_t unsafe = ballot(S.w1 + p0 >= s1) | ballot(S.w2 + p2 >= s2);
        if (__builtin_expect(unsafe != 0, 0)) {
            ++S.n_rescan;
            // exact scan on the sums themselves (Viterbi.hpp:79-89 restricted to one class)
            float bv = NEG_INF; unsigned bx = 0;
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                const float v = S.w1 + S.alpha[xx];
                const mask_t m = ballot(v > bv);
                bv = selm(m, v, bv);
                bx = selm(m, (unsigned)xx, bx);
            }
            s1 = bv; sl1 = 1u + bx;
            bv = NEG_INF; unsigned bk = yy;
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {   // ascending x == ascending k for this thread
                const float v = S.w2 + S.alpha[xx];
                const mask_t m = ballot(v > bv);
                bv = selm(m, v, bv);
                bk = selm(m, 4u * (unsigned)xx + yy, bk);
            }
            merge_lower(bv, bk, swap1(bv), swap1(bk));
            merge_lower(bv, bk, swap2(bv), swap2(bk));
            s2 = bv; sl2 = 5u + bk;
        }
    }
    X.w1[PAR * kV1Pitch] = ValSlot{s1, sl1};
    if (yy == 0) X.w2[PAR * kV2Pitch] = ValSlot{s2, sl2};
    __syncthreads();

}

// the four step-group and the four skip-group winners of a thread's cells x = 0..3: two 16-byte reads per exchange array
// (cell x takes the step group j >> 2 = (t>>2) + 64k -- produced by thread 4((t>>2) + 64y) + x -- and the skip group
// j >> 4 = (t>>4) + 16k: for x = 0..3 consecutive entries of either array)
template <int PAR>
__device__ __forceinline__ void read_exchange(const ExchLL& X, float av_[4], unsigned as_[4], float bv_[4], unsigned bs_[4])
{
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v a01 = *reinterpret_cast<const u4v*>(X.r1 + PAR * kV1Pitch), a23 = *reinterpret_cast<const u4v*>(X.r1 + PAR * kV1Pitch + 2);
    const u4v b01 = *reinterpret_cast<const u4v*>(X.r2 + PAR * kV2Pitch), b23 = *reinterpret_cast<const u4v*>(X.r2 + PAR * kV2Pitch + 2);
    // (element by element into scalars first: __builtin_bit_cast applied to a vector-element expression reads element 0)
    const unsigned a0v = a01.x, a1v = a01.z, a2v = a23.x, a3v = a23.z, b0v = b01.x, b1v = b01.z, b2v = b23.x, b3v = b23.z;
    av_[0] = __builtin_bit_cast(float, a0v); av_[1] = __builtin_bit_cast(float, a1v); av_[2] = __builtin_bit_cast(float, a2v); av_[3] = __builtin_bit_cast(float, a3v);
    as_[0] = a01.y; as_[1] = a01.w; as_[2] = a23.y; as_[3] = a23.w;
    bv_[0] = __builtin_bit_cast(float, b0v); bv_[1] = __builtin_bit_cast(float, b1v); bv_[2] = __builtin_bit_cast(float, b2v); bv_[3] = __builtin_bit_cast(float, b3v);
    bs_[0] = b01.y; bs_[1] = b01.w; bs_[2] = b23.y; bs_[3] = b23.w;
}

// exact rule of the 3-way combine when two class values are equal: first maximum in ascending predecessor order (strict >, NaN
// never wins), Viterbi.hpp:79-89
__device__ __forceinline__ void combine_exact(unsigned t, unsigned k, float s0, float av, unsigned as, float bv, unsigned bs, float& best, unsigned& slot)
{
    const unsigned j = t + 256u * k;
    const unsigned p1 = ((as - 1u) << 10) | ((t >> 2) + 64u * k);
    const unsigned p2 = ((bs - 5u) << 8) | ((t >> 4) + 16u * k);
    float bb = -__builtin_inff(); unsigned bp = (unsigned)kStates, sl = 255u;
    if (s0 > bb) { bb = s0; bp = j; sl = 0; }
    if (av > bb || (av == bb && p1 < bp)) { bb = av; bp = p1; sl = as; }
    if (bv > bb || (bv == bb && p2 < bp)) { bb = bv; bp = p2; sl = bs; }
    best = bb; slot = sl;
}

template <bool FAST, int PAR>
__device__ __forceinline__ void column_ll(StateLL& S, const ExchLL& X, uint8_t* bp_row, unsigned tau, float x, float y,
                                          float ry, float ly3, float log_2pi)
{
    const unsigned t = tau >> 2, yy = tau & 3u;
    scan_and_publish<PAR>(S, X, yy);
    float av_[4], bv_[4]; unsigned as_[4], bs_[4];
    read_exchange<PAR>(X, av_, as_, bv_, bs_);
    unsigned bpw = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned k = 4u * (unsigned)i + yy;
        const float s0 = S.w0[i] + S.alpha[i];
        float best, da = 0.f, db = 0.f, qa = 0.f, qb = 0.f, u0 = 0.f, ef = 0.f;
        mask_t e0, e1, e2;
        unsigned slot;
        if constexpr (FAST) {
            // max3, the two selects (and the back-pointer shift-or below) each carry one operation of this cell's emission
            best = max3_sub_v(s0, av_[i], bv_[i], x, S.mu[i], da);                 // d  = x - mu
            e0 = ballot(s0 == best); e1 = ballot(av_[i] == best); e2 = ballot(bv_[i] == best);
            db = y - S.eta[i]; qa = da * S.rsg[i]; qb = db * S.reta[i];
            const unsigned tmp = selm_subrev_v(e1, as_[i], bs_[i], ly3, S.cc[i], u0);   // u0 = c - 3 log y
            slot = selz_fnma(e0, tmp, qa, S.sg[i], da, ef);                        // ef = fma(-q, sigma, d)
        } else {
            best = __builtin_fmaxf(__builtin_fmaxf(s0, av_[i]), bv_[i]);
            e0 = ballot(s0 == best); e1 = ballot(av_[i] == best); e2 = ballot(bv_[i] == best);
            slot = selm_zero(e0, selm(e1, as_[i], bs_[i]));
        }
        const mask_t tie = (e0 & e1) | ((e0 | e1) & e2);
        if (__builtin_expect(tie != 0, 0)) {
            ++S.n_tie;
            combine_exact(t, k, s0, av_[i], as_[i], bv_[i], bs_[i], best, slot);
        }
        if constexpr (FAST) {
            // the rest of emission<true>() (same operations, same order of roundings)
            const float av = __builtin_fmaf(ef, S.rsg[i], qa);                       // a = (x - mu) / sigma
            const float ep = __builtin_fmaf(-qb, S.eta[i], db);
            const float bv = __builtin_fmaf(ep, S.reta[i], qb);                     // b = (y - eta) / eta
            const float tt = log_2pi + av * av;
            const float lbb = S.lam[i] * bv * bv;
            const float q3 = lbb * ry;
            const float e3 = __builtin_fmaf(-q3, y, lbb);
            const float uu = u0 - __builtin_fmaf(e3, ry, q3);
            const float nn = __builtin_fmaf(-0.5f, tt, S.nls[i]);
            const float em = __builtin_fmaf(0.5f, uu, nn);
            if (i == 0) { S.alpha[i] = best + em; bpw = slot; }
            else if (i == 1) bpw = lshlor_add<8>(slot, bpw, best, em, S.alpha[i]);
            else if (i == 2) bpw = lshlor_add<16>(slot, bpw, best, em, S.alpha[i]);
            else bpw = lshlor_add<24>(slot, bpw, best, em, S.alpha[i]);
        } else {
            const float e = emission<FAST>(x, y, ry, ly3, log_2pi, S.mu[i], S.sg[i], S.rsg[i], S.nls[i], S.eta[i], S.reta[i], S.lam[i], S.cc[i]);
            S.alpha[i] = best + e;
            bpw |= slot << (8 * i);
        }
    }
    // streaming store (the row is read once, by the traceback): byte (t<<4) | (y<<2) | x = dword tau
    __builtin_nontemporal_store(bpw, reinterpret_cast<unsigned*>(bp_row) + tau);
}

// A column of a read whose emissions were computed ahead (emission_kernel.hip): the recurrence alone -- scans, exchange, 3-way
// combine -- and one add of the row's four values.  No model parameter is touched: 60 % of a column's arithmetic is gone from
// the read's critical path.
template <int PAR>
__device__ __forceinline__ void column_ahead(StateLL& S, const ExchLL& X, uint8_t* bp_row, unsigned tau, const float4 em4)
{
    const unsigned t = tau >> 2, yy = tau & 3u;
    scan_and_publish<PAR>(S, X, yy);
    float av_[4], bv_[4]; unsigned as_[4], bs_[4];
    read_exchange<PAR>(X, av_, as_, bv_, bs_);
    const float em[4] = {em4.x, em4.y, em4.z, em4.w};
    unsigned bpw = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned k = 4u * (unsigned)i + yy;
        const float s0 = S.w0[i] + S.alpha[i];
        float best = __builtin_fmaxf(__builtin_fmaxf(s0, av_[i]), bv_[i]);
        const mask_t e0 = ballot(s0 == best), e1 = ballot(av_[i] == best), e2 = ballot(bv_[i] == best);
        unsigned slot = selm_zero(e0, selm(e1, as_[i], bs_[i]));
        const mask_t tie = (e0 & e1) | ((e0 | e1) & e2);
        if (__builtin_expect(tie != 0, 0)) {
            ++S.n_tie;
            combine_exact(t, k, s0, av_[i], as_[i], bv_[i], bs_[i], best, slot);
        }
        if (i == 0) { S.alpha[i] = best + em[i]; bpw = slot; }
        else if (i == 1) bpw = lshlor_add<8>(slot, bpw, best, em[i], S.alpha[i]);
        else if (i == 2) bpw = lshlor_add<16>(slot, bpw, best, em[i], S.alpha[i]);
        else bpw = lshlor_add<24>(slot, bpw, best, em[i], S.alpha[i]);
    }
    __builtin_nontemporal_store(bpw, reinterpret_cast<unsigned*>(bp_row) + tau);
}

}  // namespace

__global__ __launch_bounds__(kLlThreads, 4) void viterbi_ll_kernel(ViterbiArgs P)
{
    __shared__ __attribute__((aligned(16))) ValSlot sV1[2][kV1Pitch];   // step-group winners
    __shared__ __attribute__((aligned(16))) ValSlot sV2[2][kV2Pitch];   // skip-group winners
    __shared__ __attribute__((aligned(16))) float4 sEv[kLlChunk];        // per event: x, y, 3*log y, 1/y
    __shared__ TbShared<kLlThreads> sTb;
    ValSlot* const sRed = &sV1[0][0];  // the final arg-max reduction reuses the exchange buffer
    static_assert(kV1Pitch >= (unsigned)kLlThreads, "the arg-max reduction needs one entry per thread");
    __shared__ unsigned sWork, sLast;

    const unsigned tau = threadIdx.x;
    const unsigned t = tau >> 2, yy = tau & 3u;
    unsigned long long t_fwd = 0, t_tb = 0, t_all0 = 0;
    if (P.prof) t_all0 = wall_clock64();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: wave[3:0] simd[5:4] cu[11:8] sh[12] se[15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    // ---- this block's back-pointer region (one block per CU: start looking at the CU's first slot) ----
    if (tau == 0) sWork = take_region(P, xcc, 2u * (((hw >> 13) & 7u) * 16u + ((hw >> 8) & 15u)));
    __syncthreads();
    const unsigned my_slot = sWork;
    if (my_slot == kNoSlot) {
        if (tau == 0) fail_without_region(P);
        return;
    }
    uint8_t* const ws = P.ws + (uint64_t)my_slot * P.slot_bytes;

    for (;;) {
        __syncthreads();   // ends the previous read (see viterbi_kernel.hip for why the barrier comes first)
        if (tau == 0) sWork = atomicAdd(P.queue, 1u) - P.queue_base;
        __syncthreads();
        const unsigned widx = sWork;
        if (widx >= P.n_reads) break;
        const unsigned r = __builtin_amdgcn_readfirstlane(P.order ? P.order[widx] : P.first_read + widx);
        const uint64_t e0 = P.off[r];
        const unsigned n = (unsigned)(P.off[r + 1] - e0);
        if (n == 0) {
            if (tau == 0) {
                P.out_logp[r] = __builtin_nanf("");
                if (P.out_status) P.out_status[r] = 0;
            }
            continue;
        }
        unsigned long long c0 = 0;
        if (P.prof) c0 = wall_clock64();
        const int ms = P.model_slot ? P.model_slot[r] : 0;
        const int ts = P.trans_slot ? P.trans_slot[r] : 0;
        const float* __restrict__ M = P.models + (size_t)ms * kModelFloats;
        const float* __restrict__ W = P.trans + (size_t)ts * kTransFloats;
        const bool model_fast = P.model_fast[ms] != 0;
        const float* __restrict__ ex = P.cmean + e0;
        const float* __restrict__ ey = P.stdv + e0;
        const float* __restrict__ el = P.lstdv + e0;

        StateLL S;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned j = t + 256u * (4u * (unsigned)i + yy);
            S.mu[i] = M[MF_MU * kStates + j];
            S.sg[i] = M[MF_SIGMA * kStates + j];
            S.rsg[i] = M[MF_RSIGMA * kStates + j];
            S.eta[i] = M[MF_ETA * kStates + j];
            S.reta[i] = M[MF_RETA * kStates + j];
            S.lam[i] = M[MF_LAMBDA * kStates + j];
            S.nls[i] = M[MF_NEG_LOG_SIGMA * kStates + j];
            S.cc[i] = M[MF_C * kStates + j];
            S.w0[i] = W[j];
        }
        S.w1 = W[kStates + (yy << 8) + t];
        S.w2 = W[kStates + 1024 + t];
        S.n_rescan = 0; S.n_tie = 0;
        // exchange addresses: producer tau stores at v1_index(tau); cell x reads the step group j >> 2 = (t>>2) + 64(4x + y),
        // produced by thread 4((t>>2) + 64y) + x, and the skip group j >> 4 = (t>>4) + 16(4x + y): consecutive entries for x = 0..3
        ExchLL X;
        X.w1 = &sV1[0][v1_index(tau)];
        X.w2 = &sV2[0][v2_index(t)];
        X.r1 = &sV1[0][v1_index(4u * ((t >> 2) + 64u * yy))];
        X.r2 = &sV2[0][v2_index((t >> 4) + 16u * yy)];

        // (rows stated by the caller are taken only if all of them lie inside the buffer -- the same test as emission_kernel's:
        // offsets that do not match the stated total cost the read its rows ahead, never an access outside the buffer)
        const uint64_t em_row = P.em ? (P.em_row0 ? P.em_row0[r] : e0) : kNoEmRow;
        if (em_row != kNoEmRow && em_row <= P.em_rows && n <= P.em_rows - em_row) {
            // ---- the read's emissions are in memory (emission_kernel.hip ran in front of this launch): rows of 1024 float4, this
            // thread's at [tau].  Eight rows in flight per thread (16 KiB rows from HBM / L2: ~2 us away, a column takes ~0.45) ----
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v* __restrict__ emp = reinterpret_cast<const f4v*>(P.em) + em_row * (uint64_t)(kStates / 4) + tau;
            auto row = [&](unsigned i) {       // (read once: streaming load; past the end: the last row again, unused)
                const f4v v = __builtin_nontemporal_load(emp + (uint64_t)(i < n ? i : n - 1) * (kStates / 4));
                return make_float4(v.x, v.y, v.z, v.w);
            };
            {
                const float4 e = row(0);                                   // column 0 (Viterbi.hpp:55-68)
                S.alpha[0] = e.x - P.log_n_states; S.alpha[1] = e.y - P.log_n_states; S.alpha[2] = e.z - P.log_n_states; S.alpha[3] = e.w - P.log_n_states;
            }
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = row(1u + (unsigned)u);
            unsigned i = 1;
            for (; i + 8 <= n; i += 8) {                                   // (i is odd: the exchange buffer's parity is the unroll index's)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u & 1) column_ahead<0>(S, X, ws + (uint64_t)(i + u) * kStates, tau, q[u]);
                    else column_ahead<1>(S, X, ws + (uint64_t)(i + u) * kStates, tau, q[u]);
                    q[u] = row(i + 8u + (unsigned)u);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {                                  // the last n - i < 8 columns (block-uniform bound)
                if (i + (unsigned)u >= n) break;
                if (u & 1) column_ahead<0>(S, X, ws + (uint64_t)(i + u) * kStates, tau, q[u]);
                else column_ahead<1>(S, X, ws + (uint64_t)(i + u) * kStates, tau, q[u]);
            }
            __syncthreads();   // the last column's exchange reads are done before sRed (= sV1[0]) is written below
        } else
        for (unsigned base = 0; base < n; base += kLlChunk) {
            // stage the next kLlChunk events: x, y, 3 log y, 1/y (one correctly rounded divide per event)
            const unsigned ie = base + tau;
            bool ok = true;
            if (ie < n) {
                const float x = ex[ie], y = ey[ie];
                sEv[tau] = make_float4(x, y, 3.0f * el[ie], 1.0f / y);
                ok = event_in_fast_range(x, y);
            }
            const bool fast = __syncthreads_and(ok) && model_fast;
            const unsigned hi = (n - base < kLlChunk) ? n - base : kLlChunk;
            unsigned lo = 0;
            if (base == 0) {
                // ---- column 0 (Viterbi.hpp:55-68) ----
                const float4 ev = sEv[0];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = fast ? emission<true>(ev.x, ev.y, ev.w, ev.z, P.log_2pi, S.mu[i], S.sg[i], S.rsg[i], S.nls[i],
                                                          S.eta[i], S.reta[i], S.lam[i], S.cc[i])
                                         : emission<false>(ev.x, ev.y, ev.w, ev.z, P.log_2pi, S.mu[i], S.sg[i], S.rsg[i], S.nls[i],
                                                           S.eta[i], S.reta[i], S.lam[i], S.cc[i]);
                    S.alpha[i] = e - P.log_n_states;
                }
                lo = 1;
            }
            // ---- columns (Viterbi.hpp:72-96) ----
            // (two columns per trip: the exchange buffer alternates with the event's parity, a compile-time offset this way)
            uint8_t* const rows = ws + (uint64_t)base * kStates;
            auto col = [&](auto fast_tag, auto par_tag, unsigned c) {
                const float4 ev = sEv[c];
                column_ll<decltype(fast_tag)::value, decltype(par_tag)::value>(S, X, rows + (uint64_t)c * kStates, tau, ev.x, ev.y, ev.w, ev.z, P.log_2pi);
            };
            auto sweep = [&](auto fast_tag) {
                unsigned c = lo;
                if (c < hi && (c & 1u)) { col(fast_tag, std::integral_constant<int, 1>{}, c); ++c; }
                for (; c + 1 < hi; c += 2) {
                    col(fast_tag, std::integral_constant<int, 0>{}, c);
                    col(fast_tag, std::integral_constant<int, 1>{}, c + 1);
                }
                if (c < hi) col(fast_tag, std::integral_constant<int, 0>{}, c);
            };
            if (fast) sweep(std::true_type{}); else sweep(std::false_type{});
            __syncthreads();   // sEv is rewritten by the next chunk
        }

        // ---- fill_state_seq: arg-max of the last column, lowest index on ties (Viterbi.hpp:125-133) ----
        {
            float bv = -__builtin_inff();
            unsigned bi = kStates;
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // ascending x == ascending j for this thread
                const bool g = S.alpha[i] > bv;
                bv = g ? S.alpha[i] : bv;
                bi = g ? t + 256u * (4u * (unsigned)i + yy) : bi;
            }
            sRed[tau] = ValSlot{bv, bi};
        }
        __syncthreads();   // also publishes every back-pointer store of this block (vmcnt(0) + barrier)
        unsigned long long c1 = 0;
        if (P.prof) c1 = wall_clock64();
        if (tau < 64) {
            const ValSlot m = reduce_last_column<kLlThreads>(sRed, tau);
            if (tau == 0) {
                P.out_logp[r] = m.v;                  // Viterbi::path_probability(), Viterbi.hpp:133
                sLast = m.s;                          // kStates when every state is -INF/NaN
            }
        }
        __syncthreads();
        traceback_block<kLlThreads, BpQuad>(P, sTb, ws, r, e0, (int)n, sLast);
        if (P.prof) {
            const unsigned long long c2 = wall_clock64();
            t_fwd += c1 - c0;
            t_tb += c2 - c1;
            if ((tau & 63u) == 0) {   // per wave: the branches are wave-uniform
                atomicAdd(&P.prof[6], (unsigned long long)S.n_rescan);
                atomicAdd(&P.prof[7], (unsigned long long)S.n_tie);
            }
        }
        // (the top-of-loop barrier ends the traceback: the region and the exchange buffer are free for the next read)
    }
    if (tau == 0 && P.slot_owner) {
        __threadfence();
        __hip_atomic_store(P.slot_owner + my_slot, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (P.prof && tau == 0) {
        atomicAdd(&P.prof[0], t_fwd);
        atomicAdd(&P.prof[1], t_tb);
        atomicAdd(&P.prof[2], wall_clock64() - t_all0);
        atomicAdd(&P.prof[3], 1ull);
        if (blockIdx.x < 2048) {
            P.prof[8 + 2 * blockIdx.x] = t_all0;
            P.prof[8 + 4096 + blockIdx.x] = ((unsigned long long)xcc << 32) | hw;
            P.prof[9 + 2 * blockIdx.x] = wall_clock64();
        }
    }
}

// (a.em != nullptr: the reads whose em_row0 is set take the recurrence-only columns)
void launch_viterbi_ll(const ViterbiArgs& a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(viterbi_ll_kernel, dim3(grid), dim3(kLlThreads), 0, stream, a);
}

}  // namespace nchmm
