// viterbi_kernel.hip -- per-read Viterbi decode over the 4096-state 6-mer pore HMM, gfx950.
//
// Replaces Viterbi::fill + fill_state_seq (src/nanocall/Viterbi.hpp:44-99,120-142) for a batch of
// reads.  One read per thread-block (persistent blocks pull reads from a work queue).
//
// Mapping (DESIGN.md "Viterbi kernel"):
//   * 512 threads = 8 waves, two per SIMD; two blocks per CU give 4 waves/SIMD (<= 128 VGPRs).
//     Thread tau = 2*t + h owns 8 of the 16 states whose LOW 8 bits (last four bases) are t:
//     j = t + 256*k with k = 4x + y (top four bits = first two bases), y in {h, h+2}, x in 0..3.
//     alpha[8] and the per-state parameters live in VGPRs for the whole read.
//   * The predecessors of j are  j,  (x<<10)|(j>>2) x=0..3  and  (xy<<8)|(j>>4) xy=0..15
//     (Kmer::neighbour_list inverted, Kmer.hpp:128-142).  All 16 skip-predecessors of a state
//     share their low 8 bits, all 4 step-predecessors their low 10 bits.  So the 21-way max of the
//     reference (Viterbi.hpp:79-89) becomes: per thread two complete 4-member step-group scans and
//     half of a 16-member skip-group scan (merged with the partner lane by one DPP swap), then one
//     3-way combine per state, the per-group winners going through LDS (one barrier per event).
//   * Weights factor as w0[j] / w1[r] / w2[q] (nchmm_api.cpp: factor_transitions), so w + alpha
//     are exactly the floats the reference forms.  The group scans run on RAW alpha (one add per
//     group instead of one per member); because RN(w + .) is monotone the winner is the same
//     unless a smaller alpha rounds to the same sum, which is ruled out per group by probing the
//     next float below the maximum -- otherwise the wave takes the exact sum-by-sum scan.  Ties
//     resolve to the lowest predecessor index as the reference's ascending strict-> scan does.
//   * Divisions: divisors are per-state constants (sigma, eta) or per-event constants (stdv): the
//     quotient comes from a correctly rounded reciprocal + two FMA residual corrections
//     (Markstein), bit-identical to IEEE division inside the validated operand range (checked per
//     model on upload, per 512-event chunk here), true division otherwise.
//   * The combine's half-rate instructions are paired by hand with full-rate ones of the emission (max3_sub_s,
//     selm_subrev_s, selz_fnma, lshlor_add below): the second issue pass of a select / max / shift-or is free for an
//     independent add / sub / mul of the same wave, and the compiler does not schedule for that.
//   * Selects are written as v_cndmask_b32_e64 with an SGPR-pair mask: on gfx950 the VOP2 form
//     reading a VCC that was not written by the immediately preceding VALU op issues ~8x slower
//     (tools/ubench/valu_rate.hip).
//   * Back-pointers are one byte per state (0 stay, 1+x step, 5+xy skip); row i of a read at
//     ws + i*4096, state j at byte (t<<4) | (h<<3) | (x<<1) | (y>>1): each thread stores its 8
//     bytes as one dwordx2 (a wave writes 512 B contiguous), and the 21 candidates of the next
//     traceback step sit in three 16-byte groups.
//   * Traceback is a second kernel (traceback_kernel, one wave per read, every read of the batch
//     at once): the chase is a dependent pointer walk, so it is latency-bound and wants many reads
//     in flight rather than CUs parked behind a barrier.  Each round trip fetches every 16-byte
//     group that can hold the byte of rows i, i-1, i-2 (27 lanes x 16 B) and resolves three events.
//
// Float contract: -ffp-contract=off (the only FMAs are the explicit residual corrections and the
// next-float probe), denormals on, no device log/exp: every log comes from the host libm.
#include "nchmm_device.h"

#include <cstdio>
#include <cstdlib>

#pragma clang fp contract(off)

namespace nchmm {

namespace {

typedef unsigned long long mask_t;
constexpr unsigned kChunk = 256;   // events staged in LDS at a time

struct __attribute__((aligned(8))) ValSlot {
    float v;
    unsigned s;   // back-pointer slot code of the group winner (1+x or 5+xy); the state index in sRed
};

// v_cndmask_b32_e64 dst, a, b, mask : mask bit set -> b, clear -> a
__device__ __forceinline__ float selm(mask_t m, float if_set, float if_clear)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
__device__ __forceinline__ unsigned selm(mask_t m, unsigned if_set, unsigned if_clear)
{
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
// mask bit set -> 0 (inline constant: no VGPR, no v_mov), clear -> if_clear
__device__ __forceinline__ unsigned selm_zero(mask_t m, unsigned if_clear)
{
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(r) : "v"(if_clear), "s"(m));
    return r;
}
__device__ __forceinline__ mask_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ---- half-rate op + independent full-rate "rider" in one asm statement ----
// A v_cndmask / v_max / v_lshl_or occupies a gfx950 SIMD's issue port for two passes (4 cycles per wave64); the pass it leaves
// idle takes an independent full-rate op (add / sub / mul) of the SAME wave at no cost, an FMA or the op after a v_max3 at about
// half, the op after a compare into an SGPR pair at full cost (tools/ubench/sstore_rate.hip, "SEQ" lines: `max add` 1.91 ns
// against 1.80 + 1.07).  hipcc's scheduler does not model this and moves the combine's selects away from the emission
// arithmetic, so the pairs that pay are written out: four per cell, -4.7 % on the forward sweep, bit-identical results
// (profiles/r03_viterbi_isa_budget.md section 2c).  The first result is early-clobber: it must not share a register with an
// operand of the second instruction.
__device__ __forceinline__ float max3_sub_s(float a, float b, float c, float p_sgpr, float q, float& r)
{
    float m;
    asm("v_max3_f32 %0, %2, %3, %4\n\tv_sub_f32 %1, %5, %6" : "=&v"(m), "=v"(r) : "v"(a), "v"(b), "v"(c), "s"(p_sgpr), "v"(q));
    return m;
}
// sel = mask ? if_set : if_clear ;  r = q - p_sgpr
__device__ __forceinline__ unsigned selm_subrev_s(mask_t mk, unsigned if_set, unsigned if_clear, float p_sgpr, float q, float& r)
{
    unsigned d;
    asm("v_cndmask_b32_e64 %0, %2, %3, %4\n\tv_subrev_f32 %1, %5, %6" : "=&v"(d), "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mk), "s"(p_sgpr), "v"(q));
    return d;
}
// sel = mask ? 0 : if_clear ;  r = fma(-a, b, c)
__device__ __forceinline__ unsigned selz_fnma(mask_t mk, unsigned if_clear, float a, float b, float c, float& r)
{
    unsigned d;
    asm("v_cndmask_b32_e64 %0, %2, 0, %3\n\tv_fma_f32 %1, -%4, %5, %6" : "=&v"(d), "=v"(r) : "v"(if_clear), "s"(mk), "v"(a), "v"(b), "v"(c));
    return d;
}
// w = (slot << SH) | w ;  r = p + q
template <int SH>
__device__ __forceinline__ unsigned lshlor_add(unsigned slot, unsigned w, float p, float q, float& r)
{
    unsigned d;
    asm("v_lshl_or_b32 %0, %2, %3, %4\n\tv_add_f32 %1, %5, %6" : "=&v"(d), "=v"(r) : "v"(slot), "n"(SH), "v"(w), "v"(p), "v"(q));
    return d;
}
// a wave-uniform float held in an SGPR instead of a VGPR
__device__ __forceinline__ float uniform(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// swap with the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2].  Written out with its own two wait states: a DPP read of
// a VGPR needs them after the VALU write, and the values swapped here come out of asm statements (selm), which the compiler's
// hazard recogniser does not look into -- the distance must not depend on what it happens to schedule in between.
__device__ __forceinline__ float swap1(float v)
{
    float r;
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ unsigned swap1(unsigned v)
{
    unsigned r;
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
    return r;
}

// byte of state j inside its 16-byte group: k = j >> 8 = 4x + y  ->  ((y&1)<<3) | (x<<1) | (y>>1)
__device__ __forceinline__ unsigned bp_pos(unsigned k) { return ((k & 1u) << 3) | ((k >> 2) << 1) | ((k >> 1) & 1u); }

// n / d with r = RN(1/d) precomputed: q0 = RN(n r), one exact residual (FMA), one correction (FMA) -- three
// VALU ops, bit-identical to the IEEE quotient.  Markstein's theorem gives this whenever q0 is a faithful
// rounding; that it holds for EVERY pair of binary32 significands (2^23 divisors x 2^23 numerators, including
// the all-ones divisor the textbook statement excludes) was settled by enumeration on the host, 0 mismatches:
// tools/ubench/markstein_exhaustive.c, profiles/r01_markstein_exhaustive.txt.
// Exponents do not enter as long as nothing leaves the normal range: needs n == 0 or 2^-100 <= |n| <= 2^100
// and d, r normal (the range validation below; outside it the true division is used).
template <bool FAST>
__device__ __forceinline__ float quot(float n, float d, float r)
{
    if constexpr (FAST) {
        const float q = n * r;
        const float e = __builtin_fmaf(-q, d, n);
        return __builtin_fmaf(e, r, q);
    } else {
        return n / d;
    }
}

// Pore_Model_State::log_pr_corrected_emission, Pore_Model.hpp:145-149 with log_normal_pdf :24-31
// and log_invgauss_pdf :33-40, operation for operation:
//   a = (x - mu) / sigma;            N  = -log_sigma - (log_2pi + a*a) / 2
//   b = (y - eta) / eta;             IG = (log_lambda - log_2pi - 3*log_y - lambda*b*b / y) / 2
// nls = -log_sigma (exact negation), c = log_lambda - log_2pi (first subtraction of the reference's
// left-to-right expression), ly3 = 3.0f * log_y.
template <bool FAST>
__device__ __forceinline__ float emission(float x, float y, float ry, float ly3, float log_2pi, float mu, float sg,
                                          float rsg, float nls, float eta, float reta, float lam, float c)
{
    const float a = quot<FAST>(x - mu, sg, rsg);
    const float b = quot<FAST>(y - eta, eta, reta);
    const float t = log_2pi + a * a;
    const float u = c - ly3 - quot<FAST>(lam * b * b, y, ry);
    if constexpr (FAST) {
        // Halving is exact: t >= log 2pi, and u / 2 is inexact only when |u| < 2^-125 (a subnormal quotient), where
        // the lost 2^-150 cannot move RN(n + u / 2) unless |n| is itself below 2^-100 -- two O(1) expressions
        // cancelling to that depth at once.  So the reference's  n = nls - t / 2,  ig = u / 2,  n + ig  are these two
        // FMAs bit for bit, two ops fewer.
        const float n = __builtin_fmaf(-0.5f, t, nls);
        return __builtin_fmaf(0.5f, u, n);
    } else {
        const float n = nls - t / 2.0f;
        const float ig = u / 2.0f;
        return n + ig;
    }
}

struct State {
    // index i = (x<<1) | (y>>1), state j = t + 256*(4x + y), y = 2*(i&1) + h
    float mu[8], sg[8], rsg[8], eta[8], reta[8], lam[8], alpha[8];
    float w1[2];   // step groups r = (y<<8)|t for y = h, h+2
    float w2;      // skip group q = t
    // wave-uniform tallies (SGPRs) of the two exactness branches, reported through P.prof[6..7]:
    // columns that took the sum-by-sum rescan, cells' 3-way combines that took the lowest-index rule
    unsigned n_rescan, n_tie;
};

// Three per-state tables that are touched once per cell (-log sigma, log lambda - log 2pi, stay weight)
// live in LDS, thread-major: thread tau's 8 floats of table f at sTab[f][tau*8 ..], the two 16-byte
// chunks XOR-swizzled by bit 3 of tau so the ds_read_b128 lane groups hit distinct banks.
__device__ __forceinline__ unsigned tab_off(unsigned tau, unsigned chunk)
{
    return tau * 8u + ((chunk ^ ((tau >> 3) & 1u)) << 2);
}

// (value, index) merge: take b if b.v > a.v, or equal and lower index
__device__ __forceinline__ void merge_lower(float& av, unsigned& ai, float bv, unsigned bi)
{
    // three compares into SGPR masks combined on the scalar unit (a short-circuit expression makes the compiler branch
    // and round-trip the mask through a VGPR)
    const mask_t m = ballot(bv > av) | (ballot(bv == av) & ballot(bi < ai));
    av = selm(m, bv, av);
    ai = selm(m, bi, ai);
}

template <bool FAST>
__device__ __forceinline__ void column(State& S, const float (*sTab)[kStates], ValSlot* sV1, ValSlot* sV2,
                                       uint8_t* bp_row, unsigned tau, float x, float y, float ry, float ly3,
                                       float log_2pi)
{
    const float NEG_INF = -__builtin_inff();
    const unsigned t = tau >> 1, h = tau & 1u;

    // ---------------- group scans over the previous column ----------------
    // raw maxima first (strict >, ascending index => first maximum), sums once per group
    // (the scans carry the winner as its back-pointer slot code straight away -- 1 + x for a step group, 5 + 4x + y for the skip
    // group: the constants ride in the selects and in the shift-add that forms the skip code, no separate adds)
    float m4[2]; unsigned x4[2];              // x4 = 1 + winning member
#pragma unroll
    for (int g = 0; g < 2; ++g) {            // y = 2g + h, members i = 2x + g
        // starting from member 0 instead of -INF saves one compare-select; the results differ only
        // if member 0 is NaN while another member is not, which needs a NaN emission for some
        // states but not others -- no finite model/event does that (and an all-NaN column is
        // reported as NCHMM_E_NUMERIC at the end)
        float bv = S.alpha[g]; unsigned bx = 1;
#pragma unroll
        for (int xx = 1; xx < 4; ++xx) {
            const float v = S.alpha[2 * xx + g];
            const mask_t m = ballot(v > bv);
            bv = selm(m, v, bv);
            bx = selm(m, (unsigned)xx + 1u, bx);
        }
        m4[g] = bv; x4[g] = bx;
    }
    // own half of the skip group: k = 4x + y.  Fast form: strict > decides; an exact tie between the halves (the lower index
    // would win) is rare and goes to the exact rescan below
    float m8 = m4[0]; unsigned k8 = 4u * x4[0] + (h + 1u);          // 5 + 4x + h with x = x4 - 1
    mask_t tie_halves = ballot(m4[1] == m8);
    {
        const mask_t g = ballot(m4[1] > m8);
        m8 = selm(g, m4[1], m8);
        k8 = selm(g, 4u * x4[1] + (h + 3u), k8);                     // 5 + 4x + (2 + h)
    }
    // partner half
    float m16 = m8; unsigned k16 = k8;
    {
        const float pm = swap1(m8); const unsigned pk = swap1(k8);
        tie_halves |= ballot(pm == m16);
        const mask_t g = ballot(pm > m16);
        m16 = selm(g, pm, m16);
        k16 = selm(g, pk, k16);
    }

    float s1[2] = {S.w1[0] + m4[0], S.w1[1] + m4[1]};
    float s2 = S.w2 + m16;
    unsigned sl1[2] = {x4[0], x4[1]};
    unsigned sl2 = k16;
    // Is any smaller alpha rounded to the same sum?  probe the next float below the maximum
    // (exact for negative normal maxima; anything else reports "unsafe").
    {
        const float c = 0x1.8p-24f;   // 0.75 ulp relative: RN(m + m*c) is the next float below a negative m
        const float p0 = __builtin_fmaf(m4[0], c, m4[0]), p1 = __builtin_fmaf(m4[1], c, m4[1]);
        const float p2 = __builtin_fmaf(m16, c, m16);
        const mask_t unsafe = ballot(S.w1[0] + p0 >= s1[0]) | ballot(S.w1[1] + p1 >= s1[1]) | ballot(S.w2 + p2 >= s2) | tie_halves;
        if (unsafe != 0) {
            ++S.n_rescan;
            // exact scan on the sums themselves (Viterbi.hpp:79-89 restricted to one class)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float bv = NEG_INF; unsigned bx = 0;
#pragma unroll
                for (int xx = 0; xx < 4; ++xx) {
                    const float v = S.w1[g] + S.alpha[2 * xx + g];
                    const mask_t m = ballot(v > bv);
                    bv = selm(m, v, bv);
                    bx = selm(m, (unsigned)xx, bx);
                }
                s1[g] = bv; sl1[g] = 1u + bx;
            }
            float bv = NEG_INF; unsigned bk = h;
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // ascending i == ascending k for this thread
                const float v = S.w2 + S.alpha[i];
                const mask_t m = ballot(v > bv);
                bv = selm(m, v, bv);
                bk = selm(m, 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1) + h, bk);
            }
            merge_lower(bv, bk, swap1(bv), swap1(bk));
            s2 = bv; sl2 = 5u + bk;
        }
    }
    sV1[(h << 8) | t] = ValSlot{s1[0], sl1[0]};
    sV1[((2u + h) << 8) | t] = ValSlot{s1[1], sl1[1]};
    if (h == 0) sV2[t] = ValSlot{s2, sl2};
    __syncthreads();

    // ---------------- 3-way combine per state ----------------
    const unsigned r1_base = (h << 6) + (t >> 2), q_base = (h << 4) + (t >> 4);
    // one per-thread base pointer per exchange array; every cell is then an immediate offset
    const ValSlot* const pa = sV1 + r1_base;
    const ValSlot* const pb = sV2 + q_base;
    unsigned bpw[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float4 nls4 = *reinterpret_cast<const float4*>(&sTab[0][tab_off(tau, c)]);
        const float4 cc4 = *reinterpret_cast<const float4*>(&sTab[1][tab_off(tau, c)]);
        const float4 w04 = *reinterpret_cast<const float4*>(&sTab[2][tab_off(tau, c)]);
        const float nls_[4] = {nls4.x, nls4.y, nls4.z, nls4.w};
        const float cc_[4] = {cc4.x, cc4.y, cc4.z, cc4.w};
        const float w0_[4] = {w04.x, w04.y, w04.z, w04.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * c + u;
            // k = kc + h with kc a compile-time constant: the LDS addresses are one per-thread base
            // (r1_base, q_base) plus an immediate offset
            const unsigned kc = 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1);
            const unsigned k = kc + h;
            const unsigned r1 = r1_base + (kc << 6), q = q_base + (kc << 4);
            const ValSlot a = pa[kc << 6];
            const ValSlot b = pb[kc << 4];
            const float s0 = w0_[u] + S.alpha[i];
            // fast path: the winner is unique unless two class values are equal
            float best, da = 0.f, db = 0.f, qa = 0.f, qb = 0.f, u0 = 0.f, ef = 0.f;
            mask_t e0, e1, e2;
            unsigned slot;
            if constexpr (FAST) {
                // max3, the two selects (and the back-pointer shift-or below) each carry one operation of this cell's emission
                best = max3_sub_s(s0, a.v, b.v, x, S.mu[i], da);                 // d  = x - mu
                e0 = ballot(s0 == best); e1 = ballot(a.v == best); e2 = ballot(b.v == best);
                db = y - S.eta[i]; qa = da * S.rsg[i]; qb = db * S.reta[i];
                const unsigned tmp = selm_subrev_s(e1, a.s, b.s, ly3, cc_[u], u0);   // u0 = c - 3 log y
                slot = selz_fnma(e0, tmp, qa, S.sg[i], da, ef);                    // ef = fma(-q, sigma, d)
            } else {
                best = __builtin_fmaxf(__builtin_fmaxf(s0, a.v), b.v);
                e0 = ballot(s0 == best); e1 = ballot(a.v == best); e2 = ballot(b.v == best);
                slot = selm_zero(e0, selm(e1, a.s, b.s));
            }
            // two or more of the three equal the maximum?  (all three NaN cannot happen for a cell that
            // matters: the read is then reported NCHMM_E_NUMERIC by the final arg-max)
            const mask_t tie = (e0 & e1) | ((e0 | e1) & e2);
            if (__builtin_expect(tie != 0, 0)) {
                ++S.n_tie;
                // exact rule: first maximum in ascending predecessor order (strict >, NaN never wins)
                const unsigned j = t + 256u * k;
                const unsigned p1 = ((a.s - 1u) << 10) | r1;
                const unsigned p2 = ((b.s - 5u) << 8) | q;
                float bb = NEG_INF; unsigned bp = (unsigned)kStates, sl = 255u;
                if (s0 > bb) { bb = s0; bp = j; sl = 0; }
                if (a.v > bb || (a.v == bb && p1 < bp)) { bb = a.v; bp = p1; sl = a.s; }
                if (b.v > bb || (b.v == bb && p2 < bp)) { bb = b.v; bp = p2; sl = b.s; }
                best = bb; slot = sl;
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
                const float nn = __builtin_fmaf(-0.5f, tt, nls_[u]);
                const float em = __builtin_fmaf(0.5f, uu, nn);
                if (u == 0) { S.alpha[i] = best + em; bpw[c] = slot; }
                else if (u == 1) bpw[c] = lshlor_add<8>(slot, bpw[c], best, em, S.alpha[i]);
                else if (u == 2) bpw[c] = lshlor_add<16>(slot, bpw[c], best, em, S.alpha[i]);
                else bpw[c] = lshlor_add<24>(slot, bpw[c], best, em, S.alpha[i]);
            } else {
                const float e = emission<FAST>(x, y, ry, ly3, log_2pi, S.mu[i], S.sg[i], S.rsg[i], nls_[u], S.eta[i],
                                               S.reta[i], S.lam[i], cc_[u]);
                S.alpha[i] = best + e;
                bpw[c] |= slot << (8 * u);
            }
        }
    }
    const unsigned w_lo = bpw[0], w_hi = bpw[1];
    {
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(u2v{w_lo, w_hi}, reinterpret_cast<u2v*>(bp_row + tau * 8u));
    }
}

__device__ __forceinline__ bool event_in_fast_range(float x, float y)
{
    // see quot(): keeps every dividend either 0 or within [2^-100, 2^100] given a validated model
    return __builtin_fabsf(x) <= 1048576.0f && y >= 0.0078125f && y <= 1024.0f;
}

}  // namespace

#ifndef NCHMM_MIN_WAVES
#define NCHMM_MIN_WAVES 4
#endif
__global__ __launch_bounds__(kThreads, NCHMM_MIN_WAVES) void viterbi_kernel(ViterbiArgs P)
{
    __shared__ __attribute__((aligned(16))) float sTab[3][kStates];   // -log sigma | log lambda - log 2pi | w0
    __shared__ ValSlot sV1[2][1024];   // step-group winners
    __shared__ ValSlot sV2[2][256];    // skip-group winners
    __shared__ __attribute__((aligned(16))) float4 sEv[kChunk];     // per event: x, y, 3*log y, 1/y
    ValSlot* const sRed = &sV1[0][0];  // the final arg-max reduction reuses the exchange buffer
    __shared__ unsigned sWork;

    const unsigned tau = threadIdx.x;
    const unsigned t = tau >> 1, h = tau & 1u;
    unsigned long long t_fwd = 0, t_tb = 0, t_all0 = 0;
    if (P.prof) t_all0 = wall_clock64();
    // Two blocks share a CU and the older block's waves win issue arbitration (age), which makes one
    // block ~25 % faster than its neighbour; with two reads per block that idles half of every CU at
    // the end.  Priority outranks age, so every 256 events each block publishes how many events it
    // has done (one word per CU slot) and the one that is behind raises its priority.
    unsigned* my_progress = nullptr;
    const unsigned* other_progress = nullptr;
    {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: wave[3:0] simd[5:4] cu[11:8] sh[12] se[15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
        const unsigned cu = ((xcc << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u)) & 2047u;
        const unsigned upper = ((hw & 15u) >> 1) & 1u;   // this block sits in wave slots 2,3 of each SIMD
        my_progress = P.cu_progress + 2u * cu + upper;
        other_progress = P.cu_progress + 2u * cu + (upper ^ 1u);
    }
    unsigned done_events = 0;

    for (;;) {
        // barrier first, then thread 0's fetch: it ends the previous read and keeps that read's closing
        // `if (tau == 0)` apart from this one (back to back across the loop edge LLVM may thread the two tests
        // together and send every other lane straight to the barrier below -- see fwbw_scaled_kernel.hip)
        __syncthreads();
        // tickets count up across launches (no memset in front of the kernel: a fill kernel queued behind a resident
        // forward sweep waits for a free wave slot, i.e. for the whole sweep -- profiles/r04_pipeline_timeline.md)
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
                P.last_state[r] = kNoState;
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
        // back-pointer row i of this read: one 4 KiB row per event of the batch, in event order
        uint8_t* const ws = P.ws + (e0 - P.ev_base) * (uint64_t)kStates;

        State S;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned k = 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1) + h;
            const unsigned j = t + 256u * k;
            S.mu[i] = M[MF_MU * kStates + j];
            S.sg[i] = M[MF_SIGMA * kStates + j];
            S.rsg[i] = M[MF_RSIGMA * kStates + j];
            S.eta[i] = M[MF_ETA * kStates + j];
            S.reta[i] = M[MF_RETA * kStates + j];
            S.lam[i] = M[MF_LAMBDA * kStates + j];
            const unsigned o = tab_off(tau, (unsigned)i >> 2) + ((unsigned)i & 3u);
            sTab[0][o] = M[MF_NEG_LOG_SIGMA * kStates + j];
            sTab[1][o] = M[MF_C * kStates + j];
            sTab[2][o] = W[j];   // (each thread reads back only what it wrote: no barrier needed)
        }
        S.w1[0] = W[kStates + (h << 8) + t];
        S.w1[1] = W[kStates + ((2u + h) << 8) + t];
        S.w2 = W[kStates + 1024 + t];
        S.n_rescan = 0; S.n_tie = 0;

        for (unsigned base = 0; base < n; base += kChunk) {
            {
                // wave-uniform: every wave of the block reads the same two words
                if ((tau & 63u) == 0) __hip_atomic_store(my_progress, done_events, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned other = __hip_atomic_load(other_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(other) > done_events) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
                done_events += kChunk;
            }
            // stage the next kChunk events: x, y, 3 log y, 1/y (one correctly rounded divide per event)
            const unsigned ie = base + tau;
            bool ok = true;
            if (tau < kChunk && ie < n) {
                const float x = ex[ie], y = ey[ie];
                sEv[tau] = make_float4(x, y, 3.0f * el[ie], 1.0f / y);
                ok = event_in_fast_range(x, y);
            }
            const bool fast = __syncthreads_and(ok) && model_fast;
            const unsigned hi = (n - base < kChunk) ? n - base : kChunk;
            unsigned lo = 0;
            if (base == 0) {
                // ---- column 0 (Viterbi.hpp:55-68) ----
                const float4 ev = sEv[0];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned o = tab_off(tau, (unsigned)i >> 2) + ((unsigned)i & 3u);
                    const float nls = sTab[0][o], cc = sTab[1][o];
                    const float e = fast ? emission<true>(ev.x, ev.y, ev.w, ev.z, P.log_2pi, S.mu[i], S.sg[i], S.rsg[i],
                                                          nls, S.eta[i], S.reta[i], S.lam[i], cc)
                                         : emission<false>(ev.x, ev.y, ev.w, ev.z, P.log_2pi, S.mu[i], S.sg[i], S.rsg[i],
                                                           nls, S.eta[i], S.reta[i], S.lam[i], cc);
                    S.alpha[i] = e - P.log_n_states;
                }
                lo = 1;
            }
            // ---- columns (Viterbi.hpp:72-96) ----
            if (fast) {
                for (unsigned c = lo; c < hi; ++c) {
                    const float4 ev = sEv[c];
                    const unsigned i = base + c;
                    column<true>(S, sTab, sV1[i & 1u], sV2[i & 1u], ws + (uint64_t)i * kStates, tau, uniform(ev.x),
                                 uniform(ev.y), uniform(ev.w), uniform(ev.z), P.log_2pi);
                }
            } else {
                for (unsigned c = lo; c < hi; ++c) {
                    const float4 ev = sEv[c];
                    const unsigned i = base + c;
                    column<false>(S, sTab, sV1[i & 1u], sV2[i & 1u], ws + (uint64_t)i * kStates, tau, uniform(ev.x),
                                  uniform(ev.y), uniform(ev.w), uniform(ev.z), P.log_2pi);
                }
            }
            __syncthreads();   // sEv is rewritten by the next chunk
        }

        // ---- fill_state_seq: arg-max of the last column, lowest index on ties (Viterbi.hpp:125-133) ----
        {
            float bv = -__builtin_inff();
            unsigned bi = kStates;
#pragma unroll
            for (int i = 0; i < 8; ++i) {   // ascending i == ascending j for this thread
                const unsigned k = 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1) + h;
                const bool g = S.alpha[i] > bv;
                bv = g ? S.alpha[i] : bv;
                bi = g ? t + 256u * k : bi;
            }
            sRed[tau] = ValSlot{bv, bi};
        }
        __syncthreads();   // also publishes every back-pointer store of this block (vmcnt(0) + barrier)
        unsigned long long c1 = 0;
        if (P.prof) c1 = wall_clock64();
        if (tau < 64) {
            ValSlot m = sRed[tau];
#pragma unroll
            for (int w = 1; w < kThreads / 64; ++w) {
                const ValSlot o = sRed[tau + 64 * w];
                if (o.v > m.v || (o.v == m.v && o.s < m.s)) m = o;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                ValSlot o;
                o.v = __shfl_xor(m.v, d, 64);
                o.s = __shfl_xor(m.s, d, 64);
                if (o.v > m.v || (o.v == m.v && o.s < m.s)) m = o;
            }
            if (tau == 0) {
                P.out_logp[r] = m.v;                  // Viterbi::path_probability(), Viterbi.hpp:133
                P.last_state[r] = m.s;                // kStates when every state is -INF/NaN
            }
        }
        if (P.prof) {
            const unsigned long long c2 = wall_clock64();
            t_fwd += c1 - c0;
            t_tb += c2 - c1;
            if ((tau & 63u) == 0) {   // per wave: the branches are wave-uniform
                atomicAdd(&P.prof[6], (unsigned long long)S.n_rescan);
                atomicAdd(&P.prof[7], (unsigned long long)S.n_tie);
            }
        }
        // the other waves wait for the traceback at the top-of-loop barrier; the workspace is reused
    }
    if ((tau & 63u) == 0) __hip_atomic_store(my_progress, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the slot's word as found
    if (P.prof && tau == 0) {
        atomicAdd(&P.prof[0], t_fwd);
        atomicAdd(&P.prof[1], t_tb);
        atomicAdd(&P.prof[2], wall_clock64() - t_all0);
        atomicAdd(&P.prof[3], 1ull);
        if (blockIdx.x < 2048) {
            const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            P.prof[8 + 2 * blockIdx.x] = t_all0;
            P.prof[8 + 4096 + blockIdx.x] = ((unsigned long long)xcc << 32) | hwid;
            P.prof[9 + 2 * blockIdx.x] = wall_clock64();
        }
    }
}

// branch-free pred_of for the (wave-uniform) chase
__device__ __forceinline__ unsigned pred_uniform(unsigned j, unsigned slot, unsigned& shift_class)
{
    const unsigned sc = slot == 0 ? 0u : (slot < 5u ? 1u : 2u);
    const unsigned hi = (slot - (sc == 1u ? 1u : 5u)) << (12u - 2u * sc);
    shift_class = sc;
    return sc == 0 ? j : ((hi | (j >> (2u * sc))) & 4095u);
}

constexpr int kTbWaves = 8;      // traceback segments per read (one wave each)

// One wave follows the back-pointers from (event ev_hi, state s) down to event ev_lo, writing
// out_state[e] for ev_lo <= e <= min(ev_hi - 1, ev_write_hi) (the start event itself is the caller's).
// Per round trip it fetches every 16-byte group that can hold the byte it will need in rows cur,
// cur-1 (1 + 3 groups; LEVELS == 2) and cur-2 (+ 23 groups; LEVELS == 3) and resolves two or three
// events from LDS.  Three levels minimise round trips (one read per wave: latency-bound); two levels
// fetch 4.5x fewer sectors per event, which is what matters once 8 waves per read make the
// traceback bandwidth-bound (every 16-byte group costs a whole 64-byte sector).  Decoded states
// are collected in LDS and written 192 at a time so that no store sits in front of the next loads.
// Returns the state at ev_lo; *mark_state receives the state at event `mark` if the walk passes it.
template <int LEVELS>
__device__ __forceinline__ unsigned chase(const uint8_t* ws, uint16_t* os, unsigned s, int ev_hi, int ev_lo, int ev_write_hi,
                                          int mark, unsigned* mark_state, unsigned lane, uint8_t (*sStage)[16],
                                          uint16_t* sOut, unsigned& bad, int bad_hi)
{
    unsigned rowoff, sh, msk, hi;
    if (lane == 0) { rowoff = 0; sh = 0; msk = 255; hi = 0; }
    else if (lane < 4) { rowoff = 1; sh = 2 * (lane - 1); msk = 255; hi = 0; }
    else if (lane < 7) { rowoff = 2; sh = 2 * (lane - 4); msk = 255; hi = 0; }
    else if (lane < 11) { rowoff = 2; sh = 6; msk = 63; hi = (lane - 7) << 6; }
    else { rowoff = 2; sh = 8; msk = 15; hi = ((lane - 11) & 15u) << 4; }
    const bool lane_on = lane < (LEVELS == 3 ? 27u : 4u);
    int cur = ev_hi;                 // row cur holds the back-pointer from event cur to event cur-1
    int pending_top = cur - 1;       // event index of sOut[0]
    unsigned n_pending = 0;
    while (cur > ev_lo) {
        const int row = cur - (int)rowoff;
        const unsigned grp = hi | ((s >> sh) & msk);
        if (lane_on && row > ev_lo)
            *reinterpret_cast<uint4*>(&sStage[lane][0]) =
                *reinterpret_cast<const uint4*>(ws + (uint64_t)row * kStates + grp * 16u);
        __builtin_amdgcn_s_waitcnt(0);   // the loads above (nothing else is outstanding) and the LDS stores
        __builtin_amdgcn_wave_barrier();
        unsigned sc0, sc1, sc2;
        // an unreachable cell (no predecessor: every candidate -INF/NaN) carries no back-pointer.  It only counts when the
        // walk is known to be on the true path: rows above bad_hi belong to the speculative run-in of a segment.
        unsigned slot = sStage[0][bp_pos(s >> 8)];
        bad |= (slot > 20u) & (unsigned)(cur <= bad_hi);
        s = pred_uniform(s, slot > 20u ? 0u : slot, sc0);
        const unsigned s_a = s;
        unsigned s_b = s, s_c = s;
        int done = 1;
        if (cur - 1 > ev_lo) {
            slot = sStage[1 + sc0][bp_pos(s >> 8)];
            bad |= (slot > 20u) & (unsigned)(cur - 1 <= bad_hi);
            s = pred_uniform(s, slot > 20u ? 0u : slot, sc1);
            s_b = s; done = 2;
            if (LEVELS == 3 && cur - 2 > ev_lo) {
                const unsigned tot = sc0 + sc1;
                const unsigned ln = tot <= 2 ? 4u + tot : (tot == 3 ? 7u + ((s >> 6) & 3u) : 11u + ((s >> 4) & 15u));
                slot = sStage[ln][bp_pos(s >> 8)];
                bad |= (slot > 20u) & (unsigned)(cur - 2 <= bad_hi);
                s = pred_uniform(s, slot > 20u ? 0u : slot, sc2);
                s_c = s; done = 3;
            }
        }
        if (mark_state) {
            if (cur - 1 == mark) *mark_state = s_a;
            if (done > 1 && cur - 2 == mark) *mark_state = s_b;
            if (done > 2 && cur - 3 == mark) *mark_state = s_c;
        }
        if (lane == 0) {
            sOut[n_pending] = (uint16_t)s_a;
            if (done > 1) sOut[n_pending + 1] = (uint16_t)s_b;
            if (done > 2) sOut[n_pending + 2] = (uint16_t)s_c;
        }
        n_pending += (unsigned)done;
        cur -= done;
        if (n_pending + 3 > 192u || cur <= ev_lo) {
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            // sOut[k] is the state of event pending_top - k
            for (unsigned k = lane; k < n_pending; k += 64) {
                const int e = pending_top - (int)k;
                if (e <= ev_write_hi) os[e] = sOut[k];
            }
            pending_top -= (int)n_pending;
            n_pending = 0;
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
    }
    return s;
}

// Viterbi::fill_state_seq, Viterbi.hpp:134-141.  The chase is a dependent pointer walk bound by
// HBM latency (~1 us per three events), so a read is cut into up to 8 segments walked by 8 waves
// at once.  Only the top segment knows its start state; the others start tb_margin (128 by default) events above
// their segment from an arbitrary state and rely on Viterbi survivor paths coalescing: if the
// speculative walk is in the same state as the true path at the first event it owns, everything
// below is the true path (back-pointers are a function of (event, state)).  Wave 0 checks each
// boundary top-down and re-walks a segment from the true state when its speculation had not merged,
// so the result is exact either way.
__global__ __launch_bounds__(64 * kTbWaves) void traceback_kernel(ViterbiArgs P)
{
    __shared__ __attribute__((aligned(16))) uint8_t sStage[kTbWaves][32][16];
    __shared__ uint16_t sOut[kTbWaves][192];
    __shared__ unsigned sLow[kTbWaves + 1], sTent[kTbWaves], sBad[kTbWaves];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned r = P.first_read + blockIdx.x;
    const uint64_t e0 = P.off[r];
    const int n = (int)(P.off[r + 1] - e0);
    if (n == 0) {
        if (threadIdx.x == 0 && P.out_status) P.out_status[r] = 0;
        return;
    }
    const uint8_t* const ws = P.ws + (e0 - P.ev_base) * (uint64_t)kStates;
    uint16_t* __restrict__ os = P.out_state + e0;
    const unsigned s_last = P.last_state[r];
    if (s_last >= (unsigned)kStates) {
        if (threadIdx.x == 0 && P.out_status) P.out_status[r] = -6;
        return;
    }
    int K = n / 512;
    K = K < 1 ? 1 : (K > kTbWaves ? kTbWaves : K);
    const int L = (n + K - 1) / K;                 // segment w owns events [w*L, min((w+1)*L, n) - 1]
    unsigned bad = 0;
    if ((int)wave < K) {
        const int lo = (int)wave * L;
        if ((int)wave == K - 1) {
            if (lane == 0) os[n - 1] = (uint16_t)s_last;
            const unsigned s_lo = K == 1 ? chase<3>(ws, os, s_last, n - 1, lo, n - 1, -1, nullptr, lane, sStage[wave], sOut[wave], bad, n)
                                         : chase<2>(ws, os, s_last, n - 1, lo, n - 1, -1, nullptr, lane, sStage[wave], sOut[wave], bad, n);
            if (lane == 0) sLow[wave] = s_lo;
        } else {
            const int own_hi = lo + L - 1;             // highest event this segment owns
            int start = own_hi + 1 + P.tb_margin;      // speculative start event
            if (start > n - 1) start = n - 1;
            unsigned tent = 0xFFFFFFFFu;
            // the state at event own_hi+1 is the first one compared with the segment above
            // `bad` of this walk is only meaningful from the boundary row down, and only if the walk turns out to have
            // merged with the true path there (wave 0 decides; otherwise the segment is walked again)
            const unsigned s_lo = chase<2>(ws, os, 0u, start, lo, own_hi, own_hi + 1, &tent, lane, sStage[wave], sOut[wave], bad,
                                           own_hi + 1);
            if (start == own_hi + 1) tent = 0u;        // no margin left: the guess itself sits on the boundary
            if (lane == 0) { sLow[wave] = s_lo; sTent[wave] = tent; }
        }
    }
    if (lane == 0) sBad[wave] = bad;
    __syncthreads();
    if (wave == 0) {
        unsigned any_bad = sBad[K - 1];                // the top segment starts from the true last state
        unsigned refix = 0;
        for (int w = K - 2; w >= 0; --w) {
            const unsigned truth = sLow[w + 1];        // true state at event (w+1)*L
            if (sTent[w] == truth) {
                any_bad |= sBad[w];                    // merged: what it walked from the boundary down was the true path
            } else {
                // speculation had not merged: walk this segment again from the true state
                unsigned b2 = 0;
                const unsigned s_lo = chase<3>(ws, os, truth, (w + 1) * L, w * L, (w + 1) * L - 1, -1, nullptr, lane, sStage[0],
                                            sOut[0], b2, n);
                any_bad |= b2;
                if (lane == 0) sLow[w] = s_lo;
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                ++refix;
            }
        }
        if (lane == 0) {
            if (P.out_status) P.out_status[r] = any_bad ? -6 : 0;
            if (P.prof) { atomicAdd(&P.prof[4], (unsigned long long)refix); atomicAdd(&P.prof[5], (unsigned long long)(K - 1)); }
        }
    }
}

void launch_viterbi(const ViterbiArgs& a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(viterbi_kernel, dim3(grid), dim3(kThreads), 0, stream, a);
}

void launch_traceback(const ViterbiArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(traceback_kernel, dim3(a.n_reads), dim3(64 * kTbWaves), 0, stream, a);
}

int viterbi_blocks_per_cu()
{
    // Persistent blocks pull reads from a queue and never wait on each other, so an over-estimate
    // only leaves late blocks with an empty queue.  (The occupancy API prices LDS against 64 KiB;
    // gfx950 has 160 KiB per CU and 512 VGPRs per lane per SIMD.)
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(viterbi_kernel)) != hipSuccess) return 1;
    const int by_lds = fa.sharedSizeBytes > 0 ? (int)(163840 / fa.sharedSizeBytes) : 8;
    const int regs = ((fa.numRegs + 7) / 8) * 8;
    const int waves_per_simd = regs > 0 ? 512 / regs : 8;
    const int by_vgpr = waves_per_simd * 4 / (kThreads / 64);
    const int by_waves = 32 / (kThreads / 64);
    int nb = by_lds < by_vgpr ? by_lds : by_vgpr;
    if (by_waves < nb) nb = by_waves;
    if (nb < 1) nb = 1;
    if (getenv("NCHMM_DEBUG"))
        fprintf(stderr, "[nchmm] viterbi_kernel: numRegs=%d lds=%zu -> %d blocks/CU\n", fa.numRegs,
                (size_t)fa.sharedSizeBytes, nb);
    return nb;
}

}  // namespace nchmm
