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
//   * Back-pointers are one byte per state (0 stay, 1+x step, 5+xy skip); row i of the read a block is
//     sweeping at region + i*4096 (one workspace REGION per resident block, nchmm_device.h), state j at
//     byte (t<<4) | (h<<3) | (x<<1) | (y>>1): each thread stores its 8 bytes as one dwordx2 (a wave
//     writes 512 B contiguous), and the 21 candidates of the next traceback step sit in three 16-byte
//     groups.
//   * Traceback happens in the same block as soon as the last column is done (traceback_block): the
//     chase is a dependent pointer walk bound by memory latency, and it idles the block's half of the CU
//     (the co-resident block's sweep is bound by its own dependent chain: alone on a CU it runs 7 % faster,
//     not twice as fast), so it is cut into 128 segments walked at once by 4-lane groups: ~0.09 ms per
//     5000-event read against 15 ms of sweep per block.  Because a region is free again when its block has
//     walked it, the workspace is (resident blocks) x (longest read), whatever the batch size, and launches
//     on different streams roll into each other: a block of the next launch starts in the place of each
//     block that runs out of reads.
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
constexpr int kTbPrio = 3;         // wave priority during the in-block traceback (tools/ubench/vit_ab_defs.sh: 0 and 3 measure the same)

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
        // streaming store: the row is read once, by the traceback, 21 GB of other rows later -- keeping it out of L2's write-back
        // path takes 0.6 % off the sweep and 11 % off the traceback that follows (profiles/r04_vit_nt_ab.txt)
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(u2v{w_lo, w_hi}, reinterpret_cast<u2v*>(bp_row + tau * 8u));
    }
}

__device__ __forceinline__ bool event_in_fast_range(float x, float y)
{
    // see quot(): keeps every dividend either 0 or within [2^-100, 2^100] given a validated model
    return __builtin_fabsf(x) <= 1048576.0f && y >= 0.0078125f && y <= 1024.0f;
}

// predecessor of state j through back-pointer code `slot` (0 stay, 1+x step, 5+xy skip), branch-free
__device__ __forceinline__ unsigned pred_of(unsigned j, unsigned slot, unsigned& shift_class)
{
    const unsigned sc = slot == 0 ? 0u : (slot < 5u ? 1u : 2u);
    const unsigned hi = (slot - (sc == 1u ? 1u : 5u)) << (12u - 2u * sc);
    shift_class = sc;
    return sc == 0 ? j : ((hi | (j >> (2u * sc))) & 4095u);
}

// ---- traceback: Viterbi::fill_state_seq, Viterbi.hpp:134-141, by the block that has just swept the read ----
// The chase is a dependent pointer walk bound by memory latency (~0.9 us per round trip), and while a block walks, its half of
// the CU does nothing else (the co-resident block's sweep is bound by its own dependent chain and does not speed up), so the
// walk has to be SHORT.  It is cut into segments of kTbSeg events walked at once, one per group of four lanes: 16 per wave,
// 128 per block and round (10 240 events).  Only the top segment knows its start state; the others start tb_margin events
// above their boundary from an arbitrary state and rely on Viterbi survivor paths coalescing: if the speculative walk is in
// the same state as the walk above it at the boundary, everything below is the true path (back-pointers are a function of
// (event, state)).  All boundaries are compared at once; if any differs (never with the default margin on sane data), wave 0
// goes through the segments top-down and walks again, from the true state, those whose speculation had not merged -- the result
// is exact either way.  Per round trip a group fetches the 16-byte group that holds its byte of row cur and the three that can
// hold the byte of row cur-1 (stay / step / skip) and resolves two events.  Decoded states are collected in LDS and written
// out once per round, contiguously.
constexpr int kTbLanes = 4;
constexpr int kTbSegs = kThreads / kTbLanes;   // 128 segments per round
constexpr int kTbSeg = 80;                     // events a segment owns (40 / 80 / 128 measured: profiles/r04_inblock_tb_params_ab.txt)

struct __attribute__((aligned(16))) TbShared {
    uint8_t stage[kTbSegs][kTbLanes][16];
    uint16_t path[kTbSegs][kTbSeg];     // path[w][k] = state of event (boundary of w) - 1 - k
    unsigned low[kTbSegs], tent[kTbSegs], bad[kTbSegs];
};

// One segment per 4-lane group (seg, q = lane within the group; every value below is the same in the four lanes of a group):
// from state s at event `start` down to event own_lo, recording the states of events own_lo .. bnd-1 and the state met at
// event bnd.  `on` = this group has a segment.  An unreachable cell (no predecessor: every candidate -INF/NaN) carries no
// back-pointer (code > 20); it only counts on rows the segment owns (<= bnd).
__device__ __forceinline__ void tb_walk(const uint8_t* ws, TbShared& T, unsigned seg, unsigned q, bool on, unsigned s, int start, int bnd,
                                        int own_lo)
{
    unsigned tent = start == bnd ? s : 0xFFFFFFFFu, bad = 0;
    int cur = on ? start : own_lo;
    const unsigned sh = q ? 2u * (q - 1u) : 0u;
    while (ballot(cur > own_lo) != 0) {
        const bool go = cur > own_lo;
        const int row = cur - (q ? 1 : 0);       // row i holds the back-pointers from event i to event i-1
        if (go && row > own_lo)
            *reinterpret_cast<uint4*>(&T.stage[seg][q][0]) =
                *reinterpret_cast<const uint4*>(ws + (uint64_t)row * kStates + ((s >> sh) & 255u) * 16u);
        __builtin_amdgcn_s_waitcnt(0);   // the loads above (nothing else is outstanding) and the LDS stores
        __builtin_amdgcn_wave_barrier();
        if (go) {
            unsigned sc0, sc1;
            unsigned slot = T.stage[seg][0][bp_pos(s >> 8)];
            bad |= (slot > 20u) & (unsigned)(cur <= bnd);
            const unsigned s1 = pred_of(s, slot > 20u ? 0u : slot, sc0);
            const int e1 = cur - 1;
            if (e1 == bnd) tent = s1;
            if (e1 < bnd && q == 0) T.path[seg][bnd - 1 - e1] = (uint16_t)s1;
            if (e1 > own_lo) {
                slot = T.stage[seg][1 + sc0][bp_pos(s1 >> 8)];
                bad |= (slot > 20u) & (unsigned)(e1 <= bnd);
                const unsigned s2 = pred_of(s1, slot > 20u ? 0u : slot, sc1);
                const int e2 = cur - 2;
                if (e2 == bnd) tent = s2;
                if (e2 < bnd && q == 0) T.path[seg][bnd - 1 - e2] = (uint16_t)s2;
                s = s2; cur -= 2;
            } else {
                s = s1; cur -= 1;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);   // stage[] is overwritten by the next round trip
        __builtin_amdgcn_wave_barrier();
    }
    if (on && q == 0) { T.low[seg] = s; T.tent[seg] = tent; T.bad[seg] = bad; }
}

__device__ __forceinline__ void traceback_block(const ViterbiArgs& P, TbShared& T, const uint8_t* ws, unsigned r, uint64_t e0, int n,
                                                unsigned s_last)
{
    const unsigned tid = threadIdx.x, seg = tid / kTbLanes, q = tid % kTbLanes;
    uint16_t* __restrict__ os = P.out_state + e0;
    if (s_last >= (unsigned)kStates) {   // every state -INF / NaN in the last column: no path (block-uniform)
        if (tid == 0 && P.out_status) P.out_status[r] = -6;
        return;
    }
    if (tid == 0) os[n - 1] = (uint16_t)s_last;
    int top = n - 1;                     // the state of event `top` is known: s_top
    unsigned s_top = s_last;
    int any_bad = 0;
    unsigned refix = 0, spec = 0;
    while (top > 0) {
        const int bot = top > kTbSegs * kTbSeg ? top - kTbSegs * kTbSeg : 0;
        const int count = top - bot;                        // events bot .. top-1 are resolved in this round
        const int K = (count + kTbSeg - 1) / kTbSeg;        // segments (<= kTbSegs), L events each (the last may be shorter)
        const int L = (count + K - 1) / K;
        const bool on = (int)seg < K;
        const int bnd = top - (int)seg * L;                 // segment `seg` owns events max(bnd - L, bot) .. bnd - 1
        const int own_lo = on ? (bnd - L > bot ? bnd - L : bot) : 0;
        int start = bnd + P.tb_margin;                      // speculative start event
        if (start > top || seg == 0) start = top;
        tb_walk(ws, T, seg, q, on, start == top ? s_top : 0u, on ? start : 0, on ? bnd : 0, own_lo);
        __syncthreads();
        // every boundary at once: the walk below met the state the walk above ended in
        const bool differs = on && q == 0 && seg > 0 && T.tent[seg] != T.low[seg - 1];
        if (__syncthreads_or(differs)) {
            if (tid < 64) {
                unsigned truth = s_top;
                for (int w = 0; w < K; ++w) {
                    const int b = top - w * L, lo = b - L > bot ? b - L : bot;
                    if (w > 0 && T.tent[w] != truth) {      // (wave-uniform)
                        tb_walk(ws, T, (unsigned)w, q, tid < kTbLanes, truth, b, b, lo);
                        __builtin_amdgcn_s_waitcnt(0);
                        __builtin_amdgcn_wave_barrier();
                        ++refix;
                    }
                    truth = T.low[w];
                }
            }
            __syncthreads();
        }
        any_bad |= __syncthreads_or(on && q == 0 && T.bad[seg] != 0);
        for (int idx = (int)tid; idx < count; idx += kThreads) {
            const int w = idx / L;
            os[top - 1 - idx] = T.path[w][idx - w * L];
        }
        spec += (unsigned)(K - 1);
        s_top = T.low[K - 1];
        top = bot;
        __syncthreads();   // path[] / low[] are rewritten by the next round
    }
    if (tid == 0) {
        if (P.out_status) P.out_status[r] = any_bad ? -6 : 0;
        if (P.prof) { atomicAdd(&P.prof[4], (unsigned long long)refix); atomicAdd(&P.prof[5], (unsigned long long)spec); }
    }
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
    static_assert(sizeof(TbShared) <= sizeof(sTab), "the traceback's staging lives in the per-state tables (dead after the sweep)");
    TbShared& sTb = *reinterpret_cast<TbShared*>(&sTab[0][0]);
    __shared__ unsigned sWork, sLast;

    const unsigned tau = threadIdx.x;
    const unsigned t = tau >> 1, h = tau & 1u;
    unsigned long long t_fwd = 0, t_tb = 0, t_all0 = 0;
    if (P.prof) t_all0 = wall_clock64();
    // Two blocks share a CU and the older block's waves win issue arbitration (age), which lets one block run ~25 % ahead of
    // its neighbour and idles half of the CU when the queue runs dry.  Priority outranks age, so every 256 events each block
    // publishes how many events it has done in this launch (one word per CU slot) and the one that is behind raises its
    // priority.  Blocks of different launches (tags differ) leave it to age: the older launch finishes first.
    unsigned* my_progress = nullptr;
    const unsigned* other_progress = nullptr;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: wave[3:0] simd[5:4] cu[11:8] sh[12] se[15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned upper = ((hw & 15u) >> 1) & 1u;   // this block sits in wave slots 2,3 of each SIMD
    {
        const unsigned cu = ((xcc << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u)) & 2047u;
        my_progress = P.cu_progress + 2u * cu + upper;
        other_progress = P.cu_progress + 2u * cu + (upper ^ 1u);
    }
    // ---- this block's back-pointer region ----
    if (tau == 0) {
        unsigned slot = kNoSlot;
        if (!P.slot_owner) {
            slot = blockIdx.x;
        } else {
            unsigned* const own = P.slot_owner + xcc * P.slots_per_xcd;
            const unsigned first = (2u * (((hw >> 13) & 7u) * 16u + ((hw >> 8) & 15u)) + upper) % P.slots_per_xcd;
            // a region is free whenever fewer blocks are resident on this XCD than it has regions: by construction always
            // (the host sizes the pool past the residency limit), so the bound below only keeps a corrupted pool from
            // hanging the device: ~1 s, then the block reports through host_err and leaves its reads to the others
            for (unsigned spin = 0; spin < 4096u && slot == kNoSlot; ++spin) {
                for (unsigned k = 0; k < P.slots_per_xcd; ++k) {
                    unsigned i = first + k;
                    if (i >= P.slots_per_xcd) i -= P.slots_per_xcd;
                    if (atomicCAS(&own[i], 0u, 1u) == 0u) { slot = xcc * P.slots_per_xcd + i; break; }
                }
                if (slot == kNoSlot) for (int z = 0; z < 64; ++z) __builtin_amdgcn_s_sleep(127);
            }
            __threadfence();   // acquire: whatever the previous holder of the region did is behind us
        }
        sWork = slot;
    }
    __syncthreads();
    const unsigned my_slot = sWork;
    if (my_slot == kNoSlot) {
        // no region: report, and take tickets like any other block so that the lane's ticket count stays what the host expects
        // (every launch draws n_reads + grid of them) -- the reads this block draws are marked failed, not left stale
        if (tau == 0) {
            if (P.host_err) __hip_atomic_store(P.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (;;) {
                const unsigned widx = atomicAdd(P.queue, 1u) - P.queue_base;
                if (widx >= P.n_reads) break;
                const unsigned r = P.order ? P.order[widx] : P.first_read + widx;
                P.out_logp[r] = __builtin_nanf("");
                if (P.out_status) P.out_status[r] = -3;   // NCHMM_E_HIP
            }
        }
        return;
    }
    uint8_t* const ws = P.ws + (uint64_t)my_slot * P.slot_bytes;
    const unsigned tag = P.launch_tag << 20;
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
        // back-pointer row i of this read: row i of the block's region

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
                const unsigned mine = done_events < 0xFFFFFu ? done_events : 0xFFFFFu;
                if ((tau & 63u) == 0) __hip_atomic_store(my_progress, tag | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned other = __builtin_amdgcn_readfirstlane(__hip_atomic_load(other_progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if ((other & 0xFFF00000u) == tag && (other & 0xFFFFFu) > mine) __builtin_amdgcn_s_setprio(1);
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
                sLast = m.s;                          // kStates when every state is -INF/NaN
            }
        }
        __syncthreads();   // sLast is there; the tables of this read are dead: the traceback's staging takes their place
        // the walk issues a few dozen instructions per memory round trip: at top priority it loses no time to the other
        // block's sweep (which loses nothing measurable in return)
        __builtin_amdgcn_s_setprio(kTbPrio);
        traceback_block(P, sTb, ws, r, e0, (int)n, sLast);
        __builtin_amdgcn_s_setprio(0);
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
    if ((tau & 63u) == 0) __hip_atomic_store(my_progress, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the slot's word as found
    if (tau == 0 && P.slot_owner) {
        // (every load and store of this block to its region is complete: the loop's closing barrier waited for them)
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

void launch_viterbi(const ViterbiArgs& a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(viterbi_kernel, dim3(grid), dim3(kThreads), 0, stream, a);
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
