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

#include "viterbi_common.hpp"

constexpr unsigned kChunk = 256;   // events staged in LDS at a time

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
    static_assert(sizeof(TbShared<kThreads>) <= sizeof(sTab), "the traceback's staging lives in the per-state tables (dead after the sweep)");
    TbShared<kThreads>& sTb = *reinterpret_cast<TbShared<kThreads>*>(&sTab[0][0]);
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
    if (tau == 0) sWork = take_region(P, xcc, 2u * (((hw >> 13) & 7u) * 16u + ((hw >> 8) & 15u)) + upper);
    __syncthreads();
    const unsigned my_slot = sWork;
    if (my_slot == kNoSlot) {
        if (tau == 0) fail_without_region(P);
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
        traceback_block<kThreads, BpWide>(P, sTb, ws, r, e0, (int)n, sLast);
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
