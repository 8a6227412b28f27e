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
//   * Selects are written as v_cndmask_b32_e64 with an SGPR-pair mask: on gfx950 the VOP2 form
//     reading a VCC that was not written by the immediately preceding VALU op issues ~8x slower
//     (tools/ubench/valu_rate.hip).
//   * Back-pointers are BIT-PLANES (nchmm_device.h, kBpRowBytes).  The 3-way combine compares each
//     candidate with the maximum anyway; those compares leave 64-bit lane masks in SGPRs, and two of
//     them -- "stay wins", "step wins" -- are the back-pointer class of 64 cells at once.  They are
//     written as they are with one s_store_dwordx4 per cell (the scalar unit and the scalar data
//     cache are otherwise idle), instead of being turned into per-lane byte codes by two
//     v_cndmask + one v_lshl_or per cell on the VALU, which is the unit this kernel is bound by
//     (compare / select class ops issue at half the rate of add / mul / fma, profiles/r03_sstore_rate.txt).
//     Which member won inside a step / skip group is known to the thread that scanned the group:
//     one byte per thread and event, a plain vector store.  1.5 KiB per event instead of 4 KiB.
//   * Traceback is a second kernel (traceback_kernel, one wave per read segment, every read of the
//     batch at once): the chase is a dependent pointer walk, wave-uniform, so it runs on the scalar
//     unit -- three scalar loads (class planes, step byte, skip byte) and a few SALU ops per event.
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
    unsigned s;   // winning member of the group (x or 4x + y): read only by the exact tie rule; the state index in sRed
};
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: the body receives its index as a type, so that it can be an instruction immediate
template <int V> struct IC { static constexpr int value = V; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}

// Step-group exchange array: 1024 entries in 16 rows of 64, row pitch 72.  A wave's combine reads rows kc + h (h = lane & 1)
// at columns t >> 2: with a pitch of 64 entries both rows start on the same bank (the 2-way conflicts PMC counted as 75 %
// extra LDS cycles in rounds 1 and 2); 72 entries = 144 words puts the odd row 16 banks further.
constexpr unsigned kV1Pitch = 72;
constexpr unsigned kV1Entries = 16 * kV1Pitch;

// The two class masks of one cell go to the back-pointer row as they are: 16 bytes from an SGPR quad, one scalar store.
// (No "memory" clobber: nothing in this kernel reads these bytes back, and the LDS traffic around it must stay free to
// move.  The data SGPRs may be overwritten as soon as the store has issued: profiles/r03_sstore_rate.txt, test A.)
template <int OFF>
__device__ __forceinline__ void store_planes(mask_t p0, mask_t p1, const uint8_t* wave_row)
{
    const u32x4 q = {(unsigned)p0, (unsigned)(p0 >> 32), (unsigned)p1, (unsigned)(p1 >> 32)};
    asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(q), "s"(wave_row), "n"(OFF));
}

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
__device__ __forceinline__ mask_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// a wave-uniform float held in an SGPR instead of a VGPR
__device__ __forceinline__ float uniform(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// a wave-uniform pointer held in an SGPR pair (scalar loads / stores take their base there)
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p)
{
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((uintptr_t)hi << 32) | lo);
}

// swap with the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float swap1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ unsigned swap1(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

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
                                       const uint8_t* bp_row, const uint8_t* bp_wave, unsigned tau, float4& ev, const float4* next_ev,
                                       float log_2pi)
{
    const float x = ev.x, y = ev.y, ly3 = ev.z, ry = ev.w;     // staged per event: x, y, 3 log y, 1 / y (wave-uniform, in SGPRs)
    const float NEG_INF = -__builtin_inff();
    const unsigned t = tau >> 1, h = tau & 1u;

    // ---------------- group scans over the previous column ----------------
    // raw maxima first (strict >, ascending index => first maximum), sums once per group
    float m4[2]; unsigned x4[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {            // y = 2g + h, members i = 2x + g
        // starting from member 0 instead of -INF saves one compare-select; the results differ only
        // if member 0 is NaN while another member is not, which needs a NaN emission for some
        // states but not others -- no finite model/event does that (and an all-NaN column is
        // reported as NCHMM_E_NUMERIC at the end)
        float bv = S.alpha[g]; unsigned bx = 0;
#pragma unroll
        for (int xx = 1; xx < 4; ++xx) {
            const float v = S.alpha[2 * xx + g];
            const mask_t m = ballot(v > bv);
            bv = selm(m, v, bv);
            bx = selm(m, (unsigned)xx, bx);
        }
        m4[g] = bv; x4[g] = bx;
    }
    // own half of the skip group: k = 4x + y
    float m8 = m4[0]; unsigned k8 = 4u * x4[0] + h;
    merge_lower(m8, k8, m4[1], 4u * x4[1] + 2u + h);
    // partner half
    float m16 = m8; unsigned k16 = k8;
    merge_lower(m16, k16, swap1(m8), swap1(k8));

    float s1[2] = {S.w1[0] + m4[0], S.w1[1] + m4[1]};
    float s2 = S.w2 + m16;
    // Is any smaller alpha rounded to the same sum?  probe the next float below the maximum
    // (exact for negative normal maxima; anything else reports "unsafe").
    {
        const float c = 0x1.8p-24f;   // 0.75 ulp relative: RN(m + m*c) is the next float below a negative m
        const float p0 = __builtin_fmaf(m4[0], c, m4[0]), p1 = __builtin_fmaf(m4[1], c, m4[1]);
        const float p2 = __builtin_fmaf(m16, c, m16);
        const mask_t unsafe = ballot(S.w1[0] + p0 >= s1[0]) | ballot(S.w1[1] + p1 >= s1[1]) | ballot(S.w2 + p2 >= s2);
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
                s1[g] = bv; x4[g] = bx;
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
            s2 = bv; k16 = bk;
        }
    }
    // group r = (y << 8) | t is entry r of the step array: row r >> 6 = 4y + (t >> 6), column t & 63
    ValSlot* const pw = sV1 + kV1Pitch * (4u * h + (t >> 6)) + (t & 63u);
    pw[0] = ValSlot{s1[0], x4[0]};
    pw[8 * kV1Pitch] = ValSlot{s1[1], x4[1]};
    if (h == 0) sV2[t] = ValSlot{s2, k16};
    // who won inside this thread's groups: the second half of the back-pointer row, one byte per thread
    // (scalar base + 32-bit lane offset, spelled out: hipcc otherwise keeps a 64-bit per-lane address alive across the loop
    // and spills it; nothing in this kernel reads the byte back, so no "memory" clobber)
    asm volatile("global_store_byte %0, %1, %2 offset:%3" :: "v"(tau), "v"(x4[0] | (x4[1] << 2) | (k16 << 4)), "s"(bp_row), "n"(kBpGroupOff));

    // ---------------- emissions of this event ----------------
    // They depend on nothing the exchange delivers, so they sit between the LDS writes and the barrier: their table reads
    // wait for "all LDS and scalar-memory operations" (one counter), and here the previous event's scalar stores have had the
    // whole scan phase to land; right after the combine they would wait for stores issued a few cycles earlier.
    __builtin_amdgcn_sched_barrier(0);
    float em[8];
    static_for<0, 8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const unsigned o = tab_off(tau, (unsigned)i >> 2) + ((unsigned)i & 3u);
        em[i] = emission<FAST>(x, y, ry, ly3, log_2pi, S.mu[i], S.sg[i], S.rsg[i], sTab[0][o], S.eta[i], S.reta[i], S.lam[i], sTab[1][o]);
    });
    // (pins the computation here: without a use in front of the barrier LLVM sinks it to the alpha update after the combine)
    asm volatile("" : "+v"(em[0]), "+v"(em[1]), "+v"(em[2]), "+v"(em[3]), "+v"(em[4]), "+v"(em[5]), "+v"(em[6]), "+v"(em[7]));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();

    // ---------------- 3-way combine per state ----------------
    // entry r1 = 64 (kc + h) + (t >> 2) of the step array = row kc + h, column t >> 2; entry q = 16 (kc + h) + (t >> 4) of the skip array
    const unsigned r1_base = (h << 6) + (t >> 2), q_base = (h << 4) + (t >> 4);
    // one per-thread base pointer per exchange array; every cell is then an immediate offset
    const ValSlot* const pa = sV1 + h * kV1Pitch + (t >> 2);
    const ValSlot* const pb = sV2 + q_base;
    // Fast path for all eight cells first, no branch in between: the winner is unique unless two class values are equal, and
    // then the three compares with the maximum name it -- as lane masks, which go to memory as they are.  Cells where two or
    // more candidates equal the maximum are collected in `tie` and the whole column is redone by the exact rule below (one
    // wave-uniform branch per event instead of one per cell).
    float s0[8], best[8];
    mask_t tie = 0;
    // every group winner this thread consumes and its eight stay weights: 18 LDS reads issued back to back, ONE wait.  (Left
    // alone, hipcc reads the three values of a cell right before their use and waits for each cell -- and each of those
    // waits would also wait for the scalar store of the cell before.)
    float av[8], bv[8];
    static_for<0, 8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        // k = kc + h with kc a compile-time constant: the LDS addresses are one per-thread base plus an immediate offset
        constexpr unsigned kc = 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1);
        av[i] = pa[kc * kV1Pitch].v;
        bv[i] = pb[kc << 4].v;
    });
    const float4 w0lo = *reinterpret_cast<const float4*>(&sTab[2][tab_off(tau, 0)]);
    const float4 w0hi = *reinterpret_cast<const float4*>(&sTab[2][tab_off(tau, 1)]);
    const float w0[8] = {w0lo.x, w0lo.y, w0lo.z, w0lo.w, w0hi.x, w0hi.y, w0hi.z, w0hi.w};
    float4 nev = *next_ev;
    asm volatile("" : "+v"(nev.x), "+v"(nev.y), "+v"(nev.z), "+v"(nev.w));
    asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]),
                      "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
    static_for<0, 8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        s0[i] = w0[i] + S.alpha[i];
        best[i] = __builtin_fmaxf(__builtin_fmaxf(s0[i], av[i]), bv[i]);
        // the class of the winner, 64 cells at a time: these masks are the back-pointer
        const mask_t e0 = ballot(s0[i] == best[i]), e1 = ballot(av[i] == best[i]), e2 = ballot(bv[i] == best[i]);
        // two or more of the three equal the maximum?  (all three NaN cannot happen for a cell that
        // matters: the read is then reported NCHMM_E_NUMERIC by the final arg-max)
        tie |= (e0 & e1) | ((e0 | e1) & e2);
        // cell i = (x << 1) | (y >> 1) sits at position ((y >> 1) << 2) | x of the wave's 128 bytes
        store_planes<16 * (((i & 1) << 2) | (i >> 1))>(e0, e1, bp_wave);
    });
    if (tie != 0) {
        ++S.n_tie;
        // the planes of this column are written a second time: the first stores must have landed (scalar stores of one wave
        // are not ordered among themselves)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<0, 8>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            constexpr unsigned kc = 4u * (unsigned)(i >> 1) + 2u * (unsigned)(i & 1);
            // exact rule: first maximum in ascending predecessor order (strict >, NaN never wins)
            const ValSlot a = pa[kc * kV1Pitch], b = pb[kc << 4];
            const unsigned j = t + 256u * (kc + h);
            const unsigned p1 = (a.s << 10) | (r1_base + (kc << 6));
            const unsigned p2 = (b.s << 8) | (q_base + (kc << 4));
            float bb = NEG_INF; unsigned bp = (unsigned)kStates, cls = 3u;    // 3: no predecessor (every candidate -INF / NaN)
            if (s0[i] > bb) { bb = s0[i]; bp = j; cls = 0u; }
            if (a.v > bb || (a.v == bb && p1 < bp)) { bb = a.v; bp = p1; cls = 1u; }
            if (b.v > bb || (b.v == bb && p2 < bp)) { bb = b.v; bp = p2; cls = 2u; }
            best[i] = bb;
            store_planes<16 * (((i & 1) << 2) | (i >> 1))>(ballot(cls == 0u) | ballot(cls == 3u), ballot(cls == 1u) | ballot(cls == 3u), bp_wave);
        });
    }
    static_for<0, 8>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        S.alpha[i] = best[i] + em[i];
    });
    ev = make_float4(uniform(nev.x), uniform(nev.y), uniform(nev.z), uniform(nev.w));
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
    __shared__ ValSlot sV1[2][kV1Entries];   // step-group winners (16 rows of 64, pitch 72)
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
        if (tau == 0) sWork = atomicAdd(P.queue, 1u);
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
        // back-pointer row i of this read: one kBpRowBytes row per event of the batch, in event order
        uint8_t* const ws = P.ws + (e0 - P.ev_base) * (uint64_t)kBpRowBytes;
        const unsigned wave_off = __builtin_amdgcn_readfirstlane(tau >> 6) * 128u;   // this wave's class planes inside a row

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
            // The staged event of column c + 1 is fetched inside column c, in the same batch of LDS reads as the group winners
            // (one wait for all of them); fetched at the top of its own column it would be a second wait, right behind the
            // scalar stores that end the column before.
            if (lo < hi) {
                float4 ev = sEv[lo];
                ev = make_float4(uniform(ev.x), uniform(ev.y), uniform(ev.z), uniform(ev.w));
                if (fast) {
                    for (unsigned c = lo; c < hi; ++c) {
                        const unsigned i = base + c;
                        const uint8_t* const row = ws + (uint64_t)i * kBpRowBytes;
                        column<true>(S, sTab, sV1[i & 1u], sV2[i & 1u], row, row + wave_off, tau, ev, &sEv[c + 1 < kChunk ? c + 1 : c], P.log_2pi);
                    }
                } else {
                    for (unsigned c = lo; c < hi; ++c) {
                        const unsigned i = base + c;
                        const uint8_t* const row = ws + (uint64_t)i * kBpRowBytes;
                        column<false>(S, sTab, sV1[i & 1u], sV2[i & 1u], row, row + wave_off, tau, ev, &sEv[c + 1 < kChunk ? c + 1 : c], P.log_2pi);
                    }
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
        // the class planes went through the scalar data cache: write it back (the traceback is another kernel; the end of
        // this one flushes L2-bound vector stores, but a scalar store only leaves the scalar cache when told to)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
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

constexpr int kTbWaves = 8;      // traceback segments per read (one wave each)

// One step of the chase: state s at event `cur` -> its predecessor at event cur - 1, decoded from back-pointer row `cur`
// (layout: nchmm_device.h).  s is wave-uniform, so this is scalar work: the class planes of s's cell (16 bytes), the byte
// of the thread that scanned s's step group and the byte of the thread that scanned its skip group sit at addresses that
// depend on s alone -- three scalar loads in flight together, one wait, a dozen SALU ops.  `none` collects the
// "no predecessor" marker (both class bits set).
__device__ __forceinline__ unsigned tb_step(const uint8_t* row, unsigned s, unsigned& none)
{
    const unsigned t = s & 255u, k = s >> 8, y = k & 3u;
    const unsigned tau = 2u * t + (y & 1u), lane = tau & 63u;
    const unsigned o_cls = (tau >> 6) * 128u + (((y >> 1) << 2) | (k >> 2)) * 16u;
    // step group r = s >> 2 = (y1 << 8) | t1 was scanned by thread 2 t1 + (y1 & 1): member bits 2 (y1 >> 1) of its byte
    const unsigned r = s >> 2, y1 = r >> 8;
    const unsigned o_step = kBpGroupOff + 2u * (r & 255u) + (y1 & 1u);
    // skip group q = s >> 4 was scanned by threads 2q and 2q + 1, which both hold the merged winner: high nibble of byte 2q
    const unsigned o_skip = kBpGroupOff + 2u * (s >> 4);
    u32x4 cls;
    unsigned w_step, w_skip;
    asm volatile("s_load_dwordx4 %0, %3, %4\n\ts_load_dword %1, %3, %5\n\ts_load_dword %2, %3, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(cls), "=&s"(w_step), "=&s"(w_skip)
                 : "s"(row), "s"(o_cls), "s"(o_step & ~3u), "s"(o_skip & ~3u)
                 : "memory");
    const unsigned p0 = (((lane & 32u) ? cls.y : cls.x) >> (lane & 31u)) & 1u;     // stay
    const unsigned p1 = (((lane & 32u) ? cls.w : cls.z) >> (lane & 31u)) & 1u;     // step
    const unsigned b_step = (w_step >> (8u * (o_step & 3u))) & 255u, b_skip = (w_skip >> (8u * (o_skip & 3u))) & 255u;
    const unsigned pred_step = (((b_step >> (2u * (y1 >> 1))) & 3u) << 10) | r;
    const unsigned pred_skip = ((b_skip >> 4) << 8) | (s >> 4);
    none |= p0 & p1;
    return p0 ? s : (p1 ? pred_step : pred_skip);
}

// One wave follows the back-pointers from (event ev_hi, state s) down to event ev_lo, writing out_state[e] for
// ev_lo <= e <= min(ev_hi - 1, ev_write_hi) (the start event itself is the caller's).  The walk is scalar (tb_step); the
// vector lanes only collect the decoded states, one per lane, and write them 64 at a time.
// Returns the state at ev_lo; *mark_state receives the state at event `mark` if the walk passes it.  `bad` collects
// "no predecessor" cells met at rows <= bad_hi (rows above belong to the speculative run-in of a segment).
__device__ __forceinline__ unsigned chase(const uint8_t* ws, uint16_t* os, unsigned s, int ev_hi, int ev_lo, int ev_write_hi,
                                          int mark, unsigned* mark_state, unsigned lane, unsigned& bad, int bad_hi)
{
    // everything that steers the walk is wave-uniform; say so, and the loop runs on the scalar unit
    s = __builtin_amdgcn_readfirstlane(s);
    ws = uniform_ptr(ws);
    ev_lo = __builtin_amdgcn_readfirstlane(ev_lo);
    ev_write_hi = __builtin_amdgcn_readfirstlane(ev_write_hi);
    mark = __builtin_amdgcn_readfirstlane(mark);
    bad_hi = __builtin_amdgcn_readfirstlane(bad_hi);
    int cur = __builtin_amdgcn_readfirstlane(ev_hi);   // row cur holds the back-pointer from event cur to event cur - 1
    int pending_top = 0;             // event of lane 0's pending state
    unsigned n_pending = 0, acc = 0;
    while (cur > ev_lo) {
        unsigned none = 0;
        s = tb_step(ws + (uint64_t)cur * kBpRowBytes, s, none);
        // an unreachable cell (no predecessor: every candidate -INF/NaN) carries no back-pointer.  It only counts when the
        // walk is known to be on the true path: rows above bad_hi belong to the speculative run-in of a segment.
        bad |= none & (unsigned)(cur <= bad_hi);
        --cur;                       // s is the state of event cur now
        if (mark_state && cur == mark) *mark_state = s;
        if (cur <= ev_write_hi) {
            if (n_pending == 0) pending_top = cur;
            acc = lane == n_pending ? s : acc;
            ++n_pending;
        }
        if (n_pending == 64u || (cur <= ev_lo && n_pending != 0u)) {
            if (lane < n_pending) os[pending_top - (int)lane] = (uint16_t)acc;
            n_pending = 0;
        }
    }
    return s;
}

// Viterbi::fill_state_seq, Viterbi.hpp:134-141.  The chase is a dependent pointer walk bound by
// memory latency (one round trip per event), so a read is cut into up to 8 segments walked by 8 waves
// at once.  Only the top segment knows its start state; the others start tb_margin (256) events above
// their segment from an arbitrary state and rely on Viterbi survivor paths coalescing: if the
// speculative walk is in the same state as the true path at the first event it owns, everything
// below is the true path (back-pointers are a function of (event, state)).  Wave 0 checks each
// boundary top-down and re-walks a segment from the true state when its speculation had not merged,
// so the result is exact either way.
__global__ __launch_bounds__(64 * kTbWaves) void traceback_kernel(ViterbiArgs P)
{
    __shared__ unsigned sLow[kTbWaves + 1], sTent[kTbWaves], sBad[kTbWaves];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned r = P.first_read + blockIdx.x;
    const uint64_t e0 = P.off[r];
    const int n = (int)(P.off[r + 1] - e0);
    if (n == 0) {
        if (threadIdx.x == 0 && P.out_status) P.out_status[r] = 0;
        return;
    }
    const uint8_t* const ws = P.ws + (e0 - P.ev_base) * (uint64_t)kBpRowBytes;
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
            const unsigned s_lo = chase(ws, os, s_last, n - 1, lo, n - 1, -1, nullptr, lane, bad, n);
            if (lane == 0) sLow[wave] = s_lo;
        } else {
            const int own_hi = lo + L - 1;             // highest event this segment owns
            int start = own_hi + 1 + P.tb_margin;      // speculative start event
            if (start > n - 1) start = n - 1;
            unsigned tent = 0xFFFFFFFFu;
            // the state at event own_hi+1 is the first one compared with the segment above
            // `bad` of this walk is only meaningful from the boundary row down, and only if the walk turns out to have
            // merged with the true path there (wave 0 decides; otherwise the segment is walked again)
            const unsigned s_lo = chase(ws, os, 0u, start, lo, own_hi, own_hi + 1, &tent, lane, bad, own_hi + 1);
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
                const unsigned s_lo = chase(ws, os, truth, (w + 1) * L, w * L, (w + 1) * L - 1, -1, nullptr, lane, b2, n);
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
