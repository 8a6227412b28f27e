// viterbi_common.hpp -- what the two instantiations of the Viterbi sweep share: viterbi_kernel.hip (8 waves, 8 states per
// thread, two blocks per CU: the throughput form) and viterbi_ll_kernel.hip (16 waves, 4 states per thread, one block per CU:
// the low-latency form for reads that would otherwise set the duration of a launch on their own).  Select / rider / lane-swap
// primitives, the reference's emission (Pore_Model.hpp:24-40,145-149) operation for operation, and the in-block traceback
// (Viterbi::fill_state_seq, Viterbi.hpp:134-141).  Included inside an anonymous namespace of namespace nchmm by both files.
#ifndef NCHMM_VITERBI_COMMON_HPP
#define NCHMM_VITERBI_COMMON_HPP

typedef unsigned long long mask_t;
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
// ... and with the lane two over (lane ^ 2): quad_perm [2,3,0,1]
__device__ __forceinline__ float swap2(float v)
{
    float r;
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ unsigned swap2(unsigned v)
{
    unsigned r;
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
    return r;
}

// Byte of state j inside the 16-byte group of its low 8 bits (k = j >> 8 = 4x + y).  The layout follows the sweep's ownership
// map, so that a thread's back-pointers are one contiguous store:
//   BpWide  (viterbi_kernel.hip, thread 2t+h owns y in {h, h+2}, x = 0..3):  ((y&1)<<3) | (x<<1) | (y>>1)   -- 8 bytes per thread
//   BpQuad  (viterbi_ll_kernel.hip, thread 4t+y owns x = 0..3):             (y<<2) | x                      -- 4 bytes per thread
struct BpWide { static __device__ __forceinline__ unsigned pos(unsigned k) { return ((k & 1u) << 3) | ((k >> 2) << 1) | ((k >> 1) & 1u); } };
struct BpQuad { static __device__ __forceinline__ unsigned pos(unsigned k) { return ((k & 3u) << 2) | (k >> 2); } };

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

// (value, index) merge: take b if b.v > a.v, or equal and lower index
__device__ __forceinline__ void merge_lower(float& av, unsigned& ai, float bv, unsigned bi)
{
    // three compares into SGPR masks combined on the scalar unit (a short-circuit expression makes the compiler branch
    // and round-trip the mask through a VGPR)
    const mask_t m = ballot(bv > av) | (ballot(bv == av) & ballot(bi < ai));
    av = selm(m, bv, av);
    ai = selm(m, bi, ai);
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
// The walk is a template over the block size (512 threads: 128 segments per round; the 1024-thread low-latency sweep of
// viterbi_ll_kernel.hip: 256) and over the byte layout of a back-pointer group (BpLayout below).
constexpr int kTbLanes = 4;
constexpr int kTbSeg = 80;                     // events a segment owns (40 / 80 / 128 measured: profiles/r04_inblock_tb_params_ab.txt)

template <int THREADS>
struct __attribute__((aligned(16))) TbShared {
    static constexpr int kSegs = THREADS / kTbLanes;   // segments per round
    uint8_t stage[kSegs][kTbLanes][16];
    uint16_t path[kSegs][kTbSeg];       // path[w][k] = state of event (boundary of w) - 1 - k
    unsigned low[kSegs], tent[kSegs], bad[kSegs];
};

// One segment per 4-lane group (seg, q = lane within the group; every value below is the same in the four lanes of a group):
// from state s at event `start` down to event own_lo, recording the states of events own_lo .. bnd-1 and the state met at
// event bnd.  `on` = this group has a segment.  An unreachable cell (no predecessor: every candidate -INF/NaN) carries no
// back-pointer (code > 20); it only counts on rows the segment owns (<= bnd).
template <int THREADS, class BP>
__device__ __forceinline__ void tb_walk(const uint8_t* ws, TbShared<THREADS>& T, unsigned seg, unsigned q, bool on, unsigned s, int start, int bnd,
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
            unsigned slot = T.stage[seg][0][BP::pos(s >> 8)];
            bad |= (slot > 20u) & (unsigned)(cur <= bnd);
            const unsigned s1 = pred_of(s, slot > 20u ? 0u : slot, sc0);
            const int e1 = cur - 1;
            if (e1 == bnd) tent = s1;
            if (e1 < bnd && q == 0) T.path[seg][bnd - 1 - e1] = (uint16_t)s1;
            if (e1 > own_lo) {
                slot = T.stage[seg][1 + sc0][BP::pos(s1 >> 8)];
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

template <int THREADS, class BP>
__device__ __forceinline__ void traceback_block(const ViterbiArgs& P, TbShared<THREADS>& T, const uint8_t* ws, unsigned r, uint64_t e0, int n,
                                                unsigned s_last)
{
    constexpr int kTbSegs = THREADS / kTbLanes;
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
        tb_walk<THREADS, BP>(ws, T, seg, q, on, start == top ? s_top : 0u, on ? start : 0, on ? bnd : 0, own_lo);
        __syncthreads();
        // every boundary at once: the walk below met the state the walk above ended in
        const bool differs = on && q == 0 && seg > 0 && T.tent[seg] != T.low[seg - 1];
        if (__syncthreads_or(differs)) {
            if (tid < 64) {
                unsigned truth = s_top;
                for (int w = 0; w < K; ++w) {
                    const int b = top - w * L, lo = b - L > bot ? b - L : bot;
                    if (w > 0 && T.tent[w] != truth) {      // (wave-uniform)
                        tb_walk<THREADS, BP>(ws, T, (unsigned)w, q, tid < kTbLanes, truth, b, b, lo);
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
        for (int idx = (int)tid; idx < count; idx += THREADS) {
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
// ---- the block's back-pointer region (thread 0) ----
// One region per resident block, handed out per XCD (nchmm_device.h).  `first` = where this block starts looking (its CU and
// block slot: neighbours do not fight over the same word).  kNoSlot after ~1 s without a free region.
__device__ __forceinline__ unsigned take_region(const ViterbiArgs& P, unsigned xcc, unsigned first)
{
    unsigned slot = kNoSlot;
    if (!P.slot_owner) return blockIdx.x;
    unsigned* const own = P.slot_owner + xcc * P.slots_per_xcd;
    first %= P.slots_per_xcd;
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
    return slot;
}

// A block that found no region (thread 0): report, and take tickets like any other block so that the lane's ticket count stays
// what the host expects (every launch draws n_reads + grid of them) -- the reads this block draws are marked failed, not left
// stale.
__device__ __forceinline__ void fail_without_region(const ViterbiArgs& P)
{
    if (P.host_err) __hip_atomic_store(P.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (;;) {
        const unsigned widx = atomicAdd(P.queue, 1u) - P.queue_base;
        if (widx >= P.n_reads) break;
        const unsigned r = P.order ? P.order[widx] : P.first_read + widx;
        P.out_logp[r] = __builtin_nanf("");
        if (P.out_status) P.out_status[r] = -3;   // NCHMM_E_HIP
    }
}

// ---- arg-max of the last column over the block (Viterbi.hpp:125-133): lowest state index on ties ----
// sRed[tid] = each thread's own best (value, state); wave 0 reduces.  Returns in lane 0 of wave 0.
template <int THREADS>
__device__ __forceinline__ ValSlot reduce_last_column(const ValSlot* sRed, unsigned tid)
{
    ValSlot m = sRed[tid];
#pragma unroll
    for (int w = 1; w < THREADS / 64; ++w) {
        const ValSlot o = sRed[tid + 64 * w];
        if (o.v > m.v || (o.v == m.v && o.s < m.s)) m = o;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ValSlot o;
        o.v = __shfl_xor(m.v, d, 64);
        o.s = __shfl_xor(m.s, d, 64);
        if (o.v > m.v || (o.v == m.v && o.s < m.s)) m = o;
    }
    return m;
}

#endif
