// viterbi_kernel.hip -- per-read Viterbi decode over the 4096-state 6-mer pore HMM, gfx950.
//
// Replaces Viterbi::fill + fill_state_seq (src/nanocall/Viterbi.hpp:44-99,120-142) for a batch of
// reads.  One read per thread-block (persistent blocks pull reads from a work queue).
//
// Mapping (DESIGN.md "Viterbi kernel"):
//   * 256 threads = 4 waves, one per SIMD.  Thread t owns the 16 states j = t + 256*k whose LOW 8
//     bits (last four bases) are t; k = top 4 bits (first two bases).  alpha[16], the six emission
//     parameters of each owned state and the transition weights live in VGPRs for the whole read.
//   * The predecessors of j are  j,  (x<<10)|(j>>2) x=0..3  and  (xy<<8)|(j>>4) xy=0..15
//     (Kmer::neighbour_list inverted, Kmer.hpp:128-142).  All 16 skip-predecessors of a state
//     share their low 8 bits, all 4 step-predecessors their low 10 bits, so thread t holds every
//     member of skip group q=t and of step groups r=(y<<8)|t: the 21-way max of the reference
//     (Viterbi.hpp:79-89) becomes two in-register group scans per thread + one 3-way combine per
//     state, with the per-group results exchanged through LDS (one barrier per event).
//   * Weights factor as w0[j] / w1[r] / w2[q] (see nchmm_api.cpp: factor_transitions) so the sums
//     w + alpha are exactly the floats the reference forms; ties resolve to the lowest predecessor
//     index exactly as the ascending strict-> scan of the reference does.
//   * Back-pointers are one byte per state (0 stay, 1+x step, 5+xy skip), row i of a read at
//     ws + i*4096, state j at byte ((j&255)<<4)|(j>>8): each thread stores its 16 bytes as ONE
//     dwordx4 (a wave writes 1 KiB contiguous), and the 21 candidates of the next traceback step
//     sit in three 16-byte groups.
//   * Traceback: wave 0 resolves three events per memory round trip by fetching every 16-byte
//     group that can hold the byte of rows i, i-1, i-2 (27 lanes x 16 B).
//
// Float contract: -ffp-contract=off, correctly rounded fp32 divide (hipcc default), denormals on,
// no device log/exp anywhere: every log is computed by the host libm and passed in.
#include "nchmm_device.h"

#pragma clang fp contract(off)

namespace nchmm {

namespace {

struct __attribute__((aligned(8))) ValIdx {
    float v;
    unsigned i;
};

__device__ __forceinline__ unsigned pred_of(unsigned j, unsigned slot)
{
    // slot 0: stay; 1..4: step with first base x = slot-1; 5..20: skip with first two bases xy = slot-5
    if (slot == 0) return j;
    if (slot < 5) return ((slot - 1) << 10) | (j >> 2);
    return ((slot - 5) << 8) | (j >> 4);
}

__device__ __forceinline__ unsigned bp_byte_offset(unsigned j) { return ((j & 255u) << 4) | (j >> 8); }

// Pore_Model_State::log_pr_corrected_emission, Pore_Model.hpp:145-149 with log_normal_pdf :24-31
// and log_invgauss_pdf :33-40, operation for operation.  c = log_lambda - log_2pi is the first
// subtraction of the reference's left-to-right expression, ly3 = 3.0f * log_x its second operand.
__device__ __forceinline__ float emission(float x, float y, float ly3, float mu, float sigma, float lsigma,
                                          float eta, float lambda, float c, float log_2pi)
{
    float a = (x - mu) / sigma;
    float n = -lsigma - (log_2pi + a * a) / 2.0f;
    float b = (y - eta) / eta;
    float ig = (c - ly3 - lambda * b * b / y) / 2.0f;
    return n + ig;
}

}  // namespace

__global__ __launch_bounds__(kThreads) void viterbi_kernel(ViterbiArgs P)
{
    __shared__ ValIdx sV1[2][1024];   // step-group winners  (value, x)
    __shared__ ValIdx sV2[2][256];    // skip-group winners  (value, xy)
    __shared__ ValIdx sRed[kThreads];
    __shared__ __attribute__((aligned(16))) uint8_t sStage[32][16];
    __shared__ unsigned sWork;

    const unsigned t = threadIdx.x;
    uint8_t* const ws = P.ws + (uint64_t)blockIdx.x * P.ws_stride;

    for (;;) {
        if (t == 0) sWork = atomicAdd(P.queue, 1u);
        __syncthreads();
        const unsigned widx = sWork;
        __syncthreads();
        if (widx >= P.n_reads) break;
        const unsigned r = __builtin_amdgcn_readfirstlane(P.order ? P.order[widx] : widx);
        const uint64_t e0 = P.off[r];
        const unsigned n = (unsigned)(P.off[r + 1] - e0);
        if (n == 0) {
            if (t == 0) {
                P.out_logp[r] = __builtin_nanf("");
                if (P.out_status) P.out_status[r] = 0;
            }
            continue;
        }
        const int ms = P.model_slot ? P.model_slot[r] : 0;
        const int ts = P.trans_slot ? P.trans_slot[r] : 0;
        const float* __restrict__ M = P.models + (size_t)ms * kModelFloats;
        const float* __restrict__ W = P.trans + (size_t)ts * kTransFloats;
        const float* __restrict__ ex = P.cmean + e0;
        const float* __restrict__ ey = P.stdv + e0;
        const float* __restrict__ el = P.lstdv + e0;

        float mu[16], sg[16], lsg[16], eta[16], lam[16], cc[16], w0[16], alpha[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned j = t + 256u * k;
            mu[k] = M[MF_MU * kStates + j];
            sg[k] = M[MF_SIGMA * kStates + j];
            lsg[k] = M[MF_LOG_SIGMA * kStates + j];
            eta[k] = M[MF_ETA * kStates + j];
            lam[k] = M[MF_LAMBDA * kStates + j];
            cc[k] = M[MF_C * kStates + j];
            w0[k] = W[j];
        }
        float w1[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) w1[y] = W[kStates + (y << 8) + t];
        const float w2 = W[kStates + 1024 + t];

        // ---- column 0 (Viterbi.hpp:55-68) ----
        {
            const float x = ex[0], y = ey[0], ly3 = 3.0f * el[0];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                alpha[k] = emission(x, y, ly3, mu[k], sg[k], lsg[k], eta[k], lam[k], cc[k], P.log_2pi) - P.log_n_states;
        }

        // ---- columns 1..n-1 (Viterbi.hpp:72-96) ----
        for (unsigned i = 1; i < n; ++i) {
            const unsigned buf = i & 1u;
            const float x = ex[i], y = ey[i], ly3 = 3.0f * el[i];
            // in-register group scans over the previous column, ascending predecessor index,
            // strict > from -INF exactly like the reference loop
            {
                float bv = -__builtin_inff();
                unsigned bi = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float v = w2 + alpha[k];
                    if (v > bv) { bv = v; bi = k; }
                }
                sV2[buf][t] = ValIdx{bv, bi};
            }
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {
                float bv = -__builtin_inff();
                unsigned bi = 0;
#pragma unroll
                for (int xx = 0; xx < 4; ++xx) {
                    const float v = w1[yy] + alpha[4 * xx + yy];
                    if (v > bv) { bv = v; bi = xx; }
                }
                sV1[buf][(yy << 8) | t] = ValIdx{bv, bi};
            }
            __syncthreads();
            unsigned bpw[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned j = t + 256u * k;
                const unsigned r1 = ((unsigned)k << 6) | (t >> 2);
                const unsigned q = ((unsigned)k << 4) | (t >> 4);
                const ValIdx a = sV1[buf][r1];
                const ValIdx b = sV2[buf][q];
                const unsigned p1 = (a.i << 10) | r1;
                const unsigned p2 = (b.i << 8) | q;
                float best = -__builtin_inff();
                unsigned bp = kStates, slot = 255u;
                const float s0 = w0[k] + alpha[k];
                if (s0 > best) { best = s0; bp = j; slot = 0; }
                if (a.v > best || (a.v == best && p1 < bp)) { best = a.v; bp = p1; slot = 1u + a.i; }
                if (b.v > best || (b.v == best && p2 < bp)) { best = b.v; bp = p2; slot = 5u + b.i; }
                const float e = emission(x, y, ly3, mu[k], sg[k], lsg[k], eta[k], lam[k], cc[k], P.log_2pi);
                alpha[k] = best + e;
                bpw[k >> 2] |= slot << (8 * (k & 3));
            }
            *reinterpret_cast<uint4*>(ws + (uint64_t)i * kStates + t * 16u) = make_uint4(bpw[0], bpw[1], bpw[2], bpw[3]);
        }

        // ---- fill_state_seq: arg-max of the last column, lowest index on ties (Viterbi.hpp:125-133) ----
        {
            float bv = -__builtin_inff();
            unsigned bi = kStates;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (alpha[k] > bv) { bv = alpha[k]; bi = t + 256u * k; }
            sRed[t] = ValIdx{bv, bi};
        }
        __syncthreads();   // also publishes every back-pointer store of this block (vmcnt(0) + barrier)
        if (t < 64) {
            ValIdx m = sRed[t];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const ValIdx o = sRed[t + 64 * w];
                if (o.v > m.v || (o.v == m.v && o.i < m.i)) m = o;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                ValIdx o;
                o.v = __shfl_xor(m.v, d, 64);
                o.i = __shfl_xor(m.i, d, 64);
                if (o.v > m.v || (o.v == m.v && o.i < m.i)) m = o;
            }
            // every lane of wave 0 now holds (path probability, last state)
            uint16_t* __restrict__ os = P.out_state + e0;
            if (t == 0) {
                P.out_logp[r] = m.v;
                if (P.out_status) P.out_status[r] = (m.i >= (unsigned)kStates) ? -6 : 0;
            }
            if (m.i < (unsigned)kStates) {
                // this CU may still cache rows of the previous read that used this workspace
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                unsigned s = m.i;
                int cur = (int)n - 1;
                if (t == 0) os[cur] = (uint16_t)s;
                while (cur >= 1) {
                    // rows cur, cur-1, cur-2: fetch every 16-byte group that can hold the needed byte
                    int row; unsigned grp; bool act = true;
                    if (t == 0) { row = cur; grp = s & 255u; }
                    else if (t < 4) { row = cur - 1; grp = (s >> (2 * (t - 1))) & 255u; }
                    else if (t < 7) { row = cur - 2; grp = (s >> (2 * (t - 4))) & 255u; }
                    else if (t < 11) { row = cur - 2; grp = ((t - 7) << 6) | ((s >> 6) & 63u); }
                    else if (t < 27) { row = cur - 2; grp = ((t - 11) << 4) | ((s >> 8) & 15u); }
                    else { row = 0; grp = 0; act = false; }
                    if (act && row >= 1)
                        *reinterpret_cast<uint4*>(&sStage[t][0]) =
                            *reinterpret_cast<const uint4*>(ws + (uint64_t)row * kStates + grp * 16u);
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_s_waitcnt(0);   // LDS stage visible to the whole wave
                    __builtin_amdgcn_wave_barrier();
                    // resolve up to three steps (all lanes redundantly; uniform control flow)
                    unsigned slot = sStage[0][s >> 8];
                    unsigned sh0 = slot == 0 ? 0u : (slot < 5 ? 1u : 2u);
                    s = pred_of(s, slot);
                    if (t == 0) os[cur - 1] = (uint16_t)s;
                    int done = 1;
                    if (cur - 1 >= 1) {
                        slot = sStage[1 + sh0][s >> 8];
                        unsigned sh1 = slot == 0 ? 0u : (slot < 5 ? 1u : 2u);
                        s = pred_of(s, slot);
                        if (t == 0) os[cur - 2] = (uint16_t)s;
                        done = 2;
                        if (cur - 2 >= 1) {
                            const unsigned tot = sh0 + sh1;
                            const unsigned lane = tot <= 2 ? 4u + tot
                                                : (tot == 3 ? 7u + ((s >> 6) & 3u) : 11u + ((s >> 4) & 15u));
                            slot = sStage[lane][s >> 8];
                            s = pred_of(s, slot);
                            if (t == 0) os[cur - 3] = (uint16_t)s;
                            done = 3;
                        }
                    }
                    cur -= done;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        // waves 1..3 wait for the traceback at the top-of-loop barrier; the workspace is reused
    }
}

void launch_viterbi(const ViterbiArgs& a, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(viterbi_kernel, dim3(grid), dim3(kThreads), 0, stream, a);
}

int viterbi_blocks_per_cu()
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, viterbi_kernel, kThreads, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

}  // namespace nchmm
