// tables_kernel.hip -- device-side construction of scaled model images.
//
// Pore_Model_State::scale (src/nanocall/Pore_Model.hpp:126-138) on the fields the emission reads is a
// handful of IEEE mul/add operations plus logs that are ADDED (log var, log var_sd: computed once per
// model on the host by libm and passed in), so doing it on the GPU gives the same bits as the host
// loop while avoiding a 128 KiB upload per (read, model) job.  -ffp-contract=off; the reciprocals are
// correctly rounded fp32 divisions (== the host's double division rounded to float).
#include "nchmm_device.h"

#pragma clang fp contract(off)

namespace nchmm {

__global__ __launch_bounds__(256) void scale_models_kernel(const float* __restrict__ states,   // [n_tables][4096][10]
                                                           const int32_t* __restrict__ table_idx,
                                                           const float* __restrict__ params,   // [n][8]: 6 params, log var, log var_sd
                                                           float* __restrict__ models, int32_t* __restrict__ model_fast,
                                                           int first_slot, float log_2pi)
{
    const int k = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const float* p = params + 8 * k;
    const float* s = states + ((size_t)table_idx[k] * kStates + j) * 10;
    const float level_mean = s[0] * p[0] + p[1];
    const float level_stdv = s[1] * p[3];
    const float sd_mean = s[2] * p[4];
    const float sd_lambda = s[4] * p[5];
    const float log_level_stdv = s[6] + p[6];
    const float log_sd_lambda = s[9] + p[7];
    float* img = models + (size_t)(first_slot + k) * kModelFloats;
    img[MF_MU * kStates + j] = level_mean;
    img[MF_SIGMA * kStates + j] = level_stdv;
    img[MF_RSIGMA * kStates + j] = 1.0f / level_stdv;
    img[MF_NEG_LOG_SIGMA * kStates + j] = -log_level_stdv;
    img[MF_ETA * kStates + j] = sd_mean;
    img[MF_RETA * kStates + j] = 1.0f / sd_mean;
    img[MF_LAMBDA * kStates + j] = sd_lambda;
    img[MF_C * kStates + j] = log_sd_lambda - log_2pi;
    // same validated range as model_image_row (nchmm_api.cpp)
    auto in = [](float v, float lo, float hi) { return v >= lo && v <= hi; };   // false for NaN
    const bool ok = in(__builtin_fabsf(level_mean), 0x1p-10f, 0x1p20f) && in(level_stdv, 0x1p-10f, 0x1p10f)
                    && in(sd_mean, 0x1p-6f, 0x1p10f) && in(sd_lambda, 0x1p-10f, 0x1p14f);
    if (!ok) model_fast[first_slot + k] = 0;   // pre-set to 1 by the host
}

void launch_scale_models(const float* d_states, const int32_t* d_table_idx, const float* d_params, float* d_models,
                         int32_t* d_model_fast, int first_slot, size_t n, float log_2pi, hipStream_t stream)
{
    hipLaunchKernelGGL(scale_models_kernel, dim3(kStates / 256, (unsigned)n), dim3(256), 0, stream, d_states, d_table_idx, d_params,
                       d_models, d_model_fast, first_slot, log_2pi);
}

// State_Transitions::compute_transitions_fast for slot first_slot + k from the 64-entry mask -> log-weight
// table the host evaluated with libm (get_trans_prob is a function of the k-mer overlap mask only):
// Viterbi weights w0|w1|w2 by table look-up (bit-identical to the host expansion), FB per-state
// coefficients with the double-counted arcs of the low-complexity k-mers folded in (fb_weights in
// nchmm_api.cpp; tolerance-checked path, double exp/log on the device for those ~60 entries).
__global__ __launch_bounds__(256) void expand_transitions_kernel(const float* __restrict__ wm_all,     // [n][64]
                                                                 const uint8_t* __restrict__ masks,    // m0[4096] m1[1024] m2[256]
                                                                 float* __restrict__ trans, float* __restrict__ trans_fb,
                                                                 int first_slot)
{
    __shared__ float wm[64];
    const int k = blockIdx.x;
    if (threadIdx.x < 64) wm[threadIdx.x] = wm_all[64 * k + threadIdx.x];
    __syncthreads();
    const uint8_t* m0 = masks; const uint8_t* m1 = masks + kStates; const uint8_t* m2 = masks + kStates + 1024;
    float* w = trans + (size_t)(first_slot + k) * kTransFloats;
    float* fb = trans_fb + (size_t)(first_slot + k) * kFbTransFloats;
    auto fold = [](float lw0, float lw1, float lw2, bool stay_in_step, bool stay_in_skip, bool step_in_skip, float* c0, float* c1,
                   float* c2) {
        if (!stay_in_step && !stay_in_skip && !step_in_skip) { *c0 = lw0; *c1 = lw1; *c2 = lw2; return; }
        const double T0 = exp((double)lw0), W1 = exp((double)lw1), W2 = exp((double)lw2);
        *c0 = (float)log(T0 - (stay_in_step ? W1 : 0.0) - (stay_in_skip && !stay_in_step ? W2 : 0.0));
        *c1 = (float)log(W1 - (step_in_skip ? W2 : 0.0));
        *c2 = lw2;
    };
    for (unsigned j = threadIdx.x; j < (unsigned)kStates; j += 256) {
        const float w0 = wm[m0[j]];
        w[j] = w0;
        if (j < 1024) w[kStates + j] = wm[m1[j]];
        if (j < 256) w[kStates + 1024 + j] = wm[m2[j]];
        fold(w0, wm[m1[j >> 2]], wm[m2[j >> 4]], (j & 1023u) == (j >> 2), (j & 255u) == (j >> 4), ((j >> 2) & 255u) == (j >> 4),
             &fb[0 * kStates + j], &fb[1 * kStates + j], &fb[2 * kStates + j]);
        fold(w0, wm[m1[j & 1023u]], wm[m2[j & 255u]], (j >> 2) == (j & 1023u), (j >> 4) == (j & 255u),
             ((j & 1023u) >> 2) == (j & 255u), &fb[3 * kStates + j], &fb[4 * kStates + j], &fb[5 * kStates + j]);
    }
}

void launch_expand_transitions(const float* d_wm, const uint8_t* d_masks, float* d_trans, float* d_trans_fb, int first_slot,
                               size_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(expand_transitions_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_wm, d_masks, d_trans, d_trans_fb,
                       first_slot);
}

}  // namespace nchmm
