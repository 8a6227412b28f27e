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

}  // namespace nchmm
