// fwbw_kernel.hip -- placeholder until the forward-backward kernel lands (next commit).
#include "nanocall_hip.h"
#include "nchmm_device.h"
namespace nchmm {
void launch_fwbw(const FwbwArgs&, int, hipStream_t) {}
int fwbw_blocks_per_cu() { return 1; }
}
extern "C" {
int nchmm_fwbw(nchmm_ctx*, size_t, const uint64_t*, const float*, const float*, const float*, const int32_t*,
               const int32_t*, const int32_t*, const float*, float*, float*, float*, float*, float*)
{ return NCHMM_E_INVALID; }
int nchmm_fwbw_dev(nchmm_ctx*, size_t, size_t, size_t, const uint64_t*, const float*, const float*, const float*,
                   const int32_t*, const int32_t*, const int32_t*, const float*, float*, float*, float*, float*, float*)
{ return NCHMM_E_INVALID; }
}
