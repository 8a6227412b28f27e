// nchmm_probe.h -- launcher of the shader-clock probe (probe_kernel.hip)
#ifndef NCHMM_PROBE_H
#define NCHMM_PROBE_H
#include <hip/hip_runtime.h>
namespace nchmm {
// d_out: 3 x u64 {shader-clock ticks, wall-clock ticks, unused}
void launch_clock_probe(unsigned long long* d_out, int grid, int iters, hipStream_t stream);
}
#endif
