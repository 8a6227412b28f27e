// cu_stream_rate.hip -- how fast ONE CU takes in a stream of 16 KiB rows (the shape of "emissions ahead": 1024 threads, one
// float4 per thread and row, rows consumed in order), by where the rows are and how many are in flight per thread.
//   hipcc --offload-arch=gfx950 -O3 cu_stream_rate.hip -o cu_stream_rate && ./cu_stream_rate
// Columns: rows in flight (a burst of that many row loads per thread, all waited for, then the next burst), GB/s per CU and ns per row for
//   L2      a 2 MiB ring read over and over (stays in the XCD's L2)
//   MALL    a 128 MiB buffer read twice, second pass timed (fits the 256 MiB memory-side cache)
//   HBM     a 4 GiB buffer read once
// with 1 block (one CU) and with 256 blocks (every CU streaming its own part).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const f4v* __restrict__ buf, size_t rows_per_block, size_t ring_rows, float* sink)
{
    const f4v* base = buf + (size_t)blockIdx.x * ring_rows * 1024 + threadIdx.x;
    f4v acc = {0, 0, 0, 0};
    // bursts of DEPTH rows: all asked for at once, all waited for, consumed, next burst -- DEPTH rows (x 16 KiB) in flight per CU
    // and nothing else going on, so GB/s = what the CU's load path delivers at that queue depth
    for (size_t i = 0; i + DEPTH <= rows_per_block; i += DEPTH) {
        f4v q[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) q[u] = __builtin_nontemporal_load(base + ((i + u) % ring_rows) * 1024);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) acc += q[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;          // (never true: keeps the loads)
}

template <int DEPTH>
static double run(const f4v* d, int blocks, size_t rows_per_block, size_t ring_rows, float* sink, int warm)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < warm; ++w) hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(blocks), dim3(1024), 0, 0, d, rows_per_block, ring_rows, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(blocks), dim3(1024), 0, 0, d, rows_per_block, ring_rows, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const size_t total_bytes = (size_t)4 << 30;
    f4v* d; float* sink;
    if (hipMalloc(&d, total_bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { std::puts("alloc failed"); return 1; }
    hipMemset(d, 0, total_bytes);
    const size_t row = 16384;
    std::printf("%-6s %-6s %6s %10s %10s\n", "where", "blocks", "depth", "GB/s/CU", "ns/row");
    struct Case { const char* name; size_t ring_rows; size_t rows; int warm; };
    auto sweep = [&](int blocks) {
        // per block: L2 ring 2 MiB / (blocks per XCD share) -> keep the ring at 64 rows (1 MiB) per block for 1 block, 8 rows for 256
        const Case cases[3] = {{"L2", blocks == 1 ? (size_t)64 : (size_t)8, 65536, 1},
                               {"MALL", (128u << 20) / row / blocks, (128u << 20) / row / blocks * (blocks == 1 ? 4 : 16), 1},
                               {"HBM", total_bytes / row / blocks, total_bytes / row / blocks, 0}};
        for (const Case& c : cases) {
#define ONE(D) { const double ms = run<D>(d, blocks, c.rows, c.ring_rows, sink, c.warm); \
                 std::printf("%-6s %-6d %6d %10.1f %10.1f\n", c.name, blocks, D, (double)c.rows * row / ms / 1e6, ms * 1e6 / (double)c.rows); }
            ONE(1) ONE(2) ONE(4) ONE(8) ONE(16)
#undef ONE
        }
    };
    sweep(1);
    sweep(256);
    return 0;
}
