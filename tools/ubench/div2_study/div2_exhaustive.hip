// div2_exhaustive.hip -- for which binary32 divisors is the TWO-operation division by a known divisor exact?
//
//     q = fma(n, zh, RN(n * zl)),   zh = RN(1 / d),   zl = RN(1 / d - zh)        (Brisebarre, Muller, Raina 2004)
//
// For most divisors it returns RN(n / d) for every numerator; for about 1.4 % of the divisor significands exactly one
// numerator significand comes out one ulp off.  This tool settles it by enumeration on the GPU, all 2^23 divisor
// significands x all 2^23 numerator significands (exponents and signs do not enter while nothing leaves the normal range):
//   pass 1  per divisor, the number of numerators on which the formula differs from the 3-operation Markstein quotient
//           (itself proven equal to IEEE division on all 2^46 pairs: profiles/r01_markstein_exhaustive.txt);
//   pass 2  for every divisor that fails, the same count with zl moved by -1, +1, -2, +2 ulps,
//           then zh one or two floats below / above RN(1/d) with its own zl (0, -1, +1, -2, +2 ulps): the first variant with NO
//           failing numerator is that divisor's pair.
//   pass 3  for the divisors no variant rescues: the one numerator significand their PLAIN pair fails on (the kernel uses the plain
//           pair for them and tests every numerator against it: a handful of operations instead of a third one per division).
// Output: div2_table.bin.nstar -- uint32 pairs (divisor significand, failing numerator significand), sorted; and
//         div2_table.bin -- sorted uint32 entries (significand << 8) | code for the divisors whose plain pair does not work,
// code 1..24 = the variant (variant_pair below) that does, 255 = none does (such a divisor keeps the 3-operation division).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off div2_exhaustive.hip -o div2_exhaustive ; ./div2_exhaustive [out.bin]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#pragma clang fp contract(off)
#ifndef WIDE_VARIANTS
#define WIDE_VARIANTS 0
#endif
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float sig(uint32_t m) { return __builtin_bit_cast(float, (127u << 23) | m); }
__device__ __forceinline__ float ulps(float v, int k) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) + (uint32_t)k); }   // moves |v| by k ulps

__device__ __forceinline__ uint32_t count_fails(float d, float zh, float zl)
{
    const float r = 1.0f / d;      // the reference needs the CORRECTLY ROUNDED reciprocal, whatever zh the candidate pair uses
    uint32_t f = 0;
    for (uint32_t nm = 0; nm < (1u << 23); ++nm) {
        const float n = sig(nm);
        const float q2 = __builtin_fmaf(n, zh, n * zl);
        const float q0 = n * r;
        const float q3 = __builtin_fmaf(__builtin_fmaf(-q0, d, n), r, q0);
        f += (__builtin_bit_cast(uint32_t, q2) != __builtin_bit_cast(uint32_t, q3));
    }
    return f;
}

__global__ __launch_bounds__(256) void pass1(uint32_t first, uint8_t* fails)
{
    const uint32_t m = first + blockIdx.x * 256u + threadIdx.x;
    const float d = sig(m);
    const float zh = 1.0f / d;
    const float zl = __builtin_fmaf(-zh, d, 1.0f) / d;
    const uint32_t f = count_fails(d, zh, zl);
    fails[m] = (uint8_t)(f > 255u ? 255u : f);
}

// Variant v = 1 .. kVariants of the pair for divisor d (0 is the plain pair): zh = RN(1/d) moved by dzh ulps, zl = RN(1/d - zh)
// moved by dzl ulps along the value line, (dzh, dzl) in the order below -- closest to the plain pair first.
constexpr int kVariants = WIDE_VARIANTS ? 62 : 24;
__device__ __forceinline__ void variant_offsets(unsigned v, int& dzh, int& dzl)
{
#if WIDE_VARIANTS
    // exploration only: zh 0, -1, +1, ... -4, +4 (9) x zl 0, -1, +1, ... -3, +3 (7), minus the plain pair
    const int zh_order[9] = {0, -1, 1, -2, 2, -3, 3, -4, 4};
    const int zl_order[7] = {0, -1, 1, -2, 2, -3, 3};
    dzh = zh_order[v / 7u];
    dzl = zl_order[v % 7u];
#else
    // v = 1..4: zh as is, zl -1, +1, -2, +2; then zh -1, +1, -2, +2 each with zl 0, -1, +1, -2, +2
    const int order[5] = {0, -1, 1, -2, 2};
    dzh = order[v / 5u];
    dzl = order[v % 5u];
#endif
}
__device__ __forceinline__ void variant_pair(float d, unsigned v, float& zh, float& zl)
{
    int dzh = 0, dzl = 0;
    if (v) variant_offsets(v, dzh, dzl);
    zh = ulps(1.0f / d, dzh);                   // (1/d > 0: bit pattern order = value order)
    zl = __builtin_fmaf(-zh, d, 1.0f) / d;      // the residual 1 - zh d is exact for zh within a few ulps of 1/d
    if (dzl != 0) zl = zl >= 0.0f ? ulps(zl, dzl) : ulps(zl, -dzl);     // dzl ulps along the value line
}

__global__ __launch_bounds__(64) void pass2(const uint32_t* bad, uint32_t n_bad, uint8_t* code)
{
    // one wave per divisor: lane v - 1 tries variant v
    const uint32_t i = blockIdx.x;
    if (i >= n_bad) return;
    const uint32_t v = threadIdx.x + 1u;
    uint32_t f = 1;
    if (v <= (uint32_t)kVariants) {
        const float d = sig(bad[i]);
        float zh, zl;
        variant_pair(d, v, zh, zl);
        f = count_fails(d, zh, zl);
    }
    const unsigned long long ok = __builtin_amdgcn_ballot_w64(f == 0);
    if (threadIdx.x == 0) code[i] = ok ? (uint8_t)(__builtin_ctzll(ok) + 1) : 255;
}

// pass 3: the ONE numerator significand on which the plain pair of a pair-less divisor fails (pass 1 counted exactly one)
__global__ __launch_bounds__(256) void pass3(const uint32_t* none, uint32_t n_none, uint32_t* nstar)
{
    // one block per divisor, the numerators split over its 256 threads
    const uint32_t i = blockIdx.x;
    if (i >= n_none) return;
    const float d = sig(none[i]);
    const float zh = 1.0f / d, r = zh;
    const float zl = __builtin_fmaf(-zh, d, 1.0f) / d;
    for (uint32_t nm = threadIdx.x; nm < (1u << 23); nm += 256u) {
        const float n = sig(nm);
        const float q2 = __builtin_fmaf(n, zh, n * zl);
        const float q0 = n * r;
        const float q3 = __builtin_fmaf(__builtin_fmaf(-q0, d, n), r, q0);
        if (__builtin_bit_cast(uint32_t, q2) != __builtin_bit_cast(uint32_t, q3)) nstar[i] = nm;
    }
}

int main(int argc, char** argv)
{
    const char* out = argc > 1 ? argv[1] : "div2_table.bin";
    uint8_t* d_fails;
    CHECK(hipMalloc(&d_fails, 1u << 23));
    const uint32_t per_launch = 1u << 17;
    for (uint32_t first = 0; first < (1u << 23); first += per_launch) {
        hipLaunchKernelGGL(pass1, dim3(per_launch / 256), dim3(256), 0, 0, first, d_fails);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<uint8_t> fails(1u << 23);
    CHECK(hipMemcpy(fails.data(), d_fails, fails.size(), hipMemcpyDeviceToHost));
    std::vector<uint32_t> bad;
    unsigned long long total = 0, hist[4] = {0, 0, 0, 0};
    for (uint32_t m = 0; m < (1u << 23); ++m) {
        if (fails[m]) { bad.push_back(m); total += fails[m]; }
        ++hist[fails[m] > 2 ? 3 : fails[m]];
    }
    printf("pass 1: %zu of 8388608 divisor significands fail (%.4f %%), %llu failing pairs; per divisor 0: %llu, 1: %llu, 2: %llu, >2: %llu\n", bad.size(),
           100.0 * bad.size() / 8388608.0, total, hist[0], hist[1], hist[2], hist[3]);
    uint32_t* d_bad; uint8_t* d_code;
    CHECK(hipMalloc(&d_bad, bad.size() * 4 + 4)); CHECK(hipMalloc(&d_code, bad.size() + 1));
    CHECK(hipMemcpy(d_bad, bad.data(), bad.size() * 4, hipMemcpyHostToDevice));
    for (size_t first = 0; first < bad.size(); first += 16384) {
        const uint32_t n = (uint32_t)std::min<size_t>(16384, bad.size() - first);
        hipLaunchKernelGGL(pass2, dim3(n), dim3(64), 0, 0, d_bad + first, n, d_code + first);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<uint8_t> code(bad.size());
    CHECK(hipMemcpy(code.data(), d_code, code.size(), hipMemcpyDeviceToHost));
    unsigned long long by_code[256] = {0};
    std::vector<uint32_t> table(bad.size());
    for (size_t i = 0; i < bad.size(); ++i) { ++by_code[code[i]]; table[i] = (bad[i] << 8) | code[i]; }
    printf("pass 2: fixed by variant");
    for (int c = 1; c <= kVariants; ++c) printf(" %d: %llu", c, by_code[c]);
    printf("; by none: %llu\n", by_code[255]);
    FILE* f = fopen(out, "wb");
    if (!f || fwrite(table.data(), 4, table.size(), f) != table.size()) { printf("cannot write %s\n", out); return 1; }
    fclose(f);
    printf("wrote %s: %zu entries (significand << 8 | code), sorted\n", out, table.size());
    // pass 3 -> <out>.nstar: (divisor significand, failing numerator significand) for the pair-less divisors, sorted by divisor
    std::vector<uint32_t> none;
    for (size_t i = 0; i < bad.size(); ++i) if (code[i] == 255) none.push_back(bad[i]);
    uint32_t *d_none, *d_nstar;
    CHECK(hipMalloc(&d_none, none.size() * 4 + 4)); CHECK(hipMalloc(&d_nstar, none.size() * 4 + 4));
    CHECK(hipMemcpy(d_none, none.data(), none.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_nstar, 0xFF, none.size() * 4));
    hipLaunchKernelGGL(pass3, dim3((uint32_t)none.size()), dim3(256), 0, 0, d_none, (uint32_t)none.size(), d_nstar);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> nstar(none.size());
    CHECK(hipMemcpy(nstar.data(), d_nstar, nstar.size() * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> pairs;
    size_t missing = 0;
    for (size_t i = 0; i < none.size(); ++i) { pairs.push_back(none[i]); pairs.push_back(nstar[i]); missing += nstar[i] == 0xFFFFFFFFu; }
    const std::string out2 = std::string(out) + ".nstar";
    FILE* g = fopen(out2.c_str(), "wb");
    if (!g || fwrite(pairs.data(), 4, pairs.size(), g) != pairs.size()) { printf("cannot write %s\n", out2.c_str()); return 1; }
    fclose(g);
    printf("pass 3: wrote %s: %zu pair-less divisors with the numerator significand their plain pair fails on (%zu without one)\n", out2.c_str(), none.size(),
           missing);
    return 0;
}
