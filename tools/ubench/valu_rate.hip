// valu_rate.hip -- VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction of
// v_fma_f32, v_pk_fma_f32, v_add_f32 + v_cndmask, at 1/2/4 waves per SIMD (one block per CU).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(float* out, int iters)
{
    float a[8], b = 1.0001f, c = 0.5f;
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8]; for (int i = 0; i < 8; ++i) p[i] = f2{a[i], a[i] + 1};
    f2 pb = {b, b}, pc = {c, c};
    unsigned long long m2;
    unsigned long long msk = 0x5555555555555555ull ^ (unsigned long long)iters;
    if (MODE == 16) asm volatile("s_mov_b64 vcc, %0" :: "s"(msk) : "vcc");
    long long t0 = clock64();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if (MODE == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 5) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (MODE == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if (MODE == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 8) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
                if (MODE == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (MODE == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 12) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 14) asm volatile("v_cmp_gt_f32 %1, %0, %2" : : "v"(a[i]), "s"(msk), "v"(b));
                if (MODE == 15) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %1, vcc" : "+v"(a[i]), "+v"(p[i].x) : "v"(b) : "vcc");
                if (MODE == 17) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if (MODE == 18) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c) : );
                if (MODE == 19) asm volatile("v_cmp_gt_f32 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[i]), "=&s"(m2) : "v"(b));
                if (MODE == 20) asm volatile("v_cmp_gt_f32 %2, %0, %3\n\tv_cndmask_b32_e64 %0, %0, %3, %2\n\tv_cndmask_b32_e64 %1, %1, %3, %2" : "+v"(a[i]), "+v"(p[i].x), "=&s"(m2) : "v"(b));
                if (MODE == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
                if (MODE == 21) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
                if (MODE == 22) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 23) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 24) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(a[i]));
                if (MODE == 25) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 26) asm volatile("v_cmp_eq_f32 %1, %0, %2" : : "v"(a[i]), "s"(msk), "v"(b));
                if (MODE == 27) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 28) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_cmp_gt_f32 %1, %0, %2" : "+v"(a[i]), "=&s"(m2) : "v"(b), "v"(c));
                if (MODE == 29) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (MODE == 30) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ((long long*)out)[400000] = t1 - t0;                                   // shader cycles (s_memtime)
        ((long long*)out)[400001] = (long long)(wall_clock64() - w0);          // 100 MHz constant clock
    }
}

// occupancy sweep: the same per-wave instruction stream at 4 / 8 resident waves per SIMD (grid = 256 or 512 blocks of 1024
// threads on 256 CUs).  If a wave could issue only every ~N cycles, throughput per SIMD would keep rising with occupancy.
template <int MODE> void sweep(const char* name, float* d)
{
    const int iters = 20000;
    for (int grid : {256, 512, 1024, 2048}) {
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr = (double)grid * 16 * iters * 32;
        printf("sweep %-14s grid=%4d x 1024 threads: %.3f ms  -> %.3f ns per wave-instruction per SIMD (1024 SIMDs)\n", name, grid, ms,
               ms * 1e6 / (wave_instr / 1024.0));
    }
}

template <int MODE> void run(const char* name, float* d, int threads)
{
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc2[2]; hipMemcpy(cyc2, (char*)d + 400000 * 8, 16, hipMemcpyDeviceToHost);
    const long long cyc = cyc2[0];
    const double ghz = cyc2[1] > 0 ? (double)cyc2[0] / (double)cyc2[1] * 0.1 : 0.0;
    double n_inst = (double)iters * 32;   // per wave
    int waves_per_simd = threads / 256;
    printf("%-18s w/SIMD=%d  %.2f ns/instr/SIMD  %.2f cyc/instr/wave  %.2f cyc/instr/SIMD  shader clock %.2f GHz\n", name, waves_per_simd,
           ms * 1e6 / (n_inst * waves_per_simd), (double)cyc / n_inst, (double)cyc / n_inst / waves_per_simd, ghz);
}

int main()
{
    float* d; hipMalloc(&d, 8 << 20);
    sweep<0>("v_fma_f32", d);
    sweep<2>("v_add_f32", d);
    sweep<8>("cndmask_e64", d);
    sweep<26>("v_cmp_eq->sgpr", d);
    sweep<4>("v_max3_f32", d);
    sweep<22>("v_exp_f32", d);
    sweep<1>("v_pk_fma_f32", d);
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32", d, threads);
        run<1>("v_pk_fma_f32", d, threads);
        run<2>("v_add_f32", d, threads);
        run<7>("v_mul_f32", d, threads);
        run<6>("v_pk_mul_f32", d, threads);
        run<3>("v_cndmask_b32", d, threads);
        run<4>("v_max3_f32", d, threads);
        run<5>("v_cmp_gt_f32", d, threads);
        run<8>("cndmask_e64_sgpr", d, threads);
        run<16>("cndmask_vcc_init", d, threads);
        run<17>("cndmask_e64_vcc", d, threads);
        run<18>("cndmask_nodep_vcc", d, threads);
        run<19>("cmp+cnd_sgpr", d, threads);
        run<20>("cmp+2cnd_sgpr", d, threads);
        run<9>("cmp+cndmask", d, threads);
        run<15>("cmp+2cndmask", d, threads);
        run<14>("v_cmp->sgpr", d, threads);
        run<10>("v_max_f32", d, threads);
        run<11>("v_mov_b32", d, threads);
        run<12>("v_lshl_or_b32", d, threads);
        run<13>("v_rcp_f32", d, threads);
        run<21>("v_add_f32_dpp", d, threads);
        run<22>("v_exp_f32", d, threads);
        run<30>("v_log_f32", d, threads);
        run<23>("v_and_b32", d, threads);
        run<24>("v_bfe_i32", d, threads);
        run<25>("v_min_f32", d, threads);
        run<26>("v_cmp_eq->sgpr", d, threads);
        run<27>("v_add_u32", d, threads);
        run<28>("fma+cmp->sgpr", d, threads);
        run<29>("v_mov_b32_dpp", d, threads);
    }
    return 0;
}
