// valu_rate.hip -- how many cycles does one wave64 VALU instruction cost a SIMD on this chip?
// Launches enough waves to fill every SIMD with W waves, each running a long stream of independent
// v_fma_f32 (and, separately, v_mul_lo_u32 / v_rcp_f32 / v_cndmask / v_readlane), and reports
// instructions per cycle per SIMD from wall time and the shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND> __global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x * 3 + 1, u1 = u0 + 7, u2 = u0 + 9, u3 = u0 + 11;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %6, %6, %6, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 1) {
                asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            } else if (KIND == 2) {
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (KIND == 3) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (KIND == 4) {
                asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"s20", "s21", "s22", "s23");
            } else if (KIND == 6) {
                asm volatile("v_cmp_gt_u64 vcc, %0, %1\n v_cmp_gt_u64 vcc, %1, %0\n v_cmp_gt_u64 vcc, %0, %1\n v_cmp_gt_u64 vcc, %1, %0"
                             : "+v"(*(double *)&a0), "+v"(*(double *)&a2)::"vcc");
            } else if (KIND == 7) {
                asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0"
                             : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6));
            } else if (KIND == 8) {
                asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %2\n v_cmp_le_f32 vcc, %2, %3\n v_cmp_le_f32 vcc, %3, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");
            } else if (KIND == 9) {
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %2, %3, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %3, %0, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %0, %1, %3 op_sel_hi:[1,0,0]"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (KIND == 10) {
                asm volatile("v_min3_f32 %0, %1, %2, %0\n v_max3_f32 %1, %2, %3, %1\n v_med3_f32 %2, %3, %0, %2\n v_min3_f32 %3, %0, %1, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (KIND == 11) {
                asm volatile("v_sqrt_f32 %0, %0\n v_rsq_f32 %1, %1\n v_sin_f32 %2, %2\n v_cos_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (KIND == 12) {
                asm volatile("v_cvt_f32_u32 %0, %1\n v_cvt_f32_u32 %2, %3\n v_xor_b32 %1, %1, %3\n v_lshrrev_b32 %3, 15, %3"
                             : "+v"(a0), "+v"(u0), "+v"(a1), "+v"(u1));
            } else if (KIND == 5) {
                asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %1, %1, %1, %0" : "+v"(*(double *)&a0), "+v"(*(double *)&a2));
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + u2 + u3 + (float)(t1 - t0);
}

template <int KIND> void run(const char *name, int per_iter, int waves_per_simd)
{
    const int iters = 4096;
    const int blocks = 256 * waves_per_simd; // 256 CUs x (waves_per_simd blocks of 4 waves)
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 16, 1.0f);
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)iters * 16 * per_iter * waves_per_simd;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           cycles / insts_per_simd);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 4, w);
    }
    for (int w : {1, 4, 8}) run<5>("v_pk_fma_f32", 2, w);
    for (int w : {1, 4}) run<1>("v_mul_lo_u32", 4, w);
    for (int w : {1, 4}) run<2>("v_rcp_f32", 4, w);
    for (int w : {1, 4}) run<3>("cmp+cndmask", 4, w);
    for (int w : {1, 4}) run<4>("v_readlane", 4, w);
    for (int w : {4, 8}) run<6>("v_cmp_gt_u64", 4, w);
    for (int w : {4, 8}) run<7>("v_mov_b64", 4, w);
    for (int w : {4, 8}) run<8>("v_cmp_le_f32", 4, w);
    for (int w : {4, 8}) run<9>("v_fma_mix_f32", 4, w);
    for (int w : {4, 8}) run<10>("min3/max3/med3", 4, w);
    for (int w : {4, 8}) run<11>("sqrt/rsq/sin/cos", 4, w);
    for (int w : {4, 8}) run<12>("cvt/xor/shift", 4, w);
    return 0;
}
