// Issue cost, in shader cycles (s_memtime, one wave per SIMD and two), of the cross-lane and select instructions the
// kernels use beside their packed arithmetic: v_permlane32_swap, v_permlane16_swap, v_mov_b32 with DPP, v_cndmask_b32,
// v_fmac_f32 with a DPP operand, v_pk_fma_f32, v_fma_f32, v_pk_mul_f32 followed by a dependent v_pk_fma_f32.
// build: hipcc -O3 --offload-arch=gfx950 xlane_rate.hip -o xlane_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *ticks, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 2, b1 = a1 * 2, b2 = a2 * 2, b3 = a3 * 2, b4 = a4 * 2, b5 = a5 * 2, b6 = a6 * 2, b7 = a7 * 2;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile(REP8("v_permlane32_swap_b32 %0, %8\n v_permlane32_swap_b32 %1, %9\n v_permlane32_swap_b32 %2, %10\n v_permlane32_swap_b32 %3, %11\n"
                              "v_permlane32_swap_b32 %4, %12\n v_permlane32_swap_b32 %5, %13\n v_permlane32_swap_b32 %6, %14\n v_permlane32_swap_b32 %7, %15\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        } else if (MODE == 1) {
            asm volatile(REP8("v_permlane16_swap_b32 %0, %8\n v_permlane16_swap_b32 %1, %9\n v_permlane16_swap_b32 %2, %10\n v_permlane16_swap_b32 %3, %11\n"
                              "v_permlane16_swap_b32 %4, %12\n v_permlane16_swap_b32 %5, %13\n v_permlane16_swap_b32 %6, %14\n v_permlane16_swap_b32 %7, %15\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        } else if (MODE == 2) {
            asm volatile(REP8("v_mov_b32_dpp %0, %8 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %9 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %10 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %11 row_mirror row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %12 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %13 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %14 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %15 row_mirror row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        } else if (MODE == 3) {
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %10, vcc\n v_cndmask_b32 %3, %3, %11, vcc\n"
                              "v_cndmask_b32 %4, %4, %12, vcc\n v_cndmask_b32 %5, %5, %13, vcc\n v_cndmask_b32 %6, %6, %14, vcc\n v_cndmask_b32 %7, %7, %15, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) :: "vcc");
        } else if (MODE == 4) {
            asm volatile(REP8("v_fmac_f32_dpp %0, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %9, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %10, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %11, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f32_dpp %4, %12, %13 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %13, %14 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %6, %14, %15 row_ror:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %15, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        } else if (MODE == 5) {
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %8\n v_pk_fma_f32 %1, %1, %5, %9\n v_pk_fma_f32 %2, %2, %6, %10\n v_pk_fma_f32 %3, %3, %7, %11\n"
                              "v_pk_fma_f32 %0, %0, %4, %8\n v_pk_fma_f32 %1, %1, %5, %9\n v_pk_fma_f32 %2, %2, %6, %10\n v_pk_fma_f32 %3, %3, %7, %11\n")
                         : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6), "+v"(*(double *)&b0), "+v"(*(double *)&b2), "+v"(*(double *)&b4), "+v"(*(double *)&b6),
                           "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6));
        } else if (MODE == 6) {
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %11, %12\n"
                              "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %13, %14\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %15, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        } else if (MODE == 7) {
            asm volatile(REP8("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n v_mov_b32 %4, %12\n v_mov_b32 %5, %13\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; }
}
template <int MODE> void run(const char *name, int wgs_per_cu, float *out, unsigned long long *ticks)
{
    const int iters = 2000;
    k<MODE><<<256 * wgs_per_cu, 256>>>(out, ticks, 100);
    k<MODE><<<256 * wgs_per_cu, 256>>>(out, ticks, iters);
    (void)hipDeviceSynchronize();
    unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-22s waves/SIMD=%d  %.2f cycles per instruction of one wave\n", name, wgs_per_cu, (double)t / ((double)iters * 64));
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long *ticks; (void)hipMalloc(&ticks, 8);
    for (int w : {1, 2}) {
        run<0>("v_permlane32_swap", w, out, ticks); run<1>("v_permlane16_swap", w, out, ticks); run<2>("v_mov_b32_dpp", w, out, ticks);
        run<3>("v_cndmask_b32", w, out, ticks); run<4>("v_fmac_f32_dpp", w, out, ticks); run<5>("v_pk_fma_f32", w, out, ticks);
        run<6>("v_fma_f32", w, out, ticks); run<7>("v_mov_b32", w, out, ticks);
    }
    return 0;
}
