// VALU issue-rate microbenchmark: plain vs packed f32, f64, with 1..8 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    float2v A = {a, a}, B = {b, b};
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7;
    double da = a, db = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { // plain fma
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            } else if (MODE == 1) { // packed fma
                p0 = __builtin_elementwise_fma(p0, A, B); p1 = __builtin_elementwise_fma(p1, A, B);
                p2 = __builtin_elementwise_fma(p2, A, B); p3 = __builtin_elementwise_fma(p3, A, B);
                p4 = __builtin_elementwise_fma(p4, A, B); p5 = __builtin_elementwise_fma(p5, A, B);
                p6 = __builtin_elementwise_fma(p6, A, B); p7 = __builtin_elementwise_fma(p7, A, B);
            } else if (MODE == 2) { // plain add
                x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a;
            } else if (MODE == 3) { // packed add
                p0 += A; p1 += A; p2 += A; p3 += A; p4 += A; p5 += A; p6 += A; p7 += A;
            } else if (MODE == 4) { // f64 fma
                d0 = fma(d0, da, db); d1 = fma(d1, da, db); d2 = fma(d2, da, db); d3 = fma(d3, da, db);
                d4 = fma(d4, da, db); d5 = fma(d5, da, db); d6 = fma(d6, da, db); d7 = fma(d7, da, db);
            } else if (MODE == 5) { // f64 add
                d0 += da; d1 += da; d2 += da; d3 += da; d4 += da; d5 += da; d6 += da; d7 += da;
            }
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y
            + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, int wgs_per_cu, float *out)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * wgs_per_cu, 256>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<256 * wgs_per_cu, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)iters * 64 * wgs_per_cu;   // wave-instructions per SIMD (1 wave per SIMD per WG)
    printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wgs_per_cu, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("fma_f32", 1, out); run<1>("pk_fma_f32", 1, out); run<2>("add_f32", 1, out); run<3>("pk_add_f32", 1, out); run<4>("fma_f64", 1, out); run<5>("add_f64", 1, out); }
        if (w == 2) { run<0>("fma_f32", 2, out); run<1>("pk_fma_f32", 2, out); run<2>("add_f32", 2, out); run<3>("pk_add_f32", 2, out); run<4>("fma_f64", 2, out); run<5>("add_f64", 2, out); }
        if (w == 4) { run<0>("fma_f32", 4, out); run<1>("pk_fma_f32", 4, out); run<2>("add_f32", 4, out); run<3>("pk_add_f32", 4, out); run<4>("fma_f64", 4, out); run<5>("add_f64", 4, out); }
    }
    return 0;
}
