// What s_memtime counts: a wave alone on its SIMD issues dependent-free packed fmas (4 cycles each on gfx950,
// plain f32 ops 2) and reads the counter before and after; with the launch's wall time this gives the counter's
// rate and the shader clock under a pure vector load.
// build: hipcc -O3 --offload-arch=gfx950 clock_cal.hip -o clock_cal
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *ticks, int iters, float a, float b)
{
    float2v p[8];
    for (int i = 0; i < 8; ++i) { p[i] = float2v{(float)threadIdx.x + i, (float)i}; }
    const float2v A = {a, a}, B = {b, b};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (PK) { p[q] = __builtin_elementwise_fma(p[q], A, B); }
                else { p[q].x = fmaf(p[q].x, a, b); }
            }
        }
    }
    for (int i = 0; i < 8; ++i) { asm volatile("" :: "v"(p[i])); }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float r = 0;
    for (int i = 0; i < 8; ++i) { r += p[i].x + p[i].y; }
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; }
}
template <int PK> void run(const char *name, int wgs_per_cu, float *out, unsigned long long *ticks)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PK><<<256 * wgs_per_cu, 256>>>(out, ticks, 2000, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    k<PK><<<256 * wgs_per_cu, 256>>>(out, ticks, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double instr = (double)iters * 64;
    printf("%-8s waves/SIMD=%d  %.3f ms  %llu ticks: %.3f ticks per instruction of one wave, counter at %.1f MHz\n", name, wgs_per_cu, ms, t, t / instr, t / (ms * 1e3));
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long *ticks; (void)hipMalloc(&ticks, 8);
    for (int w : {1, 2}) { run<1>("pk_fma", w, out, ticks); run<0>("fma", w, out, ticks); }
    return 0;
}
