// LDS instruction cost microbenchmark: per-wave private regions, wave-level sync, like the STFT kernel
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void wsync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *cyc)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, f = lane >> 4, j = lane & 15;
    float2 *ex = reinterpret_cast<float2 *>(smem) + wv * (4 * 272);
    float2 r[16];
    for (int i = 0; i < 16; ++i) { r[i].x = lane + i; r[i].y = lane - i; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 16 x ds_write_b64, row pitch 17 (pass-1 pattern)
            float2 *p = ex + f * 272 + j;
#pragma unroll
            for (int i = 0; i < 16; ++i) { p[i * 17] = r[i]; }
            wsync();
#pragma unroll
            for (int i = 0; i < 16; ++i) { r[i].x += 1.0f; }
        } else if (MODE == 1) {   // 16 x ds_read_b64 row reads (pass-2 pattern)
            const float2 *p = ex + f * 272 + j * 17;
#pragma unroll
            for (int i = 0; i < 16; ++i) { float2 t = p[i]; r[i].x += t.x; r[i].y += t.y; }
            wsync();
        } else if (MODE == 2) {   // write + read (full transpose)
            float2 *p = ex + f * 272 + j;
#pragma unroll
            for (int i = 0; i < 16; ++i) { p[i * 17] = r[i]; }
            wsync();
            const float2 *q = ex + f * 272 + j * 17;
#pragma unroll
            for (int i = 0; i < 16; ++i) { float2 t = q[i]; r[i].x += t.x; r[i].y -= t.y; }
            wsync();
        } else if (MODE == 3) {   // 16 x ds_write_b32
            float *p = reinterpret_cast<float *>(ex) + f * 544 + j;
#pragma unroll
            for (int i = 0; i < 16; ++i) { p[i * 16] = r[i].x; }
            wsync();
#pragma unroll
            for (int i = 0; i < 16; ++i) { r[i].x += 1.0f; }
        } else if (MODE == 4) {   // 4 x ds_write_b128 contiguous (commit pattern)
            float4 *p = reinterpret_cast<float4 *>(ex);
#pragma unroll
            for (int i = 0; i < 4; ++i) { p[i * 64 + lane] = make_float4(r[i].x, r[i].y, r[i + 4].x, r[i + 4].y); }
            wsync();
#pragma unroll
            for (int i = 0; i < 8; ++i) { r[i].x += 1.0f; }
        } else if (MODE == 5) {   // 200 VALU only
#pragma unroll
            for (int u = 0; u < 12; ++u) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { r[i].x = fmaf(r[i].x, 1.0001f, r[i].y); }
            }
        } else if (MODE == 6) {   // transpose + 200 VALU (overlap test)
            float2 *p = ex + f * 272 + j;
#pragma unroll
            for (int i = 0; i < 16; ++i) { p[i * 17] = r[i]; }
            wsync();
            const float2 *q = ex + f * 272 + j * 17;
#pragma unroll
            for (int i = 0; i < 16; ++i) { float2 t = q[i]; r[i].x += t.x; r[i].y -= t.y; }
            wsync();
#pragma unroll
            for (int u = 0; u < 12; ++u) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { r[i].x = fmaf(r[i].x, 1.0001f, r[i].y); }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) { s += r[i].x + r[i].y; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; }
}
template <int MODE> void run(const char *name, int wgs_per_cu, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    const size_t lds = 4 * 4 * 272 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * wgs_per_cu, 256, lds>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<MODE><<<256 * wgs_per_cu, 256, lds>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[64]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 64; ++i) { avg += (double)h[i]; } avg /= 64;
    printf("%-28s WG/CU=%d  %.3f ms  wave-cycles/iter=%.0f   CU-cycles/iter-per-wave=%.1f\n", name, wgs_per_cu, ms, avg / iters, avg / iters / (4.0 * wgs_per_cu));
}
int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 8 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    for (int w : {1, 2, 4}) {
        auto all = [&](auto wtag) {
            constexpr int W = decltype(wtag)::value;
            run<0>("16x write_b64 pitch17", W, out, cyc); run<1>("16x read_b64 rows", W, out, cyc); run<2>("transpose w+r", W, out, cyc);
            run<3>("16x write_b32", W, out, cyc); run<4>("4x write_b128", W, out, cyc); run<5>("192 fma", W, out, cyc); run<6>("transpose + 192 fma", W, out, cyc);
        };
        if (w == 1) all(std::integral_constant<int, 1>{});
        if (w == 2) all(std::integral_constant<int, 2>{});
        if (w == 4) all(std::integral_constant<int, 4>{});
    }
    return 0;
}
