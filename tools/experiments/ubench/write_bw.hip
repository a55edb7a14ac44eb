#include <hip/hip_runtime.h>
#include <cstdio>
// grid-stride float4
__global__ __launch_bounds__(256) void w_gs4(float4 *out, size_t n4)
{ const size_t st = (size_t)gridDim.x * 256; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += st) out[i] = make_float4(1, 2, 3, 4); }
// grid-stride dword
__global__ __launch_bounds__(256) void w_gs1(float *out, size_t n)
{ const size_t st = (size_t)gridDim.x * 256; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st) out[i] = 1.f; }
// one tile per block: V float4 per thread, tile contiguous
template <int V> __global__ __launch_bounds__(256) void w_tile4(float4 *out, size_t n4)
{
    size_t base = (size_t)blockIdx.x * 256 * V + threadIdx.x;
#pragma unroll
    for (int v = 0; v < V; ++v) { if (base + v * 256 < n4) out[base + v * 256] = make_float4(1, 2, 3, 4); }
}
template <int V> __global__ __launch_bounds__(256) void w_tile1(float *out, size_t n)
{
    size_t base = (size_t)blockIdx.x * 256 * V + threadIdx.x;
#pragma unroll
    for (int v = 0; v < V; ++v) { if (base + v * 256 < n) out[base + v * 256] = 1.f; }
}
// per-wave contiguous runs: each wave writes `run` consecutive KB-blocks (like the STFT kernel: 17 dword stores per 4112 B)
__global__ __launch_bounds__(256) void w_wave_runs(float *out, size_t n, int steps_per_wave)
{
    const int lane = threadIdx.x & 63; const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); const size_t nw = (size_t)gridDim.x * 4;
    for (size_t s = gw; s * 1028 + 1028 <= n; s += nw) {
        float *y = out + s * 1028;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[lane + 64 * i] = 1.f;
        y[1024 + (lane & 3)] = 2.f;
    }
}
// same runs, but every store instruction covers one 256-byte-aligned window of the output (18 per run)
__global__ __launch_bounds__(256) void w_wave_runs_aligned(float *out, size_t n)
{
    const int lane = threadIdx.x & 63; const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); const size_t nw = (size_t)gridDim.x * 4;
    for (size_t s = gw; s * 1028 + 1028 <= n; s += nw) {
        const size_t start = s * 1028, end = start + 1028;
        const size_t a0 = start & ~(size_t)63;
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const size_t idx = a0 + 64 * i + lane;
            if (idx >= start && idx < end) out[idx] = 1.f;
        }
    }
}
// runs of exactly 1024 floats (aligned), for comparison
__global__ __launch_bounds__(256) void w_wave_runs1024(float *out, size_t n)
{
    const int lane = threadIdx.x & 63; const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); const size_t nw = (size_t)gridDim.x * 4;
    for (size_t s = gw; s * 1024 + 1024 <= n; s += nw) {
        float *y = out + s * 1024;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[lane + 64 * i] = 1.f;
    }
}
// 1028-float runs, each wave starting at a different 256-byte piece of its run
template <int MUL>
__global__ __launch_bounds__(256) void w_wave_runs_rot(float *out, size_t n)
{
    const int lane = threadIdx.x & 63; const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); const size_t nw = (size_t)gridDim.x * 4;
    for (size_t s = gw; s * 1028 + 1028 <= n; s += nw) {
        float *y = out + s * 1028;
        const int rot = (int)((s * MUL) & 15);
#pragma unroll
        for (int i = 0; i < 16; ++i) y[lane + 64 * ((i + rot) & 15)] = 1.f;
        y[1024 + (lane & 3)] = 2.f;
    }
}
// block-cooperative: a block loops over groups of 4 runs (4112 floats) and its 4 waves interleave at 256 bytes
__global__ __launch_bounds__(256) void w_blk_runs(float *out, size_t n)
{
    for (size_t s = blockIdx.x; s * 4112 + 4112 <= n; s += gridDim.x) {
        float *y = out + s * 4112;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[threadIdx.x + 256 * i] = 1.f;
        if (threadIdx.x < 16) y[4096 + threadIdx.x] = 2.f;
    }
}
// same, one group per block, no loop
__global__ __launch_bounds__(256) void w_blk_runs_once(float *out, size_t n)
{
    const size_t s = blockIdx.x;
    if (s * 4112 + 4112 <= n) {
        float *y = out + s * 4112;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[threadIdx.x + 256 * i] = 1.f;
        if (threadIdx.x < 16) y[4096 + threadIdx.x] = 2.f;
    }
}
// per-wave runs, one run per wave, no loop
__global__ __launch_bounds__(256) void w_wave_runs_once(float *out, size_t n)
{
    const int lane = threadIdx.x & 63; const size_t s = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s * 1028 + 1028 <= n) {
        float *y = out + s * 1028;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[lane + 64 * i] = 1.f;
        y[1024 + (lane & 3)] = 2.f;
    }
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f(); hipEventRecord(e0); for (int i = 0; i < 10; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main()
{
    const size_t bytes = 1ull << 30, n4 = bytes / 16, n = bytes / 4;
    float4 *b; hipMalloc(&b, bytes);
    for (int g : {256, 512, 1024, 2048, 4096, 16384}) {
        float a1 = timeit([&] { w_gs4<<<g, 256>>>(b, n4); });
        float a2 = timeit([&] { w_gs1<<<g, 256>>>((float *)b, n); });
        float a3 = timeit([&] { w_wave_runs<<<g, 256>>>((float *)b, n, 0); });
        float a4 = timeit([&] { w_wave_runs_aligned<<<g, 256>>>((float *)b, n); });
        float a5 = timeit([&] { w_wave_runs1024<<<g, 256>>>((float *)b, n); });
        float q1 = timeit([&] { w_blk_runs<<<g, 256>>>((float *)b, n); });
        printf("grid %5d: blk-runs %.0f GB/s\n", g, bytes / q1 / 1e6);
        float r1 = timeit([&] { w_wave_runs_rot<1><<<g, 256>>>((float *)b, n); });
        float r5 = timeit([&] { w_wave_runs_rot<5><<<g, 256>>>((float *)b, n); });
        float r4 = timeit([&] { w_wave_runs_rot<4><<<g, 256>>>((float *)b, n); });
        printf("grid %5d: rot1 %.0f rot5 %.0f rot4 %.0f GB/s\n", g, bytes / r1 / 1e6, bytes / r5 / 1e6, bytes / r4 / 1e6);
        printf("grid %5d: gs float4 %.0f | gs dword %.0f | wave-runs 1028 %.0f | 1028 aligned-windows %.0f | runs 1024 %.0f GB/s\n", g, bytes / a1 / 1e6, bytes / a2 / 1e6, bytes / a3 / 1e6, bytes / a4 / 1e6, bytes / a5 / 1e6);
    }
    float o1 = timeit([&] { w_blk_runs_once<<<(unsigned)(n / 4112), 256>>>((float *)b, n); });
    float o2 = timeit([&] { w_wave_runs_once<<<(unsigned)(n / 4112), 256>>>((float *)b, n); });
    printf("once: blk-runs %.0f | wave-runs %.0f GB/s\n", bytes / o1 / 1e6, bytes / o2 / 1e6);
    float t1 = timeit([&] { w_tile4<1><<<(unsigned)(n4 / 256), 256>>>(b, n4); });
    float t4 = timeit([&] { w_tile4<4><<<(unsigned)(n4 / 1024), 256>>>(b, n4); });
    float t8 = timeit([&] { w_tile4<8><<<(unsigned)(n4 / 2048), 256>>>(b, n4); });
    float s4 = timeit([&] { w_tile1<4><<<(unsigned)(n / 1024), 256>>>((float *)b, n); });
    float s16 = timeit([&] { w_tile1<16><<<(unsigned)(n / 4096), 256>>>((float *)b, n); });
    printf("tile float4 V1 %.0f V4 %.0f V8 %.0f | tile dword V4 %.0f V16 %.0f GB/s\n", bytes / t1 / 1e6, bytes / t4 / 1e6, bytes / t8 / 1e6, bytes / s4 / 1e6, bytes / s16 / 1e6);
    return 0;
}
