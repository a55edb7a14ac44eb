// memory-only model of the STFT fast kernel: per step a wave reads a 1024-float span (hop 640) and writes 1028 floats
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PREFETCH, int VIA_LDS, int RQ = 4>
__global__ __launch_bounds__(256) void mix(float *__restrict__ out, const float *__restrict__ in, int sps, int steps, int N, int W)
{
    __shared__ float lds[4][1028 + 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int lo = (int)((long long)steps * x / 8), hi = (int)((long long)steps * (x + 1) / 8);
    const int stride = nwg_x * 4;
    int s = lo + (blockIdx.x >> 3) * 4 + wv;
    if (s >= hi) return;
    auto src_of = [&](int sid) { int b = sid / sps; int w = (sid - b * sps) * 4; long long off = (long long)b * N + (long long)w * 160; if (off + 1024 > (long long)(b + 1) * N) off = (long long)(b + 1) * N - 1024; return in + off; };
    auto dst_of = [&](int sid) { int b = sid / sps; int w = (sid - b * sps) * 4; return out + ((long long)b * W + w) * 257; };
    float4 pre[4];
    const float4 *p = reinterpret_cast<const float4 *>(src_of(s));
#pragma unroll
    for (int q = 0; q < RQ; ++q) pre[q] = p[q * 64 + lane];
    while (true) {
        float4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = pre[q];
        const int ns = s + stride;
        if (PREFETCH && ns < hi) { const float4 *pn = reinterpret_cast<const float4 *>(src_of(ns));
#pragma unroll
            for (int q = 0; q < RQ; ++q) pre[q] = pn[q * 64 + lane]; }
        float *y = dst_of(s);
        if (VIA_LDS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<float4 *>(lds[wv])[q * 64 + lane] = cur[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < 16; ++i) y[lane + 64 * i] = lds[wv][(lane + 64 * i + 7) & 1023];
            y[1024 + (lane & 3)] = lds[wv][lane & 3];
        } else {
            const float *c = reinterpret_cast<const float *>(cur);
#pragma unroll
            for (int i = 0; i < 16; ++i) y[lane + 64 * i] = c[i];
            y[1024 + (lane & 3)] = c[3];
        }
        if (ns >= hi) break;
        if (!PREFETCH) { const float4 *pn = reinterpret_cast<const float4 *>(src_of(ns));
#pragma unroll
            for (int q = 0; q < RQ; ++q) pre[q] = pn[q * 64 + lane]; }
        s = ns;
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
    const int B = 1024, N = 160000, W = 1000, sps = 250, steps = B * sps;
    float *in, *out; hipMalloc(&in, (size_t)B * N * 4); hipMalloc(&out, (size_t)B * W * 257 * 4);
    hipMemset(in, 0, (size_t)B * N * 4);
    const double bytes = (double)B * N * 4 + (double)B * W * 257 * 4;
    for (int g : {256, 512, 1024, 2048}) {
        float a = timeit([&] { mix<1, 1><<<g, 256>>>(out, in, sps, steps, N, W); });
        float b = timeit([&] { mix<1, 0><<<g, 256>>>(out, in, sps, steps, N, W); });
        float c = timeit([&] { mix<0, 0><<<g, 256>>>(out, in, sps, steps, N, W); });
        float d = timeit([&] { mix<0, 0, 3><<<g, 256>>>(out, in, sps, steps, N, W); });
        float e = timeit([&] { mix<1, 0, 3><<<g, 256>>>(out, in, sps, steps, N, W); });
        printf("grid %4d: 768-float reads: no prefetch %.4f ms %.0f GB/s | prefetch %.4f ms\n", g, d, bytes / d / 1e6, e);
        printf("grid %4d: prefetch+lds %.4f ms %.0f GB/s | prefetch %.4f ms %.0f | no prefetch %.4f ms %.0f GB/s\n", g, a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6);
    }
    return 0;
}
