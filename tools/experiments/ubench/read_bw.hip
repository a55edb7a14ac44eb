#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ __launch_bounds__(256) void rd(const float4 *__restrict__ in, float *out, size_t n4)
{
    float4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = in[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) { out[0] = 1; }
}
__global__ __launch_bounds__(256) void wr(float4 *out, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) { out[i] = make_float4(1, 2, 3, 4); }
}
template <int U>
__global__ __launch_bounds__(256) void cp(const float4 *__restrict__ in, float4 *out, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = in[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { out[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) { out[i] = in[i]; }
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main()
{
    const size_t bytes = 1ull << 30, n4 = bytes / 16;
    float4 *a, *b; float *o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
    hipMemset(a, 1, bytes);
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        float m1 = timeit([&] { rd<1><<<g, 256>>>(a, o, n4); });
        float m4 = timeit([&] { rd<4><<<g, 256>>>(a, o, n4); });
        float m8 = timeit([&] { rd<8><<<g, 256>>>(a, o, n4); });
        float mw = timeit([&] { wr<<<g, 256>>>(b, n4); });
        float c1 = timeit([&] { cp<1><<<g, 256>>>(a, b, n4); });
        float c4 = timeit([&] { cp<4><<<g, 256>>>(a, b, n4); });
        printf("grid %5d: read U1 %.0f U4 %.0f U8 %.0f GB/s | write %.0f GB/s | copy U1 %.0f U4 %.0f GB/s (r+w)\n", g,
               bytes / m1 / 1e6, bytes / m4 / 1e6, bytes / m8 / 1e6, bytes / mw / 1e6, 2.0 * bytes / c1 / 1e6, 2.0 * bytes / c4 / 1e6);
    }
    return 0;
}
