// traffic model: runs of R consecutive steps per wave; reads 640 new floats per step (1024 on the first step
// of a run); writes the run's output as a stream of 256-byte aligned wave stores
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ALIGNED>
__global__ __launch_bounds__(256) void runs(float *__restrict__ out, const float *__restrict__ in, int R, int nruns, int N, int W)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int lo = (int)((long long)nruns * x / 8), hi = (int)((long long)nruns * (x + 1) / 8);
    const int stride = nwg_x * 4;
    const int rps = 250 / R;                       // runs per signal
    float acc = 0.f;
    for (int r = lo + (blockIdx.x >> 3) * 4 + wv; r < hi; r += stride) {
        const int b = r / rps, s0 = (r - b * rps) * R;
        const float *src = in + (long long)b * N + (long long)s0 * 640;
        float *dst = out + ((long long)b * W + (long long)s0 * 4) * 257;
        // first span: 1024 floats
        {
            const float4 *p = reinterpret_cast<const float4 *>(src);
            float4 v0 = p[lane], v1 = p[64 + lane], v2 = p[128 + lane], v3 = p[192 + lane];
            acc += v0.x + v1.y + v2.z + v3.w;
        }
        long long pos = 0;                          // floats written so far in this run
        const long long total = (long long)R * 1028;
        const long long base_mis = (reinterpret_cast<uintptr_t>(dst) >> 2) & 63;   // floats past a 256-B boundary
        if (ALIGNED && base_mis) {                  // head: up to the next boundary
            const int n = 64 - (int)base_mis;
            if (lane < n) dst[lane] = acc;
            pos = n;
        }
        for (int s = 0; s < R; ++s) {
            if (s + 1 < R) {                        // 640 new floats of the next step: 160 float4
                const float4 *p = reinterpret_cast<const float4 *>(src + 1024 + (long long)s * 640);
                float4 v0 = p[lane], v1 = p[64 + lane];
                float4 v2 = p[128 + (lane & 31)];
                acc += v0.x + v1.y + v2.z;
            }
            const long long end = (long long)(s + 1) * 1028;
            if (ALIGNED) {
                while (pos + 64 <= end) { dst[pos + lane] = acc; pos += 64; }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[pos + lane + 64 * i] = acc;
                dst[pos + 1024 + (lane & 3)] = acc;
                pos += 1028;
            }
        }
        if (ALIGNED && pos < total) { if (lane < total - pos) dst[pos + lane] = acc; }
    }
    if (acc == 12345.f) out[0] = acc;
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f(); hipEventRecord(e0); for (int i = 0; i < 10; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main()
{
    const int B = 1024, N = 160000, W = 1000;
    float *in, *out; hipMalloc(&in, (size_t)B * N * 4 + 4096); hipMalloc(&out, (size_t)B * W * 257 * 4);
    hipMemset(in, 0, (size_t)B * N * 4 + 4096);
    const double bytes = (double)B * N * 4 + (double)B * W * 257 * 4;
    for (int R : {1, 2, 5, 10, 25, 50, 250}) {
        const int nruns = B * (250 / R);
        for (int g : {256, 512, 1024}) {
            float a = timeit([&] { runs<1><<<g, 256>>>(out, in, R, nruns, N, W); });
            float b = timeit([&] { runs<0><<<g, 256>>>(out, in, R, nruns, N, W); });
            printf("R %3d grid %4d: aligned %.4f ms %.0f GB/s | unaligned %.4f ms %.0f GB/s\n", R, g, a, bytes / a / 1e6, b, bytes / b / 1e6);
        }
    }
    return 0;
}
