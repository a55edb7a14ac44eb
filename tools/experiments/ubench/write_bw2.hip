#include <hip/hip_runtime.h>
#include <cstdio>
// persistent per-wave runs of RUN floats; MAP selects which run a wave takes
// MAP 0: block-major (4 consecutive runs per block)   MAP 1: consecutive runs -> consecutive blocks
// MAP 2: 8 contiguous ranges by blockIdx%8, round-robin inside (the STFT kernel's scheme)
template <int MAP, int RUN>
__global__ void w_runs(float *out, size_t n)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const size_t nw = (size_t)gridDim.x * nwv, nruns = n / RUN;
    size_t first, stride = nw, lim = nruns;
    if (MAP == 0) first = (size_t)blockIdx.x * nwv + wv;
    else if (MAP == 1) first = (size_t)wv * gridDim.x + blockIdx.x;
    else { const int x = blockIdx.x & 7; const size_t lo = nruns * x / 8; lim = nruns * (x + 1) / 8; first = lo + (size_t)(blockIdx.x >> 3) * nwv + wv; stride = nw / 8; }
    for (size_t s = first; s < lim; s += stride) {
        float *y = out + s * RUN;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[lane + 64 * i] = 1.f;
        if (RUN > 1024) y[1024 + (lane & 3)] = 2.f;
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
    const size_t bytes = 1ull << 30, n = bytes / 4;
    float *b; hipMalloc(&b, bytes);
    for (int bs : {64, 256}) for (int g : {256, 1024, 4096}) {
        int gg = g * (256 / bs);
        float m0 = timeit([&] { w_runs<0, 1028><<<gg, bs>>>(b, n); });
        float m1 = timeit([&] { w_runs<1, 1028><<<gg, bs>>>(b, n); });
        float m2 = timeit([&] { w_runs<2, 1028><<<gg, bs>>>(b, n); });
        float a0 = timeit([&] { w_runs<0, 1024><<<gg, bs>>>(b, n); });
        float a1 = timeit([&] { w_runs<1, 1024><<<gg, bs>>>(b, n); });
        printf("block %3d grid %5d: 1028: blockmajor %.0f | spread %.0f | xcd-ranges %.0f || 1024: blockmajor %.0f | spread %.0f GB/s\n", bs, gg,
               bytes / m0 / 1e6, bytes / m1 / 1e6, bytes / m2 / 1e6, bytes / a0 / 1e6, bytes / a1 / 1e6);
    }
    return 0;
}
