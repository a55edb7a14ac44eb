// Memory-only model of the nfft-4096 run form: 8 waves per workgroup, one workgroup per CU; a wave works through runs of 32
// frames of one signal, per frame it loads one group of 1024 floats (eight 8-byte-per-lane wave loads, a frame ahead) and stores a
// row of 2049 floats.  STW: bytes per lane and store instruction (4: 33 dword stores per row as the kernel does; 16: 16-byte
// stores where the row's alignment allows, dword stores for the rest).  PITCH: floats between rows (2049, or 2048 to see what
// aligned rows would give).  Nothing is computed: what comes out is what the memory system sustains for this pattern.
// build: hipcc -O3 --offload-arch=gfx950 mix5.hip -o mix5 ; run: ./mix5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float v4 __attribute__((ext_vector_type(4)));

template <int STW, int PITCH, int NTL>
__global__ __launch_bounds__(512, 2) void run_form(float *__restrict__ Y, const float *__restrict__ X, int B, int W, long long N, unsigned *take)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int rps = (W + 31) / 32, units = B * rps;
    const int f_lo = (int)((long long)units * xcd / 8), f_hi = (int)((long long)units * (xcd + 1) / 8);
    const int fstride = nwg_x * 8;
    int rid = f_lo + (blockIdx.x >> 3) * 8 + wv;
    while (rid < f_hi) {
        unsigned took = 0;
        if (lane == 0) { took = atomicAdd(take + xcd, 1u); }
        const int b = rid / rps, w0 = (rid - b * rps) * 32, len = min(32, W - w0);
        const float *src = X + (long long)b * N + (long long)w0 * 1024 + 2 * lane;
        v2 pre[8];
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) { pre[r] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v2 *>(src + 128 * r)) : *reinterpret_cast<const v2 *>(src + 128 * r); }
        for (int t = 0; t < len; ++t) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc += pre[r].x + pre[r].y; }
            if (t + 1 < len) {
                const float *s2 = src + 1024 * (t + 1);
#pragma unroll
                for (int r = 0; r < 8; ++r) { pre[r] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v2 *>(s2 + 128 * r)) : *reinterpret_cast<const v2 *>(s2 + 128 * r); }
            }
            float *y = Y + ((long long)b * W + w0 + t) * PITCH;
            if constexpr (STW == 4) {
#pragma unroll
                for (int i = 0; i < 32; ++i) { y[64 * i + lane] = acc; }
                y[2048] = acc;
            } else {
                // 16-byte stores from the first 16-byte boundary of the row on; the 0..3 floats before it and what is left behind as dword stores
                const int head = (int)((16 - ((size_t)y & 15)) & 15) >> 2;
                v4 *yv = reinterpret_cast<v4 *>(y + head);
#pragma unroll
                for (int i = 0; i < 8; ++i) { yv[64 * i + lane] = v4{acc, acc, acc, acc}; }
                const int done = head + 2048;
                if (lane < head) { y[lane] = acc; }
                if (lane < PITCH - done) { y[done + lane] = acc; }
            }
        }
        rid = f_lo + fstride + (int)__builtin_amdgcn_readfirstlane(took);
    }
}

template <int STW, int PITCH, int NTL> static void go(const char *name, float *Y, const float *X, unsigned *take, int B, int W, long long N)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 30; ++rep) {
        hipMemsetAsync(take, 0, 32);
        hipEventRecord(a);
        hipLaunchKernelGGL((run_form<STW, PITCH, NTL>), dim3(256), dim3(512), 0, 0, Y, X, B, W, N, take);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 10) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)B * N * 4 + (double)B * W * 2049 * 4;
    printf("%-52s mean %.3f ms  min %.3f ms  %.2f TB/s\n", name, sum / 20, best, bytes / (sum / 20) / 1e9);
}

int main()
{
    const int B = 8; const long long N = 172800000; const int W = (int)((N - 4096) / 1024 + 1);
    float *X, *Y; unsigned *take;
    hipMalloc(&X, (size_t)B * N * 4); hipMalloc(&Y, (size_t)B * W * 2049 * 4 + 64); hipMalloc(&take, 32);
    hipMemset(X, 0, (size_t)B * N * 4);
    go<4, 2049, 1>("dword stores, pitch 2049 (the kernel's pattern)", Y, X, take, B, W, N);
    go<4, 2049, 0>("  the same, plain loads", Y, X, take, B, W, N);
    go<4, 2048, 1>("dword stores, pitch 2048", Y, X, take, B, W, N);
    go<16, 2049, 1>("16-byte stores, pitch 2049", Y, X, take, B, W, N);
    go<16, 2048, 1>("16-byte stores, pitch 2048", Y, X, take, B, W, N);
    return 0;
}
