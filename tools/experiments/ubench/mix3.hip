// Memory-only models of the 512-point STFT kernel's traffic on the benchmark shape (1024 signals x 160000
// floats, hop 160, 4 frames per step: a 1024-float span in, 1028 floats out per step), 16-byte accesses only.
//   mode 0: span one step ahead in registers, committed to LDS, rows read back and stored (the kernel's form)
//   mode 1: span D steps ahead by LDS-DMA (global_load_lds_dwordx4) into a ring of D buffers per wave
//   mode 2: as mode 1, rows stored straight from registers that were read from the span buffer (no second LDS pass)
//   mode 3: reference stream: aligned 1 KiB wave reads and writes in the same byte ratio, grid-stride
// build: hipcc -O3 --offload-arch=gfx950 mix3.hip -o mix3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void *g, void *l)
{
    // hidden from the compiler's wait counting: completion is waited for by the counted s_waitcnt below
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)l);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(la) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MODE, int D, int NT>
__global__ __launch_bounds__(256) void mix(float *__restrict__ out, const float *__restrict__ in, int sps, int steps, int N, int W)
{
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int lo = (int)((long long)steps * x / 8), hi = (int)((long long)steps * (x + 1) / 8);
    const int stride = nwg_x * 4;
    int s = lo + (blockIdx.x >> 3) * 4 + wv;
    if (s >= hi) return;
    auto src_of = [&](int sid) {
        int b = sid / sps; int w = (sid - b * sps) * 4;
        long long off = (long long)b * N + (long long)w * 160;
        if (off + 1024 > (long long)(b + 1) * N) off = (long long)(b + 1) * N - 1024;
        return in + off;
    };
    auto dst_of = [&](int sid) { int b = sid / sps; int w = (sid - b * sps) * 4; return out + ((long long)b * W + w) * 257; };
    auto store_rows = [&](float *y, const v4 *r) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int idx = (i < 4) ? lane + 64 * i : 256;
            if (NT) __builtin_nontemporal_store(r[i], reinterpret_cast<v4 *>(y) + idx);
            else reinterpret_cast<v4 *>(y)[idx] = r[i];
        }
    };
    if constexpr (MODE == 0) {
        float *xs = lds + wv * 1280;
        v4 pre[4];
        const v4 *p = reinterpret_cast<const v4 *>(src_of(s));
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = p[q * 64 + lane];
        while (true) {
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<v4 *>(xs)[q * 64 + lane] = pre[q];
            wsync();
            const int ns = s + stride;
            if (ns < hi) {
                const v4 *pn = reinterpret_cast<const v4 *>(src_of(ns));
#pragma unroll
                for (int q = 0; q < 4; ++q) pre[q] = pn[q * 64 + lane];
            }
            v4 r[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) r[i] = reinterpret_cast<const v4 *>(xs)[(lane + 64 * i + 3) & 255];
            store_rows(dst_of(s), r);
            wsync();
            if (ns >= hi) break;
            s = ns;
        }
    } else if constexpr (MODE == 1 || MODE == 2) {
        float *ring = lds + wv * (D * 1024 + 1280);
        float *rows = ring + D * 1024;
        // prologue: D spans in flight
        int sq = s;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float *src = src_of(sq < hi ? sq : s);
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16(src + (q * 64 + lane) * 4, ring + d * 1024 + q * 256);
            sq += stride;
        }
        // pad the queue so that the counted wait below is exact from the first iteration on
        int d = 0;
        int it = 0;
        while (true) {
            // operations younger than the span of this step: the D-1 spans and D-1 steps of stores after it, and
            // the stores of the step before it (the spans of the prologue have only spans behind them)
            if (it < D) { wait_vm<4 * (D - 1)>(); } else { wait_vm<9 * D - 4>(); }
            wsync();
            float *buf = ring + d * 1024;
            v4 r[5];
            if constexpr (MODE == 1) {
                v4 t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = reinterpret_cast<const v4 *>(buf)[q * 64 + lane];
                wsync();
#pragma unroll
                for (int q = 0; q < 4; ++q) reinterpret_cast<v4 *>(rows)[q * 64 + lane] = t[q];
                wsync();
#pragma unroll
                for (int i = 0; i < 5; ++i) r[i] = reinterpret_cast<const v4 *>(rows)[(lane + 64 * i + 3) & 255];
            } else {
#pragma unroll
                for (int i = 0; i < 5; ++i) r[i] = reinterpret_cast<const v4 *>(buf)[(lane + 64 * i + 3) & 255];
            }
            wsync();
            // refill this buffer with the span D steps ahead (clamped: the number of operations per step stays fixed)
            {
                const int fs = s + D * stride;
                const float *src = src_of(fs < hi ? fs : s);
#pragma unroll
                for (int q = 0; q < 4; ++q) glds16(src + (q * 64 + lane) * 4, buf + q * 256);
            }
            store_rows(dst_of(s), r);
            ++it;
            const int ns = s + stride;
            if (ns >= hi) break;
            s = ns;
            d = (d + 1 == D) ? 0 : d + 1;
        }
        wait_vm<0>();
    } else {
        // reference stream: 1 KiB chunks; chunk c of the output, and chunk c * 160/257 of the input when that index steps
        const long long nout = (long long)steps * 1028 / 256, nin = (long long)steps * 640 / 256;   // 1 KiB chunks
        const long long gw = (long long)blockIdx.x * 4 + wv, nw = (long long)gridDim.x * 4;
        v4 acc = {0.f, 0.f, 0.f, 0.f};
        for (long long c = gw; c < nout; c += nw) {
            const long long ci = c * nin / nout;
            if (c == 0 || ci != (c - nw) * nin / nout) { acc += reinterpret_cast<const v4 *>(in)[ci * 64 + lane]; }
            if (NT) __builtin_nontemporal_store(acc, reinterpret_cast<v4 *>(out) + c * 64 + lane);
            else reinterpret_cast<v4 *>(out)[c * 64 + lane] = acc;
        }
    }
}

template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 30; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < 30; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 30;
}

template <int MODE, int D, int NT>
void run(const char *name, float *out, const float *in, int g)
{
    const int B = 1024, N = 160000, W = 1000, sps = 250, steps = B * sps;
    const double bytes = (double)B * N * 4 + (double)B * W * 257 * 4;
    const size_t lds = (MODE == 0 ? 4 * 1280 : MODE == 3 ? 0 : 4 * (D * 1024 + 1280)) * sizeof(float);
    auto k = mix<MODE, D, NT>;
    if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = timeit([&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), lds, 0, out, in, sps, steps, N, W); });
    hipError_t e = hipDeviceSynchronize();
    printf("%-28s grid %4d lds %6zu: %.4f ms %5.0f GB/s %s\n", name, g, lds, ms, bytes / ms / 1e6, e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
}

int main()
{
    const int B = 1024, N = 160000, W = 1000;
    float *in, *out;
    hipMalloc(&in, (size_t)B * N * 4 + 65536); hipMalloc(&out, (size_t)B * W * 257 * 4 + 65536);
    hipMemset(in, 0, (size_t)B * N * 4 + 65536);
    // warm the clocks
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((mix<3, 1, 1>), dim3(2048), dim3(256), 0, 0, out, in, 250, B * 250, N, W);
    hipDeviceSynchronize();
    for (int g : {512, 768, 1024, 2048}) {
        run<3, 1, 1>("stream nt", out, in, g * 2);
        run<3, 1, 0>("stream plain", out, in, g * 2);
        run<0, 1, 1>("regs+lds nt", out, in, g);
        run<0, 1, 0>("regs+lds plain", out, in, g);
        run<1, 1, 1>("glds D1 +rows nt", out, in, g);
        run<1, 2, 1>("glds D2 +rows nt", out, in, g);
        run<1, 3, 1>("glds D3 +rows nt", out, in, g);
        run<2, 2, 1>("glds D2 direct nt", out, in, g);
        run<2, 3, 1>("glds D3 direct nt", out, in, g);
        run<2, 3, 0>("glds D3 direct plain", out, in, g);
        if (g <= 1024) run<2, 5, 1>("glds D5 direct nt", out, in, g);
    }
    return 0;
}
