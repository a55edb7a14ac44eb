// Memory-only models, second set: what the chip sustains for the benchmark's read/write mix, and whether a
// run-based deal (a wave works through consecutive steps of one signal: every sample fetched once, rows
// leave as 1-KiB-aligned wave stores with the remainder carried in LDS) moves more bytes per second than
// the round-robin deal of the kernel (a 1024-float span in, an unaligned 4112-byte block out per step).
// build: hipcc -O3 --offload-arch=gfx950 mix4.hip -o mix4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void *g, unsigned la)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(la) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const void *)p);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- reference streams: KIND 0 read, 1 write, 2 copy 1:1, 3 mix 5 KiB in : 8 KiB out per iteration ----
template <int KIND, int NT>
__global__ __launch_bounds__(256) void ref(float *__restrict__ out, const float *__restrict__ in, int iters_total)
{
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    constexpr int RI = KIND == 1 ? 0 : KIND == 3 ? 5 : 8, WO = KIND == 0 ? 0 : 8;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = gw; it < iters_total; it += nw) {
        const v4 *src = reinterpret_cast<const v4 *>(in) + (size_t)it * RI * 64 + lane;
        v4 *dst = reinterpret_cast<v4 *>(out) + (size_t)it * WO * 64 + lane;
        v4 r[RI > 0 ? RI : 1];
#pragma unroll
        for (int i = 0; i < RI; ++i) { r[i] = NT ? __builtin_nontemporal_load(src + 64 * i) : src[64 * i]; }
#pragma unroll
        for (int i = 0; i < RI; ++i) { acc += r[i]; }
#pragma unroll
        for (int i = 0; i < WO; ++i) {
            const v4 v = (KIND == 2) ? r[i] : acc;
            if (NT) __builtin_nontemporal_store(v, dst + 64 * i); else dst[64 * i] = v;
        }
    }
    if (KIND == 0 && acc.x == 12345.f) { out[0] = acc.y; }
}

// ---- round-robin deal as in the kernel (glds ring of D spans); TAIL 0: fifth store by all lanes to one
// address (the kernel's form), 1: fifth store by lane 0 only, 2: no fifth store; SPANV: 16-byte vectors of a span fetched (256 or 220)
template <int D, int TAIL, int SPANV>
__global__ __launch_bounds__(256) void rr(float *__restrict__ out, const float *__restrict__ in, int sps, int steps, int N, int W)
{
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    float *ring = lds + wv * (D * 1024);
    auto fetch = [&](const float *src, float *buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int v = q * 64 + lane;
            glds16(src + (v < SPANV ? v : SPANV - 1) * 4, lds_addr(buf + q * 256));
        }
    };
    int sq = s;
#pragma unroll
    for (int d = 0; d < D; ++d) { fetch(src_of(sq < hi ? sq : s), ring + d * 1024); sq += stride; }
    int d = 0, it = 0;
    constexpr int NST = TAIL == 2 ? 4 : 5;
    while (true) {
        if (it < D) { wait_vm<4 * (D - 1)>(); } else { wait_vm<(4 + NST) * D - 4>(); }
        wsync();
        float *buf = ring + d * 1024;
        v4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = reinterpret_cast<const v4 *>(buf)[(lane + 64 * i + 3) & 255];
        wsync();
        { const int fs = s + D * stride; fetch(src_of(fs < hi ? fs : s), buf); }
        v4 *y = reinterpret_cast<v4 *>(dst_of(s));
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(r[i], y + lane + 64 * i);
        if (TAIL == 0) { __builtin_nontemporal_store(r[0], y + 256); }
        if (TAIL == 1) { if (lane == 0) { __builtin_nontemporal_store(r[0], y + 256); } }
        ++it;
        const int ns = s + stride;
        if (ns >= hi) break;
        s = ns;
        d = (d + 1 == D) ? 0 : d + 1;
    }
    wait_vm<0>();
}

// ---- run-based deal: a wave takes runs of R consecutive steps of one signal.  Per step 640 new floats
// (2.5 KiB: two full wave loads and a half one) come in through a ring of LDS-DMA pieces, and the rows
// (4112 bytes per step) leave as 1-KiB-aligned wave stores, what is left over waiting for the next step.
template <int DP>   // DP: steps of loads in flight
__global__ __launch_bounds__(256) void runs(float *__restrict__ out, const float *__restrict__ in, int R, int nruns, int N, int W)
{
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int lo = (int)((long long)nruns * x / 8), hi = (int)((long long)nruns * (x + 1) / 8);
    const int stride = nwg_x * 4;
    const int rps = (250 + R - 1) / R;
    float *ring = lds + wv * (DP * 768 + 2048);      // DP pieces of 3 KiB (2.5 used), then 8 KiB of row staging
    float *rows = ring + DP * 768;
    for (int r = lo + (blockIdx.x >> 3) * 4 + wv; r < hi; r += stride) {
        const int b = r / rps, s0 = (r - b * rps) * R;
        const int len = min(R, 250 - s0);
        const float *src = in + (long long)b * N + (long long)s0 * 640;
        char *dst = reinterpret_cast<char *>(out + ((long long)b * W + (long long)s0 * 4) * 257);
        auto fetch = [&](int st, int slot) {       // the 640 floats that step st adds
            const float *p = src + 384 + (long long)st * 640;
            glds16(p + lane * 4, lds_addr(ring + slot * 768));
            glds16(p + 256 + lane * 4, lds_addr(ring + slot * 768 + 256));
            glds16(p + 512 + (lane & 31) * 4, lds_addr(ring + slot * 768 + 512));
        };
        // head of the run: the 384 floats before the first step's own (a wave load and a half)
        {
            const v4 h0 = reinterpret_cast<const v4 *>(src)[lane], h1 = reinterpret_cast<const v4 *>(src)[64 + (lane & 31)];
            reinterpret_cast<v4 *>(rows)[lane] = h0; reinterpret_cast<v4 *>(rows)[64 + lane] = h1;
        }
#pragma unroll
        for (int dstp = 0; dstp < DP; ++dstp) { fetch(dstp < len ? dstp : 0, dstp); }
        int pend = 0;                                // bytes staged and not yet stored
        long long done = 0;                          // bytes of the run stored so far
        const int head = (int)((1024 - (reinterpret_cast<uintptr_t>(dst) & 1023)) & 1023);
        bool head_done = (head == 0);
        int slot = 0;
        for (int st = 0; st < len; ++st) {
            // behind this step's loads: DP - 1 steps of loads and DP steps of stores (4 or 5 each: counted as 4,
            // which waits for at most DP operations more than needed)
            wait_vm<3 * (DP - 1) + 4 * DP>();
            wsync();
            v4 t0 = reinterpret_cast<const v4 *>(ring + slot * 768)[lane];
            v4 t1 = reinterpret_cast<const v4 *>(ring + slot * 768)[64 + lane];
            wsync();
            fetch(st + DP < len ? st + DP : st, slot);
            reinterpret_cast<v4 *>(rows)[(lane + 7) & 255] = t0 + t1;
            wsync();
            pend += 4112;
            if (!head_done && pend >= head) {
                if (lane < head / 16) { __builtin_nontemporal_store(reinterpret_cast<const v4 *>(rows)[lane], reinterpret_cast<v4 *>(dst + done) + lane); }
                done += head; pend -= head; head_done = true;
            }
            while (head_done && pend >= 1024) {
                __builtin_nontemporal_store(reinterpret_cast<const v4 *>(rows)[(lane + (int)(done >> 4)) & 511], reinterpret_cast<v4 *>(dst + done) + lane);
                done += 1024; pend -= 1024;
            }
            slot = (slot + 1 == DP) ? 0 : slot + 1;
        }
        if (pend > 0) {
            if (lane < pend / 16) { __builtin_nontemporal_store(reinterpret_cast<const v4 *>(rows)[lane], reinterpret_cast<v4 *>(dst + done) + lane); }
        }
        wait_vm<0>();
        wsync();
    }
}

template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 30; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < 30; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 30;
}
static void report(const char *name, int g, size_t lds, float ms, double bytes)
{
    hipError_t e = hipDeviceSynchronize();
    printf("%-34s grid %5d lds %6zu: %.4f ms %5.0f GB/s %s\n", name, g, lds, ms, bytes / ms / 1e6, e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
}
template <int KIND, int NT> void run_ref(const char *name, float *out, const float *in, int g)
{
    const double outb = 1024.0 * 1000 * 257 * 4;
    const int iters = (int)(outb / 8192);
    const double bytes = (double)iters * ((KIND == 1 ? 0 : KIND == 3 ? 5 : 8) + (KIND == 0 ? 0 : 8)) * 1024.0;
    float ms = timeit([&] { hipLaunchKernelGGL((ref<KIND, NT>), dim3(g), dim3(256), 0, 0, out, in, iters); });
    report(name, g, 0, ms, bytes);
}
template <int D, int TAIL, int SPANV> void run_rr(const char *name, float *out, const float *in, int g)
{
    const int B = 1024, N = 160000, W = 1000, sps = 250, steps = B * sps;
    const double bytes = (double)B * N * 4 + (double)B * W * 257 * 4;
    const size_t lds = 4 * (D * 1024) * sizeof(float);
    auto k = rr<D, TAIL, SPANV>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = timeit([&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), lds, 0, out, in, sps, steps, N, W); });
    report(name, g, lds, ms, bytes);
}
template <int DP> void run_runs(const char *name, float *out, const float *in, int g, int R)
{
    const int B = 1024, N = 160000, W = 1000;
    const double bytes = (double)B * N * 4 + (double)B * W * 257 * 4;
    const int nruns = B * ((250 + R - 1) / R);
    const size_t lds = 4 * (DP * 768 + 2048) * sizeof(float);
    auto k = runs<DP>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = timeit([&] { hipLaunchKernelGGL(k, dim3(g), dim3(256), lds, 0, out, in, R, nruns, N, W); });
    char nm[64]; snprintf(nm, sizeof nm, "%s R=%d", name, R);
    report(nm, g, lds, ms, bytes);
}

int main()
{
    const int B = 1024, N = 160000, W = 1000;
    float *in, *out;
    hipMalloc(&in, (size_t)B * W * 257 * 4 + 65536); hipMalloc(&out, (size_t)B * W * 257 * 4 + 65536);
    hipMemset(in, 0, (size_t)B * W * 257 * 4 + 65536);
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((ref<2, 1>), dim3(2048), dim3(256), 0, 0, out, in, 128000);
    hipDeviceSynchronize();
    for (int g : {1024, 2048, 4096}) {
        run_ref<0, 0>("ref read", out, in, g);
        run_ref<0, 1>("ref read nt", out, in, g);
        run_ref<1, 0>("ref write", out, in, g);
        run_ref<1, 1>("ref write nt", out, in, g);
        run_ref<2, 0>("ref copy", out, in, g);
        run_ref<2, 1>("ref copy nt", out, in, g);
        run_ref<3, 0>("ref mix 5:8", out, in, g);
        run_ref<3, 1>("ref mix 5:8 nt", out, in, g);
    }
    for (int g : {256, 512, 1024}) {
        run_rr<2, 0, 256>("rr D2 tail-all", out, in, g);
        run_rr<2, 1, 256>("rr D2 tail-lane0", out, in, g);
        run_rr<2, 2, 256>("rr D2 no-tail", out, in, g);
        run_rr<2, 0, 220>("rr D2 tail-all span220", out, in, g);
        run_rr<4, 0, 256>("rr D4 tail-all", out, in, g);
        run_rr<6, 0, 256>("rr D6 tail-all", out, in, g);
        run_rr<6, 2, 256>("rr D6 no-tail", out, in, g);
    }
    for (int g : {256, 512, 1024}) {
        for (int R : {10, 25, 50, 125}) {
            run_runs<2>("runs DP2", out, in, g, R);
            run_runs<4>("runs DP4", out, in, g, R);
        }
    }
    return 0;
}
