// Fast path of the 512-point fused STFT kernel for gfx950: integer hop with 3*hop + 512 <= 1024,
// even hop, frames addressed as start = ss0 + w*hop (the stft framing, and the stft_flt framing when
// its hop is integral).  Same arithmetic as the basic kernel in r512_kernel.hip; what differs is the
// schedule around it:
//   * the span of the next 4 frames is fetched from HBM into registers (16-byte loads) while the
//     current 4 frames are transformed, and is committed to LDS at the top of the next step, so the
//     HBM latency is off the critical path; every input sample is fetched once per run of frames;
//   * the staged span and the 16 x 16 exchange tile share one LDS region (8.7 KB per wave in
//     float), which lets four workgroups share a CU;
//   * the compile-time shape (rows of the window that hold data, mean removal, sqrt/log epilogue)
//     removes the per-element selects and branches of the general kernel.
#include "stft_common.h"
#include "fft16.h"

namespace dsp {

struct R512FastArgs {
    long long ldx;        // elements between signals
    int N;                // samples per signal
    int L;                // window length (NR rows of 32 hold data)
    int W;                // frames per signal
    int stp;              // hop
    int ss0;              // start of frame 0
    int rd_shift;         // edge == 1: samples are read at n - rd_shift
    int edge;             // 0 reflect, 1 zero
    int amp, lg;
    int chunk, nchunk;    // frames per run, runs per signal
    int total_chunks;     // B * nchunk
};

constexpr int R512F_WAVE_ELEMS = 4 * R512_EXF * 2;   // T elements of LDS per wave (>= 1024 staged samples)

template <typename T> struct vec16;                   // 16-byte vector of T
template <> struct vec16<float> { using type = float4; static constexpr int n = 4; };
template <> struct vec16<double> { using type = double2; static constexpr int n = 2; };

template <typename T, int NR, bool MN0, bool EPI>
__global__ __launch_bounds__(256) void stft_r512_fast_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                              const T *__restrict__ win, const T *__restrict__ tw,
                                                              const R512FastArgs a)
{
    using T2 = typename vec2<T>::type;
    using TV = typename vec16<T>::type;
    constexpr int VEC = vec16<T>::n, NV = 16 / VEC;        // 16 staged elements per lane
    constexpr bool WINREG = (sizeof(T) == 4);
    constexpr int M = 256, F = 257;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane >> 4, j = lane & 15;
    const int L = a.L;

    T *xs = reinterpret_cast<T *>(smem_raw) + wv * R512F_WAVE_ELEMS;   // staged span; aliases the exchange tile
    T2 *ex = reinterpret_cast<T2 *>(xs);
    T *wtab = reinterpret_cast<T *>(smem_raw) + 4 * R512F_WAVE_ELEMS;  // halved window (double only)

    // ---- per-lane constants ----
    T twr[16], twi[16];                                    // W256^(j k1) = W512^(2 j k1)
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        int idx = 2 * j * k1;
        T sg = T(1);
        if (idx >= 256) { idx -= 256; sg = T(-1); }
        twr[k1] = sg * tw[2 * idx]; twi[k1] = sg * tw[2 * idx + 1];
    }
    const T ow0r = tw[2 * lane], ow0i = tw[2 * lane + 1];
    const T ow1r = tw[2 * (lane + 64)], ow1i = tw[2 * (lane + 64) + 1];
    T hw0[NR], hw1[NR];                                    // halved window taps of this lane's columns
    if (WINREG) {
#pragma unroll
        for (int n1 = 0; n1 < NR; ++n1) {
            const int l = 32 * n1 + 2 * j;
            hw0[n1] = (l < L) ? T(0.5) * win[l] : T(0);
            hw1[n1] = (l + 1 < L) ? T(0.5) * win[l + 1] : T(0);
        }
    } else {
        for (int l = tid; l < 512; l += 256) { wtab[l] = (l < L) ? T(0.5) * win[l] : T(0); }
        __syncthreads();
    }
    const int zm0 = (M - lane) & (M - 1);                  // partner bins of lane and lane + 64
    const int zm1 = M - 64 - lane;
    const int need = 3 * a.stp + 512;
    const T *xrow = xs + f * a.stp + 2 * j;

    // ---- fetch of one span (frames w..w+3 of the signal at xb) into registers ----
    TV pre[NV];
    auto load_span = [&](const T *xb, const int w) {
        const int s0 = a.ss0 + w * a.stp;
        const T *src = xb + (s0 - (a.edge ? a.rd_shift : 0));
        const bool inside = s0 >= 0 && s0 + 1024 <= a.N && (a.edge == 0 || s0 >= a.rd_shift);
        if (inside && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
#pragma unroll
            for (int q = 0; q < NV; ++q) { pre[q] = reinterpret_cast<const TV *>(src)[q * 64 + lane]; }
        } else {
            FrameDesc d{};
            d.N = a.N; d.edge = a.edge;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                T t[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const int i = (q * 64 + lane) * VEC + v;
                    t[v] = (i < need) ? fetch_sample(xb, d, (long long)s0, (long long)a.rd_shift, i) : T(0);
                }
                if constexpr (VEC == 4) { pre[q] = TV{t[0], t[1], t[2], t[3]}; }
                else { pre[q] = TV{t[0], t[1]}; }
            }
        }
    };

    int cid = blockIdx.x * 4 + wv;
    if (cid >= a.total_chunks) { return; }
    const int cstride = gridDim.x * 4;
    int b = cid / a.nchunk;
    int w = (cid - b * a.nchunk) * a.chunk;
    int wend = min(w + a.chunk, a.W);
    const T *xb = X + (long long)b * a.ldx;
    T *yb = Y + (long long)b * a.W * F;
    load_span(xb, w);

    while (true) {
        // ---- commit the fetched span to LDS, then start fetching the next one ----
#pragma unroll
        for (int q = 0; q < NV; ++q) { reinterpret_cast<TV *>(xs)[q * 64 + lane] = pre[q]; }
        wave_sync();
        int nw = w + 4, nwend = wend, ncid = cid;
        const T *nxb = xb;
        T *nyb = yb;
        bool more = true;
        if (nw >= wend) {
            ncid = cid + cstride;
            more = ncid < a.total_chunks;
            if (more) {
                const int nb = ncid / a.nchunk;
                nw = (ncid - nb * a.nchunk) * a.chunk;
                nwend = min(nw + a.chunk, a.W);
                nxb = X + (long long)nb * a.ldx;
                nyb = Y + (long long)nb * a.W * F;
            }
        }
        if (more) { load_span(nxb, nw); }

        // ---- pass 1: lane (frame f, column n2 = j): 16-point FFT over the rows n1 ----
        T re[16], im[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            if (n1 < NR) {
                const T2 t = *reinterpret_cast<const T2 *>(xrow + 32 * n1);
                T p = t.x, c = t.y;
                if (NR == 16 || n1 == NR - 1) {          // rows that may reach past the window
                    const int l = 32 * n1 + 2 * j;
                    p = (l < L) ? p : T(0);
                    c = (l + 1 < L) ? c : T(0);
                }
                T h0, h1;
                if (WINREG) { h0 = hw0[n1]; h1 = hw1[n1]; }
                else { const T2 h = *reinterpret_cast<const T2 *>(wtab + 32 * n1 + 2 * j); h0 = h.x; h1 = h.y; }
                re[n1] = p * h0; im[n1] = c * h1;
            } else {
                re[n1] = T(0); im[n1] = T(0);
            }
        }
        if (MN0) {
            T sum = T(0);
#pragma unroll
            for (int n1 = 0; n1 < NR; ++n1) { sum += re[n1] + im[n1]; }
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
            const T mean = sum / T(L);
#pragma unroll
            for (int n1 = 0; n1 < NR; ++n1) {
                const int l = 32 * n1 + 2 * j;
                re[n1] -= (l < L) ? mean : T(0);
                im[n1] -= (l + 1 < L) ? mean : T(0);
            }
        }
        dft16<T, NR>(re, im);
        wave_sync();                                       // all lanes have read the span: the tile may overwrite it
        {
            T2 *exf = ex + f * R512_EXF + j;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                T r = re[slot16(k1)], i = im[slot16(k1)];
                if (k1 > 0) { cmul(r, i, twr[k1], twi[k1]); }
                T2 o; o.x = r; o.y = i;
                exf[k1 * 17] = o;
            }
        }
        wave_sync();

        // ---- pass 2: lane (frame f, output residue k1 = j): 16-point FFT over n2 ----
        {
            const T2 *exf = ex + f * R512_EXF + j * 17;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) { const T2 t = exf[n2]; re[n2] = t.x; im[n2] = t.y; }
        }
        dft16<T, 16>(re, im);
        wave_sync();
        {
            T2 *zf = ex + f * R512_EXF + j;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                T2 o; o.x = re[slot16(k2)]; o.y = im[slot16(k2)];
                zf[16 * k2] = o;                           // Z[k1 + 16 k2]
            }
        }
        wave_sync();

        // ---- real-FFT split fused with power; lane handles bins k and 256 - k of every frame ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (w + g < wend) {
                const T2 *zf = ex + g * R512_EXF;
                T *y = yb + (unsigned)((w + g) * F);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int k = lane + 64 * m;
                    const T2 z = zf[k], zm = zf[m ? zm1 : zm0];
                    const T wr = m ? ow1r : ow0r, wi = m ? ow1i : ow0i;
                    const T ar = z.x + zm.x, ai = z.y - zm.y, br = z.x - zm.x, bi = z.y + zm.y;
                    const T tr = br * wi + bi * wr, ti = bi * wi - br * wr;
                    const T pr = ar + tr, pi = ai + ti, qr = ar - tr, qi = ai - ti;
                    T p = pr * pr + pi * pi, q = qr * qr + qi * qi;
                    if (EPI) { p = power_epilogue(p, a.amp, a.lg); q = power_epilogue(q, a.amp, a.lg); }
                    y[k] = p;
                    y[M - k] = q;
                }
                if (lane == 0) {
                    const T2 z = zf[128];                  // bin 128 pairs with itself (Z holds half values)
                    T p = T(4) * (z.x * z.x + z.y * z.y);
                    if (EPI) { p = power_epilogue(p, a.amp, a.lg); }
                    y[128] = p;
                }
            }
        }
        if (!more) { break; }
        wave_sync();                                       // tile reads are done before the next span lands
        cid = ncid; w = nw; wend = nwend; xb = nxb; yb = nyb;
    }
}

// Conditions of the fast path: frames at ss0 + w*hop, hop even with 3*hop + 512 <= 1024, and every
// index of one signal's input and output fits 31 bits.
bool r512_fast_supported(const FrameDesc &d)
{
    return d.tbl == nullptr && d.nfft == 512 && d.L >= 1 && d.L <= 512 && !(d.stp & 1) && 3ll * d.stp + 512 <= 1024 &&
           d.N < (1ll << 31) - 4096 && (long long)d.W * 257 < (1ll << 30) && d.ss0 > -(1ll << 30) && d.ss0 < (1ll << 30) &&
           d.rd_shift >= 0 && d.rd_shift < (1ll << 30);
}

template <typename T, int NR, bool MN0, bool EPI>
static hipError_t launch_fast_inst(T *Y, const T *X, const T *win, const T *tw, const R512FastArgs &a, hipStream_t s)
{
    const size_t lds = (4 * (size_t)R512F_WAVE_ELEMS + (sizeof(T) == 4 ? 0 : 512)) * sizeof(T);
    static size_t granted = 0;
    (void)hipGetLastError();
    if (lds > 48 * 1024 && lds > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&stft_r512_fast_kernel<T, NR, MN0, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { return e; }
        granted = lds;
    }
    long long grid = ((long long)a.total_chunks + 3) / 4;
    if (grid > 256 * 4) { grid = 256 * 4; }
    hipLaunchKernelGGL((stft_r512_fast_kernel<T, NR, MN0, EPI>), dim3((unsigned)grid), dim3(256), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_stft_r512_fast(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    R512FastArgs a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.amp = d.amp; a.lg = d.lg;
    // runs of consecutive frames per wave: long enough to amortise the per-wave constants and the
    // first (unhidden) fetch, short enough that every wave slot of the chip gets several runs
    const long long frames = (long long)d.B * d.W;
    long long chunk = frames / (256ll * 16 * 8);
    chunk = chunk < 8 ? 8 : (chunk > 64 ? 64 : chunk);
    chunk &= ~3ll;
    a.chunk = (int)chunk;
    a.nchunk = (int)((d.W + chunk - 1) / chunk);
    a.total_chunks = d.B * a.nchunk;
    const bool epi = d.amp || d.lg;
    const bool nr13 = d.L > 384 && d.L <= 416;
#define DSP_R512F(NR_, MN0_, EPI_) launch_fast_inst<T, NR_, MN0_, EPI_>(Y, X, win, tw, a, s)
    if (nr13) {
        if (d.mn0) { return epi ? DSP_R512F(13, true, true) : DSP_R512F(13, true, false); }
        return epi ? DSP_R512F(13, false, true) : DSP_R512F(13, false, false);
    }
    if (d.mn0) { return epi ? DSP_R512F(16, true, true) : DSP_R512F(16, true, false); }
    return epi ? DSP_R512F(16, false, true) : DSP_R512F(16, false, false);
#undef DSP_R512F
}

template hipError_t launch_stft_r512_fast<float>(float *, const float *, const float *, const float *, const FrameDesc &, hipStream_t);
template hipError_t launch_stft_r512_fast<double>(double *, const double *, const double *, const double *, const FrameDesc &, hipStream_t);

}  // namespace dsp
