// Register-resident 512-point fused STFT kernel for gfx950 (see generic_kernels.hip for the other
// kernel families and the overall description).
//
// nfft == 512.  16 lanes per frame, 4 frames per wavefront.  The 512 real samples of a frame are
// packed as 256 complex points z_n = x_{2n} + i x_{2n+1} and transformed as 16 x 16 (four-step):
// lane n2 runs a 16-point FFT over n1 in registers, multiplies by W256^(n2 k1), the 16 x 16 tile is
// transposed through LDS, lane k1 runs the second 16-point FFT, and the real-FFT split is fused with
// the power step.  Each wave walks a run of consecutive frames of one signal, stages the span of its
// 4 frames in LDS once per step and stores spectrogram rows as whole 256-byte wave stores.
#include "stft_common.h"
#include "fft16.h"

namespace dsp {


template <typename T, bool WINREG>
__global__ __launch_bounds__(256) void stft_r512_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                         const T *__restrict__ win, const T *__restrict__ tw,
                                                         const FrameDesc d, const int xs_len, const int span_mode,
                                                         const int chunk, const int nchunk)
{
    using T2 = typename vec2<T>::type;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int M = 256, F = 257;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = lane >> 4, j = lane & 15;
    const int L = d.L;

    // LDS: [window table, 512 T, only when !WINREG] then per wave [xs: xs_len T][ex: 4*EXF complex]
    T *wtab = reinterpret_cast<T *>(smem_raw);
    T *wave_base = wtab + (WINREG ? 0 : 512) + (size_t)wv * ((size_t)xs_len + 4 * R512_EXF * 2);
    T *xs = wave_base;
    T2 *ex = reinterpret_cast<T2 *>(wave_base + xs_len);

    // per-lane constants: pass-1 twiddles W256^(j k1) = W512^(2 j k1), output twiddles W512^k
    T twr[16], twi[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        int idx = 2 * j * k1;
        T sg = T(1);
        if (idx >= 256) { idx -= 256; sg = T(-1); }
        twr[k1] = sg * tw[2 * idx]; twi[k1] = sg * tw[2 * idx + 1];
    }
    const T ow0r = tw[2 * lane], ow0i = tw[2 * lane + 1];
    const T ow1r = tw[2 * (lane + 64)], ow1i = tw[2 * (lane + 64) + 1];

    // window, halved (power-of-two scaling is exact and lets the split step drop its 1/2 factors)
    T hw0[16], hw1[16];
    if (WINREG) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int l = 32 * n1 + 2 * j;
            hw0[n1] = (l < L) ? T(0.5) * win[l] : T(0);
            hw1[n1] = (l + 1 < L) ? T(0.5) * win[l + 1] : T(0);
        }
    } else {
        for (int l = tid; l < 512; l += 256) { wtab[l] = (l < L) ? T(0.5) * win[l] : T(0); }
        __syncthreads();
    }

    const int total_chunks = d.B * nchunk;
    for (int cid = blockIdx.x * 4 + wv; cid < total_chunks; cid += gridDim.x * 4) {
        const int b = cid / nchunk;
        const int w0 = (cid - b * nchunk) * chunk;
        const int wend = min(w0 + chunk, d.W);
        const T *xb = X + (long long)b * d.ldx;

        for (int w = w0; w < wend; w += 4) {
            // ---------------- stage the samples of frames w..w+3 in LDS ----------------
            if (span_mode) {
                const long long s0 = d.ss0 + (long long)w * d.stp;
                const int need = 3 * d.stp + 512;
                if (s0 >= 0 && s0 + need <= d.N && (d.edge == 0 || s0 >= d.rd_shift)) {
                    const T *src = xb + (s0 - (d.edge ? d.rd_shift : 0));
                    constexpr int SB = sizeof(T) == 4 ? 8 : 4;     // loads per lane in flight at a time
                    for (int i0 = 0; i0 < need; i0 += 64 * SB) {
                        T t[SB];
#pragma unroll
                        for (int q = 0; q < SB; ++q) { const int i = i0 + 64 * q + lane; t[q] = (i < need) ? src[i] : T(0); }
#pragma unroll
                        for (int q = 0; q < SB; ++q) { const int i = i0 + 64 * q + lane; if (i < need) { xs[i] = t[q]; } }
                    }
                } else {
                    for (int i = lane; i < need; i += 64) { xs[i] = fetch_sample(xb, d, s0, d.rd_shift, i); }
                }
            } else {
                // frame starts: from the table with one load for the 4 frames (lane g: start, lane 4 + g:
                // read position), or from the hop
                long long ssg[4], rdg[4];
                if (d.tbl) {
                    const int fr = w + (lane & 3);
                    const int v = (fr < d.W) ? d.tbl[2 * fr + ((lane >> 2) & 1)] : 0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int sv = __shfl(v, g), rv = __shfl(v, 4 + g);
                        ssg[g] = sv; rdg[g] = (long long)sv - rv;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) { frame_origin(d, min(w + g, d.W - 1), ssg[g], rdg[g]); }
                }
                // frames inside their signal: the loads of GB frames (8 per lane and frame) go out before the
                // first is stored (as many as the registers take); the others sample by sample through the
                // edge rule
                constexpr int GB = sizeof(T) == 4 ? 2 : 1;
#pragma unroll
                for (int g0 = 0; g0 < 4; g0 += GB) {
                    T v[GB][8];
                    bool in[GB];
#pragma unroll
                    for (int gg = 0; gg < GB; ++gg) {
                        const int g = g0 + gg;
                        in[gg] = w + g < d.W && ssg[g] >= 0 && ssg[g] + L <= d.N && (d.edge == 0 || ssg[g] >= rdg[g]);
                        if (in[gg]) {
                            const T *src = xb + (ssg[g] - (d.edge ? rdg[g] : 0));
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int l = lane + 64 * q;
                                v[gg][q] = (l < L) ? src[l] : T(0);
                            }
                        }
                    }
#pragma unroll
                    for (int gg = 0; gg < GB; ++gg) {
                        const int g = g0 + gg;
                        if (in[gg]) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) { xs[g * 512 + lane + 64 * q] = v[gg][q]; }
                        } else if (w + g < d.W) {
                            for (int l = lane; l < 512; l += 64) {
                                xs[g * 512 + l] = (l < L) ? fetch_sample(xb, d, ssg[g], rdg[g], l) : T(0);
                            }
                        }
                    }
                }
            }
            wave_sync();

            // ---------------- pass 1: lane (f, n2 = j) ----------------
            T re[16], im[16];
            {
                const T *xf = xs + (span_mode ? f * d.stp : f * 512) + 2 * j;
                const bool vec_ok = !span_mode || !(d.stp & 1);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    T a = T(0), c = T(0);
                    if (32 * n1 < L) {
                        if (vec_ok) { const T2 t = *reinterpret_cast<const T2 *>(xf + 32 * n1); a = t.x; c = t.y; }
                        else { a = xf[32 * n1]; c = xf[32 * n1 + 1]; }
                        if (32 * (n1 + 1) > L) {      // row straddles the end of the window
                            const int l = 32 * n1 + 2 * j;
                            a = (l < L) ? a : T(0);
                            c = (l + 1 < L) ? c : T(0);
                        }
                    }
                    T h0, h1;
                    if (WINREG) { h0 = hw0[n1]; h1 = hw1[n1]; }
                    else { const T2 t = *reinterpret_cast<const T2 *>(wtab + 32 * n1 + 2 * j); h0 = t.x; h1 = t.y; }
                    re[n1] = a * h0; im[n1] = c * h1;
                }
            }
            if (d.mn0) {
                T sum = T(0);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) { sum += re[n1] + im[n1]; }
                sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
                sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
                const T mean = sum / T(L);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const int l = 32 * n1 + 2 * j;
                    re[n1] -= (l < L) ? mean : T(0);
                    im[n1] -= (l + 1 < L) ? mean : T(0);
                }
            }
            dft16(re, im);
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

            // ---------------- pass 2: lane (f, k1 = j) ----------------
            {
                const T2 *exf = ex + f * R512_EXF + j * 17;
#pragma unroll
                for (int n2 = 0; n2 < 16; ++n2) { const T2 t = exf[n2]; re[n2] = t.x; im[n2] = t.y; }
            }
            dft16(re, im);
            wave_sync();    // every lane has read its row before the region is reused for Z
            {
                T2 *zf = ex + f * R512_EXF + j;
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {
                    T2 o; o.x = re[slot16(k2)]; o.y = im[slot16(k2)];
                    zf[16 * k2] = o;            // Z[k1 + 16 k2]
                }
            }
            wave_sync();

            // ---------------- split + power + store: lane handles bins k and 256-k ----------------
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (w + g >= wend) { break; }
                const T2 *zf = ex + g * R512_EXF;
                T *y = Y + ((long long)b * d.wy() + (w + g)) * F;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int k = lane + 64 * m;
                    const T2 z = zf[k], zm = zf[(M - k) & (M - 1)];
                    const T wr = m ? ow1r : ow0r, wi = m ? ow1i : ow0i;
                    const T ar = z.x + zm.x, ai = z.y - zm.y, br = z.x - zm.x, bi = z.y + zm.y;
                    const T tr = br * wi + bi * wr, ti = bi * wi - br * wr;
                    const T pr = ar + tr, pi = ai + ti, qr = ar - tr, qi = ai - ti;
                    y[k] = power_epilogue(pr * pr + pi * pi, d.epi);
                    y[M - k] = power_epilogue(qr * qr + qi * qi, d.epi);
                }
                if (lane == 0) {
                    const T2 z = zf[128];       // bin 128 pairs with itself: X = conj(Z) (Z holds half values)
                    y[128] = power_epilogue(T(4) * (z.x * z.x + z.y * z.y), d.epi);
                }
            }
            wave_sync();
        }
    }
}

constexpr int R512_XS_CAP = 2048;   // staged samples per wave at most (4 whole frames; keeps LDS under 160 KB for double)

template <typename T>
bool r512_supported(const FrameDesc &d)
{
    return d.nfft == 512 && d.L >= 1 && d.L <= 512 && (long long)d.B * ((d.W + 3) / 4) < (1ll << 30);
}

template <typename T>
hipError_t launch_stft_r512(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    if (r512_fast_supported(d)) { return launch_stft_r512_fast<T>(Y, X, win, tw, d, s); }
    constexpr bool WINREG = (sizeof(T) == 4);
    const bool span = (d.tbl == nullptr) && (3ll * d.stp + 512 <= R512_XS_CAP);
    int xs_len = span ? 3 * d.stp + 512 : 2048;
    xs_len = (xs_len + 3) & ~3;                       // keeps the exchange region 16-byte aligned
    const size_t lds = ((WINREG ? 0 : 512) + 4 * ((size_t)xs_len + 4 * R512_EXF * 2)) * sizeof(T);
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_r512_kernel<T, WINREG>), lds, grant); e != hipSuccess) { return e; }
    // runs of `chunk` consecutive frames per wave: long enough to amortise the per-wave constants,
    // short enough that every CU gets several waves' worth of runs
    const long long frames = (long long)d.B * d.W;
    long long chunk = frames / (256ll * 16 * 4);
    chunk = chunk < 4 ? 4 : (chunk > 64 ? 64 : chunk);
    chunk &= ~3ll;
    const int nchunk = (int)((d.W + chunk - 1) / chunk);
    const long long total_chunks = (long long)d.B * nchunk;
    long long grid = (total_chunks + 3) / 4;
    if (grid > 256 * 16) { grid = 256 * 16; }
    hipLaunchKernelGGL((stft_r512_kernel<T, WINREG>), dim3((unsigned)grid), dim3(256), lds, s, Y, X, win, tw, d,
                       xs_len, span ? 1 : 0, (int)chunk, nchunk);
    return hipGetLastError();
}

template hipError_t launch_stft_r512<float>(float *, const float *, const float *, const float *, const FrameDesc &, hipStream_t);
template hipError_t launch_stft_r512<double>(double *, const double *, const double *, const double *, const FrameDesc &, hipStream_t);
template bool r512_supported<float>(const FrameDesc &);
template bool r512_supported<double>(const FrameDesc &);

}  // namespace dsp
