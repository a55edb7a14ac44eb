// Fused STFT power-spectrogram kernels for gfx950: framing + window + [mean removal] + real FFT +
// power + [sqrt] + [log] in one launch (the fused loop of reference c/stft.c:60-96, c/stft_flt.c:63-124).
//
// Three families, chosen on the host:
//   r512   : nfft == 512.  16 lanes per frame, 4 frames per wavefront.  The 512 real samples are
//            packed as 256 complex points and transformed as 16 x 16: two register-resident
//            16-point FFTs per lane with one transpose through LDS in between, then the
//            real-FFT split fused with the power step.  HBM-bound by design: each wave walks a
//            run of consecutive frames, stages the span of its 4 frames in LDS once and stores
//            spectrogram rows as whole 256-byte wave stores.
//   pow2   : any power-of-two nfft.  One 256-thread workgroup per frame, radix-2 passes in LDS.
//   direct : any other nfft.  O(nfft * L) evaluation of the DFT sum with a table of exact twiddles.
#include "stft_common.h"

namespace dsp {

// =================================================================================================
// generic power-of-two family
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void stft_pow2_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                         const T *__restrict__ win, const T *__restrict__ tw,
                                                         const FrameDesc d)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *buf = reinterpret_cast<T *>(smem_raw);   // nfft reals, read as M = nfft/2 complex points
    T *red = buf + d.nfft;                      // 4 partial sums
    const int tid = threadIdx.x;
    const int nfft = d.nfft, M = nfft >> 1, F = M + 1, L = d.L;
    const int lgM = 31 - __clz(M);
    const long long total = (long long)d.B * d.W;

    for (long long fi = blockIdx.x; fi < total; fi += gridDim.x) {
        const long long b = fi / d.W;
        const int w = (int)(fi - b * d.W);
        long long ss, rdoff;
        frame_origin(d, w, ss, rdoff);
        const T *xb = X + b * d.ldx;

        // window the frame into bit-reversed complex order: z_j = v_{2j} + i v_{2j+1}
        T part = T(0);
        for (int l = tid; l < nfft; l += 256) {
            T v = T(0);
            if (l < L) { v = fetch_sample(xb, d, ss, rdoff, l) * win[l]; part += v; }
            const int j = l >> 1;
            const int r = (M > 1) ? (int)(__brev((unsigned)j) >> (32 - lgM)) : 0;
            buf[2 * r + (l & 1)] = v;
        }
        if (d.mn0) {
            for (int off = 32; off > 0; off >>= 1) { part += __shfl_down(part, off); }
            if ((tid & 63) == 0) { red[tid >> 6] = part; }
            __syncthreads();
            const T mean = (red[0] + red[1] + red[2] + red[3]) / T(L);
            for (int l = tid; l < L; l += 256) {
                const int j = l >> 1;
                const int r = (M > 1) ? (int)(__brev((unsigned)j) >> (32 - lgM)) : 0;
                buf[2 * r + (l & 1)] -= mean;
            }
        }
        __syncthreads();

        // radix-2 decimation-in-time passes over the M complex points
        for (int h = 1; h < M; h <<= 1) {
            const int tstep = M / h;   // W_{2h}^j = W_nfft^{j nfft/(2h)}
            for (int t = tid; t < (M >> 1); t += 256) {
                const int j = t & (h - 1);
                const int i = ((t - j) << 1) + j;
                const T wr = tw[2 * (j * tstep)], wi = tw[2 * (j * tstep) + 1];
                const T ar = buf[2 * i], ai = buf[2 * i + 1];
                const T br = buf[2 * (i + h)], bi = buf[2 * (i + h) + 1];
                const T tr = br * wr - bi * wi, ti = br * wi + bi * wr;
                buf[2 * i] = ar + tr; buf[2 * i + 1] = ai + ti;
                buf[2 * (i + h)] = ar - tr; buf[2 * (i + h) + 1] = ai - ti;
            }
            __syncthreads();
        }

        // real-FFT split fused with power: with A = Z_k + conj Z_{M-k}, B = Z_k - conj Z_{M-k},
        // X_k = (A - i W^k B)/2 and X_{M-k} = conj((A + i W^k B)/2)
        T *y = Y + (b * d.wy() + w) * (long long)F;
        for (int k = tid; k <= (M >> 1); k += 256) {
            const int km = (M - k) & (M - 1);
            const T zr = buf[2 * k], zi = buf[2 * k + 1], mr = buf[2 * km], mi = buf[2 * km + 1];
            const T wr = tw[2 * k], wi = tw[2 * k + 1];
            const T ar = zr + mr, ai = zi - mi, br = zr - mr, bi = zi + mi;
            const T tr = br * wi + bi * wr, ti = bi * wi - br * wr;
            const T pr = ar + tr, pi = ai + ti, qr = ar - tr, qi = ai - ti;
            y[k] = power_epilogue(T(0.25) * (pr * pr + pi * pi), d.epi);
            y[M - k] = power_epilogue(T(0.25) * (qr * qr + qi * qi), d.epi);
        }
        __syncthreads();
    }
}

template <typename T>
hipError_t launch_stft_pow2(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    const size_t lds = (size_t)d.nfft * sizeof(T) + 64;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_pow2_kernel<T>), lds, grant); e != hipSuccess) { return e; }
    const long long total = (long long)d.B * d.W;
    const int grid = (int)(total < 8192 ? total : 8192);
    hipLaunchKernelGGL(stft_pow2_kernel<T>, dim3(grid), dim3(256), lds, s, Y, X, win, tw, d);
    return hipGetLastError();
}

// =================================================================================================
// direct family (nfft not a power of two)
// =================================================================================================
// tw holds nfft entries (cos, -sin)(2 pi a / nfft).  The power pairing keeps the reference's loops
// (c/stft.c:80-82): Y[f] = r_f^2, then Y[F-2-j] += Yw[F+j]^2; for even nfft that is r_f^2 + i_f^2,
// for odd nfft the half-complex slot F+j holds i_{F-1-j}, i.e. Y[f] += i_{f+1}^2 for 1 <= f <= F-2.
template <typename T>
__global__ __launch_bounds__(256) void stft_direct_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                           const T *__restrict__ win, const T *__restrict__ tw,
                                                           const FrameDesc d)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *v = reinterpret_cast<T *>(smem_raw);     // L windowed samples
    T *red = v + d.L;
    const int tid = threadIdx.x;
    const int nfft = d.nfft, F = nfft / 2 + 1, L = d.L, odd = nfft & 1;
    const long long total = (long long)d.B * d.W;

    for (long long fi = blockIdx.x; fi < total; fi += gridDim.x) {
        const long long b = fi / d.W;
        const int w = (int)(fi - b * d.W);
        long long ss, rdoff;
        frame_origin(d, w, ss, rdoff);
        const T *xb = X + b * d.ldx;
        T part = T(0);
        for (int l = tid; l < L; l += 256) {
            const T s = fetch_sample(xb, d, ss, rdoff, l) * win[l];
            v[l] = s; part += s;
        }
        if (d.mn0) {
            for (int off = 32; off > 0; off >>= 1) { part += __shfl_down(part, off); }
            if ((tid & 63) == 0) { red[tid >> 6] = part; }
            __syncthreads();
            const T mean = (red[0] + red[1] + red[2] + red[3]) / T(L);
            for (int l = tid; l < L; l += 256) { v[l] -= mean; }
        }
        __syncthreads();
        T *y = Y + (b * d.wy() + w) * (long long)F;
        for (int f = tid; f < F; f += 256) {
            const int k2 = f + odd;
            int a1 = 0, a2 = 0;
            double re = 0.0, im = 0.0;
            for (int l = 0; l < L; ++l) {
                const double s = (double)v[l];
                re += s * (double)tw[2 * a1];
                im += s * (double)tw[2 * a2 + 1];
                a1 += f; if (a1 >= nfft) { a1 -= nfft; }
                a2 += k2; if (a2 >= nfft) { a2 -= nfft; }
            }
            const T r = (T)re, i = (T)im;
            T p = r * r;
            if (f >= 1 && f + 2 <= F) { p += i * i; }
            y[f] = power_epilogue(p, d.epi);
        }
        __syncthreads();
    }
}

template <typename T>
hipError_t launch_stft_direct(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    const size_t lds = (size_t)d.L * sizeof(T) + 64;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_direct_kernel<T>), lds, grant); e != hipSuccess) { return e; }
    const long long total = (long long)d.B * d.W;
    const int grid = (int)(total < 8192 ? total : 8192);
    hipLaunchKernelGGL(stft_direct_kernel<T>, dim3(grid), dim3(256), lds, s, Y, X, win, tw, d);
    return hipGetLastError();
}

template hipError_t launch_stft_pow2<float>(float *, const float *, const float *, const float *, const FrameDesc &, hipStream_t);
template hipError_t launch_stft_pow2<double>(double *, const double *, const double *, const double *, const FrameDesc &, hipStream_t);
template hipError_t launch_stft_direct<float>(float *, const float *, const float *, const float *, const FrameDesc &, hipStream_t);
template hipError_t launch_stft_direct<double>(double *, const double *, const double *, const double *, const FrameDesc &, hipStream_t);


// =================================================================================================
// unfused steps: fft_power, frame_univar[_flt], apply_win, fft_fftw_r2hc
// =================================================================================================

// Signals of a multivariate tensor laid out with a sample stride (reference tensor convention: stride K along the
// transformed axis, c/fft.fftw.r2hc.c:63-65, c/apply_win.c:46-74; a row-major [N][C] tensor has incx = C, ldx = 1)
// gathered into the dense [B][N] layout the fused kernels stream: Y[b][n] = X[b ldx + n incx].  Tiles of 64 signals
// x 64 samples through LDS: reads run along the signals (one sample of 64 interleaved channels is contiguous),
// writes along the samples; 2 elements of traffic per sample.
template <typename T>
__global__ __launch_bounds__(256) void gather_signals_kernel(T *__restrict__ Y, const T *__restrict__ X, const size_t N, const size_t B,
                                                           const size_t ldx, const size_t incx, const size_t tiles_n, const size_t tiles)
{
    __shared__ T tile[64][65];
    const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t b0 = (t / tiles_n) * 64, n0 = (t % tiles_n) * 64;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const size_t n = n0 + hi + 4 * i, b = b0 + lo;
            tile[lo][hi + 4 * i] = (n < N && b < B) ? X[b * ldx + n * incx] : T(0);
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const size_t b = b0 + hi + 4 * i, n = n0 + lo;
            if (b < B && n < N) { Y[b * N + n] = tile[hi + 4 * i][lo]; }
        }
        __syncthreads();
    }
}

// The same gather for interleaved channels (ldx == 1: the B signals of a sample are adjacent) with 16-byte accesses on both
// sides: a workgroup takes 64 channels x TN samples (1 KiB of output per channel), reads vectors of 16 / sizeof(T) adjacent
// channels of one sample (a wave: 4 samples x 64 channels, 1 KiB where incx == 64), writes them to the LDS tile element by
// element (row pitch TN + 1: conflict-free), reads the tile along the samples and stores vectors of consecutive samples, a
// wave a whole KiB of one channel.  All of a thread's loads are issued before the first is used.
template <typename T>
__global__ __launch_bounds__(256) void gather_interleaved_kernel(T *__restrict__ Y, const T *__restrict__ X, const size_t N, const size_t B,
                                                               const size_t incx, const size_t tiles_n, const size_t tiles)
{
    constexpr int VEC = 16 / (int)sizeof(T), TN = 1024 / (int)sizeof(T), CV = 64 / VEC, NPP = 256 / CV, PASSES = TN / NPP;
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    __shared__ T tile[64][TN + 1];
    const int tid = threadIdx.x;
    const int cv = tid % CV, nl = tid / CV;                // load side: channel vector, sample within a pass
    const int q = tid % (TN / VEC), cl = tid / (TN / VEC); // store side: sample vector, channel within a pass
    constexpr int CPP = 256 / (TN / VEC), SPASSES = 64 / CPP;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t b0 = (t / tiles_n) * 64, n0 = (t % tiles_n) * TN;
        vecT v[PASSES];
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const size_t n = n0 + nl + NPP * i, b = b0 + cv * VEC;
            v[i] = vecT(0);
            if (n < N && b < B) { v[i] = *reinterpret_cast<const vecT *>(X + n * incx + b); }
        }
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { tile[cv * VEC + k][nl + NPP * i] = v[i][k]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPASSES; ++j) {
            const int ch = cl + CPP * j;
            const size_t b = b0 + ch, n = n0 + (size_t)q * VEC;
            vecT o;
#pragma unroll
            for (int k = 0; k < VEC; ++k) { o[k] = tile[ch][q * VEC + k]; }
            if (b < B && n < N) { *reinterpret_cast<vecT *>(Y + b * N + n) = o; }
        }
        __syncthreads();
    }
}

template <typename T>
hipError_t launch_gather_signals(T *Y, const T *X, size_t N, size_t B, size_t ldx, size_t incx, hipStream_t s)
{
    if (N == 0 || B == 0) { return hipSuccess; }
    constexpr size_t VEC = 16 / sizeof(T), TN = 1024 / sizeof(T);
    if (ldx == 1 && incx % VEC == 0 && B % VEC == 0 && N % VEC == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(Y) % 16 == 0) {
        const size_t tiles_n = (N + TN - 1) / TN, tiles = tiles_n * ((B + 63) / 64);
        const size_t grid = tiles < 256 * 2 ? tiles : 256 * 2;
        hipLaunchKernelGGL(gather_interleaved_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, Y, X, N, B, incx, tiles_n, tiles);
        return hipGetLastError();
    }
    const size_t tiles_n = (N + 63) / 64, tiles = tiles_n * ((B + 63) / 64);
    const size_t grid = tiles < 256 * 16 ? tiles : 256 * 16;
    hipLaunchKernelGGL(gather_signals_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, Y, X, N, B, ldx, incx, tiles_n, tiles);
    return hipGetLastError();
}
template hipError_t launch_gather_signals<float>(float *, const float *, size_t, size_t, size_t, size_t, hipStream_t);
template hipError_t launch_gather_signals<double>(double *, const double *, size_t, size_t, size_t, size_t, hipStream_t);

// reference c/fft_power.c:21 -- Y[n] = re^2 + im^2; pure streaming, 3 elements of traffic per value
template <typename T>
__global__ __launch_bounds__(256) void fft_power_kernel(T *__restrict__ Y, const T *__restrict__ X, const size_t N)
{
    using T2 = typename vec2<T>::type;
    const T2 *x2 = reinterpret_cast<const T2 *>(X);
    for (size_t n = (size_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (size_t)gridDim.x * 256) {
        const T2 v = x2[n];
        T y = v.x * v.x;
        y += v.y * v.y;
        Y[n] = y;
    }
}

template <typename T>
hipError_t launch_fft_power(T *Y, const T *X, size_t N, hipStream_t s)
{
    if (N == 0) { return hipSuccess; }
    size_t grid = (N + 255) / 256;
    if (grid > 256 * 32) { grid = 256 * 32; }
    hipLaunchKernelGGL(fft_power_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, Y, X, N);
    return hipGetLastError();
}

// reference c/pow_compress.c:23-50 -- Y[n] = f(X[n] + preg), f by the exponent; 2 elements of traffic per value
template <typename T>
__global__ __launch_bounds__(256) void pow_compress_kernel(T *__restrict__ Y, const T *__restrict__ X, const size_t N, const Epilogue e)
{
    for (size_t n = (size_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (size_t)gridDim.x * 256) { Y[n] = power_epilogue(X[n], e); }
}

template <typename T>
hipError_t launch_pow_compress(T *Y, const T *X, size_t N, const Epilogue &e, hipStream_t s)
{
    if (N == 0) { return hipSuccess; }
    size_t grid = (N + 255) / 256;
    if (grid > 256 * 32) { grid = 256 * 32; }
    hipLaunchKernelGGL(pow_compress_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, Y, X, N, e);
    return hipGetLastError();
}

// Window generators (reference c/hann.c:21-41, c/hamming.c:20-40, c/povey.c:21-41,
// c/blackmanharris.c:22-42, c/blackman.c:26-54, c/bartlett.c:18-41, c/triangular.c:18-38,
// c/rectangular.c:17-37, c/flattop.c:23-43, c/gauss.c:17-39, c/tukey.c:21-48, c/planck.c:17-45):
// symmetric windows, first half from the formula in T arithmetic, centre of an odd length forced to 1
// (tukey and planck: from the formula as well), second half mirrored; norm 1 divides by the sum (flattop:
// of magnitudes), 2 by the root of the sum of squares, 3 by the element L/2.  par: gauss a, tukey r,
// planck epsilon; flag: blackman "exact" coefficients.  One workgroup; sums in double.
template <typename T>
__global__ __launch_bounds__(256) void window_kernel(T *__restrict__ Y, const int L, const int kind, const int norm, const T par, const int flag)
{
    __shared__ double red[256];
    const int tid = threadIdx.x, h = L / 2, odd = L & 1;
    const bool taper = kind == WIN_TUKEY || kind == WIN_PLANCK;      // the formula runs through the centre
    const T p = (T)(6.283185307179586476925286766559 / (double)(L - 1));
    // tukey: period of the cosine lobes and their length; planck: length of the tapers
    const T pt = (kind == WIN_TUKEY) ? (T)(6.283185307179586476925286766559 / ((double)(L - 1) * (double)par)) : par * (T)L;
    const int Lt = (kind == WIN_TUKEY) ? (int)ceil(T(0.5) * par * (T)L) : (int)ceil(pt);
    for (int l = tid; l < h + (taper ? odd : 0); l += 256) {
        T v;
        const T c1 = cos((T)l * p);
        switch (kind) {
        case WIN_HANN: v = T(0.5) - T(0.5) * c1; break;
        case WIN_HAMMING: v = T(0.54) - T(0.46) * c1; break;
        case WIN_POVEY: v = pow(T(0.5) - T(0.5) * c1, T(0.85)); break;
        case WIN_BLACKMANHARRIS: v = T(0.35875) - T(0.48829) * c1 + T(0.14128) * cos((T)(2 * l) * p) - T(0.01168) * cos((T)(3 * l) * p); break;
        case WIN_BLACKMAN:
            if (flag) { v = T(7938.0) / T(18608.0) - T(9240.0) / T(18608.0) * c1 + T(1430.0) / T(18608.0) * cos((T)(2 * l) * p); }
            else { v = T(0.42) - T(0.5) * c1 + T(0.08) * cos((T)(2 * l) * p); }
            break;
        case WIN_BARTLETT: v = (T)(l * (2 - odd)) * (T(1) / (T)(L - odd * h - 1)); break;
        case WIN_TRIANGULAR: v = (T)(l * (2 - odd) + 1) * (T(1) / (T)(L - odd * h)); break;
        case WIN_RECTANGULAR: v = T(1); break;
        case WIN_FLATTOP:
            v = T(0.21557895) - T(0.41663158) * c1 + T(0.277263158) * cos((T)(2 * l) * p) - T(0.083578947) * cos((T)(3 * l) * p) + T(0.006947368) * cos((T)(4 * l) * p);
            break;
        case WIN_GAUSS: {
            const T q = T(-2) * (par * par / (T)((long long)(L - 1) * (L - 1)));
            const T m = T(0.5) * (T)(L - 1);
            v = exp(((T)l - m) * ((T)l - m) * q);
            break;
        }
        case WIN_TUKEY: v = (L <= 2 || l >= Lt) ? T(1) : T(0.5) - T(0.5) * cos((T)l * pt); break;
        default:   // WIN_PLANCK
            v = (L <= 2 || l >= Lt) ? T(1) : (l == 0 ? T(0) : T(1) / (T(1) + exp(pt / (T)l - pt / (pt - (T)l))));
            break;
        }
        Y[l] = v;
        Y[L - 1 - l] = v;
    }
    if (odd && !taper && tid == 0) { Y[h] = T(1); }
    if (norm == 0) { return; }
    __syncthreads();
    double nrm;
    if (norm == 3) {
        nrm = (double)Y[h];
    } else {
        double acc = 0.0;
        for (int l = tid; l < L; l += 256) { const double v = (double)Y[l]; acc += (norm == 1) ? (kind == WIN_FLATTOP ? fabs(v) : v) : v * v; }
        red[tid] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) { red[tid] += red[tid + o]; }
            __syncthreads();
        }
        nrm = (norm == 1) ? red[0] : sqrt(red[0]);
    }
    __syncthreads();
    const T d = (T)nrm;
    for (int l = tid; l < L; l += 256) { Y[l] = Y[l] / d; }
}

template <typename T>
hipError_t launch_window(T *Y, size_t L, int kind, int norm, T par, int flag, hipStream_t s)
{
    if (L == 0) { return hipSuccess; }
    hipLaunchKernelGGL(window_kernel<T>, dim3(1), dim3(256), 0, s, Y, (int)L, kind, norm, par, flag);
    return hipGetLastError();
}
template hipError_t launch_window<float>(float *, size_t, int, int, float, int, hipStream_t);
template hipError_t launch_window<double>(double *, size_t, int, int, double, int, hipStream_t);

// reference c/frame_univar.c:41-95, c/frame_univar_flt.c:29-91 -- Y[w][l] = X[index(w,l)], E values
// of T per sample (1 real, 2 complex)
template <typename T>
__global__ __launch_bounds__(256) void frames_kernel(T *__restrict__ Y, const T *__restrict__ X, const T *__restrict__ win, const FrameDesc d, const int E)
{
    const long long per_frame = (long long)d.L * E;
    const long long total = (long long)d.W * per_frame;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int w = (int)(i / per_frame);
        const int r = (int)(i - (long long)w * per_frame);
        const int l = r / E, e = r - l * E;
        long long ss, rdoff;
        frame_origin(d, w, ss, rdoff);
        long long n = ss + l;
        T v = T(0);
        if (d.edge == 0) {
            while (n < 0 || n >= d.N) { n = (n < 0) ? -n - 1 : 2 * d.N - 1 - n; }
            v = X[n * E + e];
        } else if (n >= 0 && n < d.N) {
            v = X[(n - rdoff) * E + e];
        }
        Y[i] = win ? v * win[l] : v;            // window_univar*: real frames times a real window
    }
}

template <typename T>
hipError_t launch_frames(T *Y, const T *X, const T *win, const FrameDesc &d, int E, hipStream_t s)
{
    const long long total = (long long)d.W * d.L * E;
    if (total == 0) { return hipSuccess; }
    long long grid = (total + 255) / 256;
    if (grid > 256 * 32) { grid = 256 * 32; }
    hipLaunchKernelGGL(frames_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, Y, X, win, d, E);
    return hipGetLastError();
}

// reference c/apply_win.c:26-108 -- Y[n] = X1[n] * X2[(n / stride_dim) % len_dim]
template <typename T>
__global__ __launch_bounds__(256) void apply_win_kernel(T *Y, const T *X1, const T *__restrict__ X2,
                                                         const size_t N, const size_t stride_dim, const size_t len_dim,
                                                         const int cplx)
{
    using T2 = typename vec2<T>::type;
    for (size_t n = (size_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (size_t)gridDim.x * 256) {
        const size_t l = (n / stride_dim) % len_dim;
        if (cplx) {
            const T2 a = reinterpret_cast<const T2 *>(X1)[n], b = reinterpret_cast<const T2 *>(X2)[l];
            T2 o;
            o.x = a.x * b.x - a.y * b.y;
            o.y = a.x * b.y + a.y * b.x;
            reinterpret_cast<T2 *>(Y)[n] = o;
        } else {
            Y[n] = X1[n] * X2[l];
        }
    }
}

template <typename T>
hipError_t launch_apply_win(T *Y, const T *X1, const T *X2, size_t N, size_t stride_dim, size_t len_dim, int cplx, hipStream_t s)
{
    if (N == 0) { return hipSuccess; }
    size_t grid = (N + 255) / 256;
    if (grid > 256 * 32) { grid = 256 * 32; }
    hipLaunchKernelGGL(apply_win_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, Y, X1, X2, N, stride_dim, len_dim, cplx);
    return hipGetLastError();
}

// reference c/fft.fftw.r2hc.c:24-109 -- real FFT of V vectors of length Lx (element stride K, vector
// v = (o, i): X + o K Lx + i), zero-padded to nfft, output nfft/2+1 interleaved complex values with
// the same stride.  pow2 != 0: radix-2 in LDS (tw: nfft/2 entries); else direct sum (tw: nfft entries).
template <typename T>
__global__ __launch_bounds__(256) void r2hc_kernel(T *__restrict__ Y, const T *__restrict__ X, const T *__restrict__ tw,
                                                    const size_t V, const size_t K, const size_t Lx, const int nfft,
                                                    const int pow2, const T scale)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *buf = reinterpret_cast<T *>(smem_raw);
    const int tid = threadIdx.x;
    const int M = nfft >> 1, Ly = nfft / 2 + 1;
    for (size_t v = blockIdx.x; v < V; v += gridDim.x) {
        const size_t o = v / K, i = v - o * K;
        const T *x = X + o * K * Lx + i;
        T *y = Y + 2 * (o * K * (size_t)Ly + i);
        if (pow2) {
            const int lgM = 31 - __clz(M);
            for (int l = tid; l < nfft; l += 256) {
                const T s = ((size_t)l < Lx) ? x[(size_t)l * K] : T(0);
                const int j = l >> 1;
                const int r = (M > 1) ? (int)(__brev((unsigned)j) >> (32 - lgM)) : 0;
                buf[2 * r + (l & 1)] = s;
            }
            __syncthreads();
            for (int h = 1; h < M; h <<= 1) {
                const int tstep = M / h;
                for (int t = tid; t < (M >> 1); t += 256) {
                    const int j = t & (h - 1);
                    const int a = ((t - j) << 1) + j;
                    const T wr = tw[2 * (j * tstep)], wi = tw[2 * (j * tstep) + 1];
                    const T ar = buf[2 * a], ai = buf[2 * a + 1];
                    const T br = buf[2 * (a + h)], bi = buf[2 * (a + h) + 1];
                    const T tr = br * wr - bi * wi, ti = br * wi + bi * wr;
                    buf[2 * a] = ar + tr; buf[2 * a + 1] = ai + ti;
                    buf[2 * (a + h)] = ar - tr; buf[2 * (a + h) + 1] = ai - ti;
                }
                __syncthreads();
            }
            for (int k = tid; k <= (M >> 1); k += 256) {
                const int km = (M - k) & (M - 1);
                const T zr = buf[2 * k], zi = buf[2 * k + 1], mr = buf[2 * km], mi = buf[2 * km + 1];
                const T wr = tw[2 * k], wi = tw[2 * k + 1];
                const T ar = zr + mr, ai = zi - mi, br = zr - mr, bi = zi + mi;
                const T tr = br * wi + bi * wr, ti = bi * wi - br * wr;
                const T h = T(0.5) * scale;
                T xr = h * (ar + tr), xi = h * (ai + ti), ur = h * (ar - tr), ui = -h * (ai - ti);
                if (k == 0) { xi = T(0); ui = T(0); }       // imag(DC) = imag(Nyquist) = 0
                y[2 * (size_t)k * K] = xr; y[2 * (size_t)k * K + 1] = xi;
                y[2 * (size_t)(M - k) * K] = ur; y[2 * (size_t)(M - k) * K + 1] = ui;
            }
        } else {
            for (int l = tid; (size_t)l < Lx; l += 256) { buf[l] = x[(size_t)l * K]; }
            __syncthreads();
            for (int k = tid; k < Ly; k += 256) {
                int a = 0;
                double re = 0.0, im = 0.0;
                for (int l = 0; (size_t)l < Lx; ++l) {
                    const double s = (double)buf[l];
                    re += s * (double)tw[2 * a];
                    im += s * (double)tw[2 * a + 1];
                    a += k; if (a >= nfft) { a -= nfft; }
                }
                if (k == 0 || 2 * k == nfft) { im = 0.0; }
                y[2 * (size_t)k * K] = (T)re * scale; y[2 * (size_t)k * K + 1] = (T)im * scale;
            }
        }
        __syncthreads();
    }
}

template <typename T>
hipError_t launch_r2hc(T *Y, const T *X, const T *tw, size_t V, size_t K, size_t Lx, size_t nfft, int pow2, T scale, hipStream_t s)
{
    if (V == 0) { return hipSuccess; }
    const size_t lds = (pow2 ? nfft : Lx) * sizeof(T) + 64;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&r2hc_kernel<T>), lds, grant); e != hipSuccess) { return e; }
    const size_t grid = V < 8192 ? V : 8192;
    hipLaunchKernelGGL(r2hc_kernel<T>, dim3((unsigned)grid), dim3(256), lds, s, Y, X, tw, V, K, Lx, (int)nfft, pow2, scale);
    return hipGetLastError();
}

// reference c/fft.fftw.c:194-272 -- complex FFT of V vectors of Lx interleaved complex values (element
// stride K complex values, vector v = (o, i): X + 2 (o K Lx + i)), zero-padded to nfft, nfft complex
// outputs with the same stride.  pow2 != 0: radix-2 in LDS (tw: nfft/2 entries); else direct sum in
// double (tw: nfft entries).
template <typename T>
__global__ __launch_bounds__(256) void c2c_kernel(T *__restrict__ Y, const T *__restrict__ X, const T *__restrict__ tw,
                                                   const size_t V, const size_t K, const size_t Lx, const int nfft,
                                                   const int pow2, const T scale)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *buf = reinterpret_cast<T *>(smem_raw);
    const int tid = threadIdx.x;
    for (size_t v = blockIdx.x; v < V; v += gridDim.x) {
        const size_t o = v / K, i = v - o * K;
        const T *x = X + 2 * (o * K * Lx + i);
        T *y = Y + 2 * (o * K * (size_t)nfft + i);
        if (pow2) {
            const int lg = 31 - __clz(nfft);
            for (int l = tid; l < nfft; l += 256) {
                const int r = (int)(__brev((unsigned)l) >> (32 - lg));
                const bool in = (size_t)l < Lx;
                buf[2 * r] = in ? x[2 * (size_t)l * K] : T(0);
                buf[2 * r + 1] = in ? x[2 * (size_t)l * K + 1] : T(0);
            }
            __syncthreads();
            for (int h = 1; h < nfft; h <<= 1) {
                const int tstep = (nfft >> 1) / h;
                for (int t = tid; t < (nfft >> 1); t += 256) {
                    const int j = t & (h - 1);
                    const int a = ((t - j) << 1) + j;
                    const T wr = tw[2 * (j * tstep)], wi = tw[2 * (j * tstep) + 1];
                    const T ar = buf[2 * a], ai = buf[2 * a + 1];
                    const T br = buf[2 * (a + h)], bi = buf[2 * (a + h) + 1];
                    const T tr = br * wr - bi * wi, ti = br * wi + bi * wr;
                    buf[2 * a] = ar + tr; buf[2 * a + 1] = ai + ti;
                    buf[2 * (a + h)] = ar - tr; buf[2 * (a + h) + 1] = ai - ti;
                }
                __syncthreads();
            }
            for (int k = tid; k < nfft; k += 256) {
                y[2 * (size_t)k * K] = buf[2 * k] * scale;
                y[2 * (size_t)k * K + 1] = buf[2 * k + 1] * scale;
            }
        } else {
            for (int l = tid; (size_t)l < 2 * Lx; l += 256) { buf[l] = x[2 * (size_t)(l >> 1) * K + (l & 1)]; }
            __syncthreads();
            for (int k = tid; k < nfft; k += 256) {
                int a = 0;
                double re = 0.0, im = 0.0;
                for (int l = 0; (size_t)l < Lx; ++l) {
                    const double sr = (double)buf[2 * l], si = (double)buf[2 * l + 1];
                    const double wr = (double)tw[2 * a], wi = (double)tw[2 * a + 1];
                    re += sr * wr - si * wi;
                    im += sr * wi + si * wr;
                    a += k; if (a >= nfft) { a -= nfft; }
                }
                y[2 * (size_t)k * K] = (T)re * scale; y[2 * (size_t)k * K + 1] = (T)im * scale;
            }
        }
        __syncthreads();
    }
}

template <typename T>
hipError_t launch_c2c(T *Y, const T *X, const T *tw, size_t V, size_t K, size_t Lx, size_t nfft, int pow2, T scale, hipStream_t s)
{
    if (V == 0) { return hipSuccess; }
    const size_t lds = 2 * (pow2 ? nfft : Lx) * sizeof(T) + 64;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&c2c_kernel<T>), lds, grant); e != hipSuccess) { return e; }
    const size_t grid = V < 8192 ? V : 8192;
    hipLaunchKernelGGL(c2c_kernel<T>, dim3((unsigned)grid), dim3(256), lds, s, Y, X, tw, V, K, Lx, (int)nfft, pow2, scale);
    return hipGetLastError();
}
template hipError_t launch_c2c<float>(float *, const float *, const float *, size_t, size_t, size_t, size_t, int, float, hipStream_t);
template hipError_t launch_c2c<double>(double *, const double *, const double *, size_t, size_t, size_t, size_t, int, double, hipStream_t);

template hipError_t launch_pow_compress<float>(float *, const float *, size_t, const Epilogue &, hipStream_t);
template hipError_t launch_pow_compress<double>(double *, const double *, size_t, const Epilogue &, hipStream_t);
template hipError_t launch_fft_power<float>(float *, const float *, size_t, hipStream_t);
template hipError_t launch_fft_power<double>(double *, const double *, size_t, hipStream_t);
template hipError_t launch_frames<float>(float *, const float *, const float *, const FrameDesc &, int, hipStream_t);
template hipError_t launch_frames<double>(double *, const double *, const double *, const FrameDesc &, int, hipStream_t);
template hipError_t launch_apply_win<float>(float *, const float *, const float *, size_t, size_t, size_t, int, hipStream_t);
template hipError_t launch_apply_win<double>(double *, const double *, const double *, size_t, size_t, size_t, int, hipStream_t);
template hipError_t launch_r2hc<float>(float *, const float *, const float *, size_t, size_t, size_t, size_t, int, float, hipStream_t);
template hipError_t launch_r2hc<double>(double *, const double *, const double *, size_t, size_t, size_t, size_t, int, double, hipStream_t);
}  // namespace dsp
