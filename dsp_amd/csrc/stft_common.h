// Shared declarations of the HIP STFT library (host + device).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <atomic>
#include <cstdint>

namespace dsp {

// Where frame w of a signal starts and how its samples are fetched.
//   uniform mode (tbl == nullptr): start ss_w = ss0 + w*stp
//   table mode  (tbl != nullptr):  ss_w = tbl[2w], read start rd_w = tbl[2w+1]
// edge == 0: sample index n = ss_w + l is reflected into [0,N) (reference c/stft.c:113)
// edge == 1: value is 0 unless 0 <= n < N, else X[n - (ss_w - rd_w)]  (reference c/stft_flt.c:66-88)
// What happens to a power value before it is stored: [sqrt] -> [log] -> [power compression].
// pc selects the compression of reference c/pow_compress.c:23-50 applied to v + preg:
// 0 none, 1 identity (v + preg), 2 log, 3 sqrt, 4 cbrt, 5 pow(., p).
// Division of a step number by a divisor that is fixed for the launch: multiplication by a 33-bit reciprocal made on the host
// (Granlund and Montgomery's round-up form), exact for every 32-bit dividend.  On wave-uniform operands it is five scalar
// instructions where the compiler's division by a run-time value takes about 25, in kernels that draw a step number per step.
struct FastDiv {
    unsigned m = 1;
    int sh1 = 0, sh2 = 0;
    static FastDiv of(unsigned d)
    {
        FastDiv r;
        int s = 0;
        while ((1ull << s) < d) { ++s; }
        r.m = (unsigned)((((1ull << s) - d) << 32) / d + 1);
        r.sh1 = s < 1 ? s : 1;
        r.sh2 = s > 1 ? s - 1 : 0;
        return r;
    }
    __host__ __device__ unsigned div(unsigned n) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned t = __umulhi(n, m);
#else
        const unsigned t = (unsigned)(((unsigned long long)n * m) >> 32);
#endif
        return (t + ((n - t) >> sh1)) >> sh2;
    }
};

struct Epilogue {
    int amp, lg, pc;
    double p, preg;
    bool any() const { return amp || lg || pc; }
};

struct FrameDesc {
    long long N;         // samples per signal
    long long ldx;       // elements between the starts of consecutive signals
    long long ss0;       // uniform mode: start sample of frame 0
    long long rd_shift;  // uniform mode, edge == 1: ss_w - rd_w (the same for every frame)
    const int *tbl;      // table mode: 2 ints per frame
    int tbl_reach;       // table mode: max over w of tbl[2(w+3)] - tbl[2w] when the starts never decrease and
                         // every frame is read where it starts (tbl[2w+1] == tbl[2w]); -1 otherwise
    int L;               // window length
    int nfft;            // transform length
    int W;               // frames per signal
    int stp;             // uniform mode: hop in samples
    int edge;
    int mn0;
    Epilogue epi;
    int B;               // signals
    int Wy;              // frames per signal in the OUTPUT array: the rows of signal b start at row b * Wy (0: W).  A launch
                         // over a range of a batch's frames writes W frames per signal into rows that are Wy apart.
    // frames per signal of the output array
    __host__ __device__ int wy() const { return Wy ? Wy : W; }
    // stft_flt framings whose first frames hang over the start of the signal while the later ones are read with a shift
    // (reference c/stft_flt.c:66-88) go in two launches: frames [0, w_head) by the table, frames [w_head, W) as frames of
    // the signal cut to n_tail samples, read where they start, at tail_stp-spaced starts from tail_ss0 (tail_stp > 0) or at
    // the starts in tail_tbl (reach tail_reach).  w_head == 0: one launch.
    int w_head, tail_stp, tail_reach;
    long long n_tail, tail_ss0;
    const int *tail_tbl;
};

template <typename T>
__device__ __forceinline__ T fetch_sample(const T *__restrict__ xb, const FrameDesc &d, long long ss, long long rdoff, int l)
{
    long long n = ss + l;
    if (d.edge == 0) {
        while (n < 0 || n >= d.N) { n = (n < 0) ? -n - 1 : 2 * d.N - 1 - n; }
        return xb[n];
    }
    // (n >= rdoff: span staging also asks for samples between frames that no frame uses; with a read shift
    // those just past the start of the signal would lie before it)
    return (n >= 0 && n < d.N && n >= rdoff) ? xb[n - rdoff] : T(0);
}

__device__ __forceinline__ void frame_origin(const FrameDesc &d, int w, long long &ss, long long &rdoff)
{
    if (d.tbl) { ss = d.tbl[2 * w]; rdoff = ss - (long long)d.tbl[2 * w + 1]; }
    else { ss = d.ss0 + (long long)w * d.stp; rdoff = d.rd_shift; }
}

// sqrt and natural log of a power value.  float: the hardware's 1-ulp v_sqrt_f32 and v_log_f32 (log2)
// with the change of base carried in two parts, a handful of instructions instead of the library's ~12 each
// (16 values per lane and step: the library forms double the kernel's instruction count).  Both
// instructions take a denormal input for zero, so values below 2^-96 are scaled by 2^64 first and the
// result is corrected (sqrt: times 2^-32, log2: minus 64): powers of very quiet signals come out as the
// library's sqrtf / logf give them, not as 0 / -inf.  double: the library functions.
__device__ __forceinline__ float epi_sqrt(float v)
{
    const bool tiny = v < 0x1p-96f;
    return __builtin_amdgcn_sqrtf(tiny ? v * 0x1p64f : v) * (tiny ? 0x1p-32f : 1.0f);
}
__device__ __forceinline__ double epi_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float epi_log2(float v)
{
    const bool tiny = v < 0x1p-96f;
    return __builtin_amdgcn_logf(tiny ? v * 0x1p64f : v) - (tiny ? 64.0f : 0.0f);
}
__device__ __forceinline__ float epi_log(float v)
{
    const float l2 = epi_log2(v);
    // ln 2 = hi + lo with hi the float BELOW ln 2, so that both parts have the sign of l2 and log(0) = -inf,
    // log(inf) = inf come out of the same two instructions (no inf - inf)
    return fmaf(l2, 0.6931471228599548f, l2 * 5.769998878690785e-08f);
}
__device__ __forceinline__ double epi_log(double v) { return log(v); }
// Cube root and power of the same values.  float: from the hardware's log2 and exp2, ~10 instructions
// instead of the library's 40 to 80, which made the fused cube-root epilogue cost two thirds of the
// whole 512-point kernel (0.65 ms against 0.40).  cbrt: exp2(log2|v| / 3) and one Newton step (to about
// 1 ulp).  pow: log2 v refined by one step (v 2^-l0 - 1) / ln 2, the product with p carried in two parts,
// exp2 of the high part times 1 + ln 2 (low part): about 2 ulp.  Zero, infinities and NaN as the library;
// denormal arguments and results through the same scaling as above.
__device__ __forceinline__ float epi_cbrt(float v)
{
    const float a0 = fabsf(v);
    const bool tiny = a0 < 0x1p-96f;
    const float a = tiny ? a0 * 0x1p96f : a0;             // cbrt(2^96) = 2^32
    const float y0 = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(a) * 0.3333333432674407958984375f);
    const float r = __builtin_amdgcn_rcpf(y0 * y0);
    const float y = fmaf(a * r, 0.3333333432674407958984375f, y0 * 0.666666686534881591796875f);
    return copysignf(((y0 == 0.0f || y0 == __builtin_inff()) ? y0 : y) * (tiny ? 0x1p-32f : 1.0f), v);
}
__device__ __forceinline__ double epi_cbrt(double v) { return cbrt(v); }
__device__ __forceinline__ float epi_pow(float v, float p)
{
    const bool tiny = v < 0x1p-96f;
    const float vs = tiny ? v * 0x1p64f : v;
    const float ls = __builtin_amdgcn_logf(vs);                       // NaN for v < 0, as pow with a fractional p
    const float r = vs * __builtin_amdgcn_exp2f(-ls);                 // 1 + eps
    const float corr = (r - 1.0f) * 1.44269502162933349609375f;       // log2(1 + eps)
    const float l0 = ls - (tiny ? 64.0f : 0.0f);
    const float eh = p * l0;
    const float el = fmaf(p, l0, -eh) + p * corr;
    const bool low = eh < -100.0f;                                    // the result is near or below 2^-126: exp2 of a shifted exponent
    const float yh = __builtin_amdgcn_exp2f(low ? eh + 64.0f : eh);
    const float y = fmaf(yh * 0.693147182464599609375f, el, yh) * (low ? 0x1p-64f : 1.0f);
    return (v == 0.0f || fabsf(l0) == __builtin_inff()) ? ((v == 0.0f) ? 0.0f : v) : y;
}
__device__ __forceinline__ double epi_pow(double v, double p) { return pow(v, p); }

// epilogue of the power value (reference c/stft.c:86-95, then c/pow_compress.c:28-47)
template <typename T>
__device__ __forceinline__ T power_epilogue(T v, const Epilogue &e)
{
    if (e.amp) { v = epi_sqrt(v); }
    if (e.lg) { v = epi_log(v); }
    if (e.pc) {
        v += (T)e.preg;
        if (e.pc == 2) { v = epi_log(v); }
        else if (e.pc == 3) { v = epi_sqrt(v); }
        else if (e.pc == 4) { v = epi_cbrt(v); }
        else if (e.pc == 5) { v = epi_pow(v, (T)e.p); }
    }
    return v;
}

// the same for N values of a lane, with the (uniform) branches outside the loops over the values: per value
// they cost the 512-point kernel more than the arithmetic they select
template <typename T, int N>
__device__ __forceinline__ void power_epilogue_n(T (&v)[N], const Epilogue &e)
{
    if (e.amp) {
#pragma unroll
        for (int i = 0; i < N; ++i) { v[i] = epi_sqrt(v[i]); }
    }
    if (e.lg) {
#pragma unroll
        for (int i = 0; i < N; ++i) { v[i] = epi_log(v[i]); }
    }
    if (e.pc) {
        const T preg = (T)e.preg;
        if (e.pc == 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) { v[i] += preg; }
        } else if (e.pc == 2) {
#pragma unroll
            for (int i = 0; i < N; ++i) { v[i] = epi_log(v[i] + preg); }
        } else if (e.pc == 3) {
#pragma unroll
            for (int i = 0; i < N; ++i) { v[i] = epi_sqrt(v[i] + preg); }
        } else if (e.pc == 4) {
#pragma unroll
            for (int i = 0; i < N; ++i) { v[i] = epi_cbrt(v[i] + preg); }
        } else {
            const T p = (T)e.p;
#pragma unroll
            for (int i = 0; i < N; ++i) { v[i] = epi_pow(v[i] + preg, p); }
        }
    }
}

template <typename T> struct vec2;
template <> struct vec2<float> { using type = float2; };
template <> struct vec2<double> { using type = double2; };

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A kernel that needs more than 48 KB of dynamic LDS must be granted the size once per device (the
// attribute belongs to the device's copy of the kernel).  One LdsGrant per kernel instantiation.
struct LdsGrant { std::atomic<size_t> bytes[64]; };
inline hipError_t ensure_lds(const void *kernel, size_t lds, LdsGrant &g)
{
    (void)hipGetLastError();                      // a stale error must not be taken for this launch's
    if (lds <= 48 * 1024) { return hipSuccess; }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { return e; }
    if (dev < 0 || dev >= 64) { return hipErrorInvalidDevice; }
    if (lds <= g.bytes[dev].load()) { return hipSuccess; }
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { return e; }
    g.bytes[dev].store(lds);
    return hipSuccess;
}

// ---- launchers (stft_kernels.hip) -------------------------------------------------------------
// tw: device table of W_nfft^k = (cos, -sin)(2 pi k / nfft), k < nfft/2 (power-of-two families) or
//     k < nfft (direct family), as interleaved pairs of T.
template <typename T> hipError_t launch_stft_pow2(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s);
template <typename T> hipError_t launch_stft_direct(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s);
template <typename T> hipError_t launch_stft_r512(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s);
template <typename T> bool r512_supported(const FrameDesc &d);
bool r512_fast_supported(const FrameDesc &d);
bool r4096_supported(const FrameDesc &d);
// take: 8 counters (32 bytes of device memory) for the run form, or nullptr for the library's own sets
hipError_t launch_stft_r4096(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s, unsigned *take = nullptr);
hipError_t launch_stft_r4096(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, hipStream_t s, unsigned *take = nullptr);   // 4 waves per workgroup
// nfft 8192, float32 (r8192_kernel.hip)
bool r8192_supported(const FrameDesc &d);
hipError_t launch_stft_r8192(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s);
// nfft 1024 and 2048, float32, equally spaced frames (r512x_kernel.hip)
bool r512x_supported(const FrameDesc &d);
hipError_t launch_stft_r512x(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s);
hipError_t launch_stft_r512x(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, hipStream_t s);   // 4 waves per workgroup
// nfft 256, 128 and 64 (r256_kernel.hip)
bool r256_supported(const FrameDesc &d);
hipError_t launch_stft_r256(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s);
hipError_t launch_stft_r256(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, hipStream_t s);   // 4 waves per workgroup
template <typename T> bool wave_fft_supported(const FrameDesc &d);
template <typename T> hipError_t launch_stft_wave(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s);
template <typename T> hipError_t launch_stft_r512_fast(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s);
// nfft 512, float32: the d.B signals are adjacent channels of a [N][pitch] array (sample n of channel c at X[n pitch + c]), read in place
bool r512_fast_ilv_supported(const FrameDesc &d, const float *X, size_t pitch);
hipError_t launch_stft_r512_fast_ilv(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, size_t pitch, hipStream_t s);
bool r512_fast_cplx_supported(const FrameDesc &d);
bool r512x_cplx_supported(const FrameDesc &d);
bool r256_cplx_supported(const FrameDesc &d);
bool r4096_cplx_supported(const FrameDesc &d);
hipError_t launch_stft_r4096_cplx(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, double scale, hipStream_t s);
hipError_t launch_stft_r4096_cplx(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, double scale, hipStream_t s);
hipError_t launch_stft_r256_cplx(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, double scale, hipStream_t s);
hipError_t launch_stft_r256_cplx(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, double scale, hipStream_t s);
hipError_t launch_stft_r512x_cplx(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, double scale, hipStream_t s);
hipError_t launch_stft_r512x_cplx(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, double scale, hipStream_t s);
template <typename T> hipError_t launch_stft_r512_fast_cplx(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, double scale, hipStream_t s);

// ---- unfused kernels (unfused_kernels.hip) ----------------------------------------------------
enum WindowKind { WIN_HANN = 0, WIN_HAMMING, WIN_POVEY, WIN_BLACKMANHARRIS, WIN_BLACKMAN, WIN_BARTLETT, WIN_TRIANGULAR,
                  WIN_RECTANGULAR, WIN_FLATTOP, WIN_GAUSS, WIN_TUKEY, WIN_PLANCK, WIN_COUNT };
template <typename T> hipError_t launch_window(T *Y, size_t L, int kind, int norm, T par, int flag, hipStream_t s);
template <typename T> hipError_t launch_fft_power(T *Y, const T *X, size_t N, hipStream_t s);
// Y[b][n] = X[b ldx + n incx], dense [B][N]
template <typename T> hipError_t launch_gather_signals(T *Y, const T *X, size_t N, size_t B, size_t ldx, size_t incx, hipStream_t s);
template <typename T> hipError_t launch_pow_compress(T *Y, const T *X, size_t N, const Epilogue &e, hipStream_t s);
// win != nullptr (E == 1 only): every frame is multiplied by the L window values
template <typename T> hipError_t launch_frames(T *Y, const T *X, const T *win, const FrameDesc &d, int E, hipStream_t s);
template <typename T> hipError_t launch_apply_win(T *Y, const T *X1, const T *X2, size_t N, size_t stride_dim, size_t len_dim, int cplx, hipStream_t s);
template <typename T> hipError_t launch_r2hc(T *Y, const T *X, const T *tw, size_t V, size_t K, size_t Lx, size_t nfft, int pow2, T scale, hipStream_t s);
template <typename T> hipError_t launch_c2c(T *Y, const T *X, const T *tw, size_t V, size_t K, size_t Lx, size_t nfft, int pow2, T scale, hipStream_t s);

}  // namespace dsp
