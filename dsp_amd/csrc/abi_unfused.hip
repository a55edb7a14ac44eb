// C ABI of libdsp_stft_hip.so, host side: the unfused steps (power compression, windows, frequency axes, fft_power, framing,
// apply_win, the batched FFTs along a dimension).
#include "abi_internal.h"

namespace dsp {
namespace abi {

template <typename T>
int pow_compress_host(const char *fn, T *Y, const T *X, size_t N, T p, T preg)
{
    Epilogue e{0, 0, 0, 0.0, 0.0};
    if (pc_epilogue<T>(fn, p, preg, e)) { return 1; }
    if (N == 0) { return 0; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    void *dx;
    if (ws_reserve(fn, st, 0, N * sizeof(T), dx)) { return 1; }
    HIPCHK(fn, hipMemcpy(dx, X, N * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(fn, launch_pow_compress<T>((T *)dx, (const T *)dx, N, e, nullptr));
    HIPCHK(fn, hipMemcpy(Y, dx, N * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

template <typename T>
int pow_compress_dev(const char *fn, T *Yd, const T *Xd, size_t N, T p, T preg, void *stream)
{
    Epilogue e{0, 0, 0, 0.0, 0.0};
    if (pc_epilogue<T>(fn, p, preg, e)) { return 1; }
    if (N == 0) { return 0; }
    if (!Yd || !Xd) { fprintf(stderr, "error in %s: null device pointer\n", fn); return 1; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    HIPCHK(fn, launch_pow_compress<T>(Yd, Xd, N, e, (hipStream_t)stream));
    return 0;
}

// window generators: checks and message of reference c/hann.c:23
template <typename T>
bool window_par_ok(const char *fn, int kind, T par)
{
    // reference c/gauss.c:20, c/tukey.c:24, c/planck.c:20
    if (kind == WIN_GAUSS && !(par > T(0))) { fprintf(stderr, "error in %s: a must be positive \n", fn); return false; }
    if (kind == WIN_TUKEY && !(par >= T(0) && par <= T(1))) { fprintf(stderr, "error in %s: r must be in [0.0 1.0] \n", fn); return false; }
    if (kind == WIN_PLANCK && !(par >= T(0) && par <= T(0.5))) { fprintf(stderr, "error in %s: epsilon must be in [0.0 0.5] \n", fn); return false; }
    return true;
}

template <typename T>
int window_host(const char *fn, T *Y, size_t L, size_t norm, int kind, T par, int flag)
{
    if (norm > 3u) { fprintf(stderr, "error in %s: norm must be in {0,1,2,3}\n", fn); return 1; }
    if (!window_par_ok<T>(fn, kind, par)) { return 1; }
    if (L > (size_t)INT_MAX / 4) { fprintf(stderr, "error in %s: problem size exceeds the supported index range\n", fn); return 1; }
    if (L == 0) { return 0; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    void *dy;
    if (ws_reserve(fn, st, 1, L * sizeof(T), dy)) { return 1; }
    HIPCHK(fn, launch_window<T>((T *)dy, L, kind, (int)norm, par, flag, nullptr));
    HIPCHK(fn, hipMemcpy(Y, dy, L * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

// the complex forms (reference c/hann.c:67-113 and the like): the same L values, each followed by a zero imaginary part
template <typename T>
int window_host_cplx(const char *fn, T *Y, size_t L, size_t norm, int kind, T par, int flag)
{
    if (L > (size_t)INT_MAX / 4) { fprintf(stderr, "error in %s: problem size exceeds the supported index range\n", fn); return 1; }
    std::vector<T> w;
    try { w.resize(L); } catch (const std::exception &) { fprintf(stderr, "error in %s: problem with malloc. \n", fn); return 1; }
    if (int rc = window_host<T>(fn, w.data(), L, norm, kind, par, flag)) { return rc; }
    for (size_t l = 0; l < L; ++l) { Y[2 * l] = w[l]; Y[2 * l + 1] = T(0); }
    return 0;
}

template <typename T>
int window_dev(const char *fn, T *Yd, int kind, size_t L, size_t norm, void *stream, T par, int flag)
{
    if (kind < 0 || kind >= WIN_COUNT) { fprintf(stderr, "error in %s: window must be in 0..%d (hann, hamming, povey, blackmanharris, blackman, bartlett, triangular, rectangular, flattop, gauss, tukey, planck)\n", fn, WIN_COUNT - 1); return 1; }
    if (norm > 3u) { fprintf(stderr, "error in %s: norm must be in {0,1,2,3}\n", fn); return 1; }
    if (!window_par_ok<T>(fn, kind, par)) { return 1; }
    if (L > (size_t)INT_MAX / 4) { fprintf(stderr, "error in %s: problem size exceeds the supported index range\n", fn); return 1; }
    if (L == 0) { return 0; }
    if (!Yd) { fprintf(stderr, "error in %s: null device pointer\n", fn); return 1; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    HIPCHK(fn, launch_window<T>(Yd, L, kind, (int)norm, par, flag, (hipStream_t)stream));
    return 0;
}

// frequency axis of the spectrogram, F values computed on the host (reference c/get_stft_freqs.c:21-33)
template <typename T>
int stft_freqs(const char *fn, T *Y, size_t nfft, size_t F, T fs)
{
    if (F > nfft) { fprintf(stderr, "error in %s: F must be <= nfft\n", fn); return 1; }
    if (F == 0) { return 0; }
    const size_t nyq = nfft / 2u + 1u;
    const T finc = fs / (T)nfft;
    Y[0] = T(0);
    for (size_t f = 1; f < F && f < nyq; ++f) { Y[f] = (T)f * finc; }
    for (size_t f = nyq > 1 ? nyq : 1; f < F; ++f) { Y[f] = -finc * (T)(nfft - f); }
    return 0;
}

// Frequency scales of the axis, in place on the host like the axis itself (reference
// c/convert_freqs.c:24-127): hz, sqrt, cbrt, octave (log2).  The reference tells the input scale by its
// first two characters ("octave": six) and the output scale by four ("hz": two, "octave": six).
template <typename T>
int convert_freqs_host(const char *fn, T *frqs, size_t F, const char *in_scale, const char *out_scale)
{
    if (!in_scale || strlen(in_scale) < 2u) { fprintf(stderr, "error in %s: input freq scale must be string with length > 1\n", fn); return 1; }
    if (!out_scale || strlen(out_scale) < 2u) { fprintf(stderr, "error in %s: output freq scale must be string with length > 1\n", fn); return 1; }
    if (F == 0u) { return 0; }
    int in, out;
    if (strncmp(in_scale, "hz", 2u) == 0) { in = 0; }
    else if (strncmp(in_scale, "sq", 2u) == 0) { in = 1; }
    else if (strncmp(in_scale, "cb", 2u) == 0) { in = 2; }
    else if (strncmp(in_scale, "octave", 6u) == 0) { in = 3; }
    else { fprintf(stderr, "error in %s: input frequency scale not recognized\n", fn); return 1; }
    if (strncmp(out_scale, "hz", 2u) == 0) { out = 0; }
    else if (strncmp(out_scale, "sqrt", 4u) == 0) { out = 1; }
    else if (strncmp(out_scale, "cbrt", 4u) == 0) { out = 2; }
    else if (strncmp(out_scale, "octave", 6u) == 0) { out = 3; }
    else { fprintf(stderr, "error in %s: output frequency scale not recognized\n", fn); return 1; }
    if (in == out) { return 0; }
    for (size_t f = 0; f < F; ++f) {
        const T x = frqs[f];
        T y;
        if (in == 0) { y = (out == 1) ? std::sqrt(x) : (out == 2) ? std::cbrt(x) : std::log2(x); }
        else if (in == 1) { y = (out == 0) ? x * x : (out == 2) ? std::cbrt(x * x) : std::log2(x * x); }
        else if (in == 2) { y = (out == 0) ? x * (x * x) : (out == 1) ? std::sqrt(x * x * x) : std::log2(x * x * x); }
        else { y = (out == 0) ? std::pow(T(2), x) : (out == 1) ? std::pow(T(2), T(0.5) * x) : std::cbrt(std::pow(T(2), x)); }
        frqs[f] = y;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// unfused entry points
// ------------------------------------------------------------------------------------------------
template <typename T>
int fft_power_host(const char *fn, T *Y, const T *X, size_t N)
{
    if (N == 0) { return 0; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    void *dx, *dy;
    if (ws_reserve(fn, st, 0, 2 * N * sizeof(T), dx)) { return 1; }
    if (ws_reserve(fn, st, 1, N * sizeof(T), dy)) { return 1; }
    HIPCHK(fn, hipMemcpy(dx, X, 2 * N * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(fn, launch_fft_power<T>((T *)dy, (const T *)dx, N, nullptr));
    HIPCHK(fn, hipMemcpy(Y, dy, N * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

template <typename T>
int frames_host(const char *fn, T *Y, const T *X, const FrameDesc &d, int E, int dev, const T *win)
{
    if (d.W == 0) { return 0; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    void *dx, *dy, *dw = nullptr;
    const size_t xb = (size_t)d.N * E * sizeof(T), yb = (size_t)d.W * d.L * E * sizeof(T);
    if (ws_reserve(fn, st, 0, xb, dx)) { return 1; }
    if (ws_reserve(fn, st, 1, yb, dy)) { return 1; }
    HIPCHK(fn, hipMemcpy(dx, X, xb, hipMemcpyHostToDevice));
    if (win) {
        if (ws_reserve(fn, st, 2, (size_t)d.L * sizeof(T), dw)) { return 1; }
        HIPCHK(fn, hipMemcpy(dw, win, (size_t)d.L * sizeof(T), hipMemcpyHostToDevice));
    }
    HIPCHK(fn, launch_frames<T>((T *)dy, (const T *)dx, (const T *)dw, d, E, nullptr));
    HIPCHK(fn, hipMemcpy(Y, dy, yb, hipMemcpyDeviceToHost));
    return 0;
}

// reference c/frame_univar.c:43-48
template <typename T>
int frame_univar_host(const char *fn, T *Y, const T *X, size_t N, size_t L, size_t stp, int snip_edges, int E, const T *win)
{
    if (L < 1u) { fprintf(stderr, "error in %s: L must be positive\n", fn); return 1; }
    if (stp < 1u) { fprintf(stderr, "error in %s: stp must be positive\n", fn); return 1; }
    if (snip_edges && L > N) { fprintf(stderr, "error in %s: L must be < N if snip_edges\n", fn); return 1; }
    FrameDesc d; size_t W;
    size_t nfft = 2; while (nfft < L) { nfft *= 2; }
    if (stft_desc(fn, d, 1, N, N, L, nfft, stp, snip_edges, 0, 0, 0, W)) { return 1; }
    if (W == 0) { return 0; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    return frames_host<T>(fn, Y, X, d, E, dev, win);
}

template <typename T>
int frame_univar_flt_host(const char *fn, T *Y, const T *X, size_t N, size_t L, size_t W, T c0, T stp, int E, const T *win)
{
    FrameDesc d;
    int dev = 0;
    std::shared_ptr<int> hold;
    if (W > N || L > N || L < 1u || N < 1u || c0 > (T)(N - 1u) || stp < type_eps<T>() || W == 0) {
        return flt_desc<T>(fn, 0, d, 1, N, N, L, W, 2, c0, stp, 0, 0, 0, 0, false, hold);
    }
    if (use_device(fn, dev)) { return 1; }
    if (flt_desc<T>(fn, dev, d, 1, N, N, L, W, 2, c0, stp, 0, 0, 0, 0, false, hold)) { return 1; }
    return frames_host<T>(fn, Y, X, d, E, dev, win);
}

// reference c/apply_win.c:28-34
template <typename T>
int apply_win_host(const char *fn, T *Y, const T *X1, const T *X2, size_t R, size_t C, size_t S, size_t H, size_t L, size_t dim,
                   int iscolmajor, int cplx)
{
    if (dim > 3u) { fprintf(stderr, "error in %s: dim must be in [0 3]\n", fn); return 1; }
    if (dim == 0u && R != L) { fprintf(stderr, "error in %s: L (winlength) must equal R (nrows X) for dim=0\n", fn); return 1; }
    if (dim == 1u && C != L) { fprintf(stderr, "error in %s: L (winlength) must equal C (ncols X) for dim=1\n", fn); return 1; }
    if (dim == 2u && S != L) { fprintf(stderr, "error in %s: L (winlength) must equal S (nslices X) for dim=2\n", fn); return 1; }
    if (dim == 3u && H != L) { fprintf(stderr, "error in %s: L (winlength) must equal H (nhyperslices X) for dim=3\n", fn); return 1; }
    const size_t N = R * C * S * H;
    if (N == 0) { return 0; }
    const size_t sz[4] = {R, C, S, H};
    size_t stride = 1;
    if (iscolmajor) { for (size_t k = 0; k < dim; ++k) { stride *= sz[k]; } }
    else { for (size_t k = 3; k > dim; --k) { stride *= sz[k]; } }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    const size_t E = cplx ? 2 : 1;
    void *dx, *dw;
    if (ws_reserve(fn, st, 0, N * E * sizeof(T), dx)) { return 1; }
    if (ws_reserve(fn, st, 2, L * E * sizeof(T), dw)) { return 1; }
    HIPCHK(fn, hipMemcpy(dx, X1, N * E * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(fn, hipMemcpy(dw, X2, L * E * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(fn, launch_apply_win<T>((T *)dx, (const T *)dx, (const T *)dw, N, stride, L, cplx, nullptr));
    HIPCHK(fn, hipMemcpy(Y, dx, N * E * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

// reference c/fft.fftw.r2hc.c:26-39
template <typename T>
int r2hc_host(const char *fn, T *Y, const T *X, size_t R, size_t C, size_t S, size_t H, int iscolmajor, size_t dim, size_t nfft, int sc)
{
    if (dim > 3u) { fprintf(stderr, "error in %s: dim must be in [0 3]\n", fn); return 1; }
    const size_t sz[4] = {R, C, S, H};
    const size_t N = R * C * S * H, Lx = sz[dim], Ly = nfft / 2u + 1u;
    if (nfft < Lx) {                              // c/fft.fftw.r2hc.c:32 and c/fft.fftw.c:36 word it differently
        const bool r2hc = strncmp(fn, "fft_fftw_r2hc", 13) == 0;
        fprintf(stderr, r2hc ? "error in %s: nfft must be >= L (vec length)\n" : "error in %s: nfft must be >= Lx (vec length of vecs in X)\n", fn);
        return 1;
    }
    if (nfft == 0u || N == 0u) { return 0; }
    if (nfft > (size_t)INT_MAX / 4) { fprintf(stderr, "error in %s: nfft exceeds the supported index range\n", fn); return 1; }
    size_t K = 1;
    if (iscolmajor) { for (size_t k = 0; k < dim; ++k) { K *= sz[k]; } }
    else { for (size_t k = 3; k > dim; --k) { K *= sz[k]; } }
    const size_t V = N / Lx;
    const int pow2 = (nfft >= 2 && is_pow2(nfft)) ? 1 : 0;
    if ((pow2 ? nfft : Lx) * sizeof(T) > LDS_LIMIT) { fprintf(stderr, "error in %s: nfft too large (limit %zu)\n", fn, LDS_LIMIT / sizeof(T)); return 1; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    const T *tw = nullptr;
    if (get_twiddles<T>(fn, dev, nfft, pow2 ? 0 : 1, tw)) { return 1; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    void *dx, *dy;
    const size_t yb = 2 * V * Ly * sizeof(T);
    if (ws_reserve(fn, st, 0, N * sizeof(T), dx)) { return 1; }
    if (ws_reserve(fn, st, 1, yb, dy)) { return 1; }
    HIPCHK(fn, hipMemcpy(dx, X, N * sizeof(T), hipMemcpyHostToDevice));
    const T scale = sc ? (T)(1.0 / sqrt((double)(2u * nfft))) : T(1);
    HIPCHK(fn, launch_r2hc<T>((T *)dy, (const T *)dx, tw, V, K, Lx, nfft, pow2, scale, nullptr));
    HIPCHK(fn, hipMemcpy(Y, dy, yb, hipMemcpyDeviceToHost));
    return 0;
}

// reference c/fft.fftw.c:194-272 (complex input, nfft complex outputs along dim; sc: 1/sqrt(2 nfft))
template <typename T>
int c2c_host(const char *fn, T *Y, const T *X, size_t R, size_t C, size_t S, size_t H, int iscolmajor, size_t dim, size_t nfft, int sc)
{
    if (dim > 3u) { fprintf(stderr, "error in %s: dim must be in [0 3]\n", fn); return 1; }
    const size_t sz[4] = {R, C, S, H};
    const size_t N = R * C * S * H, Lx = sz[dim];
    if (nfft < Lx) { fprintf(stderr, "error in %s: nfft must be >= Lx (vec length of vecs in X)\n", fn); return 1; }
    if (nfft == 0u || N == 0u) { return 0; }
    if (nfft > (size_t)INT_MAX / 4) { fprintf(stderr, "error in %s: nfft exceeds the supported index range\n", fn); return 1; }
    size_t K = 1;
    if (iscolmajor) { for (size_t k = 0; k < dim; ++k) { K *= sz[k]; } }
    else { for (size_t k = 3; k > dim; --k) { K *= sz[k]; } }
    const size_t V = N / Lx;
    const int pow2 = (nfft >= 2 && is_pow2(nfft)) ? 1 : 0;
    if (2 * (pow2 ? nfft : Lx) * sizeof(T) > LDS_LIMIT) { fprintf(stderr, "error in %s: nfft too large (limit %zu)\n", fn, LDS_LIMIT / (2 * sizeof(T))); return 1; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    const T *tw = nullptr;
    if (get_twiddles<T>(fn, dev, nfft, pow2 ? 0 : 1, tw)) { return 1; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    void *dx, *dy;
    const size_t yb = 2 * V * nfft * sizeof(T);
    if (ws_reserve(fn, st, 0, 2 * N * sizeof(T), dx)) { return 1; }
    if (ws_reserve(fn, st, 1, yb, dy)) { return 1; }
    HIPCHK(fn, hipMemcpy(dx, X, 2 * N * sizeof(T), hipMemcpyHostToDevice));
    const T scale = sc ? (T)(1.0 / sqrt((double)(2u * nfft))) : T(1);
    HIPCHK(fn, launch_c2c<T>((T *)dy, (const T *)dx, tw, V, K, Lx, nfft, pow2, scale, nullptr));
    HIPCHK(fn, hipMemcpy(Y, dy, yb, hipMemcpyDeviceToHost));
    return 0;
}

// the instantiations the other translation units link against
template int pow_compress_host<float>(const char *fn, float *Y, const float *X, size_t N, float p, float preg);
template int pow_compress_host<double>(const char *fn, double *Y, const double *X, size_t N, double p, double preg);
template int pow_compress_dev<float>(const char *fn, float *Yd, const float *Xd, size_t N, float p, float preg, void *stream);
template int pow_compress_dev<double>(const char *fn, double *Yd, const double *Xd, size_t N, double p, double preg, void *stream);
template bool window_par_ok<float>(const char *fn, int kind, float par);
template bool window_par_ok<double>(const char *fn, int kind, double par);
template int window_host<float>(const char *fn, float *Y, size_t L, size_t norm, int kind, float par, int flag);
template int window_host<double>(const char *fn, double *Y, size_t L, size_t norm, int kind, double par, int flag);
template int window_host_cplx<float>(const char *fn, float *Y, size_t L, size_t norm, int kind, float par, int flag);
template int window_host_cplx<double>(const char *fn, double *Y, size_t L, size_t norm, int kind, double par, int flag);
template int window_dev<float>(const char *fn, float *Yd, int kind, size_t L, size_t norm, void *stream, float par, int flag);
template int window_dev<double>(const char *fn, double *Yd, int kind, size_t L, size_t norm, void *stream, double par, int flag);
template int stft_freqs<float>(const char *fn, float *Y, size_t nfft, size_t F, float fs);
template int stft_freqs<double>(const char *fn, double *Y, size_t nfft, size_t F, double fs);
template int convert_freqs_host<float>(const char *fn, float *frqs, size_t F, const char *in_scale, const char *out_scale);
template int convert_freqs_host<double>(const char *fn, double *frqs, size_t F, const char *in_scale, const char *out_scale);
template int fft_power_host<float>(const char *fn, float *Y, const float *X, size_t N);
template int fft_power_host<double>(const char *fn, double *Y, const double *X, size_t N);
template int frames_host<float>(const char *fn, float *Y, const float *X, const FrameDesc &d, int E, int dev, const float *win);
template int frames_host<double>(const char *fn, double *Y, const double *X, const FrameDesc &d, int E, int dev, const double *win);
template int frame_univar_host<float>(const char *fn, float *Y, const float *X, size_t N, size_t L, size_t stp, int snip_edges, int E, const float *win);
template int frame_univar_host<double>(const char *fn, double *Y, const double *X, size_t N, size_t L, size_t stp, int snip_edges, int E, const double *win);
template int frame_univar_flt_host<float>(const char *fn, float *Y, const float *X, size_t N, size_t L, size_t W, float c0, float stp, int E, const float *win);
template int frame_univar_flt_host<double>(const char *fn, double *Y, const double *X, size_t N, size_t L, size_t W, double c0, double stp, int E, const double *win);
template int apply_win_host<float>(const char *fn, float *Y, const float *X1, const float *X2, size_t R, size_t C, size_t S, size_t H, size_t L, size_t dim,
                   int iscolmajor, int cplx);
template int apply_win_host<double>(const char *fn, double *Y, const double *X1, const double *X2, size_t R, size_t C, size_t S, size_t H, size_t L, size_t dim,
                   int iscolmajor, int cplx);
template int r2hc_host<float>(const char *fn, float *Y, const float *X, size_t R, size_t C, size_t S, size_t H, int iscolmajor, size_t dim, size_t nfft, int sc);
template int r2hc_host<double>(const char *fn, double *Y, const double *X, size_t R, size_t C, size_t S, size_t H, int iscolmajor, size_t dim, size_t nfft, int sc);
template int c2c_host<float>(const char *fn, float *Y, const float *X, size_t R, size_t C, size_t S, size_t H, int iscolmajor, size_t dim, size_t nfft, int sc);
template int c2c_host<double>(const char *fn, double *Y, const double *X, size_t R, size_t C, size_t S, size_t H, int iscolmajor, size_t dim, size_t nfft, int sc);

}  // namespace abi
}  // namespace dsp
