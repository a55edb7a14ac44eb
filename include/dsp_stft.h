/*
 * dsp_stft.h -- C ABI of libdsp_stft_hip.so: the STFT power-spectrogram path of the dsp library
 * (framing, windowing, real FFT, power) computed by hand-written HIP kernels for gfx950 (MI355X).
 *
 * Part 1 keeps the reference's entry points byte for byte (names, argument order and meaning,
 * 0 / 1 return code, one "error in <fn>: ..." line on stderr).  All pointers in part 1 are HOST
 * pointers; results are in Y when the call returns.  The reference has no headers: each of its
 * command-line tools #includes the .c file that holds these declarations, so the citation on each
 * block is the declaration this one replaces.
 *
 * Part 2 is additional: batched and device-resident forms of the same computation, plus set-up and
 * diagnostics.  Nothing in part 2 exists in the reference.
 *
 * There is no CPU fallback: when no HIP device is usable every compute entry point returns 1 with an
 * "error in <fn>: hip..." line.
 */
#ifndef DSP_STFT_H
#define DSP_STFT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ===================== Part 1: the reference's entry points (host pointers) ===================== */

/* reference c/stft.c:29-30.  Y[W][F] (frame-major), F = nfft/2+1,
 * W = snip_edges ? 1+(N-L)/stp : (N+stp/2)/stp.  mn0: subtract the frame mean after windowing;
 * amp: sqrt of power; lg: natural log of the result.  Differences from the reference, both where it
 * has undefined behaviour: nfft < L and nfft < 2 are rejected with return 1. */
int stft_s (float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg);
int stft_d (double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg);

/* reference c/stft_flt.c:29-30.  W frames centred at round(c0 + w*stp) (centre accumulated in the
 * data type, rounding half away from zero), zeros outside [0,N-1].  The default framing reproduces
 * what the reference's code actually computes (see dsp_stft_flt_set_mode). */
int stft_flt_s (float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg);
int stft_flt_d (double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg);

/* reference c/fft_power.c:13-14.  Y[n] = re^2 + im^2 over N interleaved complex values. */
int fft_power_c (float *Y, const float *X, const size_t N);
int fft_power_z (double *Y, const double *X, const size_t N);

/* reference c/frame_univar.c:35-38.  Y[W][L] frames of X with the stft framing (edge reflection). */
int frame_univar_s (float *Y, const float *X, const size_t N, const size_t L, const size_t stp, const int snip_edges);
int frame_univar_d (double *Y, const double *X, const size_t N, const size_t L, const size_t stp, const int snip_edges);
int frame_univar_c (float *Y, const float *X, const size_t N, const size_t L, const size_t stp, const int snip_edges);
int frame_univar_z (double *Y, const double *X, const size_t N, const size_t L, const size_t stp, const int snip_edges);

/* reference c/frame_univar_flt.c:23-26.  Y[W][L] frames with the _flt framing (zeros outside X). */
int frame_univar_flt_s (float *Y, const float *X, const size_t N, const size_t L, const size_t W, const float c0, const float stp);
int frame_univar_flt_d (double *Y, const double *X, const size_t N, const size_t L, const size_t W, const double c0, const double stp);
int frame_univar_flt_c (float *Y, const float *X, const size_t N, const size_t L, const size_t W, const float c0, const float stp);
int frame_univar_flt_z (double *Y, const double *X, const size_t N, const size_t L, const size_t W, const double c0, const double stp);

/* reference c/window_univar.c:35-36, c/window_univar_flt.c:23-24.  Framing and windowing in one
 * step: Y[W][L] = frame_univar*(X1) times X2 along each frame (the real forms; X2 has L values). */
int window_univar_s (float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t stp, const int snip_edges);
int window_univar_d (double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t stp, const int snip_edges);
int window_univar_flt_s (float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t W, const float c0, const float stp);
int window_univar_flt_d (double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t W, const double c0, const double stp);
/* reference c/window_univar.c:37-38, c/window_univar_flt.c:25-26.  The same for a complex signal
 * (N interleaved values) and a REAL window of L values: both parts of a sample are multiplied by the
 * window value.  (After a frame that lies inside the signal the reference's window_univar_c/_z steps
 * back L-stp floats instead of complex values, c/window_univar.c:156-159,184-186, and reads past the
 * signal unless stp == L; the frames of frame_univar_c/_z times the window are returned here.) */
int window_univar_c (float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t stp, const int snip_edges);
int window_univar_z (double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t stp, const int snip_edges);
int window_univar_flt_c (float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t W, const float c0, const float stp);
int window_univar_flt_z (double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t W, const double c0, const double stp);

/* reference c/apply_win.c:15-23.  Y = X1 .* X2, X2 (length L) broadcast along dim of the
 * R x C x S x H tensor X1, row- or column-major; the _c/_z forms multiply complex by complex.
 * The map is index-exact (X2 indexed by the coordinate along dim); the reference's strided branch
 * walks X2 past its end for tensors of more than two non-singleton axes, which is not reproduced. */
int apply_win_s (float *Y, const float *X1, const float *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor);
int apply_win_d (double *Y, const double *X1, const double *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor);
int apply_win_c (float *Y, const float *X1, const float *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor);
int apply_win_z (double *Y, const double *X1, const double *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor);
int apply_win_inplace_s (float *X1, const float *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor);
int apply_win_inplace_d (double *X1, const double *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor);
int apply_win_inplace_c (float *X1, const float *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor);
int apply_win_inplace_z (double *X1, const double *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor);

/* reference c/fft.fftw.r2hc.c:20-21.  Real FFT of every vector along dim, zero-padded to nfft;
 * Y has nfft/2+1 interleaved complex values along dim, imag(DC) = imag(Nyquist) = 0.  sc scales
 * by 1/sqrt(2 nfft).  (The reference's single-vector branch and its sc loop read/write out of
 * bounds, c/fft.fftw.r2hc.c:56-59,101-105; the mathematically intended result is returned.) */
int fft_fftw_r2hc_s (float *Y, const float *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc);
int fft_fftw_r2hc_d (double *Y, const double *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc);
/* reference c/fft.fftw.c:23-26.  FFT of every vector along dim of a tensor of up to 4 dimensions,
 * zero-padded to nfft.  _s/_d: real input, nfft/2+1 interleaved complex values along dim (the values
 * of fft_fftw_r2hc); _c/_z: interleaved complex input, nfft complex values along dim.  sc scales by
 * 1/sqrt(2 nfft). */
int fft_fftw_s (float *Y, const float *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc);
int fft_fftw_d (double *Y, const double *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc);
int fft_fftw_c (float *Y, const float *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc);
int fft_fftw_z (double *Y, const double *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc);

/* reference c/hann.c:15-16, c/hamming.c:15-16, c/povey.c:15-16, c/blackmanharris.c:16-17.  The
 * window X2 of stft: symmetric, period L-1, L values; norm 0 none, 1 unit sum, 2 unit energy,
 * 3 divided by element L/2.  The _c/_z forms (reference c/hann.c:17-18 and the like) write the same L values as
 * interleaved complex numbers with a zero imaginary part (2 L elements). */
int hann_s (float *Y, const size_t L, const size_t norm);
int hann_d (double *Y, const size_t L, const size_t norm);
int hamming_s (float *Y, const size_t L, const size_t norm);
int hamming_d (double *Y, const size_t L, const size_t norm);
int povey_s (float *Y, const size_t L, const size_t norm);
int povey_d (double *Y, const size_t L, const size_t norm);
int blackmanharris_s (float *Y, const size_t L, const size_t norm);
int blackmanharris_d (double *Y, const size_t L, const size_t norm);
/* reference c/blackman.c:20-21 (exact != 0: the coefficients 7938, 9240, 1430 over 18608),
 * c/bartlett.c:12-13, c/triangular.c:12-13, c/rectangular.c:11-12, c/flattop.c:17-18 (norm 1: sum of
 * magnitudes), c/gauss.c:11-12 (exp(-2 (a (l - (L-1)/2) / (L-1))^2), a > 0), c/tukey.c:15-16 (cosine
 * lobes over ceil(r L / 2) values at each end, r in [0,1]), c/planck.c:11-12 (Planck tapers over
 * ceil(epsilon L) values, epsilon in [0,0.5]).  Two cases in which the reference writes outside the
 * L values or leaves one unwritten are defined here: planck with epsilon 0 is all ones, and with an
 * odd L and ceil(epsilon L) > L/2 the centre value comes from the taper formula. */
int blackman_s (float *Y, const size_t L, const int exact, const size_t norm);
int blackman_d (double *Y, const size_t L, const int exact, const size_t norm);
int bartlett_s (float *Y, const size_t L, const size_t norm);
int bartlett_d (double *Y, const size_t L, const size_t norm);
int triangular_s (float *Y, const size_t L, const size_t norm);
int triangular_d (double *Y, const size_t L, const size_t norm);
int rectangular_s (float *Y, const size_t L, const size_t norm);
int rectangular_d (double *Y, const size_t L, const size_t norm);
int flattop_s (float *Y, const size_t L, const size_t norm);
int flattop_d (double *Y, const size_t L, const size_t norm);
int gauss_s (float *Y, const size_t L, const float r, const size_t norm);
int gauss_d (double *Y, const size_t L, const double r, const size_t norm);
int tukey_s (float *Y, const size_t L, const float r, const size_t norm);
int tukey_d (double *Y, const size_t L, const double r, const size_t norm);
int planck_s (float *Y, const size_t L, const float epsilon, const size_t norm);
int planck_d (double *Y, const size_t L, const double epsilon, const size_t norm);
/* the complex forms of the twelve generators (reference c/<window>.c, the _c and _z functions).  gauss_c/_z return the
 * symmetric window; the reference mirrors their second half with the real form's offset (c/gauss.c:78,107) and
 * copies the wrong elements there. */
int hann_c (float *Y, const size_t L, const size_t norm);
int hann_z (double *Y, const size_t L, const size_t norm);
int hamming_c (float *Y, const size_t L, const size_t norm);
int hamming_z (double *Y, const size_t L, const size_t norm);
int povey_c (float *Y, const size_t L, const size_t norm);
int povey_z (double *Y, const size_t L, const size_t norm);
int blackmanharris_c (float *Y, const size_t L, const size_t norm);
int blackmanharris_z (double *Y, const size_t L, const size_t norm);
int bartlett_c (float *Y, const size_t L, const size_t norm);
int bartlett_z (double *Y, const size_t L, const size_t norm);
int triangular_c (float *Y, const size_t L, const size_t norm);
int triangular_z (double *Y, const size_t L, const size_t norm);
int rectangular_c (float *Y, const size_t L, const size_t norm);
int rectangular_z (double *Y, const size_t L, const size_t norm);
int flattop_c (float *Y, const size_t L, const size_t norm);
int flattop_z (double *Y, const size_t L, const size_t norm);
int blackman_c (float *Y, const size_t L, const int exact, const size_t norm);
int blackman_z (double *Y, const size_t L, const int exact, const size_t norm);
int gauss_c (float *Y, const size_t L, const float r, const size_t norm);
int gauss_z (double *Y, const size_t L, const double r, const size_t norm);
int tukey_c (float *Y, const size_t L, const float r, const size_t norm);
int tukey_z (double *Y, const size_t L, const double r, const size_t norm);
int planck_c (float *Y, const size_t L, const float epsilon, const size_t norm);
int planck_z (double *Y, const size_t L, const double epsilon, const size_t norm);

/* reference c/pow_compress.c:17-20.  Y[n] = f(X[n] + preg) over N values: p == 1 identity, p == 0
 * natural log, p == 0.5 sqrt, |p - 1/3| <= FLT_EPSILON cbrt, otherwise pow(., p); p in [0,1],
 * preg >= 0.  The step after stft in the reference's pipeline (the fused form is in part 2). */
int pow_compress_s (float *Y, const float *X, const size_t N, const float p, const float preg);
int pow_compress_d (double *Y, const double *X, const size_t N, const double p, const double preg);
int pow_compress_inplace_s (float *X, const size_t N, const float p, const float preg);
int pow_compress_inplace_d (double *X, const size_t N, const double p, const double preg);

/* reference c/get_stft_freqs.c:13-14.  The first F bin frequencies of an nfft-point STFT at sample
 * rate fs: f*fs/nfft up to Nyquist, negative frequencies after it.  F <= nfft values, computed on
 * the host (metadata, not part of the transform). */
int get_stft_freqs_s (float *Y, const size_t nfft, const size_t F, const float fs);
int get_stft_freqs_d (double *Y, const size_t nfft, const size_t F, const double fs);
/* reference c/convert_freqs.c:14-15.  The F values of a frequency axis, in place, from one scale to
 * another: "hz", "sqrt", "cbrt", "octave" (log2).  Host array, computed on the host like the axis. */
int convert_freqs_s (float *frqs, const size_t F, const char in_scale[], const char out_scale[]);
int convert_freqs_d (double *frqs, const size_t F, const char in_scale[], const char out_scale[]);


/* ===================== Part 2: additions (no counterpart in the reference) ===================== */

/* Number of frames of stft_s/_d (reference formula, c/stft.c:40); 0 when the arguments are invalid. */
size_t dsp_stft_num_frames (const size_t N, const size_t L, const size_t stp, const int snip_edges);

/* Selects the HIP device used by the calling thread's later calls (default: the current device).  This,
 * dsp_stft_flt_set_mode and dsp_stft_force_generic are settings of the CALLING THREAD: threads that want
 * different devices or framings do not disturb each other, and a new thread starts with the defaults. */
int dsp_set_device (const int device);
/* Frees what the host-pointer entry points keep between calls on the calling thread's device: the device workspace
 * (as large as the largest unstreamed call so far), the pinned stages of the streamed path and the cached stft_flt
 * frame tables.  For long-running hosts; the next call allocates again what it needs. */
int dsp_stft_release_buffers (void);

/* stft_flt framing: 0 (default) = what the reference's code computes: once frames stop overlapping
 * the left edge every frame is read `shift` samples early, shift = start of the first frame whose
 * start is >= 0 (c/stft_flt.c:72-82 never advances X1 to that start); 1 = the documented framing of
 * c/window_univar_flt.c / c/frame_univar_flt.c.  Both agree when that first start is 0. */
int dsp_stft_flt_set_mode (const int intended);

/* Batched, DEVICE-resident STFT: B signals of N samples, signal b at X_dev + b*ldx; one window;
 * output Y_dev[b][W][F] dense.  Pointers are device pointers; the work is enqueued on `stream`
 * (a hipStream_t, NULL = default stream) and NOT synchronised.  Flags as stft_s.
 * The first use of an nfft on a device builds its twiddle table (an allocation and a blocking copy; the first use
 * of a non-integral stft_flt framing likewise builds its table of frame starts); calls after that only enqueue.
 * Every entry point of this library returns, for a given signal, values that depend on that signal and the
 * arguments only: the host-pointer, batched, strided (but see the in-place case there) and multi-GPU forms give the same bits as this one (the
 * allocation-free form computes its twiddle tables on the device instead of taking the host's: the same values
 * unless a table entry lies within 1e-16 of a rounding boundary, so equal bits in practice, the tolerance of the
 * path by contract).  One exception: nfft 4096 with a full window and hop 1024 walks a signal in runs of 32 frames when the
 * batch has at least 4096 such runs and frame by frame when it has fewer; the two walks differ in the last bit, so
 * a signal transformed alone (few runs) and inside a large batch may differ by one ulp. */
int dsp_stft_batch_dev_s (float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream);
int dsp_stft_batch_dev_d (double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream);

/* No counterpart in the reference.  dsp_stft_batch_dev_s for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): float32, nfft a
 * power of two <= 4096 (any other nfft returns 1 with an error line).  It calls no hipMalloc / hipFree:
 * the twiddle tables and the work counters its kernels use live in one 33 KB array of the library's code
 * object (each device the library runs on holds its copy until the process ends).  The first call on a
 * device computes the tables with a kernel in `stream` and waits for that stream once -- so the FIRST call on a
 * device must be made outside a graph capture (a stream that is being captured cannot be waited for); later
 * calls only enqueue and may be captured.  Every call on a device shares that array: the counters are cleared in `stream` before each
 * 4096-point launch, so calls that OVERLAP on one device (different streams) race; calls in one stream,
 * or one after another, do not.  Same arguments, output and error behaviour as dsp_stft_batch_dev_s. */
int dsp_stft_batch_dev_noalloc_s (float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream);

/* Batched, device-resident stft_flt (same conventions; W frames per signal). */
int dsp_stft_flt_batch_dev_s (float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg, void *stream);
int dsp_stft_flt_batch_dev_d (double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg, void *stream);

/* The same for signals with a sample stride: sample n of signal b at X_dev[b*ldx + n*incx].  A multivariate
 * tensor with interleaved channels (row-major [N][C]: the reference's tensors have stride K along the
 * transformed axis, c/fft.fftw.r2hc.c:63-65) is B = C, ldx = 1, incx = C.  The signals are first gathered into a
 * dense [B][N] buffer that lives for the call (allocated and freed in `stream`'s order, B*N elements), so the
 * output has the bits of the dense entry point; incx = 1 is the dense entry point itself.
 * One case is read in place, without the buffer: dsp_stft_batch_dev_strided_s with nfft 512 on 4, 8, ... 32 adjacent
 * channels (ldx = 1, B and incx multiples of 4, X_dev 16-byte aligned), where the kernel fetches one frame of 4
 * channels per step (2 to 3.5 times faster than the gather up to 16 channels, equal at 32).  Its values agree with
 * the dense entry point's to rounding (for windows of 385 to 416 samples the dense kernel is another instantiation);
 * the environment variable DSP_ILV_MAX_CHANNELS moves the limit of 32 (0: always gather). */
int dsp_stft_batch_dev_strided_s (float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream);
int dsp_stft_batch_dev_strided_d (double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream);
int dsp_stft_flt_batch_dev_strided_s (float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg, void *stream);
int dsp_stft_flt_batch_dev_strided_d (double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg, void *stream);

/* The complex spectrogram of B signals: the frames of stft (same W, same edge reflection) times the window, then the
 * real FFT of each as fft_fftw_r2hc_s/_d returns it: Y[b][w][F] interleaved (re, im) pairs, F = nfft/2+1, the imaginary
 * parts of bins 0 and nfft/2 zero; sc != 0 scales by 1/sqrt(2 nfft) (reference c/window_univar.c:41-89 followed by
 * c/fft.fftw.r2hc.c:24-109 along the frames, fused).  nfft 64 / 128 / 256 / 512 / 1024 / 2048 / 4096 run on the
 * fused kernels (the rows leave as complex values instead of powers), anything else on the unfused kernels. */
int dsp_stft_cplx_batch_dev_s (float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int sc, void *stream);
int dsp_stft_cplx_batch_dev_d (double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int sc, void *stream);

/* Batched host-pointer forms: X[b*ldx + n], Y[b][W][F]; copies in, computes, copies out.  Calls that move 64 MB
 * and more (these, and stft_s/_d and stft_flt_s/_d on one long signal) are streamed through pinned
 * stages in chunks -- groups of signals, or ranges of the frames of one signal with the samples they read -- with
 * copy in, kernel and copy out overlapped; the staging copies run on up to 12 threads, DSP_COPY_THREADS overrides. */
int dsp_stft_batch_s (float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg);
int dsp_stft_batch_d (double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg);

/* The same for the stft_flt framing (one window, one W, c0 and stp for all B signals: the 64 channels of a multivariate
 * recording in one call instead of 64 stft_flt_d calls); streamed like the above. */
int dsp_stft_flt_batch_s (float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg);
int dsp_stft_flt_batch_d (double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg);

/* Host-pointer forms for signals with a sample stride (sample n of signal b at X[b*ldx + n*incx]): what the tools call
 * for a multivariate file whose channels are interleaved (row-major [N][C]: B = C, ldx = 1, incx = C; the layout of
 * c/fft.fftw.r2hc.c:63-65).  The span of X that holds the signals is copied to the device as it lies, the signals are
 * gathered there (or read in place, as dsp_stft_batch_dev_strided_s describes), Y[b][W][F] comes back dense.  incx = 1
 * is dsp_stft_batch_s/_d and dsp_stft_flt_batch_s/_d. */
int dsp_stft_batch_strided_s (float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg);
int dsp_stft_batch_strided_d (double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg);
int dsp_stft_flt_batch_strided_s (float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg);
int dsp_stft_flt_batch_strided_d (double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg);

/* Several GPUs from one process (the reference is one process, src/stft.cpp:219).  The B signals are split in
 * contiguous blocks, GPU g of ngpu taking signals [lo, hi) = dsp_shard_range(B, g, ngpu): blocks of
 * ceil(B / ngpu), the last ones short or empty.  No data crosses between GPUs on the way: every signal is
 * transformed where it lies.  devices: ngpu device indices, or NULL for 0 .. ngpu-1.
 * dsp_stft_multi_s/_d: host pointers as dsp_stft_batch_s/_d; every GPU runs its block on a thread of its own
 * and writes it to its place in Y (the gather is the copy out).
 * dsp_stft_multi_dev_s/_d: X_dev[g], win_dev[g], Y_dev[g] are GPU g's input block, window and output block
 * (device memory of devices[g]); Y_all_dev, if not NULL, is memory of devices[0] for the whole Y[B][W][F], and
 * every block is copied into it over the GPU-to-GPU links as soon as it is complete (a block that already
 * lies there is not copied): that copy needs peer access of devices[g] to devices[0], which the call enables
 * and, where the machine does not grant it, reports as an error (return 1) before any work starts.  The
 * kernels run on streams of the call's own on every GPU, not ordered with any stream of the caller: whatever
 * produces X_dev[g] and win_dev[g] must have finished on devices[g] before the call.  The calling thread's
 * current device is the same after the call as before it.  Both return when all work has finished. */
void dsp_shard_range (const size_t B, const size_t rank, const size_t world, size_t *lo, size_t *hi);
int dsp_stft_multi_s (float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const int *devices, const int ngpu);
int dsp_stft_multi_d (double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const int *devices, const int ngpu);
/* dsp_stft_flt_multi_s/_d: the same split for the stft_flt framing, host pointers as dsp_stft_flt_batch_s/_d. */
int dsp_stft_flt_multi_s (float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg, const int *devices, const int ngpu);
int dsp_stft_flt_multi_d (double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg, const int *devices, const int ngpu);
int dsp_stft_multi_dev_s (float *const *Y_dev, const float *const *X_dev, const float *const *win_dev, float *Y_all_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const int *devices, const int ngpu);
int dsp_stft_multi_dev_d (double *const *Y_dev, const double *const *X_dev, const double *const *win_dev, double *Y_all_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const int *devices, const int ngpu);

/* STFT with the power compression of pow_compress_s/_d fused into the store of every value
 * (power -> [sqrt] -> [log] -> compress): the spectrogram is written once instead of written,
 * re-read and re-written.  Device-pointer and host-pointer batched forms. */
int dsp_stft_pc_batch_dev_s (float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const float p, const float preg, void *stream);
int dsp_stft_pc_batch_dev_d (double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const double p, const double preg, void *stream);
int dsp_stft_pc_batch_s (float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const float p, const float preg);
int dsp_stft_pc_batch_d (double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const double p, const double preg);

/* Device-resident pow_compress over N values (Y_dev may equal X_dev), enqueued on `stream`. */
int dsp_pow_compress_dev_s (float *Y_dev, const float *X_dev, const size_t N, const float p, const float preg, void *stream);
int dsp_pow_compress_dev_d (double *Y_dev, const double *X_dev, const size_t N, const double p, const double preg, void *stream);

/* Window generated straight into device memory (window: 0 hann, 1 hamming, 2 povey,
 * 3 blackmanharris, 4 blackman, 5 bartlett, 6 triangular, 7 rectangular, 8 flattop, 9 gauss, 10 tukey,
 * 11 planck), enqueued on `stream`: the X2 of the *_dev entry points without a host array.  The _par
 * form takes the generator's parameter (gauss a, tukey r, planck epsilon) and flag (blackman exact);
 * the plain form uses a = 2.5, r = 0.5, epsilon = 0.1, exact = 0. */
int dsp_window_dev_s (float *Y_dev, const int window, const size_t L, const size_t norm, void *stream);
int dsp_window_dev_d (double *Y_dev, const int window, const size_t L, const size_t norm, void *stream);
int dsp_window_par_dev_s (float *Y_dev, const int window, const size_t L, const float par, const int flag, const size_t norm, void *stream);
int dsp_window_par_dev_d (double *Y_dev, const int window, const size_t L, const double par, const int flag, const size_t norm, void *stream);

/* Device-resident fft_power over N interleaved complex values, enqueued on `stream`. */
int dsp_fft_power_dev_c (float *Y_dev, const float *X_dev, const size_t N, void *stream);
int dsp_fft_power_dev_z (double *Y_dev, const double *X_dev, const size_t N, void *stream);

/* Which kernel family the fused path would use for (nfft, element size 4 or 8):
 * 0 = none (invalid), 1 = direct DFT (nfft not a power of two), 2 = generic power-of-two,
 * 3 = register-resident 512-point, 4 = register-resident 4096-point (float32), 5 = the other powers of two from 16
 * to 8192 (float32) / 4096 (float64), 6 = 1024- and 2048-point, 7 = 64-, 128- and 256-point.  Families 6 and 7
 * take frames (equally spaced, or the stft_flt framing of any hop) that keep a step (2, 1 and 8 frames) inside 2048
 * samples; other framings of those lengths run on family 5.  For tests and benchmarks. */
int dsp_stft_kernel_family (const size_t nfft, const size_t elem_size);

/* Forces the generic kernel family (2, or 1 for non-powers-of-two) when on != 0; for A/B tests. */
int dsp_stft_force_generic (const int on);

/* Library version string. */
const char *dsp_stft_version (void);

#ifdef __cplusplus
}
#endif
#endif
