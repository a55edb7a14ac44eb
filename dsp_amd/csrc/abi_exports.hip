// C ABI of libdsp_stft_hip.so: the exported entry points (declarations and reference citations: include/dsp_stft.h), one-line
// forwarders into the host-side templates of abi_core.hip, abi_stream.hip, abi_multi.hip and abi_unfused.hip.
#include "abi_internal.h"

using namespace dsp;
using namespace dsp::abi;

// ================================================================================================
extern "C" {

int dsp_stft_cplx_batch_dev_s(float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int sc, void *stream)
{ return stft_cplx_dev<float>("dsp_stft_cplx_batch_dev_s", Y_dev, X_dev, win_dev, B, N, ldx, L, nfft, stp, snip_edges, sc, stream); }
int dsp_stft_cplx_batch_dev_d(double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int sc, void *stream)
{ return stft_cplx_dev<double>("dsp_stft_cplx_batch_dev_d", Y_dev, X_dev, win_dev, B, N, ldx, L, nfft, stp, snip_edges, sc, stream); }

int stft_s(float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg)
{ return stft_host<float>("stft_s", Y, X1, X2, 1, N, N, L, nfft, stp, snip_edges, mn0, amp, lg); }
int stft_d(double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg)
{ return stft_host<double>("stft_d", Y, X1, X2, 1, N, N, L, nfft, stp, snip_edges, mn0, amp, lg); }

int stft_flt_s(float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg)
{ return stft_flt_host<float>("stft_flt_s", Y, X1, X2, N, L, W, nfft, c0, stp, mn0, amp, lg); }
int stft_flt_d(double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg)
{ return stft_flt_host<double>("stft_flt_d", Y, X1, X2, N, L, W, nfft, c0, stp, mn0, amp, lg); }

int window_univar_s(float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return frame_univar_host<float>("window_univar_s", Y, X1, N, L, stp, snip_edges, 1, X2); }
int window_univar_d(double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return frame_univar_host<double>("window_univar_d", Y, X1, N, L, stp, snip_edges, 1, X2); }
int window_univar_flt_s(float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t W, const float c0, const float stp)
{ return frame_univar_flt_host<float>("window_univar_flt_s", Y, X1, N, L, W, c0, stp, 1, X2); }
int window_univar_flt_d(double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t W, const double c0, const double stp)
{ return frame_univar_flt_host<double>("window_univar_flt_d", Y, X1, N, L, W, c0, stp, 1, X2); }
int window_univar_c(float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return frame_univar_host<float>("window_univar_c", Y, X1, N, L, stp, snip_edges, 2, X2); }
int window_univar_z(double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return frame_univar_host<double>("window_univar_z", Y, X1, N, L, stp, snip_edges, 2, X2); }
int window_univar_flt_c(float *Y, const float *X1, const float *X2, const size_t N, const size_t L, const size_t W, const float c0, const float stp)
{ return frame_univar_flt_host<float>("window_univar_flt_c", Y, X1, N, L, W, c0, stp, 2, X2); }
int window_univar_flt_z(double *Y, const double *X1, const double *X2, const size_t N, const size_t L, const size_t W, const double c0, const double stp)
{ return frame_univar_flt_host<double>("window_univar_flt_z", Y, X1, N, L, W, c0, stp, 2, X2); }
int hann_s(float *Y, const size_t L, const size_t norm) { return window_host<float>("hann_s", Y, L, norm, 0); }
int hann_d(double *Y, const size_t L, const size_t norm) { return window_host<double>("hann_d", Y, L, norm, 0); }
int hamming_s(float *Y, const size_t L, const size_t norm) { return window_host<float>("hamming_s", Y, L, norm, 1); }
int hamming_d(double *Y, const size_t L, const size_t norm) { return window_host<double>("hamming_d", Y, L, norm, 1); }
int povey_s(float *Y, const size_t L, const size_t norm) { return window_host<float>("povey_s", Y, L, norm, 2); }
int povey_d(double *Y, const size_t L, const size_t norm) { return window_host<double>("povey_d", Y, L, norm, 2); }
int blackmanharris_s(float *Y, const size_t L, const size_t norm) { return window_host<float>("blackmanharris_s", Y, L, norm, 3); }
int blackmanharris_d(double *Y, const size_t L, const size_t norm) { return window_host<double>("blackmanharris_d", Y, L, norm, 3); }
int blackman_s(float *Y, const size_t L, const int exact, const size_t norm) { return window_host<float>("blackman_s", Y, L, norm, WIN_BLACKMAN, 0.f, exact != 0); }
int blackman_d(double *Y, const size_t L, const int exact, const size_t norm) { return window_host<double>("blackman_d", Y, L, norm, WIN_BLACKMAN, 0.0, exact != 0); }
int bartlett_s(float *Y, const size_t L, const size_t norm) { return window_host<float>("bartlett_s", Y, L, norm, WIN_BARTLETT); }
int bartlett_d(double *Y, const size_t L, const size_t norm) { return window_host<double>("bartlett_d", Y, L, norm, WIN_BARTLETT); }
int triangular_s(float *Y, const size_t L, const size_t norm) { return window_host<float>("triangular_s", Y, L, norm, WIN_TRIANGULAR); }
int triangular_d(double *Y, const size_t L, const size_t norm) { return window_host<double>("triangular_d", Y, L, norm, WIN_TRIANGULAR); }
int rectangular_s(float *Y, const size_t L, const size_t norm) { return window_host<float>("rectangular_s", Y, L, norm, WIN_RECTANGULAR); }
int rectangular_d(double *Y, const size_t L, const size_t norm) { return window_host<double>("rectangular_d", Y, L, norm, WIN_RECTANGULAR); }
int flattop_s(float *Y, const size_t L, const size_t norm) { return window_host<float>("flattop_s", Y, L, norm, WIN_FLATTOP); }
int flattop_d(double *Y, const size_t L, const size_t norm) { return window_host<double>("flattop_d", Y, L, norm, WIN_FLATTOP); }
int gauss_s(float *Y, const size_t L, const float r, const size_t norm) { return window_host<float>("gauss_s", Y, L, norm, WIN_GAUSS, r); }
int gauss_d(double *Y, const size_t L, const double r, const size_t norm) { return window_host<double>("gauss_d", Y, L, norm, WIN_GAUSS, r); }
int tukey_s(float *Y, const size_t L, const float r, const size_t norm) { return window_host<float>("tukey_s", Y, L, norm, WIN_TUKEY, r); }
int tukey_d(double *Y, const size_t L, const double r, const size_t norm) { return window_host<double>("tukey_d", Y, L, norm, WIN_TUKEY, r); }
int planck_s(float *Y, const size_t L, const float epsilon, const size_t norm) { return window_host<float>("planck_s", Y, L, norm, WIN_PLANCK, epsilon); }
int planck_d(double *Y, const size_t L, const double epsilon, const size_t norm) { return window_host<double>("planck_d", Y, L, norm, WIN_PLANCK, epsilon); }
int hann_c(float *Y, const size_t L, const size_t norm) { return window_host_cplx<float>("hann_c", Y, L, norm, 0); }
int hann_z(double *Y, const size_t L, const size_t norm) { return window_host_cplx<double>("hann_z", Y, L, norm, 0); }
int hamming_c(float *Y, const size_t L, const size_t norm) { return window_host_cplx<float>("hamming_c", Y, L, norm, 1); }
int hamming_z(double *Y, const size_t L, const size_t norm) { return window_host_cplx<double>("hamming_z", Y, L, norm, 1); }
int povey_c(float *Y, const size_t L, const size_t norm) { return window_host_cplx<float>("povey_c", Y, L, norm, 2); }
int povey_z(double *Y, const size_t L, const size_t norm) { return window_host_cplx<double>("povey_z", Y, L, norm, 2); }
int blackmanharris_c(float *Y, const size_t L, const size_t norm) { return window_host_cplx<float>("blackmanharris_c", Y, L, norm, 3); }
int blackmanharris_z(double *Y, const size_t L, const size_t norm) { return window_host_cplx<double>("blackmanharris_z", Y, L, norm, 3); }
int bartlett_c(float *Y, const size_t L, const size_t norm) { return window_host_cplx<float>("bartlett_c", Y, L, norm, WIN_BARTLETT); }
int bartlett_z(double *Y, const size_t L, const size_t norm) { return window_host_cplx<double>("bartlett_z", Y, L, norm, WIN_BARTLETT); }
int triangular_c(float *Y, const size_t L, const size_t norm) { return window_host_cplx<float>("triangular_c", Y, L, norm, WIN_TRIANGULAR); }
int triangular_z(double *Y, const size_t L, const size_t norm) { return window_host_cplx<double>("triangular_z", Y, L, norm, WIN_TRIANGULAR); }
int rectangular_c(float *Y, const size_t L, const size_t norm) { return window_host_cplx<float>("rectangular_c", Y, L, norm, WIN_RECTANGULAR); }
int rectangular_z(double *Y, const size_t L, const size_t norm) { return window_host_cplx<double>("rectangular_z", Y, L, norm, WIN_RECTANGULAR); }
int flattop_c(float *Y, const size_t L, const size_t norm) { return window_host_cplx<float>("flattop_c", Y, L, norm, WIN_FLATTOP); }
int flattop_z(double *Y, const size_t L, const size_t norm) { return window_host_cplx<double>("flattop_z", Y, L, norm, WIN_FLATTOP); }
int blackman_c(float *Y, const size_t L, const int exact, const size_t norm) { return window_host_cplx<float>("blackman_c", Y, L, norm, WIN_BLACKMAN, 0.f, exact != 0); }
int blackman_z(double *Y, const size_t L, const int exact, const size_t norm) { return window_host_cplx<double>("blackman_z", Y, L, norm, WIN_BLACKMAN, 0.0, exact != 0); }
int gauss_c(float *Y, const size_t L, const float r, const size_t norm) { return window_host_cplx<float>("gauss_c", Y, L, norm, WIN_GAUSS, r); }
int gauss_z(double *Y, const size_t L, const double r, const size_t norm) { return window_host_cplx<double>("gauss_z", Y, L, norm, WIN_GAUSS, r); }
int tukey_c(float *Y, const size_t L, const float r, const size_t norm) { return window_host_cplx<float>("tukey_c", Y, L, norm, WIN_TUKEY, r); }
int tukey_z(double *Y, const size_t L, const double r, const size_t norm) { return window_host_cplx<double>("tukey_z", Y, L, norm, WIN_TUKEY, r); }
int planck_c(float *Y, const size_t L, const float epsilon, const size_t norm) { return window_host_cplx<float>("planck_c", Y, L, norm, WIN_PLANCK, epsilon); }
int planck_z(double *Y, const size_t L, const double epsilon, const size_t norm) { return window_host_cplx<double>("planck_z", Y, L, norm, WIN_PLANCK, epsilon); }
int dsp_window_dev_s(float *Y_dev, const int window, const size_t L, const size_t norm, void *stream) { return window_dev<float>("dsp_window_dev_s", Y_dev, window, L, norm, stream, window == WIN_GAUSS ? 2.5f : window == WIN_TUKEY ? 0.5f : window == WIN_PLANCK ? 0.1f : 0.f); }
int dsp_window_dev_d(double *Y_dev, const int window, const size_t L, const size_t norm, void *stream) { return window_dev<double>("dsp_window_dev_d", Y_dev, window, L, norm, stream, window == WIN_GAUSS ? 2.5 : window == WIN_TUKEY ? 0.5 : window == WIN_PLANCK ? 0.1 : 0.0); }
int dsp_window_par_dev_s(float *Y_dev, const int window, const size_t L, const float par, const int flag, const size_t norm, void *stream) { return window_dev<float>("dsp_window_par_dev_s", Y_dev, window, L, norm, stream, par, flag != 0); }
int dsp_window_par_dev_d(double *Y_dev, const int window, const size_t L, const double par, const int flag, const size_t norm, void *stream) { return window_dev<double>("dsp_window_par_dev_d", Y_dev, window, L, norm, stream, par, flag != 0); }
int convert_freqs_s(float *frqs, const size_t F, const char in_scale[], const char out_scale[]) { return convert_freqs_host<float>("convert_freqs_s", frqs, F, in_scale, out_scale); }
int convert_freqs_d(double *frqs, const size_t F, const char in_scale[], const char out_scale[]) { return convert_freqs_host<double>("convert_freqs_d", frqs, F, in_scale, out_scale); }
int pow_compress_s(float *Y, const float *X, const size_t N, const float p, const float preg) { return pow_compress_host<float>("pow_compress_s", Y, X, N, p, preg); }
int pow_compress_d(double *Y, const double *X, const size_t N, const double p, const double preg) { return pow_compress_host<double>("pow_compress_d", Y, X, N, p, preg); }
int pow_compress_inplace_s(float *X, const size_t N, const float p, const float preg) { return pow_compress_host<float>("pow_compress_inplace_s", X, X, N, p, preg); }
int pow_compress_inplace_d(double *X, const size_t N, const double p, const double preg) { return pow_compress_host<double>("pow_compress_inplace_d", X, X, N, p, preg); }
int get_stft_freqs_s(float *Y, const size_t nfft, const size_t F, const float fs) { return stft_freqs<float>("get_stft_freqs_s", Y, nfft, F, fs); }
int get_stft_freqs_d(double *Y, const size_t nfft, const size_t F, const double fs) { return stft_freqs<double>("get_stft_freqs_d", Y, nfft, F, fs); }
int dsp_pow_compress_dev_s(float *Y_dev, const float *X_dev, const size_t N, const float p, const float preg, void *stream)
{ return pow_compress_dev<float>("dsp_pow_compress_dev_s", Y_dev, X_dev, N, p, preg, stream); }
int dsp_pow_compress_dev_d(double *Y_dev, const double *X_dev, const size_t N, const double p, const double preg, void *stream)
{ return pow_compress_dev<double>("dsp_pow_compress_dev_d", Y_dev, X_dev, N, p, preg, stream); }
int dsp_stft_pc_batch_dev_s(float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const float p, const float preg, void *stream)
{ return stft_pc_dev<float>("dsp_stft_pc_batch_dev_s", Y_dev, X_dev, win_dev, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, p, preg, stream); }
int dsp_stft_pc_batch_dev_d(double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const double p, const double preg, void *stream)
{ return stft_pc_dev<double>("dsp_stft_pc_batch_dev_d", Y_dev, X_dev, win_dev, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, p, preg, stream); }
int dsp_stft_pc_batch_s(float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const float p, const float preg)
{ return stft_pc_host<float>("dsp_stft_pc_batch_s", Y, X, win, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, p, preg); }
int dsp_stft_pc_batch_d(double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const double p, const double preg)
{ return stft_pc_host<double>("dsp_stft_pc_batch_d", Y, X, win, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, p, preg); }

int fft_power_c(float *Y, const float *X, const size_t N) { return fft_power_host<float>("fft_power_c", Y, X, N); }
int fft_power_z(double *Y, const double *X, const size_t N) { return fft_power_host<double>("fft_power_z", Y, X, N); }

int frame_univar_s(float *Y, const float *X, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return frame_univar_host<float>("frame_univar_s", Y, X, N, L, stp, snip_edges, 1); }
int frame_univar_d(double *Y, const double *X, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return frame_univar_host<double>("frame_univar_d", Y, X, N, L, stp, snip_edges, 1); }
int frame_univar_c(float *Y, const float *X, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return frame_univar_host<float>("frame_univar_c", Y, X, N, L, stp, snip_edges, 2); }
int frame_univar_z(double *Y, const double *X, const size_t N, const size_t L, const size_t stp, const int snip_edges)
{ return frame_univar_host<double>("frame_univar_z", Y, X, N, L, stp, snip_edges, 2); }

int frame_univar_flt_s(float *Y, const float *X, const size_t N, const size_t L, const size_t W, const float c0, const float stp)
{ return frame_univar_flt_host<float>("frame_univar_flt_s", Y, X, N, L, W, c0, stp, 1); }
int frame_univar_flt_d(double *Y, const double *X, const size_t N, const size_t L, const size_t W, const double c0, const double stp)
{ return frame_univar_flt_host<double>("frame_univar_flt_d", Y, X, N, L, W, c0, stp, 1); }
int frame_univar_flt_c(float *Y, const float *X, const size_t N, const size_t L, const size_t W, const float c0, const float stp)
{ return frame_univar_flt_host<float>("frame_univar_flt_c", Y, X, N, L, W, c0, stp, 2); }
int frame_univar_flt_z(double *Y, const double *X, const size_t N, const size_t L, const size_t W, const double c0, const double stp)
{ return frame_univar_flt_host<double>("frame_univar_flt_z", Y, X, N, L, W, c0, stp, 2); }

int apply_win_s(float *Y, const float *X1, const float *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return apply_win_host<float>("apply_win_s", Y, X1, X2, R, C, S, H, L, dim, iscolmajor, 0); }
int apply_win_d(double *Y, const double *X1, const double *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return apply_win_host<double>("apply_win_d", Y, X1, X2, R, C, S, H, L, dim, iscolmajor, 0); }
int apply_win_c(float *Y, const float *X1, const float *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return apply_win_host<float>("apply_win_c", Y, X1, X2, R, C, S, H, L, dim, iscolmajor, 1); }
int apply_win_z(double *Y, const double *X1, const double *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return apply_win_host<double>("apply_win_z", Y, X1, X2, R, C, S, H, L, dim, iscolmajor, 1); }
int apply_win_inplace_s(float *X1, const float *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return apply_win_host<float>("apply_win_inplace_s", X1, X1, X2, R, C, S, H, L, dim, iscolmajor, 0); }
int apply_win_inplace_d(double *X1, const double *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return apply_win_host<double>("apply_win_inplace_d", X1, X1, X2, R, C, S, H, L, dim, iscolmajor, 0); }
int apply_win_inplace_c(float *X1, const float *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return apply_win_host<float>("apply_win_inplace_c", X1, X1, X2, R, C, S, H, L, dim, iscolmajor, 1); }
int apply_win_inplace_z(double *X1, const double *X2, const size_t R, const size_t C, const size_t S, const size_t H, const size_t L, const size_t dim, const int iscolmajor)
{ return apply_win_host<double>("apply_win_inplace_z", X1, X1, X2, R, C, S, H, L, dim, iscolmajor, 1); }

int fft_fftw_r2hc_s(float *Y, const float *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc)
{ return r2hc_host<float>("fft_fftw_r2hc_s", Y, X, R, C, S, H, iscolmajor, dim, nfft, sc); }
int fft_fftw_r2hc_d(double *Y, const double *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc)
{ return r2hc_host<double>("fft_fftw_r2hc_d", Y, X, R, C, S, H, iscolmajor, dim, nfft, sc); }

int fft_fftw_s(float *Y, const float *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc)
{ return r2hc_host<float>("fft_fftw_s", Y, X, R, C, S, H, iscolmajor, dim, nfft, sc); }
int fft_fftw_d(double *Y, const double *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc)
{ return r2hc_host<double>("fft_fftw_d", Y, X, R, C, S, H, iscolmajor, dim, nfft, sc); }
int fft_fftw_c(float *Y, const float *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc)
{ return c2c_host<float>("fft_fftw_c", Y, X, R, C, S, H, iscolmajor, dim, nfft, sc); }
int fft_fftw_z(double *Y, const double *X, const size_t R, const size_t C, const size_t S, const size_t H, const int iscolmajor, const size_t dim, const size_t nfft, const int sc)
{ return c2c_host<double>("fft_fftw_z", Y, X, R, C, S, H, iscolmajor, dim, nfft, sc); }

// ---------------------------------------- additions ----------------------------------------------
size_t dsp_stft_num_frames(const size_t N, const size_t L, const size_t stp, const int snip_edges)
{
    if (L < 1u || stp < 1u || (snip_edges && L > N)) { return 0u; }
    return snip_edges ? 1u + (N - L) / stp : (N + stp / 2u) / stp;
}

int dsp_set_device(const int device)
{
    int n = 0;
    HIPCHK("dsp_set_device", hipGetDeviceCount(&n));
    if (device < 0 || device >= n || device >= MAX_DEV) { fprintf(stderr, "error in dsp_set_device: device %d not available\n", device); return 1; }
    tl_device = device;
    HIPCHK("dsp_set_device", hipSetDevice(device));
    return 0;
}

// Gives back what the host-pointer entry points keep between calls on the calling thread's device: the device
// workspace, the pinned stages and their device twins, and the cached stft_flt frame-start tables.  (The twiddle
// tables, a few KB per transform length, and the streams stay.)  Calls in progress on other threads finish first.
int dsp_stft_release_buffers(void)
{
    const char *fn = "dsp_stft_release_buffers";
    int dev;
    if (use_device(fn, dev)) { return 1; }
    DevState &st = g_dev[dev];
    std::list<std::pair<FltKey, FltFraming>> dropped;          // tables are freed once no call holds them
    {
        std::lock_guard<std::mutex> lk(st.mu);
        dropped.swap(st.flt);
    }
    dropped.clear();
    tl_flt_htbl.reset();
    std::lock_guard<std::mutex> lk(st.ws_mu);
    int rc = 0;
    for (int i = 0; i < 4; ++i) {
        if (st.ws[i] && hipFree(st.ws[i]) != hipSuccess) { rc = 1; }
        st.ws[i] = nullptr; st.cap[i] = 0;
    }
    for (int i = 0; i < DevState::SLOTS; ++i) {
        if (st.pin_in[i] && hipHostFree(st.pin_in[i]) != hipSuccess) { rc = 1; }
        if (st.pin_out[i] && hipHostFree(st.pin_out[i]) != hipSuccess) { rc = 1; }
        if (st.dev_in[i] && hipFree(st.dev_in[i]) != hipSuccess) { rc = 1; }
        if (st.dev_out[i] && hipFree(st.dev_out[i]) != hipSuccess) { rc = 1; }
        st.pin_in[i] = st.pin_out[i] = st.dev_in[i] = st.dev_out[i] = nullptr;
    }
    st.cap_in = st.cap_out = 0;
    if (rc) { fprintf(stderr, "error in %s: hip call failed (free)\n", fn); }
    return rc;
}

int dsp_stft_flt_set_mode(const int intended) { tl_flt_intended = intended != 0; return 0; }
int dsp_stft_force_generic(const int on) { tl_force_generic = on != 0; return 0; }

int dsp_stft_kernel_family(const size_t nfft, const size_t elem_size)
{
    if (elem_size == 4) { return family_of<float>(nfft); }
    if (elem_size == 8) { return family_of<double>(nfft); }
    return 0;
}

int dsp_stft_batch_dev_s(float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream)
{ return stft_dev<float>("dsp_stft_batch_dev_s", Y_dev, X_dev, win_dev, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, stream); }
int dsp_stft_batch_dev_d(double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream)
{ return stft_dev<double>("dsp_stft_batch_dev_d", Y_dev, X_dev, win_dev, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, stream); }

int dsp_stft_batch_dev_noalloc_s(float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream)
{ return stft_dev_noalloc("dsp_stft_batch_dev_noalloc_s", Y_dev, X_dev, win_dev, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, stream); }

int dsp_stft_batch_dev_strided_s(float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream)
{
    const char *fn = "dsp_stft_batch_dev_strided_s";
    if (B == 0 || N == 0 || !X_dev) { return stft_dev<float>(fn, Y_dev, X_dev, win_dev, B, N, N, L, nfft, stp, snip_edges, mn0, amp, lg, stream); }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    if (const int rc = stft_dev_in_place<float>(fn, Y_dev, X_dev, win_dev, B, N, ldx, incx, L, nfft, stp, snip_edges, mn0, amp, lg, stream); rc >= 0) { return rc; }
    return with_dense_signals<float>(fn, X_dev, B, N, ldx, incx, stream, [&](const float *xd, size_t ld) { return stft_dev<float>(fn, Y_dev, xd, win_dev, B, N, ld, L, nfft, stp, snip_edges, mn0, amp, lg, stream); });
}
int dsp_stft_batch_dev_strided_d(double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, void *stream)
{
    const char *fn = "dsp_stft_batch_dev_strided_d";
    if (B == 0 || N == 0 || !X_dev) { return stft_dev<double>(fn, Y_dev, X_dev, win_dev, B, N, N, L, nfft, stp, snip_edges, mn0, amp, lg, stream); }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    return with_dense_signals<double>(fn, X_dev, B, N, ldx, incx, stream, [&](const double *xd, size_t ld) { return stft_dev<double>(fn, Y_dev, xd, win_dev, B, N, ld, L, nfft, stp, snip_edges, mn0, amp, lg, stream); });
}
int dsp_stft_flt_batch_dev_strided_s(float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg, void *stream)
{
    const char *fn = "dsp_stft_flt_batch_dev_strided_s";
    if (B == 0 || N == 0 || !X_dev) { return stft_flt_dev<float>(fn, Y_dev, X_dev, win_dev, B, N, N, L, W, nfft, c0, stp, mn0, amp, lg, stream); }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    return with_dense_signals<float>(fn, X_dev, B, N, ldx, incx, stream, [&](const float *xd, size_t ld) { return stft_flt_dev<float>(fn, Y_dev, xd, win_dev, B, N, ld, L, W, nfft, c0, stp, mn0, amp, lg, stream); });
}
int dsp_stft_flt_batch_dev_strided_d(double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg, void *stream)
{
    const char *fn = "dsp_stft_flt_batch_dev_strided_d";
    if (B == 0 || N == 0 || !X_dev) { return stft_flt_dev<double>(fn, Y_dev, X_dev, win_dev, B, N, N, L, W, nfft, c0, stp, mn0, amp, lg, stream); }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    return with_dense_signals<double>(fn, X_dev, B, N, ldx, incx, stream, [&](const double *xd, size_t ld) { return stft_flt_dev<double>(fn, Y_dev, xd, win_dev, B, N, ld, L, W, nfft, c0, stp, mn0, amp, lg, stream); });
}

int dsp_stft_flt_batch_dev_s(float *Y_dev, const float *X_dev, const float *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg, void *stream)
{ return stft_flt_dev<float>("dsp_stft_flt_batch_dev_s", Y_dev, X_dev, win_dev, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, stream); }
int dsp_stft_flt_batch_dev_d(double *Y_dev, const double *X_dev, const double *win_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg, void *stream)
{ return stft_flt_dev<double>("dsp_stft_flt_batch_dev_d", Y_dev, X_dev, win_dev, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, stream); }

int dsp_stft_flt_batch_s(float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg)
{ return stft_flt_host<float>("dsp_stft_flt_batch_s", Y, X, win, N, L, W, nfft, c0, stp, mn0, amp, lg, B, ldx); }
int dsp_stft_flt_batch_d(double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg)
{ return stft_flt_host<double>("dsp_stft_flt_batch_d", Y, X, win, N, L, W, nfft, c0, stp, mn0, amp, lg, B, ldx); }

int dsp_stft_batch_s(float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg)
{
    if (B > 1 && ldx < N) { fprintf(stderr, "error in dsp_stft_batch_s: ldx must be >= N\n"); return 1; }
    return stft_host<float>("dsp_stft_batch_s", Y, X, win, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg);
}
int dsp_stft_batch_d(double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg)
{
    if (B > 1 && ldx < N) { fprintf(stderr, "error in dsp_stft_batch_d: ldx must be >= N\n"); return 1; }
    return stft_host<double>("dsp_stft_batch_d", Y, X, win, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg);
}

int dsp_stft_batch_strided_s(float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg)
{ return stft_strided_host<float>("dsp_stft_batch_strided_s", Y, X, win, B, N, ldx, incx, L, nfft, stp, snip_edges, mn0, amp, lg); }
int dsp_stft_batch_strided_d(double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg)
{ return stft_strided_host<double>("dsp_stft_batch_strided_d", Y, X, win, B, N, ldx, incx, L, nfft, stp, snip_edges, mn0, amp, lg); }
int dsp_stft_flt_batch_strided_s(float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg)
{ return stft_flt_strided_host<float>("dsp_stft_flt_batch_strided_s", Y, X, win, B, N, ldx, incx, L, W, nfft, c0, stp, mn0, amp, lg); }
int dsp_stft_flt_batch_strided_d(double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t incx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg)
{ return stft_flt_strided_host<double>("dsp_stft_flt_batch_strided_d", Y, X, win, B, N, ldx, incx, L, W, nfft, c0, stp, mn0, amp, lg); }

void dsp_shard_range(const size_t B, const size_t rank, const size_t world, size_t *lo, size_t *hi)
{
    size_t a, b;
    shard_range(B, rank, world, a, b);
    if (lo) { *lo = a; }
    if (hi) { *hi = b; }
}
int dsp_stft_multi_s(float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const int *devices, const int ngpu)
{ return stft_multi_host<float>("dsp_stft_multi_s", Y, X, win, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, devices, ngpu); }
int dsp_stft_multi_d(double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const int *devices, const int ngpu)
{ return stft_multi_host<double>("dsp_stft_multi_d", Y, X, win, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, devices, ngpu); }
int dsp_stft_flt_multi_s(float *Y, const float *X, const float *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const float c0, const float stp, const int mn0, const int amp, const int lg, const int *devices, const int ngpu)
{ return stft_flt_multi_host<float>("dsp_stft_flt_multi_s", Y, X, win, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, devices, ngpu); }
int dsp_stft_flt_multi_d(double *Y, const double *X, const double *win, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t W, const size_t nfft, const double c0, const double stp, const int mn0, const int amp, const int lg, const int *devices, const int ngpu)
{ return stft_flt_multi_host<double>("dsp_stft_flt_multi_d", Y, X, win, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, devices, ngpu); }
int dsp_stft_multi_dev_s(float *const *Y_dev, const float *const *X_dev, const float *const *win_dev, float *Y_all_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const int *devices, const int ngpu)
{ return stft_multi_dev<float>("dsp_stft_multi_dev_s", Y_dev, X_dev, win_dev, Y_all_dev, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, devices, ngpu); }
int dsp_stft_multi_dev_d(double *const *Y_dev, const double *const *X_dev, const double *const *win_dev, double *Y_all_dev, const size_t B, const size_t N, const size_t ldx, const size_t L, const size_t nfft, const size_t stp, const int snip_edges, const int mn0, const int amp, const int lg, const int *devices, const int ngpu)
{ return stft_multi_dev<double>("dsp_stft_multi_dev_d", Y_dev, X_dev, win_dev, Y_all_dev, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, devices, ngpu); }

int dsp_fft_power_dev_c(float *Y_dev, const float *X_dev, const size_t N, void *stream)
{
    int dev;
    if (use_device("dsp_fft_power_dev_c", dev)) { return 1; }
    HIPCHK("dsp_fft_power_dev_c", launch_fft_power<float>(Y_dev, X_dev, N, (hipStream_t)stream));
    return 0;
}
int dsp_fft_power_dev_z(double *Y_dev, const double *X_dev, const size_t N, void *stream)
{
    int dev;
    if (use_device("dsp_fft_power_dev_z", dev)) { return 1; }
    HIPCHK("dsp_fft_power_dev_z", launch_fft_power<double>(Y_dev, X_dev, N, (hipStream_t)stream));
    return 0;
}

const char *dsp_stft_version(void) { return "dsp_stft_hip 0.1 (gfx950)"; }

}  // extern "C"

