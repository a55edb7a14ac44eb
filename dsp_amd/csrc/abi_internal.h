// Declarations shared by the host-side translation units of the C ABI (abi_core.hip, abi_stream.hip, abi_multi.hip,
// abi_unfused.hip, abi_exports.hip).  Function templates are defined in one of them and instantiated there for float and double.
#pragma once
#include "stft_common.h"
#include "../../include/dsp_stft.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

namespace dsp {
namespace abi {

constexpr int MAX_DEV = 64;
constexpr size_t LDS_LIMIT = 128 * 1024;     // bytes of LDS one generic workgroup may stage

extern thread_local int tl_device;             // -1: whatever device is current
// per calling thread, like the device: two threads that want different framings do not disturb each other
extern thread_local int tl_flt_intended;
extern thread_local int tl_force_generic;
// the host copy of the frame-start table that the calling thread's last flt_desc() came to (empty: equally spaced frames)
extern thread_local std::shared_ptr<std::vector<int>> tl_flt_htbl;

#define HIPCHK(fn, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "error in %s: hip call failed (%s): %s\n", fn, #call, hipGetErrorString(e_)); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

// What stft_flt's framing comes to for one (type, N, L, W, c0, stp, mode): equally spaced frames, or a table of
// frame starts in device memory.  Cached per device, so that a repeated call neither walks the W centres on the
// host again nor allocates or copies; a table stays alive while a call that looked it up is still enqueueing
// its launch (the shared_ptr), and is freed with hipFree, which waits for the device, once it has been dropped
// from the cache and no call holds it.
struct FltKey {
    int esz, as_ref;
    size_t N, L, W;
    unsigned long long c0, stp;               // bit patterns
    bool operator==(const FltKey &o) const { return esz == o.esz && as_ref == o.as_ref && N == o.N && L == o.L && W == o.W && c0 == o.c0 && stp == o.stp; }
};
struct FltFraming {
    long long ss0 = 0, rd_shift = 0;
    int stp = 1, tbl_reach = -1;
    // two-launch form (FrameDesc::w_head ...): head frames by the table, the others as frames of the signal cut to n_tail
    int w_head = 0, tail_stp = 0, tail_reach = -1;
    long long n_tail = 0, tail_ss0 = 0;
    size_t tail_off = 0;                      // ints from the start of tbl to the tail's table (0: none)
    std::shared_ptr<int> tbl;                 // empty: equally spaced frames
    std::shared_ptr<std::vector<int>> htbl;   // the table on the host (the streamed host path cuts it into ranges of frames)
};
constexpr size_t FLT_CACHE = 32, FLT_TABLES = 8;

struct DevState {
    std::mutex mu;                            // guards everything below
    std::map<std::tuple<int, size_t, int>, void *> tw;   // (element size, nfft, full) -> table
    std::list<std::pair<FltKey, FltFraming>> flt;   // most recently used first
    // workspace of the host-pointer entry points (one call at a time per device)
    std::mutex ws_mu;
    void *ws[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    // streamed form of the batched host-pointer entry points: per slot a pinned input and output stage and
    // their device twins, three streams (copy in, compute, copy out), one event of each kind per slot
    static constexpr int SLOTS = 3;
    void *pin_in[SLOTS] = {}, *pin_out[SLOTS] = {}, *dev_in[SLOTS] = {}, *dev_out[SLOTS] = {};
    size_t cap_in = 0, cap_out = 0;
    hipStream_t st_in = nullptr, st_k = nullptr, st_out = nullptr;
    hipEvent_t ev_in[SLOTS] = {}, ev_k[SLOTS] = {}, ev_out[SLOTS] = {};
};
extern DevState g_dev[MAX_DEV];

template <typename T> T round_away(T x);
template <> inline float round_away<float>(float x) { return roundf(x); }
template <> inline double round_away<double>(double x) { return round(x); }
template <typename T> constexpr T type_eps();
template <> constexpr float type_eps<float>() { return FLT_EPSILON; }
template <> constexpr double type_eps<double>() { return DBL_EPSILON; }

inline bool is_pow2(size_t n) { return n >= 1 && !(n & (n - 1)); }

int use_device(const char *fn, int &dev);
template <typename T>
int get_twiddles(const char *fn, int dev, size_t nfft, int full, const T *&out);
int ws_reserve(const char *fn, DevState &st, int slot, size_t bytes, void *&out);
int stft_desc(const char *fn, FrameDesc &d, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
              int snip_edges, int mn0, int amp, int lg, size_t &W);
template <typename T>
int flt_desc(const char *fn, int dev, FrameDesc &d, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft, T c0, T stp,
             int mn0, int amp, int lg, int as_reference, bool check_nfft, std::shared_ptr<int> &hold, T stp_min = type_eps<T>());
template <typename T>
int family_of(size_t nfft);
template <typename T>
int run_fused(const char *fn, int dev, T *Yd, const T *Xd, const T *wd, const FrameDesc &d, hipStream_t s,
              const T *tw_fixed = nullptr, unsigned *take = nullptr);
template <typename T>
int stft_dev(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
             int snip_edges, int mn0, int amp, int lg, void *stream);
template <typename T>
int stft_flt_dev(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft,
                 T c0, T stp, int mn0, int amp, int lg, void *stream);
template <typename T>
int stft_dev_in_place(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp,
                      int snip_edges, int mn0, int amp, int lg, void *stream);
int na_setup(const char *fn, int dev, hipStream_t s, unsigned char *&ptr);
int stft_dev_noalloc(const char *fn, float *Yd, const float *Xd, const float *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft,
                     size_t stp, int snip_edges, int mn0, int amp, int lg, void *stream);
template <typename T>
int pc_epilogue(const char *fn, T p, T preg, Epilogue &e);
template <typename T>
int stft_pc_dev(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                int snip_edges, int mn0, int amp, int lg, T p, T preg, void *stream);
template <typename T>
int stft_pc_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                 int snip_edges, int mn0, int amp, int lg, T p, T preg);
template <typename T>
int stft_cplx_dev(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                  int snip_edges, int sc, void *stream);
template <typename T>
int run_fused_host_streamed(const char *fn, int dev, T *Y, const T *X, const T *win, const FrameDesc &d, const int *htbl);
template <typename T>
int run_fused_host(const char *fn, int dev, T *Y, const T *X, const T *win, FrameDesc &d, size_t x_elems, const int *htbl = nullptr);
template <typename T>
int stft_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
              int snip_edges, int mn0, int amp, int lg);
template <typename T>
int stft_flt_host(const char *fn, T *Y, const T *X, const T *win, size_t N, size_t L, size_t W, size_t nfft, T c0, T stp,
                  int mn0, int amp, int lg, size_t B = 1, size_t ldx = 0);
template <typename T>
int stft_strided_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp,
                      int snip_edges, int mn0, int amp, int lg);
template <typename T>
int stft_flt_strided_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t W, size_t nfft,
                          T c0, T stp, int mn0, int amp, int lg);
void shard_range(size_t B, size_t rank, size_t world, size_t &lo, size_t &hi);
template <typename T>
int stft_multi_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                    int snip_edges, int mn0, int amp, int lg, const int *devices, int ngpu);
template <typename T>
int stft_flt_multi_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft,
                        T c0, T stp, int mn0, int amp, int lg, const int *devices, int ngpu);
template <typename T>
int stft_multi_dev(const char *fn, T *const *Y_dev, const T *const *X_dev, const T *const *win_dev, T *Y_all_dev, size_t B, size_t N, size_t ldx,
                   size_t L, size_t nfft, size_t stp, int snip_edges, int mn0, int amp, int lg, const int *devices, int ngpu);
template <typename T>
int pow_compress_host(const char *fn, T *Y, const T *X, size_t N, T p, T preg);
template <typename T>
int pow_compress_dev(const char *fn, T *Yd, const T *Xd, size_t N, T p, T preg, void *stream);
template <typename T>
bool window_par_ok(const char *fn, int kind, T par);
template <typename T>
int window_host(const char *fn, T *Y, size_t L, size_t norm, int kind, T par = T(0), int flag = 0);
template <typename T>
int window_host_cplx(const char *fn, T *Y, size_t L, size_t norm, int kind, T par = T(0), int flag = 0);
template <typename T>
int window_dev(const char *fn, T *Yd, int kind, size_t L, size_t norm, void *stream, T par = T(0), int flag = 0);
template <typename T>
int stft_freqs(const char *fn, T *Y, size_t nfft, size_t F, T fs);
template <typename T>
int convert_freqs_host(const char *fn, T *frqs, size_t F, const char *in_scale, const char *out_scale);
template <typename T>
int fft_power_host(const char *fn, T *Y, const T *X, size_t N);
template <typename T>
int frames_host(const char *fn, T *Y, const T *X, const FrameDesc &d, int E, int dev, const T *win = nullptr);
template <typename T>
int frame_univar_host(const char *fn, T *Y, const T *X, size_t N, size_t L, size_t stp, int snip_edges, int E, const T *win = nullptr);
template <typename T>
int frame_univar_flt_host(const char *fn, T *Y, const T *X, size_t N, size_t L, size_t W, T c0, T stp, int E, const T *win = nullptr);
template <typename T>
int apply_win_host(const char *fn, T *Y, const T *X1, const T *X2, size_t R, size_t C, size_t S, size_t H, size_t L, size_t dim,
                   int iscolmajor, int cplx);
template <typename T>
int r2hc_host(const char *fn, T *Y, const T *X, size_t R, size_t C, size_t S, size_t H, int iscolmajor, size_t dim, size_t nfft, int sc);
template <typename T>
int c2c_host(const char *fn, T *Y, const T *X, size_t R, size_t C, size_t S, size_t H, int iscolmajor, size_t dim, size_t nfft, int sc);

// Device-resident forms for signals with a sample stride (multivariate tensors with interleaved channels): the
// signals are gathered into a dense [B][N] buffer allocated and freed in the stream's order, then transformed
// as a dense batch (same bits as the dense entry point on the gathered signals).
template <typename T, typename Run>
int with_dense_signals(const char *fn, const T *Xd, size_t B, size_t N, size_t ldx, size_t incx, void *stream, Run run)
{
    if (incx == 1) { return run(Xd, ldx); }
    if (incx == 0) { fprintf(stderr, "error in %s: incx must be positive\n", fn); return 1; }
    hipStream_t s = (hipStream_t)stream;
    void *dense = nullptr;
    HIPCHK(fn, hipMallocAsync(&dense, B * N * sizeof(T), s));
    int rc = 0;
    if (hipError_t e = launch_gather_signals<T>((T *)dense, Xd, N, B, ldx, incx, s); e != hipSuccess) {
        fprintf(stderr, "error in %s: hip call failed (gather of the signals): %s\n", fn, hipGetErrorString(e));
        rc = 1;
    }
    if (!rc) { rc = run((const T *)dense, N); }
    HIPCHK(fn, hipFreeAsync(dense, s));
    return rc;
}

// Host-pointer forms of the same (the tools' path for a multivariate file with interleaved channels): the whole
// span of the tensor that holds the signals goes to the device as it lies, the signals are gathered there, the
// spectrograms come back dense.  run(Yd, Xd, wd) enqueues the device-pointer form on the null stream.
template <typename T, typename RunDev>
int strided_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t y_elems, RunDev run)
{
    if (incx == 0) { fprintf(stderr, "error in %s: incx must be positive\n", fn); return 1; }
    if (B == 0 || N == 0 || y_elems == 0) { return 0; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    const size_t span = (B - 1) * ldx + (N - 1) * incx + 1;
    void *dx, *dy, *dw;
    if (ws_reserve(fn, st, 0, span * sizeof(T), dx)) { return 1; }
    if (ws_reserve(fn, st, 1, y_elems * sizeof(T), dy)) { return 1; }
    if (ws_reserve(fn, st, 2, L * sizeof(T), dw)) { return 1; }
    HIPCHK(fn, hipMemcpy(dx, X, span * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(fn, hipMemcpy(dw, win, L * sizeof(T), hipMemcpyHostToDevice));
    if (run((T *)dy, (const T *)dx, (const T *)dw)) { return 1; }
    HIPCHK(fn, hipMemcpy(Y, dy, y_elems * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace abi
}  // namespace dsp
