// C ABI of libdsp_stft_hip.so, host side: several GPUs from one process.
#include "abi_internal.h"

namespace dsp {
namespace abi {

// ------------------------------------------------------------------------------------------------
// several GPUs from one process: signals split in contiguous blocks, no reduction
// ------------------------------------------------------------------------------------------------
void shard_range(size_t B, size_t rank, size_t world, size_t &lo, size_t &hi)
{
    const size_t per = world ? (B + world - 1) / world : B;
    lo = std::min(B, rank * per);
    hi = std::min(B, lo + per);
}

// host pointers: GPU g transforms signals [lo_g, hi_g) on a thread of its own (the streamed host path), each
// writing its block of Y: the gather is the copy out
template <typename T>
int stft_multi_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                    int snip_edges, int mn0, int amp, int lg, const int *devices, int ngpu)
{
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }
    if (ngpu < 1 || ngpu > MAX_DEV) { fprintf(stderr, "error in %s: ngpu must be in [1 %d]\n", fn, MAX_DEV); return 1; }
    if (B > 1 && ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
    if (W == 0 || B == 0) { return 0; }
    int ndev = 0;
    HIPCHK(fn, hipGetDeviceCount(&ndev));
    for (int g = 0; g < ngpu; ++g) {
        const int dv = devices ? devices[g] : g;
        if (dv < 0 || dv >= ndev) { fprintf(stderr, "error in %s: device %d not available\n", fn, dv); return 1; }
    }
    const size_t F = nfft / 2 + 1;
    const int intended = tl_flt_intended, generic = tl_force_generic;
    std::vector<int> rcs(ngpu, 0);
    std::vector<std::thread> th;
    for (int g = 0; g < ngpu; ++g) {
        th.emplace_back([&, g] {
            size_t lo, hi;
            shard_range(B, (size_t)g, (size_t)ngpu, lo, hi);
            if (hi == lo) { return; }
            tl_device = devices ? devices[g] : g;            // thread-local settings of this worker
            tl_flt_intended = intended; tl_force_generic = generic;
            rcs[g] = stft_host<T>(fn, Y + lo * W * F, X + lo * ldx, win, hi - lo, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg);
        });
    }
    for (auto &t : th) { t.join(); }
    for (int rc : rcs) { if (rc) { return 1; } }
    return 0;
}

// the same for the stft_flt framing (a multivariate recording over the GPUs of the node in one call)
template <typename T>
int stft_flt_multi_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft,
                        T c0, T stp, int mn0, int amp, int lg, const int *devices, int ngpu)
{
    const T stp_min = (T)FLT_EPSILON;
    if (W > N || L > N || L < 1u || N < 1u || c0 > (T)(N - 1u) || stp < stp_min || nfft < L || nfft < 2u || W == 0) {
        FrameDesc d; std::shared_ptr<int> hold;                       // (the argument checks and their messages)
        return flt_desc<T>(fn, 0, d, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, !tl_flt_intended, true, hold, stp_min);
    }
    if (ngpu < 1 || ngpu > MAX_DEV) { fprintf(stderr, "error in %s: ngpu must be in [1 %d]\n", fn, MAX_DEV); return 1; }
    if (B > 1 && ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
    if (B == 0) { return 0; }
    int ndev = 0;
    HIPCHK(fn, hipGetDeviceCount(&ndev));
    for (int g = 0; g < ngpu; ++g) {
        const int dv = devices ? devices[g] : g;
        if (dv < 0 || dv >= ndev) { fprintf(stderr, "error in %s: device %d not available\n", fn, dv); return 1; }
    }
    const size_t F = nfft / 2 + 1;
    const int intended = tl_flt_intended, generic = tl_force_generic;
    std::vector<int> rcs(ngpu, 0);
    std::vector<std::thread> th;
    for (int g = 0; g < ngpu; ++g) {
        th.emplace_back([&, g] {
            size_t lo, hi;
            shard_range(B, (size_t)g, (size_t)ngpu, lo, hi);
            if (hi == lo) { return; }
            tl_device = devices ? devices[g] : g;            // thread-local settings of this worker
            tl_flt_intended = intended; tl_force_generic = generic;
            rcs[g] = stft_flt_host<T>(fn, Y + lo * W * F, X + lo * ldx, win, N, L, W, nfft, c0, stp, mn0, amp, lg, hi - lo, ldx);
        });
    }
    for (auto &t : th) { t.join(); }
    for (int rc : rcs) { if (rc) { return 1; } }
    return 0;
}

// device pointers: GPU g holds its block X_dev[g] (signals [lo_g, hi_g) of the batch, dense or with ldx) and its
// output block Y_dev[g]; if Y_all_dev (memory of devices[0] for the whole [B][W][F]) is given, every block is
// copied there over the GPU-to-GPU links as soon as it is complete.  Returns when everything has finished.
template <typename T>
int stft_multi_dev(const char *fn, T *const *Y_dev, const T *const *X_dev, const T *const *win_dev, T *Y_all_dev, size_t B, size_t N, size_t ldx,
                   size_t L, size_t nfft, size_t stp, int snip_edges, int mn0, int amp, int lg, const int *devices, int ngpu)
{
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }
    if (ngpu < 1 || ngpu > MAX_DEV) { fprintf(stderr, "error in %s: ngpu must be in [1 %d]\n", fn, MAX_DEV); return 1; }
    if (!Y_dev || !X_dev || !win_dev) { fprintf(stderr, "error in %s: null pointer array\n", fn); return 1; }
    if (W == 0 || B == 0) { return 0; }
    const size_t F = nfft / 2 + 1;
    const int dev0 = devices ? devices[0] : 0;
    const int keep = tl_device;
    int prev = 0, ndev = 0;
    HIPCHK(fn, hipGetDevice(&prev));                     // the calling thread's device, current again on every way out
    HIPCHK(fn, hipGetDeviceCount(&ndev));
    for (int g = 0; g < ngpu; ++g) {
        const int dv = devices ? devices[g] : g;
        if (dv < 0 || dv >= ndev) { fprintf(stderr, "error in %s: device %d not available\n", fn, dv); return 1; }
    }
    if (Y_all_dev) {
        // the gather copies a block from its GPU straight into devices[0]'s memory: that needs peer access, asked
        // for here and reported when the machine does not grant it (no silent detour through the host)
        for (int g = 0; g < ngpu; ++g) {
            const int dv = devices ? devices[g] : g;
            if (dv == dev0) { continue; }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, dv, dev0) != hipSuccess || !can) {
                fprintf(stderr, "error in %s: device %d has no peer access to device %d (needed to gather the blocks)\n", fn, dv, dev0);
                (void)hipSetDevice(prev);
                return 1;
            }
            hipError_t e = hipSetDevice(dv);
            if (e == hipSuccess) { e = hipDeviceEnablePeerAccess(dev0, 0); }
            if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            if (e != hipSuccess) {
                fprintf(stderr, "error in %s: hip call failed (peer access of device %d to device %d): %s\n", fn, dv, dev0, hipGetErrorString(e));
                (void)hipSetDevice(prev);
                return 1;
            }
        }
    }
    std::vector<hipStream_t> streams(ngpu, nullptr);
    int rc = 0;
    for (int g = 0; g < ngpu && !rc; ++g) {
        size_t lo, hi;
        shard_range(B, (size_t)g, (size_t)ngpu, lo, hi);
        if (hi == lo) { continue; }
        const int dv = devices ? devices[g] : g;
        tl_device = dv;
        int cur;
        if (use_device(fn, cur)) { rc = 1; break; }
        if (hipStreamCreateWithFlags(&streams[g], hipStreamNonBlocking) != hipSuccess) { fprintf(stderr, "error in %s: hip call failed (stream)\n", fn); rc = 1; break; }
        rc = stft_dev<T>(fn, Y_dev[g], X_dev[g], win_dev[g], hi - lo, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, streams[g]);
        if (!rc && Y_all_dev && !(dv == dev0 && Y_dev[g] == Y_all_dev + lo * W * F)) {
            if (hipMemcpyPeerAsync(Y_all_dev + lo * W * F, dev0, Y_dev[g], dv, (hi - lo) * W * F * sizeof(T), streams[g]) != hipSuccess) {
                fprintf(stderr, "error in %s: hip call failed (copy of a block to device %d)\n", fn, dev0); rc = 1;
            }
        }
    }
    for (int g = 0; g < ngpu; ++g) {
        if (!streams[g]) { continue; }
        (void)hipSetDevice(devices ? devices[g] : g);
        if (hipStreamSynchronize(streams[g]) != hipSuccess) { fprintf(stderr, "error in %s: hip call failed (synchronize)\n", fn); rc = 1; }
        (void)hipStreamDestroy(streams[g]);
    }
    tl_device = keep;
    (void)hipSetDevice(prev);
    return rc;
}

// the instantiations the other translation units link against
template int stft_multi_host<float>(const char *fn, float *Y, const float *X, const float *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                    int snip_edges, int mn0, int amp, int lg, const int *devices, int ngpu);
template int stft_multi_host<double>(const char *fn, double *Y, const double *X, const double *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                    int snip_edges, int mn0, int amp, int lg, const int *devices, int ngpu);
template int stft_flt_multi_host<float>(const char *fn, float *Y, const float *X, const float *win, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft,
                        float c0, float stp, int mn0, int amp, int lg, const int *devices, int ngpu);
template int stft_flt_multi_host<double>(const char *fn, double *Y, const double *X, const double *win, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft,
                        double c0, double stp, int mn0, int amp, int lg, const int *devices, int ngpu);
template int stft_multi_dev<float>(const char *fn, float *const *Y_dev, const float *const *X_dev, const float *const *win_dev, float *Y_all_dev, size_t B, size_t N, size_t ldx,
                   size_t L, size_t nfft, size_t stp, int snip_edges, int mn0, int amp, int lg, const int *devices, int ngpu);
template int stft_multi_dev<double>(const char *fn, double *const *Y_dev, const double *const *X_dev, const double *const *win_dev, double *Y_all_dev, size_t B, size_t N, size_t ldx,
                   size_t L, size_t nfft, size_t stp, int snip_edges, int mn0, int amp, int lg, const int *devices, int ngpu);

}  // namespace abi
}  // namespace dsp
