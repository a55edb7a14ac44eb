// C ABI of libdsp_stft_hip.so, host side: the host-pointer forms (pinned, chunked, overlapped copies) and the strided host forms.
#include "abi_internal.h"

namespace dsp {
namespace abi {

// ------------------------------------------------------------------------------------------------
// host-pointer forms
// ------------------------------------------------------------------------------------------------
// Copies between the caller's (pageable) arrays and pinned staging memory, split over a few threads: one
// thread moves ~10 GB/s, the PCIe link takes ~55 GB/s in each direction at once.
class CopyPool {
public:
    static CopyPool &get() { static CopyPool p; return p; }
    // dst[r][0..row) = src[r][0..row) for r < rows (row strides in bytes); returns when all of it is done
    void copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t row, size_t rows)
    {
        std::lock_guard<std::mutex> one_caller(call_mu_);      // (the threads of a multi-GPU call take turns)
        const size_t total = row * rows;
        if (total < (size_t)4 << 20 || workers_.empty()) { do_range(dst, dst_stride, src, src_stride, row, 0, total); return; }
        const size_t parts = workers_.size() + 1;
        const size_t per = ((total + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = [=](size_t i) { const size_t a = std::min(total, i * per), b = std::min(total, a + per); if (b > a) { do_range(dst, dst_stride, src, src_stride, row, a, b); } };
            pending_ = workers_.size();
            ++gen_;
        }
        cv_.notify_all();
        job_(workers_.size());                              // the caller takes the last part
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &t : workers_) { t.join(); }
    }
private:
    CopyPool()
    {
        unsigned n = std::thread::hardware_concurrency();
        if (const char *e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0 && (unsigned)v < n) { n = (unsigned)v; } }
        if (const char *e = getenv("DSP_COPY_THREADS")) { const int v = atoi(e); if (v > 0) { n = (unsigned)v; } }
        n = std::min(n, 12u);
        for (unsigned i = 0; i + 1 < n; ++i) { workers_.emplace_back([this, i] { loop(i); }); }
    }
    // bytes [a, b) of the dense sequence of rows
    static void do_range(void *dst, size_t ds, const void *src, size_t ss, size_t row, size_t a, size_t b)
    {
        if (ds == row && ss == row) { memcpy((char *)dst + a, (const char *)src + a, b - a); return; }
        while (a < b) {
            const size_t r = a / row, o = a - r * row, n = std::min(row - o, b - a);
            memcpy((char *)dst + r * ds + o, (const char *)src + r * ss + o, n);
            a += n;
        }
    }
    void loop(size_t i)
    {
        unsigned long long seen = 0;
        for (;;) {
            std::function<void(size_t)> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) { return; }
                job = job_;
            }
            job(i);
            { std::lock_guard<std::mutex> lk(mu_); if (--pending_ == 0) { done_.notify_all(); } }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_;
    std::function<void(size_t)> job_;
    size_t pending_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};


// Streamed form of the host-pointer entry points: the work goes through in chunks, chunk c being staged and copied in
// while chunk c - 1 is transformed and chunk c - 2 is copied out and unstaged (pinned stages, three streams).
// A chunk is a group of whole signals (batches of short signals; the kernel sees a dense batch of its own) or, when
// one signal alone is larger than a chunk (the reference's own calls on hour-long signals), a range of the frames
// of one signal together with the samples those frames read -- reflected samples at the ends of the signal
// included.  The kernel is then given the chunk as "the signal from sample lo on" (pointer moved back by lo, true
// length N, start of the first frame moved on by w0 hops), so every frame sees the samples, and takes the branches,
// it would in one launch over the whole signal.
constexpr size_t STREAM_MIN_BYTES = (size_t)64 << 20;      // below this the plain three copies are faster
constexpr size_t STREAM_CHUNK_BYTES = (size_t)48 << 20;    // input + output bytes of a chunk
struct StreamChunk { size_t b0, nb, w0, nw, lo, n_in; };   // signals [b0, b0 + nb), frames [w0, w0 + nw), samples [lo, lo + n_in) of each
template <typename T>
int run_fused_host_streamed(const char *fn, int dev, T *Y, const T *X, const T *win, const FrameDesc &d, const int *htbl)
{
    DevState &st = g_dev[dev];
    const size_t F = (size_t)d.nfft / 2 + 1, N = (size_t)d.N, W = (size_t)d.W, B = (size_t)d.B;
    const size_t in_sig = N * sizeof(T), out_sig = W * F * sizeof(T);
    std::vector<StreamChunk> chunks;
    size_t max_in = 0, max_out = 0;                        // elements of the largest chunk
    if (in_sig + out_sig <= STREAM_CHUNK_BYTES && B >= 2) {
        size_t cb = STREAM_CHUNK_BYTES / (in_sig + out_sig);
        cb = std::max<size_t>(1, std::min(cb, (B + 1) / 2));
        for (size_t b0 = 0; b0 < B; b0 += cb) { chunks.push_back({b0, std::min(cb, B - b0), 0, W, 0, N}); }
        max_in = cb * N; max_out = cb * W * F;
    } else {
        // by frames.  Beyond the last frame's window the kernels may read on to the end of a vector, a row or a
        // transform length: that much is staged as well.  htbl: the frames' (start, read start) pairs when they are
        // not equally spaced (stft_flt with a non-integral hop; zero edges).
        const long long L = d.L, shift = d.edge ? d.rd_shift : 0, slack = 2ll * d.nfft + 4096;       // (a step's span reaches up to 1536 samples past its first frame whatever its frames)
        const long long stp = htbl ? std::max(1ll, ((long long)htbl[2 * (W - 1)] - htbl[0]) / (long long)std::max<size_t>(1, W - 1)) : d.stp;
        size_t fw = STREAM_CHUNK_BYTES / (((size_t)stp + F) * sizeof(T));
        fw = std::max<size_t>(128, fw & ~(size_t)127);
        if (fw * 2 > W) { fw = std::max<size_t>(128, ((W + 1) / 2 + 127) & ~(size_t)127); }   // at least two chunks: something to overlap
        for (size_t b = 0; b < B; ++b) {
            for (size_t w0 = 0; w0 < W; w0 += fw) {
                const size_t nw = std::min(fw, W - w0);
                long long lo, hi;
                if (htbl) {
                    lo = (long long)N; hi = 0;
                    for (size_t w = w0; w < w0 + nw; ++w) {
                        // the frame reads X[n - (ss - rs)] for its n inside the signal
                        const long long ss = htbl[2 * w], off = ss - htbl[2 * w + 1];
                        const long long a = std::max(ss, 0ll), b = std::min(ss + L, (long long)N);
                        if (b > a) { lo = std::min(lo, a - off); hi = std::max(hi, b - off); }
                    }
                    if (hi <= lo) { lo = 0; hi = 0; }
                    hi += slack;
                }
                const long long first = d.ss0 + (long long)w0 * stp, last_end = d.ss0 + (long long)(w0 + nw - 1) * stp + L;
                if (!htbl) { lo = first - shift; hi = last_end - shift + slack; }
                if (!htbl && d.edge == 0) {
                    if (first < 0) { hi = std::max(hi, -first); }                                  // sample -n-1 for n < 0
                    if (last_end > (long long)N) { lo = std::min(lo, 2 * (long long)N - last_end); } // sample 2N-1-n for n >= N
                    // The span-staging kernels gather a whole step's span once it crosses the end of the signal, samples
                    // behind the last window included (unused, but read): they reflect down to 2N - (last_end + slack).
                    if (last_end + slack > (long long)N) { lo = std::min(lo, 2 * (long long)N - (last_end + slack)); }
                }
                lo = std::max(0ll, std::min(lo, (long long)N)) & ~3ll;     // (a multiple of 16 bytes: loads keep the alignment they have in one launch)
                hi = std::max(lo, std::min(hi, (long long)N));
                chunks.push_back({b, 1, w0, nw, (size_t)lo, (size_t)(hi - lo)});
                max_in = std::max(max_in, (size_t)(hi - lo)); max_out = std::max(max_out, nw * F);
            }
        }
    }
    max_in += 16;
    const size_t nchunks = chunks.size();
    constexpr int S = DevState::SLOTS;
    if (st.cap_in < max_in * sizeof(T) || st.cap_out < max_out * sizeof(T)) {
        HIPCHK(fn, hipDeviceSynchronize());
        for (int i = 0; i < S; ++i) {
            if (st.pin_in[i]) { (void)hipHostFree(st.pin_in[i]); (void)hipHostFree(st.pin_out[i]); (void)hipFree(st.dev_in[i]); (void)hipFree(st.dev_out[i]); }
            st.pin_in[i] = st.pin_out[i] = st.dev_in[i] = st.dev_out[i] = nullptr;
        }
        st.cap_in = st.cap_out = 0;
        const size_t ci = std::max(st.cap_in, max_in * sizeof(T)), co = std::max(st.cap_out, max_out * sizeof(T));
        for (int i = 0; i < S; ++i) {
            HIPCHK(fn, hipHostMalloc(&st.pin_in[i], ci, hipHostMallocDefault));
            HIPCHK(fn, hipHostMalloc(&st.pin_out[i], co, hipHostMallocDefault));
            HIPCHK(fn, hipMalloc(&st.dev_in[i], ci));
            HIPCHK(fn, hipMalloc(&st.dev_out[i], co));
        }
        st.cap_in = ci; st.cap_out = co;
    }
    if (!st.st_in) {
        HIPCHK(fn, hipStreamCreateWithFlags(&st.st_in, hipStreamNonBlocking));
        HIPCHK(fn, hipStreamCreateWithFlags(&st.st_k, hipStreamNonBlocking));
        HIPCHK(fn, hipStreamCreateWithFlags(&st.st_out, hipStreamNonBlocking));
        for (int i = 0; i < S; ++i) {
            HIPCHK(fn, hipEventCreateWithFlags(&st.ev_in[i], hipEventDisableTiming));
            HIPCHK(fn, hipEventCreateWithFlags(&st.ev_k[i], hipEventDisableTiming));
            HIPCHK(fn, hipEventCreateWithFlags(&st.ev_out[i], hipEventDisableTiming));
        }
    }
    void *dw;
    if (ws_reserve(fn, st, 2, (size_t)d.L * sizeof(T), dw)) { return 1; }
    HIPCHK(fn, hipMemcpy(dw, win, (size_t)d.L * sizeof(T), hipMemcpyHostToDevice));
    CopyPool &pool = CopyPool::get();
    constexpr size_t LAG = S - 1;
    int rc = 0;
    for (size_t c = 0; c < nchunks + LAG; ++c) {
        if (c < nchunks && !rc) {
            const int sl = (int)(c % S);
            const StreamChunk &k = chunks[c];
            // (the slot's stages and device buffers are free: chunk c - S was unstaged in the iteration before)
            const size_t row = k.n_in * sizeof(T);
            pool.copy_rows(st.pin_in[sl], row, X + k.b0 * (size_t)d.ldx + k.lo, (size_t)d.ldx * sizeof(T), row, k.nb);
            HIPCHK(fn, hipMemcpyAsync(st.dev_in[sl], st.pin_in[sl], k.nb * row, hipMemcpyHostToDevice, st.st_in));
            HIPCHK(fn, hipEventRecord(st.ev_in[sl], st.st_in));
            HIPCHK(fn, hipStreamWaitEvent(st.st_k, st.ev_in[sl], 0));
            FrameDesc dc = d;
            dc.B = (int)k.nb; dc.ldx = (long long)k.n_in; dc.W = (int)k.nw;
            dc.ss0 = d.ss0 + (long long)k.w0 * d.stp;
            if (d.tbl) { dc.tbl = d.tbl + 2 * k.w0; }
            if (d.n_tail > 0) {                        // a range of frames of a two-launch framing: what is left of its head, and its tail
                const long long skip = std::max(0ll, (long long)k.w0 - d.w_head);
                dc.w_head = (int)std::min<long long>(std::max(0ll, (long long)d.w_head - (long long)k.w0), (long long)k.nw);
                dc.tail_ss0 = d.tail_ss0 + skip * d.tail_stp;
                if (d.tail_tbl) { dc.tail_tbl = d.tail_tbl + 2 * skip; }
            }
            rc = run_fused<T>(fn, dev, (T *)st.dev_out[sl], (const T *)st.dev_in[sl] - k.lo, (const T *)dw, dc, st.st_k, nullptr, nullptr);
            HIPCHK(fn, hipEventRecord(st.ev_k[sl], st.st_k));
            HIPCHK(fn, hipStreamWaitEvent(st.st_out, st.ev_k[sl], 0));
            HIPCHK(fn, hipMemcpyAsync(st.pin_out[sl], st.dev_out[sl], k.nb * k.nw * F * sizeof(T), hipMemcpyDeviceToHost, st.st_out));
            HIPCHK(fn, hipEventRecord(st.ev_out[sl], st.st_out));
        }
        if (c >= LAG) {
            const StreamChunk &k = chunks[c - LAG];
            const int sl = (int)((c - LAG) % S);
            HIPCHK(fn, hipEventSynchronize(st.ev_out[sl]));
            const size_t row = k.nw * F * sizeof(T);
            if (!rc) { pool.copy_rows(Y + (k.b0 * W + k.w0) * F, out_sig, st.pin_out[sl], row, row, k.nb); }
        }
    }
    return rc;
}

// host-pointer form: copy in, compute on the null stream, copy out (blocking)
template <typename T>
int run_fused_host(const char *fn, int dev, T *Y, const T *X, const T *win, FrameDesc &d, size_t x_elems, const int *htbl)
{
    if (d.W == 0 || d.B == 0) { return 0; }
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.ws_mu);
    const size_t F = (size_t)d.nfft / 2 + 1;
    const size_t y_elems = (size_t)d.B * (size_t)d.W * F;
    if ((d.tbl == nullptr || htbl != nullptr) && (x_elems + y_elems) * sizeof(T) >= STREAM_MIN_BYTES &&
        (d.B >= 2 || d.W >= 256) && d.N >= d.L) {
        return run_fused_host_streamed<T>(fn, dev, Y, X, win, d, d.tbl ? htbl : nullptr);
    }
    void *dx, *dy, *dw;
    if (ws_reserve(fn, st, 0, x_elems * sizeof(T), dx)) { return 1; }
    if (ws_reserve(fn, st, 1, y_elems * sizeof(T), dy)) { return 1; }
    if (ws_reserve(fn, st, 2, (size_t)d.L * sizeof(T), dw)) { return 1; }
    HIPCHK(fn, hipMemcpy(dx, X, x_elems * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(fn, hipMemcpy(dw, win, (size_t)d.L * sizeof(T), hipMemcpyHostToDevice));
    if (run_fused<T>(fn, dev, (T *)dy, (const T *)dx, (const T *)dw, d, nullptr)) { return 1; }
    HIPCHK(fn, hipMemcpy(Y, dy, y_elems * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

template <typename T>
int stft_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
              int snip_edges, int mn0, int amp, int lg)
{
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }
    if (W == 0 || B == 0) { return 0; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    const size_t x_elems = (B - 1) * ldx + N;
    return run_fused_host<T>(fn, dev, Y, X, win, d, x_elems);
}

template <typename T>
int stft_flt_host(const char *fn, T *Y, const T *X, const T *win, size_t N, size_t L, size_t W, size_t nfft, T c0, T stp,
                  int mn0, int amp, int lg, size_t B, size_t ldx)
{
    if (B <= 1) { ldx = N; }
    // argument checks come before any device work, as in the reference
    FrameDesc d;
    int dev = 0;
    std::shared_ptr<int> hold;
    const T stp_min = (T)FLT_EPSILON;             // both precisions: reference c/stft_flt.c:40,139
    if (W > N || L > N || L < 1u || N < 1u || c0 > (T)(N - 1u) || stp < stp_min || nfft < L || nfft < 2u || W == 0) {
        return flt_desc<T>(fn, 0, d, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, !tl_flt_intended, true, hold, stp_min);
    }
    if (B == 0) { return 0; }
    if (ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
    if (use_device(fn, dev)) { return 1; }
    if (flt_desc<T>(fn, dev, d, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, !tl_flt_intended, true, hold, stp_min)) { return 1; }
    const std::shared_ptr<std::vector<int>> htbl = tl_flt_htbl;      // (held for the call: the cache may drop its entry meanwhile)
    return run_fused_host<T>(fn, dev, Y, X, win, d, (B - 1) * ldx + N, (d.tbl && htbl) ? htbl->data() : nullptr);
}

template <typename T>
int stft_strided_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp,
                      int snip_edges, int mn0, int amp, int lg)
{
    if (incx == 1) {
        if (B > 1 && ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
        return stft_host<T>(fn, Y, X, win, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg);
    }
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, N, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }     // the argument checks, before any device work
    return strided_host<T>(fn, Y, X, win, B, N, ldx, incx, L, B * W * (nfft / 2 + 1), [&](T *yd, const T *xd, const T *wd) {
        if (const int rc = stft_dev_in_place<T>(fn, yd, xd, wd, B, N, ldx, incx, L, nfft, stp, snip_edges, mn0, amp, lg, nullptr); rc >= 0) { return rc; }
        return with_dense_signals<T>(fn, xd, B, N, ldx, incx, nullptr, [&](const T *dense, size_t ld) {
            return stft_dev<T>(fn, yd, dense, wd, B, N, ld, L, nfft, stp, snip_edges, mn0, amp, lg, nullptr); });
    });
}

template <typename T>
int stft_flt_strided_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t W, size_t nfft,
                          T c0, T stp, int mn0, int amp, int lg)
{
    if (incx == 1) { return stft_flt_host<T>(fn, Y, X, win, N, L, W, nfft, c0, stp, mn0, amp, lg, B, ldx); }
    const T stp_min = (T)FLT_EPSILON;
    if (W > N || L > N || L < 1u || N < 1u || c0 > (T)(N - 1u) || stp < stp_min || nfft < L || nfft < 2u || W == 0) {
        FrameDesc d; std::shared_ptr<int> hold;                       // (the argument checks and their messages)
        return flt_desc<T>(fn, 0, d, B, N, N, L, W, nfft, c0, stp, mn0, amp, lg, !tl_flt_intended, true, hold, stp_min);
    }
    return strided_host<T>(fn, Y, X, win, B, N, ldx, incx, L, B * W * (nfft / 2 + 1), [&](T *yd, const T *xd, const T *wd) {
        return with_dense_signals<T>(fn, xd, B, N, ldx, incx, nullptr, [&](const T *dense, size_t ld) {
            return stft_flt_dev<T>(fn, yd, dense, wd, B, N, ld, L, W, nfft, c0, stp, mn0, amp, lg, nullptr); });
    });
}

// the instantiations the other translation units link against
template int run_fused_host_streamed<float>(const char *fn, int dev, float *Y, const float *X, const float *win, const FrameDesc &d, const int *htbl);
template int run_fused_host_streamed<double>(const char *fn, int dev, double *Y, const double *X, const double *win, const FrameDesc &d, const int *htbl);
template int run_fused_host<float>(const char *fn, int dev, float *Y, const float *X, const float *win, FrameDesc &d, size_t x_elems, const int *htbl);
template int run_fused_host<double>(const char *fn, int dev, double *Y, const double *X, const double *win, FrameDesc &d, size_t x_elems, const int *htbl);
template int stft_host<float>(const char *fn, float *Y, const float *X, const float *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
              int snip_edges, int mn0, int amp, int lg);
template int stft_host<double>(const char *fn, double *Y, const double *X, const double *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
              int snip_edges, int mn0, int amp, int lg);
template int stft_flt_host<float>(const char *fn, float *Y, const float *X, const float *win, size_t N, size_t L, size_t W, size_t nfft, float c0, float stp,
                  int mn0, int amp, int lg, size_t B, size_t ldx);
template int stft_flt_host<double>(const char *fn, double *Y, const double *X, const double *win, size_t N, size_t L, size_t W, size_t nfft, double c0, double stp,
                  int mn0, int amp, int lg, size_t B, size_t ldx);
template int stft_strided_host<float>(const char *fn, float *Y, const float *X, const float *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp,
                      int snip_edges, int mn0, int amp, int lg);
template int stft_strided_host<double>(const char *fn, double *Y, const double *X, const double *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp,
                      int snip_edges, int mn0, int amp, int lg);
template int stft_flt_strided_host<float>(const char *fn, float *Y, const float *X, const float *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t W, size_t nfft,
                          float c0, float stp, int mn0, int amp, int lg);
template int stft_flt_strided_host<double>(const char *fn, double *Y, const double *X, const double *win, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t W, size_t nfft,
                          double c0, double stp, int mn0, int amp, int lg);

}  // namespace abi
}  // namespace dsp
