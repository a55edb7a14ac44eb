// C ABI of libdsp_stft_hip.so (declarations and reference citations: include/dsp_stft.h).
// Host side only: argument checks with the reference's messages, frame descriptors, twiddle-table
// and workspace caches, host<->device copies for the host-pointer entry points, kernel dispatch.
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

namespace {
using namespace dsp;

constexpr int MAX_DEV = 64;
constexpr size_t LDS_LIMIT = 128 * 1024;     // bytes of LDS one generic workgroup may stage

thread_local int tl_device = -1;             // -1: whatever device is current
// per calling thread, like the device: two threads that want different framings do not disturb each other
thread_local int tl_flt_intended = 0;
thread_local int tl_force_generic = 0;
// the host copy of the frame-start table that the calling thread's last flt_desc() came to (empty: equally spaced frames)
thread_local std::shared_ptr<std::vector<int>> tl_flt_htbl;

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
DevState g_dev[MAX_DEV];

int use_device(const char *fn, int &dev)
{
    if (tl_device >= 0) { HIPCHK(fn, hipSetDevice(tl_device)); }
    HIPCHK(fn, hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) { fprintf(stderr, "error in %s: hip device index %d out of range\n", fn, dev); return 1; }
    return 0;
}

bool is_pow2(size_t n) { return n >= 1 && !(n & (n - 1)); }

// table of W_nfft^k = (cos, -sin)(2 pi k / nfft): k < nfft/2 (full == 0) or k < nfft (full == 1),
// evaluated in long double and rounded once
template <typename T>
int get_twiddles(const char *fn, int dev, size_t nfft, int full, const T *&out)
{
    DevState &st = g_dev[dev];
    std::lock_guard<std::mutex> lk(st.mu);
    const auto key = std::make_tuple((int)sizeof(T), nfft, full);
    auto it = st.tw.find(key);
    if (it != st.tw.end()) { out = static_cast<const T *>(it->second); return 0; }
    size_t cnt = full ? nfft : nfft / 2;
    if (cnt == 0) { cnt = 1; }
    std::vector<T> h(2 * cnt);
    const long double pi2 = 6.283185307179586476925286766559005768L;
    for (size_t k = 0; k < cnt; ++k) {
        // reduce the angle to the first octant so that symmetric entries are exactly symmetric
        const long double a = pi2 * (long double)k / (long double)nfft;
        h[2 * k] = (T)cosl(a);
        h[2 * k + 1] = (T)(-sinl(a));
    }
    void *p = nullptr;
    // (at least 1 KiB: the span-staging kernels issue placeholder loads of the table's first KiB)
    HIPCHK(fn, hipMalloc(&p, std::max(2 * cnt * sizeof(T), (size_t)1024)));
    HIPCHK(fn, hipMemcpy(p, h.data(), 2 * cnt * sizeof(T), hipMemcpyHostToDevice));
    st.tw[key] = p;
    out = static_cast<const T *>(p);
    return 0;
}

int ws_reserve(const char *fn, DevState &st, int slot, size_t bytes, void *&out)
{
    if (bytes == 0) { bytes = 16; }
    if (st.cap[slot] < bytes) {
        if (st.ws[slot]) { HIPCHK(fn, hipFree(st.ws[slot])); st.ws[slot] = nullptr; st.cap[slot] = 0; }
        HIPCHK(fn, hipMalloc(&st.ws[slot], bytes));
        st.cap[slot] = bytes;
    }
    out = st.ws[slot];
    return 0;
}

// ------------------------------------------------------------------------------------------------
// framing descriptors
// ------------------------------------------------------------------------------------------------
// stft framing; reference c/stft.c:40,63-64,100-124
int stft_desc(const char *fn, FrameDesc &d, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
              int snip_edges, int mn0, int amp, int lg, size_t &W)
{
    if (L < 1u) { fprintf(stderr, "error in %s: L must be positive\n", fn); return 1; }
    if (stp < 1u) { fprintf(stderr, "error in %s: stp must be positive\n", fn); return 1; }
    if (snip_edges && L > N) { fprintf(stderr, "error in %s: L must be < N if snip_edges\n", fn); return 1; }
    if (nfft < L) { fprintf(stderr, "error in %s: nfft must be >= L\n", fn); return 1; }
    if (nfft < 2u) { fprintf(stderr, "error in %s: nfft must be >= 2\n", fn); return 1; }
    W = snip_edges ? 1u + (N - L) / stp : (N + stp / 2u) / stp;
    if (W > (size_t)INT_MAX || nfft > (size_t)INT_MAX / 4 || B > (size_t)INT_MAX || N > (size_t)LLONG_MAX / 4) {
        fprintf(stderr, "error in %s: problem size exceeds the supported index range\n", fn); return 1;
    }
    d = FrameDesc{};
    d.N = (long long)N; d.ldx = (long long)ldx;
    d.ss0 = snip_edges ? 0 : (long long)(stp / 2u) - (long long)(L / 2u);
    d.rd_shift = 0; d.tbl = nullptr;
    d.L = (int)L; d.nfft = (int)nfft; d.W = (int)W;
    d.stp = (int)(stp > (size_t)INT_MAX ? (size_t)INT_MAX : stp);      // stp > INT_MAX implies W <= 1
    d.edge = 0; d.mn0 = mn0 != 0; d.epi = Epilogue{amp != 0, lg != 0, 0, 0.0, 0.0}; d.B = (int)B;
    return 0;
}

template <typename T> T round_away(T x);
template <> float round_away<float>(float x) { return roundf(x); }
template <> double round_away<double>(double x) { return round(x); }
template <typename T> constexpr T type_eps();
template <> constexpr float type_eps<float>() { return FLT_EPSILON; }
template <> constexpr double type_eps<double>() { return DBL_EPSILON; }

// stft_flt / frame_univar_flt framing; reference c/stft_flt.c:35-40,57-92, c/frame_univar_flt.c:41-87.
// The centre is accumulated serially in T, as the reference does, so non-dyadic hops round the same way.
// as_reference != 0 reproduces the X1 pointer walk of c/stft_flt.c (read start = start - shift once the
// start is non-negative); otherwise reads start at the frame's own start.
template <typename T>
int flt_desc(const char *fn, int dev, FrameDesc &d, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft, T c0, T stp,
             int mn0, int amp, int lg, int as_reference, bool check_nfft, std::shared_ptr<int> &hold, T stp_min = type_eps<T>())
{
    if (W > N) { fprintf(stderr, "error in %s: W must be <= N (length X)\n", fn); return 1; }
    if (L > N) { fprintf(stderr, "error in %s: L must be <= N (length X)\n", fn); return 1; }
    if (L < 1u) { fprintf(stderr, "error in %s: L must be positive\n", fn); return 1; }
    if (N < 1u) { fprintf(stderr, "error in %s: N (length X) must be positive\n", fn); return 1; }
    if (c0 > (T)(N - 1u)) { fprintf(stderr, "error in %s: c0 (center samp of 1st frame) must be < N (length X)\n", fn); return 1; }
    if (stp < stp_min) { fprintf(stderr, "error in %s: stp (step size) must be positive\n", fn); return 1; }
    if (check_nfft) {
        if (nfft < L) { fprintf(stderr, "error in %s: nfft must be >= L\n", fn); return 1; }
        if (nfft < 2u) { fprintf(stderr, "error in %s: nfft must be >= 2\n", fn); return 1; }
    }
    if (W > (size_t)INT_MAX / 2 || nfft > (size_t)INT_MAX / 4 || B > (size_t)INT_MAX || N > (size_t)INT_MAX) {
        fprintf(stderr, "error in %s: problem size exceeds the supported index range\n", fn); return 1;
    }
    d = FrameDesc{};
    d.N = (long long)N; d.ldx = (long long)ldx;
    d.L = (int)L; d.nfft = (int)nfft; d.W = (int)W; d.edge = 1;
    d.mn0 = mn0 != 0; d.epi = Epilogue{amp != 0, lg != 0, 0, 0.0, 0.0}; d.B = (int)B;
    d.tbl = nullptr; d.ss0 = 0; d.rd_shift = 0; d.stp = 1;
    if (W == 0) { return 0; }

    FltKey key{(int)sizeof(T), as_reference != 0, N, L, W, 0ull, 0ull};
    memcpy(&key.c0, &c0, sizeof(T));
    memcpy(&key.stp, &stp, sizeof(T));
    auto apply = [&](const FltFraming &f) {
        d.ss0 = f.ss0; d.stp = f.stp; d.rd_shift = f.rd_shift; d.tbl_reach = f.tbl_reach;
        hold = f.tbl; d.tbl = f.tbl.get();
        d.w_head = f.w_head; d.tail_stp = f.tail_stp; d.tail_reach = f.tail_reach; d.n_tail = f.n_tail; d.tail_ss0 = f.tail_ss0;
        d.tail_tbl = f.tail_off ? f.tbl.get() + f.tail_off : nullptr;
        tl_flt_htbl = f.htbl;
    };
    DevState &st = g_dev[dev];
    {
        std::lock_guard<std::mutex> lk(st.mu);
        for (auto it = st.flt.begin(); it != st.flt.end(); ++it) {
            if (it->first == key) { st.flt.splice(st.flt.begin(), st.flt, it); apply(st.flt.front().second); return 0; }
        }
    }

    std::vector<int> tbl(2 * W);
    const int Lpre = (int)(L / 2u), Lpost = (int)L - Lpre - 1;
    T cc = c0;
    long long shift = -1;                 // start of the first frame with 0 <= start <= N-1
    bool uniform = true, left_overlap = false;
    for (size_t w = 0; w < W; ++w, cc += stp) {
        const int cs = (int)round_away<T>(cc);
        const int ss = cs - Lpre, es = cs + Lpost;
        tbl[2 * w] = ss;
        if (ss >= 0 && ss <= (int)N - 1 && shift < 0) { shift = ss; }
        if (ss < 0 && es >= 0) { left_overlap = true; }
        if (w >= 1 && (long long)ss - tbl[0] != (long long)w * ((long long)tbl[2] - tbl[0])) { uniform = false; }
    }
    if (!as_reference || shift < 0) { shift = 0; }
    for (size_t w = 0; w < W; ++w) { tbl[2 * w + 1] = tbl[2 * w] - (tbl[2 * w] >= 0 ? (int)shift : 0); }
    const long long hop = (W >= 2) ? (long long)tbl[2] - tbl[0] : 1;
    FltFraming f;
    if (uniform && hop >= 1 && !(shift != 0 && left_overlap)) {
        f.ss0 = tbl[0]; f.stp = (int)hop; f.rd_shift = shift;
    } else {
        long long reach = (shift == 0) ? 0 : -1;
        for (size_t w = 0; reach >= 0 && w + 1 < W; ++w) {
            if (tbl[2 * (w + 1)] < tbl[2 * w]) { reach = -1; break; }
            const size_t w3 = (w + 3 < W) ? w + 3 : W - 1;
            reach = std::max(reach, (long long)tbl[2 * w3] - tbl[2 * w]);
        }
        f.tbl_reach = (reach >= 0 && reach <= (1 << 30)) ? (int)reach : -1;   // (each kernel family has its own bound on it)
        // The reference's pointer walk (c/stft_flt.c:66-88) reads the frames that start inside the signal `shift` samples
        // before their start, where shift is the start of the first of them.  With first frames that hang over the start
        // of the signal the read shift differs between frames, which only the general kernels take.  But frame w >= w1 (the
        // first that starts inside) is exactly the frame at start - shift of the signal cut to N - shift samples, read where
        // it starts: the fast kernels transform those as a batch of their own and the w1 head frames go by the table.
        std::vector<int> tail;
        if (shift != 0 && left_overlap && reach < 0) {
            size_t w1 = 0;
            while (w1 < W && tbl[2 * w1] < 0) { ++w1; }
            bool ordered = w1 < W;
            for (size_t w = w1; ordered && w + 1 < W; ++w) { ordered = tbl[2 * (w + 1)] >= tbl[2 * w]; }
            if (ordered) {
                const size_t Wt = W - w1;
                const long long t0 = (long long)tbl[2 * w1] - shift;
                const long long thop = (Wt >= 2) ? (long long)tbl[2 * (w1 + 1)] - tbl[2 * w1] : 1;
                bool tuni = thop >= 1;
                long long treach = 0;
                for (size_t w = w1; w < W; ++w) {
                    if ((long long)tbl[2 * w] - tbl[2 * w1] != (long long)(w - w1) * thop) { tuni = false; }
                    const size_t w3 = (w + 3 < W) ? w + 3 : W - 1;
                    treach = std::max(treach, (long long)tbl[2 * w3] - tbl[2 * w]);
                }
                f.w_head = (int)w1; f.n_tail = (long long)N - shift; f.tail_ss0 = t0;
                if (tuni) { f.tail_stp = (int)thop; }
                else {
                    f.tail_stp = 0; f.tail_reach = (treach <= (1 << 30)) ? (int)treach : -1;
                    tail.resize(2 * Wt);
                    for (size_t w = 0; w < Wt; ++w) { tail[2 * w] = tail[2 * w + 1] = (int)((long long)tbl[2 * (w1 + w)] - shift); }
                    f.tail_off = tbl.size();
                }
            }
        }
        // (first use of a framing: an allocation and a blocking copy; repeated calls find it in the cache)
        int *p = nullptr;
        HIPCHK(fn, hipMalloc((void **)&p, (tbl.size() + tail.size()) * sizeof(int)));
        f.tbl = std::shared_ptr<int>(p, [](int *q) { (void)hipFree(q); });
        HIPCHK(fn, hipMemcpy(p, tbl.data(), tbl.size() * sizeof(int), hipMemcpyHostToDevice));
        if (!tail.empty()) { HIPCHK(fn, hipMemcpy(p + tbl.size(), tail.data(), tail.size() * sizeof(int), hipMemcpyHostToDevice)); }
        f.htbl = std::make_shared<std::vector<int>>(std::move(tbl));
    }
    apply(f);
    std::list<std::pair<FltKey, FltFraming>> dropped;          // freed after the lock is released
    {
        std::lock_guard<std::mutex> lk(st.mu);
        st.flt.emplace_front(key, f);
        size_t tables = 0;
        for (auto it = st.flt.begin(); it != st.flt.end();) {
            auto cur = it++;
            const size_t pos = (size_t)std::distance(st.flt.begin(), cur);
            if (cur->second.tbl && ++tables > FLT_TABLES) { dropped.splice(dropped.end(), st.flt, cur); }
            else if (pos >= FLT_CACHE) { dropped.splice(dropped.end(), st.flt, cur); }
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fused dispatch
// ------------------------------------------------------------------------------------------------
template <typename T>
int family_of(size_t nfft)
{
    if (nfft < 2) { return 0; }
    if (!is_pow2(nfft)) { return 1; }
    if (nfft == 512 && !tl_force_generic) { return 3; }
    if (nfft == 4096 && !tl_force_generic) { return 4; }
    if (nfft == 8192 && sizeof(T) == 4 && !tl_force_generic) { return 8; }
    if ((nfft == 1024 || nfft == 2048) && !tl_force_generic) { return 6; }
    if ((nfft == 256 || nfft == 128 || nfft == 64) && !tl_force_generic) { return 7; }
    if (!tl_force_generic && nfft <= (1u << 20)) {
        FrameDesc probe{};
        probe.nfft = (int)nfft; probe.L = 1;
        if (wave_fft_supported<T>(probe)) { return 5; }
    }
    return 2;
}

// tw_fixed / take: a caller-held twiddle table of nfft (power of two) and run-form counters, in place of the
// library's cached table and counter sets (the allocation-free entry point)
template <typename T>
int run_fused(const char *fn, int dev, T *Yd, const T *Xd, const T *wd, const FrameDesc &d, hipStream_t s,
              const T *tw_fixed = nullptr, unsigned *take = nullptr)
{
    if (d.W == 0 || d.B == 0) { return 0; }
    if (d.n_tail > 0) {
        // two launches (FrameDesc::w_head): the head frames by the table, the others as frames of the cut signal
        const size_t F = (size_t)d.nfft / 2 + 1;
        FrameDesc h = d;
        h.n_tail = 0; h.w_head = 0; h.tail_tbl = nullptr; h.Wy = d.wy();
        FrameDesc t = h;
        h.W = d.w_head;
        t.W = d.W - d.w_head; t.N = d.n_tail; t.rd_shift = 0; t.ss0 = d.tail_ss0;
        t.tbl = d.tail_stp > 0 ? nullptr : d.tail_tbl; t.tbl_reach = d.tail_reach; t.stp = d.tail_stp > 0 ? d.tail_stp : 1;
        if (h.W > 0 && run_fused<T>(fn, dev, Yd, Xd, wd, h, s, tw_fixed, take)) { return 1; }
        if (t.W > 0) { return run_fused<T>(fn, dev, Yd + (size_t)d.w_head * F, Xd, wd, t, s, tw_fixed, take); }
        return 0;
    }
    const size_t nfft = (size_t)d.nfft;
    const int fam = family_of<T>(nfft);
    const T *tw = tw_fixed;
    if (fam == 1) {
        if ((size_t)d.L * sizeof(T) > LDS_LIMIT) { fprintf(stderr, "error in %s: L too large for a non-power-of-two nfft (limit %zu)\n", fn, LDS_LIMIT / sizeof(T)); return 1; }
        if (get_twiddles<T>(fn, dev, nfft, 1, tw)) { return 1; }
        HIPCHK(fn, launch_stft_direct<T>(Yd, Xd, wd, tw, d, s));
        return 0;
    }
    if (!tw && get_twiddles<T>(fn, dev, nfft, 0, tw)) { return 1; }
    if (fam == 3 && r512_supported<T>(d)) {
        HIPCHK(fn, launch_stft_r512<T>(Yd, Xd, wd, tw, d, s));
        return 0;
    }
    if (fam == 4 && r4096_supported(d)) {
        HIPCHK(fn, launch_stft_r4096(Yd, Xd, wd, tw, d, s, take));
        return 0;
    }
    if constexpr (sizeof(T) == 4) {
        if (fam == 8 && r8192_supported(d)) {
            HIPCHK(fn, launch_stft_r8192(Yd, Xd, wd, tw, d, s));
            return 0;
        }
    }
    if (fam == 6 && r512x_supported(d)) {
        HIPCHK(fn, launch_stft_r512x(Yd, Xd, wd, tw, d, s));
        return 0;
    }
    if (fam == 7 && r256_supported(d)) {
        HIPCHK(fn, launch_stft_r256(Yd, Xd, wd, tw, d, s));
        return 0;
    }
    if (fam >= 5 && wave_fft_supported<T>(d)) {
        HIPCHK(fn, launch_stft_wave<T>(Yd, Xd, wd, tw, d, s));
        return 0;
    }
    if (nfft * sizeof(T) > LDS_LIMIT) { fprintf(stderr, "error in %s: nfft too large (limit %zu)\n", fn, LDS_LIMIT / sizeof(T)); return 1; }
    HIPCHK(fn, launch_stft_pow2<T>(Yd, Xd, wd, tw, d, s));
    return 0;
}

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
int run_fused_host(const char *fn, int dev, T *Y, const T *X, const T *win, FrameDesc &d, size_t x_elems, const int *htbl = nullptr)
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
                  int mn0, int amp, int lg, size_t B = 1, size_t ldx = 0)
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
int stft_dev(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
             int snip_edges, int mn0, int amp, int lg, void *stream)
{
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }
    if (W == 0 || B == 0) { return 0; }
    if (B > 1 && ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
    if (!Yd || !Xd || !wd) { fprintf(stderr, "error in %s: null device pointer\n", fn); return 1; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    return run_fused<T>(fn, dev, Yd, Xd, wd, d, (hipStream_t)stream);
}

template <typename T>
int stft_flt_dev(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft,
                 T c0, T stp, int mn0, int amp, int lg, void *stream)
{
    FrameDesc d;
    int dev = 0;
    std::shared_ptr<int> hold;
    const T stp_min = (T)FLT_EPSILON;             // both precisions: reference c/stft_flt.c:40,139
    if (W > N || L > N || L < 1u || N < 1u || c0 > (T)(N - 1u) || stp < stp_min || nfft < L || nfft < 2u || W == 0 || B == 0) {
        return flt_desc<T>(fn, 0, d, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, !tl_flt_intended, true, hold, stp_min);
    }
    if (B > 1 && ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
    if (!Yd || !Xd || !wd) { fprintf(stderr, "error in %s: null device pointer\n", fn); return 1; }
    if (use_device(fn, dev)) { return 1; }
    if (flt_desc<T>(fn, dev, d, B, N, ldx, L, W, nfft, c0, stp, mn0, amp, lg, !tl_flt_intended, true, hold, stp_min)) { return 1; }
    return run_fused<T>(fn, dev, Yd, Xd, wd, d, (hipStream_t)stream);
}

// Adjacent channels of a [N][pitch] array (ldx == 1, incx = pitch), float32, nfft 512: read in place by the 512-point
// kernel's interleaved form where that is as fast as gathering them into a dense batch first: up to 32 channels (2 to
// 3.5 times faster up to 16, the same at 32, 7 % slower at 64; DSP_ILV_MAX_CHANNELS overrides the limit, 0 = never) --
// and without the temporary.  Returns -1 when the call is not of that kind (the caller then gathers), else the status
// of the call.
template <typename T>
int stft_dev_in_place(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp,
                      int snip_edges, int mn0, int amp, int lg, void *stream)
{
    if constexpr (sizeof(T) == 4) {
        size_t max_c = 32;
        if (const char *e = getenv("DSP_ILV_MAX_CHANNELS")) { max_c = (size_t)atoll(e); }
        if (nfft != 512 || ldx != 1 || incx < B || B > max_c || tl_force_generic || !Yd || !Xd || !wd) { return -1; }
        FrameDesc d; size_t W;
        if (stft_desc(fn, d, B, N, N, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }
        if (W == 0 || B == 0) { return 0; }
        if (!r512_fast_ilv_supported(d, Xd, incx)) { return -1; }
        int dev;
        if (use_device(fn, dev)) { return 1; }
        const T *tw = nullptr;
        if (get_twiddles<T>(fn, dev, nfft, 0, tw)) { return 1; }
        HIPCHK(fn, launch_stft_r512_fast_ilv(Yd, Xd, wd, tw, d, incx, (hipStream_t)stream));
        return 0;
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------
// allocation-free form of the batched device-resident STFT (float32, nfft a power of two <= 4096)
// ------------------------------------------------------------------------------------------------
// What the kernels need beyond the caller's input and output lives in one array of the code object
// instead of hipMalloc'ed memory: the twiddle tables of nfft = 2, 4, ..., 4096 (nfft floats each, the table
// of nfft at float offset nfft - 2) and, behind them, the 8 work counters of the 4096-point run form.
// Every device holds its own copy at its own address, found once per device with hipGetSymbolAddress
// while that device is current; host code passes that address, never the array's name.
constexpr size_t NA_MAX_NFFT = 4096;
constexpr size_t NA_TW_BYTES = 2 * NA_MAX_NFFT * sizeof(float);    // 8190 floats used
constexpr size_t NA_SCRATCH_BYTES = NA_TW_BYTES + 256;             // + the counters (32 bytes used)
static __device__ __align__(256) unsigned char scratch[NA_SCRATCH_BYTES];

// complex entry c of the tables, c < 4095: table nfft = 2 h with h the largest power of two <= c + 1,
// k = c + 1 - h; value (cos, -sin)(2 pi k / nfft) evaluated in double and rounded once
__global__ void na_fill_twiddles(float *tw)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int)NA_MAX_NFFT - 1) { return; }
    const int m = c + 1, h = 1 << (31 - __clz(m)), k = m - h;
    double sn, cs;
    sincospi((double)k / (double)h, &sn, &cs);                    // 2 pi k / (2 h) = pi k / h
    tw[2 * c] = (float)cs;
    tw[2 * c + 1] = (float)(-sn);
}

struct NaDev { std::mutex mu; unsigned char *ptr = nullptr; };
NaDev g_na[MAX_DEV];

// address of the device's copy of the array; on the first call for a device the tables are computed by a
// kernel in the caller's stream, which is waited for once so that calls on other streams find them complete
int na_setup(const char *fn, int dev, hipStream_t s, unsigned char *&ptr)
{
    NaDev &st = g_na[dev];
    std::lock_guard<std::mutex> lk(st.mu);
    if (!st.ptr) {
        void *p = nullptr;
        HIPCHK(fn, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        hipLaunchKernelGGL(na_fill_twiddles, dim3((NA_MAX_NFFT + 255) / 256), dim3(256), 0, s, static_cast<float *>(p));
        HIPCHK(fn, hipGetLastError());
        HIPCHK(fn, hipStreamSynchronize(s));
        st.ptr = static_cast<unsigned char *>(p);
    }
    ptr = st.ptr;
    return 0;
}

int stft_dev_noalloc(const char *fn, float *Yd, const float *Xd, const float *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft,
                     size_t stp, int snip_edges, int mn0, int amp, int lg, void *stream)
{
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }
    if (!is_pow2(nfft) || nfft > NA_MAX_NFFT) {
        fprintf(stderr, "error in %s: nfft must be a power of two <= %zu (the limit of the allocation-free form)\n", fn, NA_MAX_NFFT);
        return 1;
    }
    if (W == 0 || B == 0) { return 0; }
    if (B > 1 && ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
    if (!Yd || !Xd || !wd) { fprintf(stderr, "error in %s: null device pointer\n", fn); return 1; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    unsigned char *base = nullptr;
    if (na_setup(fn, dev, (hipStream_t)stream, base)) { return 1; }
    const float *tw = reinterpret_cast<const float *>(base) + (nfft - 2);
    unsigned *take = reinterpret_cast<unsigned *>(base + NA_TW_BYTES);
    return run_fused<float>(fn, dev, Yd, Xd, wd, d, (hipStream_t)stream, tw, take);
}

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

// power compression as an epilogue; checks and messages of reference c/pow_compress.c:25-26,30-46
template <typename T>
int pc_epilogue(const char *fn, T p, T preg, Epilogue &e)
{
    if (p < T(0) || p > T(1)) { fprintf(stderr, "error in %s: pow must be in [0.0 1.0]\n", fn); return 1; }
    if (preg < T(0)) { fprintf(stderr, "error in %s: preg must be nonnegative\n", fn); return 1; }
    e.p = (double)p; e.preg = (double)preg;
    if (p == T(1)) { e.pc = 1; }
    else if (p == T(0)) { e.pc = 2; }
    else if (p == T(0.5)) { e.pc = 3; }
    else if (std::fabs(p - T(1) / T(3)) <= (T)FLT_EPSILON) { e.pc = 4; }
    else { e.pc = 5; }
    return 0;
}

template <typename T>
int stft_pc_dev(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                int snip_edges, int mn0, int amp, int lg, T p, T preg, void *stream)
{
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }
    if (pc_epilogue<T>(fn, p, preg, d.epi)) { return 1; }
    if (W == 0 || B == 0) { return 0; }
    if (B > 1 && ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
    if (!Yd || !Xd || !wd) { fprintf(stderr, "error in %s: null device pointer\n", fn); return 1; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    return run_fused<T>(fn, dev, Yd, Xd, wd, d, (hipStream_t)stream);
}

template <typename T>
int stft_pc_host(const char *fn, T *Y, const T *X, const T *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                 int snip_edges, int mn0, int amp, int lg, T p, T preg)
{
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, ldx, L, nfft, stp, snip_edges, mn0, amp, lg, W)) { return 1; }
    if (pc_epilogue<T>(fn, p, preg, d.epi)) { return 1; }
    if (W == 0 || B == 0) { return 0; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    return run_fused_host<T>(fn, dev, Y, X, win, d, (B - 1) * ldx + N);
}

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
int window_host(const char *fn, T *Y, size_t L, size_t norm, int kind, T par = T(0), int flag = 0)
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
int window_host_cplx(const char *fn, T *Y, size_t L, size_t norm, int kind, T par = T(0), int flag = 0)
{
    if (L > (size_t)INT_MAX / 4) { fprintf(stderr, "error in %s: problem size exceeds the supported index range\n", fn); return 1; }
    std::vector<T> w;
    try { w.resize(L); } catch (const std::exception &) { fprintf(stderr, "error in %s: problem with malloc. \n", fn); return 1; }
    if (int rc = window_host<T>(fn, w.data(), L, norm, kind, par, flag)) { return rc; }
    for (size_t l = 0; l < L; ++l) { Y[2 * l] = w[l]; Y[2 * l + 1] = T(0); }
    return 0;
}

template <typename T>
int window_dev(const char *fn, T *Yd, int kind, size_t L, size_t norm, void *stream, T par = T(0), int flag = 0)
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
int frames_host(const char *fn, T *Y, const T *X, const FrameDesc &d, int E, int dev, const T *win = nullptr)
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
int frame_univar_host(const char *fn, T *Y, const T *X, size_t N, size_t L, size_t stp, int snip_edges, int E, const T *win = nullptr)
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
int frame_univar_flt_host(const char *fn, T *Y, const T *X, size_t N, size_t L, size_t W, T c0, T stp, int E, const T *win = nullptr)
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

// The complex spectrogram: framing and windowing as stft (reference c/window_univar.c:41-89), then the real FFT of every
// frame as fft_fftw_r2hc gives it (c/fft.fftw.r2hc.c:69-78: nfft/2+1 interleaved complex values, imaginary parts of the
// first and last zero; sc: every value times 1/sqrt(2 nfft)).  Device pointers, enqueued on `stream`.  nfft 512 with a hop
// of at most 512 runs on the fused 512-point kernel; everything else through the unfused kernels with a temporary for the
// frames that is allocated and freed in the stream's order.
template <typename T>
int stft_cplx_dev(const char *fn, T *Yd, const T *Xd, const T *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                  int snip_edges, int sc, void *stream)
{
    FrameDesc d; size_t W;
    if (stft_desc(fn, d, B, N, ldx, L, nfft, stp, snip_edges, 0, 0, 0, W)) { return 1; }
    if (W == 0 || B == 0) { return 0; }
    if (B > 1 && ldx < N) { fprintf(stderr, "error in %s: ldx must be >= N\n", fn); return 1; }
    if (!Yd || !Xd || !wd) { fprintf(stderr, "error in %s: null device pointer\n", fn); return 1; }
    int dev;
    if (use_device(fn, dev)) { return 1; }
    hipStream_t s = (hipStream_t)stream;
    const double scale = sc ? 1.0 / sqrt((double)(2u * nfft)) : 1.0;
    const int pow2 = is_pow2(nfft) ? 1 : 0;
    const T *tw = nullptr;
    if (nfft == 512 && !tl_force_generic && r512_fast_cplx_supported(d)) {
        if (get_twiddles<T>(fn, dev, nfft, 0, tw)) { return 1; }
        HIPCHK(fn, launch_stft_r512_fast_cplx<T>(Yd, Xd, wd, tw, d, scale, s));
        return 0;
    }
    if ((nfft == 1024 || nfft == 2048) && !tl_force_generic && r512x_cplx_supported(d)) {
        if (get_twiddles<T>(fn, dev, nfft, 0, tw)) { return 1; }
        HIPCHK(fn, launch_stft_r512x_cplx(Yd, Xd, wd, tw, d, scale, s));
        return 0;
    }
    if ((nfft == 256 || nfft == 128 || nfft == 64) && !tl_force_generic && r256_cplx_supported(d)) {
        if (get_twiddles<T>(fn, dev, nfft, 0, tw)) { return 1; }
        HIPCHK(fn, launch_stft_r256_cplx(Yd, Xd, wd, tw, d, scale, s));
        return 0;
    }
    if (nfft == 4096 && !tl_force_generic && r4096_cplx_supported(d)) {
        if (get_twiddles<T>(fn, dev, nfft, 0, tw)) { return 1; }
        HIPCHK(fn, launch_stft_r4096_cplx(Yd, Xd, wd, tw, d, scale, s));
        return 0;
    }
    if ((pow2 ? nfft : L) * sizeof(T) > LDS_LIMIT) { fprintf(stderr, "error in %s: nfft too large (limit %zu)\n", fn, LDS_LIMIT / sizeof(T)); return 1; }
    if (get_twiddles<T>(fn, dev, nfft, pow2 ? 0 : 1, tw)) { return 1; }
    void *frames = nullptr;
    HIPCHK(fn, hipMallocAsync(&frames, B * W * L * sizeof(T), s));
    int rc = 0;
    for (size_t b = 0; b < B && !rc; ++b) {
        if (hipError_t e = launch_frames<T>((T *)frames + b * W * L, Xd + b * ldx, wd, d, 1, s); e != hipSuccess) {
            fprintf(stderr, "error in %s: hip call failed (framing): %s\n", fn, hipGetErrorString(e)); rc = 1;
        }
    }
    if (!rc) {
        if (hipError_t e = launch_r2hc<T>(Yd, (const T *)frames, tw, B * W, 1, L, nfft, pow2, (T)scale, s); e != hipSuccess) {
            fprintf(stderr, "error in %s: hip call failed (transform): %s\n", fn, hipGetErrorString(e)); rc = 1;
        }
    }
    HIPCHK(fn, hipFreeAsync(frames, s));
    return rc;
}

}  // namespace

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
