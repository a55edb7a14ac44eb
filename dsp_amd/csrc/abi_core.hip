// C ABI of libdsp_stft_hip.so, host side (declarations shared by the abi_*.hip files: abi_internal.h): the state per device and
// per calling thread, twiddle and framing caches, frame descriptors, the dispatch of the fused kernels, the device-pointer
// forms, the allocation-free form, the fused power compression and the complex spectrogram.
#include "abi_internal.h"

namespace dsp {
namespace abi {

thread_local int tl_device = -1;
thread_local int tl_flt_intended = 0;
thread_local int tl_force_generic = 0;
thread_local std::shared_ptr<std::vector<int>> tl_flt_htbl;
DevState g_dev[MAX_DEV];

int use_device(const char *fn, int &dev)
{
    if (tl_device >= 0) { HIPCHK(fn, hipSetDevice(tl_device)); }
    HIPCHK(fn, hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) { fprintf(stderr, "error in %s: hip device index %d out of range\n", fn, dev); return 1; }
    return 0;
}


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


// stft_flt / frame_univar_flt framing; reference c/stft_flt.c:35-40,57-92, c/frame_univar_flt.c:41-87.
// The centre is accumulated serially in T, as the reference does, so non-dyadic hops round the same way.
// as_reference != 0 reproduces the X1 pointer walk of c/stft_flt.c (read start = start - shift once the
// start is non-negative); otherwise reads start at the frame's own start.
template <typename T>
int flt_desc(const char *fn, int dev, FrameDesc &d, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft, T c0, T stp,
             int mn0, int amp, int lg, int as_reference, bool check_nfft, std::shared_ptr<int> &hold, T stp_min)
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
              const T *tw_fixed, unsigned *take)
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

// the instantiations the other translation units link against
template int get_twiddles<float>(const char *fn, int dev, size_t nfft, int full, const float *&out);
template int get_twiddles<double>(const char *fn, int dev, size_t nfft, int full, const double *&out);
template int flt_desc<float>(const char *fn, int dev, FrameDesc &d, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft, float c0, float stp,
             int mn0, int amp, int lg, int as_reference, bool check_nfft, std::shared_ptr<int> &hold, float stp_min);
template int flt_desc<double>(const char *fn, int dev, FrameDesc &d, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft, double c0, double stp,
             int mn0, int amp, int lg, int as_reference, bool check_nfft, std::shared_ptr<int> &hold, double stp_min);
template int family_of<float>(size_t nfft);
template int family_of<double>(size_t nfft);
template int run_fused<float>(const char *fn, int dev, float *Yd, const float *Xd, const float *wd, const FrameDesc &d, hipStream_t s,
              const float *tw_fixed, unsigned *take);
template int run_fused<double>(const char *fn, int dev, double *Yd, const double *Xd, const double *wd, const FrameDesc &d, hipStream_t s,
              const double *tw_fixed, unsigned *take);
template int stft_dev<float>(const char *fn, float *Yd, const float *Xd, const float *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
             int snip_edges, int mn0, int amp, int lg, void *stream);
template int stft_dev<double>(const char *fn, double *Yd, const double *Xd, const double *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
             int snip_edges, int mn0, int amp, int lg, void *stream);
template int stft_flt_dev<float>(const char *fn, float *Yd, const float *Xd, const float *wd, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft,
                 float c0, float stp, int mn0, int amp, int lg, void *stream);
template int stft_flt_dev<double>(const char *fn, double *Yd, const double *Xd, const double *wd, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft,
                 double c0, double stp, int mn0, int amp, int lg, void *stream);
template int stft_dev_in_place<float>(const char *fn, float *Yd, const float *Xd, const float *wd, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp,
                      int snip_edges, int mn0, int amp, int lg, void *stream);
template int stft_dev_in_place<double>(const char *fn, double *Yd, const double *Xd, const double *wd, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp,
                      int snip_edges, int mn0, int amp, int lg, void *stream);
template int pc_epilogue<float>(const char *fn, float p, float preg, Epilogue &e);
template int pc_epilogue<double>(const char *fn, double p, double preg, Epilogue &e);
template int stft_pc_dev<float>(const char *fn, float *Yd, const float *Xd, const float *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                int snip_edges, int mn0, int amp, int lg, float p, float preg, void *stream);
template int stft_pc_dev<double>(const char *fn, double *Yd, const double *Xd, const double *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                int snip_edges, int mn0, int amp, int lg, double p, double preg, void *stream);
template int stft_pc_host<float>(const char *fn, float *Y, const float *X, const float *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                 int snip_edges, int mn0, int amp, int lg, float p, float preg);
template int stft_pc_host<double>(const char *fn, double *Y, const double *X, const double *win, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                 int snip_edges, int mn0, int amp, int lg, double p, double preg);
template int stft_cplx_dev<float>(const char *fn, float *Yd, const float *Xd, const float *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                  int snip_edges, int sc, void *stream);
template int stft_cplx_dev<double>(const char *fn, double *Yd, const double *Xd, const double *wd, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp,
                  int snip_edges, int sc, void *stream);

}  // namespace abi
}  // namespace dsp
