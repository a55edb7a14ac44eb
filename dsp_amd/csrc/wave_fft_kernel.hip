// Fused STFT kernels for gfx950 for the power-of-two transform lengths that have no kernel of their own
// (reference c/stft.c:33-162: frame, window, [mean-0], real FFT, power, [sqrt][log]).
//
// The nfft real samples of a frame are packed as M = nfft/2 complex points z_n = x_2n + i x_2n+1 and
// transformed by a Stockham autosort FFT: every pass reads R-point butterflies from an LDS buffer,
// applies the pass twiddles, transforms them in registers (radix 16, 8, 4 or 2, the packed helpers of
// fft16.h) and writes them in the order the next pass reads (to a second buffer, or in place behind a
// fence); the result of the last pass is in natural order.  A frame belongs to a group of lanes: 8 to 32
// lanes for the short transforms (a wave then works on 2 to 8 frames side by side), one wavefront in the
// middle, and a workgroup of 256 threads for the longest.  Up to one wavefront the passes are separated by
// wave-level fences only (the generic kernel of generic_kernels.hip pays log2(M) workgroup barriers per
// frame).  nfft is a template parameter: the radix plan, the loop counts and all LDS offsets are
// compile-time constants.  The real-FFT split is fused with the power step and the epilogue; rows are
// stored lane-linear in both directions.  The first pairs of the next frame (8 per lane) are loaded while
// the passes of the current one run.
#include "stft_common.h"
#include "fft16.h"

namespace dsp {

template <typename T> struct vec2u;                      // a pair of samples at sample alignment
template <> struct vec2u<float> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct vec2u<double> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };

template <typename T> __device__ __forceinline__ cx<T> cx_make(T a, T b) { return cx<T>{a, b}; }

// index of element i in a buffer: one slot of padding per 16, so that writes with a stride of R between
// lanes spread over the banks
__host__ __device__ constexpr int wpad(int i) { return i + (i >> 4); }

constexpr int wave_cap(int M, int elem) { return M / 64 < 4 ? 4 : (M / 64 > (elem == 4 ? 16 : 8) ? (elem == 4 ? 16 : 8) : M / 64); }
constexpr int wave_radix(int rem, int cap) { return rem < cap ? rem : cap; }
// lanes that share a frame: the whole wave from M = 256 on (float64: 128, measured); below, M / 4 (the butterflies of a radix-4
// pass), at least 8: a wave then transforms 2 to 8 short frames side by side instead of idling lanes
// The longest transform left to this kernel (float64 nfft 2048: a wave alone has neither the registers nor the LDS
// for it) takes a whole workgroup of 256 threads per frame, with workgroup barriers between the passes.
constexpr bool wave_wg(int M, int elem) { return M * elem >= 8192; }
constexpr int wave_gl(int M, int elem) { return wave_wg(M, elem) ? 256 : ((M >= 256 || (elem == 8 && M >= 128)) ? 64 : (M / 4 < 8 ? 8 : M / 4)); }
template <int GL> __device__ __forceinline__ void frame_sync() { if constexpr (GL > 64) { __syncthreads(); } else { wave_sync(); } }
constexpr int wave_slots(int M) { return wpad(M) + 1; }
// Pass twiddles: the pass that starts at sub-transform length ns (radix R) multiplies input t of butterfly
// j by W_{ns R}^{k t}, k = j mod ns.  They are laid out per pass as [t - 1][k], so that the lanes of a wave
// read consecutive entries at a compile-time offset.  Offset of the pass that starts at ns_target (the
// first pass has none); with ns_target = M: the size of the whole table.
constexpr int wave_tw_off(int M, int elem, int ns_target)
{
    int off = 0;
    for (int ns = 1; ns < ns_target;) {
        const int R = wave_radix(M / ns, wave_cap(M, elem));
        if (ns > 1) { off += (R - 1) * ns; }
        ns *= R;
    }
    return off;
}
// A pass can work in place (all reads of the wave, a fence, then the writes): one buffer per wave instead
// of two, i.e. twice the waves per CU for the LDS, but no overlap of a pass's reads and writes.  It pays
// where the LDS limits the waves and the registers allow more: float32 nfft 512 and 1024 (nfft 1024: 0.80 -> 0.70 ms at 16
// waves per CU; nfft 2048 float32 0.76 -> 0.86 and nfft 1024 float64 1.73 -> 1.99 with it: kept out).
constexpr bool wave_inplace(int M, int elem) { return (elem == 4 && M / 64 >= 4 && M / 64 <= 8) || M * elem >= 8192; }
// waves per CU worth having: the kernels wait on LDS round trips more than they compute, so as many as the
// registers allow: 16 for the short transforms and the in-place ones, 12 above (measured, nfft 256: 0.92 ms at 8 waves, 0.59 at
// 16; nfft 1024 at 129 registers: 1.07 ms at 8, 0.82 at 12, 1.18 at 14 in two workgroups of 7)
constexpr int wave_target(int M, int elem) { return (M <= 256 || wave_inplace(M, elem)) ? 16 : 12; }
// LDS of a workgroup of wpw waves: W_nfft^k, k < M, for the split; the pass twiddles; the window (M
// pairs); one or two buffers per wave
constexpr long wave_lds_of(int M, int elem, int wpw) { return (2L * M + wave_tw_off(M, elem, M) + (wave_inplace(M, elem) ? 1L : 2L) * wave_slots(M) * (wave_wg(M, elem) ? 1 : wpw * (64 / wave_gl(M, elem)))) * 2 * elem + 64; }
// waves per workgroup (1..8): the smallest count that puts the most waves (up to wave_target) on a CU's
// 160 KB; 0: the buffers do not fit at all
constexpr int wave_wpw(int M, int elem)
{
    if (wave_wg(M, elem)) { return wave_lds_of(M, elem, 4) <= 160 * 1024 ? 4 : 0; }
    int best = 0, best_waves = 0;
    for (int wpw = 1; wpw <= 8; ++wpw) {
        const long lds = wave_lds_of(M, elem, wpw);
        if (lds > 160 * 1024) { break; }
        long waves = (160 * 1024 / lds) * wpw;
        if (waves > wave_target(M, elem)) { waves = wave_target(M, elem); }
        if (waves > best_waves) { best_waves = (int)waves; best = wpw; }
    }
    return best;
}
constexpr long wave_lds(int M, int elem) { return wave_lds_of(M, elem, wave_wpw(M, elem)); }

template <typename T>
__device__ __forceinline__ void dft8c(cx<T> (&v)[8])
{
    constexpr T R2 = T(0.70710678118654752440084436210484904L);
    dft4c<T>(v[0], v[2], v[4], v[6]);
    dft4c<T>(v[1], v[3], v[5], v[7]);
    const cx<T> o1 = cx_mul<T>(v[3], R2, -R2);                       // W8^1
    const cx<T> o2 = cx_swap(v[5]) * cx<T>{T(1), T(-1)};             // W8^2 = -i
    const cx<T> o3 = cx_mul<T>(v[7], -R2, -R2);                      // W8^3
    const cx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}

// One Stockham pass of radix R over M points: butterfly j reads src[j + t M/R], multiplies input t by
// W_{NS R}^{k t} (k = j mod NS, NS = product of the radices of the passes before) and writes output t to
// dst[(j - k) R + k + t NS].  tw: the pass twiddles (wave_tw_off).
template <typename T, int M, int R, int NS>
__device__ __forceinline__ void wave_pass(const cx<T> *src, cx<T> *dst, const cx<T> *tw, const int lane)
{
    using C = cx<T>;
    constexpr int GL = wave_gl(M, (int)sizeof(T));                         // lanes on this frame; lane: index among them
    constexpr int NB = M / R;                              // butterflies
    constexpr int IT = NB >= GL ? NB / GL : 1;
    constexpr int OFF = wave_tw_off(M, (int)sizeof(T), NS);
#pragma unroll 1
    for (int i = 0; i < IT; ++i) {                         // rolled: unrolled, the copies' registers add up
        const int j = lane + GL * i;
        if (NB < GL && lane >= NB) { continue; }
        const int k = j & (NS - 1);
        C v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) { v[t] = src[wpad(j + t * NB)]; }
        if (NS > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) { v[t] = cx_mul<T>(v[t], tw[OFF + (t - 1) * NS + k]); }
        }
        const int base = (j - k) * R + k;
        if constexpr (R == 16) {
            dft16c<T, 16>(v);
#pragma unroll
            for (int t = 0; t < 16; ++t) { dst[wpad(base + t * NS)] = v[slot16(t)]; }
        } else {
            if constexpr (R == 8) { dft8c<T>(v); }
            else if constexpr (R == 4) { dft4c<T>(v[0], v[1], v[2], v[3]); }
            else { const C a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; }
#pragma unroll
            for (int t = 0; t < R; ++t) { dst[wpad(base + t * NS)] = v[t]; }
        }
    }
}

// The same pass in place: every lane first reads the inputs of all its butterflies, then, behind a wave
// fence, transforms and writes them.
template <typename T, int M, int R, int NS>
__device__ __forceinline__ void wave_pass_inplace(cx<T> *buf, const cx<T> *tw, const int lane)
{
    using C = cx<T>;
    constexpr int GL = wave_gl(M, (int)sizeof(T));
    constexpr int NB = M / R;
    constexpr int IT = NB >= GL ? NB / GL : 1;
    constexpr int OFF = wave_tw_off(M, (int)sizeof(T), NS);
    const bool active = NB >= GL || lane < NB;
    C v[IT][R];
    if (active) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
#pragma unroll
            for (int t = 0; t < R; ++t) { v[i][t] = buf[wpad(lane + GL * i + t * NB)]; }
        }
    }
    frame_sync<GL>();
    if (active) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int j = lane + GL * i;
            const int k = j & (NS - 1);
            if (NS > 1) {
#pragma unroll
                for (int t = 1; t < R; ++t) { v[i][t] = cx_mul<T>(v[i][t], tw[OFF + (t - 1) * NS + k]); }
            }
            const int base = (j - k) * R + k;
            if constexpr (R == 16) {
                dft16c<T, 16>(v[i]);
#pragma unroll
                for (int t = 0; t < 16; ++t) { buf[wpad(base + t * NS)] = v[i][slot16(t)]; }
            } else {
                if constexpr (R == 8) { dft8c<T>(v[i]); }
                else if constexpr (R == 4) { dft4c<T>(v[i][0], v[i][1], v[i][2], v[i][3]); }
                else { const C a = v[i][0], b = v[i][1]; v[i][0] = a + b; v[i][1] = a - b; }
#pragma unroll
                for (int t = 0; t < R; ++t) { buf[wpad(base + t * NS)] = v[i][t]; }
            }
        }
    }
}

// the pass twiddles from sub-transform length NS on, from the table W_2M^m, m < M, in global memory
template <typename T, int M, int NS>
__device__ __forceinline__ void wave_fill_tw(cx<T> *ptw, const T *__restrict__ tw, const int tid, const int nthr)
{
    if constexpr (NS < M) {
        constexpr int R = wave_radix(M / NS, wave_cap(M, (int)sizeof(T)));
        if constexpr (NS > 1) {
            constexpr int OFF = wave_tw_off(M, (int)sizeof(T), NS);
            constexpr int TMUL = 2 * M / (NS * R);
            for (int idx = tid; idx < (R - 1) * NS; idx += nthr) {
                const int t = idx / NS + 1, k = idx - (t - 1) * NS;
                const int e = k * t * TMUL;                  // < 2 M; the second half of the circle is the first negated
                const int h = e < M ? e : e - M;
                const T sg = e < M ? T(1) : T(-1);
                ptw[OFF + idx] = cx_make<T>(sg * tw[2 * h], sg * tw[2 * h + 1]);
            }
        }
        wave_fill_tw<T, M, NS * R>(ptw, tw, tid, nthr);
    }
}

// the passes from sub-transform length NS on; returns the buffer that holds the result
template <typename T, int M, int NS>
__device__ __forceinline__ cx<T> *wave_passes(cx<T> *src, cx<T> *dst, const cx<T> *tw, const int lane)
{
    if constexpr (NS >= M) {
        return src;
    } else {
        constexpr int R = wave_radix(M / NS, wave_cap(M, (int)sizeof(T)));
        if constexpr (wave_inplace(M, (int)sizeof(T))) {
            wave_pass_inplace<T, M, R, NS>(src, tw, lane);
            frame_sync<wave_gl(M, (int)sizeof(T))>();
            return wave_passes<T, M, NS * R>(src, dst, tw, lane);
        } else {
            wave_pass<T, M, R, NS>(src, dst, tw, lane);
            frame_sync<wave_gl(M, (int)sizeof(T))>();
            return wave_passes<T, M, NS * R>(dst, src, tw, lane);
        }
    }
}

template <typename T, int NFFT>
__global__ __launch_bounds__(64 * wave_wpw(NFFT / 2, (int)sizeof(T)), 2) void stft_wave_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                                                           const T *__restrict__ win, const T *__restrict__ tw,
                                                                                           const FrameDesc d)
{
    using C = cx<T>;
    using T2u = typename vec2u<T>::type;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int M = NFFT / 2, F = M + 1;
    constexpr int WPW = wave_wpw(M, (int)sizeof(T));
    constexpr int SLOTS = wave_slots(M);
    constexpr int GL = wave_gl(M, (int)sizeof(T));         // lanes per frame
    constexpr bool WG = GL > 64;                           // the workgroup shares one frame
    constexpr int FPW = WG ? 1 : 64 / GL;                  // frames per wave
    constexpr int E = M >= GL ? M / GL : 1;                // pairs per lane
    constexpr int U = E < 8 ? E : 8;                       // pairs per lane and block of loads
    const int tid = threadIdx.x;
    const int lane = tid & (GL - 1);                       // index among the lanes of the frame
    const int grp = WG ? 0 : (tid & 63) / GL;              // which of the wave's frames
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = d.L;

    // LDS: [W_nfft^m, m < M, for the split][pass twiddles][window as M pairs, zero past L] then per wave two
    // buffers of M (padded) complex points
    constexpr int TWN = wave_tw_off(M, (int)sizeof(T), M);
    C *twl = reinterpret_cast<C *>(smem_raw);
    C *ptw = twl + M;
    C *wl = ptw + TWN;
    constexpr int NBUF = wave_inplace(M, (int)sizeof(T)) ? 1 : 2;
    C *buf0 = wl + M + (WG ? 0 : (size_t)(wv * FPW + grp) * NBUF * SLOTS);
    C *buf1 = buf0 + (NBUF - 1) * SLOTS;
    T *red = reinterpret_cast<T *>(wl + M + (size_t)(WG ? 1 : WPW * FPW) * NBUF * SLOTS);   // WG: 4 partial sums
    for (int m = tid; m < M; m += 64 * WPW) {
        twl[m] = cx_make<T>(tw[2 * m], tw[2 * m + 1]);
        wl[m] = cx_make<T>(2 * m < L ? win[2 * m] : T(0), 2 * m + 1 < L ? win[2 * m + 1] : T(0));
    }
    wave_fill_tw<T, M, 1>(ptw, tw, tid, 64 * WPW);
    __syncthreads();

    const long long total = (long long)d.B * d.W;
    const long long stride = WG ? (long long)gridDim.x : (long long)gridDim.x * WPW * FPW;

    // where a frame lies
    struct Where { const T *xb; long long ss, rdoff, row; bool inside; };
    auto locate = [&](long long fi) {
        Where q;
        const long long b = fi / d.W;
        const int w = (int)(fi - b * d.W);
        frame_origin(d, w, q.ss, q.rdoff);
        if (!d.edge) { q.rdoff = 0; }
        q.xb = X + b * d.ldx;
        q.row = b * d.wy() + w;
        q.inside = q.ss >= 0 && q.ss + L <= d.N && q.ss >= q.rdoff;
        return q;
    };
    // The pairs m0 + GL u + lane, u < U, of a frame: the loads of a block go out together, one round
    // trip to memory per block.  Frames across a signal edge: sample by sample.
    auto load_block = [&](const Where &q, const int m0, C (&x)[U]) {
        if (q.inside) {
            const T *src = q.xb + (q.ss - q.rdoff);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int l = 2 * (m0 + GL * u + lane);
                x[u] = cx_make<T>(T(0), T(0));
                if (l + 1 < L) { const T2u v = *reinterpret_cast<const T2u *>(src + l); x[u] = cx_make<T>(v.x, v.y); }
                else if (l < L) { x[u].x = src[l]; }
            }
        } else {
#pragma unroll 1
            for (int u = 0; u < U; ++u) {
                const int l = 2 * (m0 + GL * u + lane);
                const T v0 = (l < L) ? fetch_sample(q.xb, d, q.ss, q.rdoff, l) : T(0);
                const T v1 = (l + 1 < L) ? fetch_sample(q.xb, d, q.ss, q.rdoff, l + 1) : T(0);
#pragma unroll
                for (int r = 0; r < U; ++r) { if (r == u) { x[r] = cx_make<T>(v0, v1); } }
            }
        }
    };

    long long fi = WG ? (long long)blockIdx.x : ((long long)blockIdx.x * WPW + wv) * FPW + grp;
    if (fi >= total) { return; }
    Where cur = locate(fi);
    C xn[U];                                                // first block of the frame to come
    load_block(cur, 0, xn);
    for (;;) {
        // window the frame into buf0 as complex pairs, the zero padding up to nfft included
        T part = T(0);
#pragma unroll
        for (int blk = 0; blk < E / U; ++blk) {
            C x[U];
            if (blk == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) { x[u] = xn[u]; }
            } else {
                load_block(cur, GL * U * blk, x);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = GL * U * blk + GL * u + lane;
                if (M >= GL || m < M) {
                    const C z = x[u] * wl[m];
                    part += z.x + z.y;
                    buf0[wpad(m)] = z;
                }
            }
        }
        const long long row = cur.row;
        fi += stride;
        const bool more = fi < total;
        if (more) {
            cur = locate(fi);
            load_block(cur, 0, xn);
        }
        if (d.mn0) {
#pragma unroll
            for (int o = 1; o < (WG ? 64 : GL); o <<= 1) { part += __shfl_xor(part, o); }
            if constexpr (WG) {
                if ((tid & 63) == 0) { red[wv] = part; }
                __syncthreads();
                part = red[0] + red[1] + red[2] + red[3];
            }
            const T mean = part / T(L);
            frame_sync<GL>();
            for (int m = lane; 2 * m < L; m += GL) {
                C z = buf0[wpad(m)];
                z.x -= mean;
                if (2 * m + 1 < L) { z.y -= mean; }
                buf0[wpad(m)] = z;
            }
        }
        frame_sync<GL>();

        const C *res = wave_passes<T, M, 1>(buf0, buf1, ptw, lane);

        // real-FFT split fused with power: with A = Z_k + conj Z_{M-k}, B = Z_k - conj Z_{M-k},
        // X_k = (A - i W^k B)/2 and X_{M-k} = conj((A + i W^k B)/2)
        T *y = Y + row * (long long)F;
        constexpr int KI = (M / 2 + 1 + GL - 1) / GL;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int k = lane + GL * i;
            if (k <= M / 2) {
                const int km = (M - k) & (M - 1);
                const C z = res[wpad(k)], zm = res[wpad(km)];
                const C A = cx_add_conj<T>(z, zm);
                const C Bv = cx_sub_conj<T>(z, zm);
                const C wk = twl[k];
                const C Tw = cx_mul<T>(Bv, wk.y, -wk.x);         // -i W^k B
                const C Pp = A + Tw, Qq = A - Tw;
                y[k] = power_epilogue(T(0.25) * cx_norm<T>(Pp), d.epi);
                y[M - k] = power_epilogue(T(0.25) * cx_norm<T>(Qq), d.epi);
            }
        }
        if (!more) { break; }
        frame_sync<GL>();
    }
}

template <typename T, int NFFT>
static hipError_t launch_wave_inst(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    constexpr int M = NFFT / 2;
    constexpr int WPW = wave_wpw(M, (int)sizeof(T));
    constexpr size_t lds = (size_t)wave_lds(M, (int)sizeof(T));
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_wave_kernel<T, NFFT>), lds, grant); e != hipSuccess) { return e; }
    const long long total = (long long)d.B * d.W;
    const long long per_cu = std::max<long long>(1, std::min<long long>(wave_target(M, (int)sizeof(T)) / WPW > 0 ? wave_target(M, (int)sizeof(T)) / WPW : 1, (long long)(160 * 1024 / lds)));   // resident workgroups per CU
    constexpr bool WG = wave_wg(M, (int)sizeof(T));
    constexpr int FPW = WG ? 1 : 64 / wave_gl(M, (int)sizeof(T));
    long long grid = WG ? total : (total + WPW * FPW - 1) / (WPW * FPW);
    grid = std::min(grid, 256 * per_cu);
    hipLaunchKernelGGL((stft_wave_kernel<T, NFFT>), dim3((unsigned)grid), dim3(64 * WPW), lds, s, Y, X, win, tw, d);
    return hipGetLastError();
}

template <typename T>
bool wave_fft_supported(const FrameDesc &d)
{
    const int n = d.nfft;
    if (n < 16 || (n & (n - 1)) != 0 || d.L < 1 || d.L > n) { return false; }
    return n <= 2048;                                    // 4096 and float32 8192 have kernels of their own
}

template <typename T>
hipError_t launch_stft_wave(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    switch (d.nfft) {
    case 16: return launch_wave_inst<T, 16>(Y, X, win, tw, d, s);
    case 32: return launch_wave_inst<T, 32>(Y, X, win, tw, d, s);
    case 64: return launch_wave_inst<T, 64>(Y, X, win, tw, d, s);
    case 128: return launch_wave_inst<T, 128>(Y, X, win, tw, d, s);
    case 256: return launch_wave_inst<T, 256>(Y, X, win, tw, d, s);
    case 512: return launch_wave_inst<T, 512>(Y, X, win, tw, d, s);
    case 1024: return launch_wave_inst<T, 1024>(Y, X, win, tw, d, s);
    case 2048: return launch_wave_inst<T, 2048>(Y, X, win, tw, d, s);
    default: return hipErrorInvalidValue;
    }
}

template bool wave_fft_supported<float>(const FrameDesc &);
template bool wave_fft_supported<double>(const FrameDesc &);
template hipError_t launch_stft_wave<float>(float *, const float *, const float *, const float *, const FrameDesc &, hipStream_t);
template hipError_t launch_stft_wave<double>(double *, const double *, const double *, const double *, const FrameDesc &, hipStream_t);

}  // namespace dsp
