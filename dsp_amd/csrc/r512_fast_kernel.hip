// Fast path of the 512-point fused STFT kernel for gfx950: integer hop with 3*hop + 512 <= 2048 (hop <= 512),
// frames addressed as start = ss0 + w*hop (the stft framing, and the stft_flt framing when
// its hop is integral), or (TBL) frames whose starts come from a table, never decrease and lie at most 1536
// samples apart for frames w and w + 3 (the stft_flt framing with a non-integral hop).  Same arithmetic as the basic kernel in r512_kernel.hip; what differs is the
// schedule around it:
//   * the span of the next 4 frames is fetched from HBM into registers (16-byte loads) while the
//     current 4 frames are transformed, and is committed to LDS at the top of the next step, so the
//     HBM latency is off the critical path; every input sample is fetched once per run of frames;
//   * the staged span and the 16 x 16 exchange tile share one LDS region (8.7 KB per wave in
//     float); two workgroups of four waves run on a CU (the registers, not the LDS, set that), swap
//     issue priority every step and keep their waves in step with a barrier every fourth step;
//   * rows leave as 16-byte non-temporal stores of the step's 4 x 257 values, staged through LDS in
//     the order they have in the output;
//   * the compile-time shape (rows of the window that hold data, mean removal, sqrt/log epilogue)
//     removes the per-element selects and branches of the general kernel.
#include "stft_common.h"
#include "fft16.h"
#include <cstdlib>

namespace dsp {

struct R512FastArgs {
    long long ldx;        // elements between signals
    int N;                // samples per signal
    int L;                // window length (NR rows of 32 hold data)
    int W;                // frames per signal
    int stp;              // hop (table form: the largest distance between the starts of frames w and w + 3)
    int ss0;              // start of frame 0
    int rd_shift;         // edge == 1: samples are read at n - rd_shift
    int edge;             // 0 reflect, 1 zero
    Epilogue epi;
    int sps;              // steps (groups of 4 consecutive frames) per signal
    int steps;            // B * sps
    const int *tbl;       // table form: start of frame w at tbl[2 w] (frames need not be equally spaced)
};

constexpr int R512F_WAVE_ELEMS = 4 * R512_EXF * 2;   // T elements of LDS per wave (>= 1280 staged samples)

// value of lane (16 - i) % 16 of the same 16-lane row (row_mirror, then row_ror:1)
__device__ __forceinline__ int dpp_partner16_i(int v)
{
    const int m = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
    return __builtin_amdgcn_update_dpp(0, m, 0x121, 0xf, 0xf, false);
}
__device__ __forceinline__ float dpp_partner16(float v) { return __int_as_float(dpp_partner16_i(__float_as_int(v))); }
__device__ __forceinline__ double dpp_partner16(double v)
{
    const int lo = dpp_partner16_i(__double2loint(v)), hi = dpp_partner16_i(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <typename T> struct vec16;                   // 16-byte vector of T
template <> struct vec16<float> { using type = float4; static constexpr int n = 4; };
template <> struct vec16<double> { using type = double2; static constexpr int n = 2; };
template <typename T> struct vec2e;                   // a pair of T at element alignment
template <> struct vec2e<float> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct vec2e<double> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };
template <typename T> struct vec16u;                  // the same at element alignment
template <> struct vec16u<float> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); };
template <> struct vec16u<double> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };

#ifndef R512F_TW_LDS
#define R512F_TW_LDS 0          // 1: pass-1 twiddles come from an LDS table instead of 30 registers
#endif
#ifndef R512F_WIN_LDS
#define R512F_WIN_LDS 0         // 1: float window taps come from LDS too (double always does)
#endif
#ifndef R512F_ABLATE
#define R512F_ABLATE 0          // diagnostic builds: 1 = memory traffic only, 2 = no scalar-path stores, 3 = no FFT arithmetic, 4 = no exchange through LDS
#endif
#ifndef R512F_NT_STORE
#define R512F_NT_STORE 1        // 1: spectrogram rows are stored with the non-temporal hint
#endif
#ifndef R512F_NT_LOAD
#define R512F_NT_LOAD 0         // 1: signal spans are loaded with the non-temporal hint
#endif
#if R512F_NT_STORE
#define R512F_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define R512F_ST(ptr, val) (*(ptr) = (val))
#endif
#ifndef R512F_ALT_PRIO
#define R512F_ALT_PRIO 1        // 1 = the two workgroups of a CU swap issue priority every step
#endif
#ifndef R512F_STAMPS
#define R512F_STAMPS 0          // diagnostic builds: 1 = per-phase s_memtime sums into g_r512f_stamps
#endif
#if R512F_STAMPS
__device__ unsigned long long g_r512f_stamps[8192 * 8];
#define R512F_STAMP(i)                                                                            \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        st_acc[i] += t_ - st_last; st_last = t_;                                                  \
    } while (0)
#else
#define R512F_STAMP(i) do {} while (0)
#endif
#ifndef R512F_TW_LDS_F64
#define R512F_TW_LDS_F64 1      // 1: the double kernel reads its pass-1 twiddles from LDS (they do not fit in registers)
#endif
#ifndef R512F_NOLOAD
#define R512F_NOLOAD 0          // diagnostic builds: 1 = the span loads are compiled out
#endif
#ifndef R512F_NOSTORE
#define R512F_NOSTORE 0         // diagnostic builds: 1 = the row stores are compiled out
#endif
#ifndef R512F_LOCK_EVERY
#define R512F_LOCK_EVERY 4      // n > 0 (power of two): the waves of a workgroup meet at a barrier every n steps
#endif
#ifndef R512F_ST4
#define R512F_ST4 1             // 1: rows leave as 16-byte vector stores when the step's block is 16-byte aligned
#endif
#ifndef R512F_MINWAVES_F32
#define R512F_MINWAVES_F32 2
#endif

template <typename T, int NR, bool MN0, bool EPI, int WIDE, bool TBL>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? R512F_MINWAVES_F32 : 2)) void stft_r512_fast_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                              const T *__restrict__ win, const T *__restrict__ tw,
                                                              const R512FastArgs a)
{
    using T2 = typename vec2<T>::type;
    using TV = typename vec16<T>::type;
    // staged elements per lane: 16 (spans of up to 1024 samples: hop <= 170), WIDE 1: 20 (1280: hop <= 256),
    // WIDE 2: 32 (2048: hop <= 512)
    constexpr int VEC = vec16<T>::n, SPAN = WIDE == 2 ? 2048 : WIDE ? 1280 : 1024, NV = SPAN / 64 / VEC;
    static_assert(SPAN <= R512F_WAVE_ELEMS, "the staged span lives in the wave's exchange region");
    constexpr bool WINREG = (sizeof(T) == 4) && !R512F_WIN_LDS;
    constexpr bool TWREG = !R512F_TW_LDS && !(R512F_TW_LDS_F64 && sizeof(T) == 8);
    constexpr int M = 256, F = 257;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane >> 4, j = lane & 15;
    const int L = a.L;

    T *xs = reinterpret_cast<T *>(smem_raw) + wv * R512F_WAVE_ELEMS;   // staged span; aliases the exchange tile
    T2 *ex = reinterpret_cast<T2 *>(xs);
    T *wtab = reinterpret_cast<T *>(smem_raw) + 4 * R512F_WAVE_ELEMS;  // halved window, 512 T
    T2 *ttab = reinterpret_cast<T2 *>(wtab + 512);                     // W256^(j k1) at [k1][j], 256 T2

    // ---- per-lane constants ----
    using C = cx<T>;
    C twd[16];                                             // W256^(j k1) = W512^(2 j k1)
    if (TWREG) {
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
            int idx = 2 * j * k1;
            T sg = T(1);
            if (idx >= 256) { idx -= 256; sg = T(-1); }
            twd[k1] = C{sg * tw[2 * idx], sg * tw[2 * idx + 1]};
        }
    } else {
        const int k1 = tid >> 4, jj = tid & 15;
        int idx = 2 * jj * k1;
        T sg = T(1);
        if (idx >= 256) { idx -= 256; sg = T(-1); }
        T2 t; t.x = sg * tw[2 * idx]; t.y = sg * tw[2 * idx + 1];
        ttab[tid] = t;
    }
    C ow[8];                                               // -i W512^(j + 16 k2), k2 < 8
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) { ow[k2] = C{tw[2 * (j + 16 * k2) + 1], -tw[2 * (j + 16 * k2)]}; }
    C hw[NR];                                              // halved window taps of this lane's columns
    if (WINREG) {
#pragma unroll
        for (int n1 = 0; n1 < NR; ++n1) {
            const int l = 32 * n1 + 2 * j;
            hw[n1] = C{(l < L) ? T(0.5) * win[l] : T(0), (l + 1 < L) ? T(0.5) * win[l + 1] : T(0)};
        }
    } else {
        for (int l = tid; l < 512; l += 256) { wtab[l] = (l < L) ? T(0.5) * win[l] : T(0); }
    }
    if (!WINREG || !TWREG) { __syncthreads(); }
    static_assert(!TBL || (WIDE >= 1 && NR == 16), "the table form uses the wide, 16-row instantiations");
    const int reach = TBL ? a.stp : 3 * a.stp;               // start of the step's last frame relative to its first
    const int need = reach + 512;
    const int nvec = (reach + 32 * NR + VEC - 1) / VEC;      // 16-byte vectors of a span that pass 1 reads
    const T *xrow = xs + f * a.stp + 2 * j;                  // (table form: set per step)
    // table form: lane i holds the start of frame w + (i & 3) of a step (clamped to the last frame)
    auto tbl_fetch = [&](const int w_) { return a.tbl[2 * min(w_ + (lane & 3), a.W - 1)]; };
    auto tbl_xrow = [&](const int tv) { return xs + (__shfl(tv, f) - __builtin_amdgcn_readfirstlane(tv)) + 2 * j; };

    // ---- fetch of one span (frames w..w+3 of the signal at xb) into registers ----
    TV pre[NV];
    auto load_span = [&](const T *xb, const int s0) {
        const T *src = xb + (s0 - (a.edge ? a.rd_shift : 0));
        const bool inside = s0 >= 0 && s0 + SPAN <= a.N && (a.edge == 0 || s0 >= a.rd_shift);
        if (inside && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                // only 3*hop + 32*NR samples are ever read by pass 1: lanes past that re-read the last
                // needed vector (same cache line, no extra traffic) so that the number of loads stays fixed
#if R512F_NOLOAD
                if (a.N == -12345)
#endif
                pre[q] = reinterpret_cast<const TV *>(src)[min(q * 64 + lane, nvec - 1)];
            }
        } else if (inside) {
            // a span that does not start on a 16-byte boundary (odd ss0, a hop that is no multiple of 4
            // elements, an unaligned signal): the same loads at element alignment
            using TVu = typename vec16u<T>::type;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const TVu v = reinterpret_cast<const TVu *>(src)[min(q * 64 + lane, nvec - 1)];
                if constexpr (VEC == 4) { pre[q] = TV{v.x, v.y, v.z, v.w}; }
                else { pre[q] = TV{v.x, v.y}; }
            }
        } else {
            FrameDesc d{};
            d.N = a.N; d.edge = a.edge;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                T t[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const int i = (q * 64 + lane) * VEC + v;
                    t[v] = (i < need) ? fetch_sample(xb, d, (long long)s0, (long long)a.rd_shift, i) : T(0);
                }
                if constexpr (VEC == 4) { pre[q] = TV{t[0], t[1], t[2], t[3]}; }
                else { pre[q] = TV{t[0], t[1]}; }
            }
        }
    };

    // Work distribution: the steps of the whole batch are cut into 8 contiguous ranges, one per group
    // of workgroups that share an XCD (blockIdx % 8), and inside a range the waves take steps
    // round-robin.  The waves of a range therefore read and write one moving window of adjacent
    // addresses (good DRAM page locality, and the samples two neighbouring steps share come from the
    // same L2) instead of thousands of separate streams.
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int s_lo = (int)((long long)a.steps * xcd / 8), s_hi = (int)((long long)a.steps * (xcd + 1) / 8);
    const int sstride = nwg_x * 4;
    int sid = s_lo + (blockIdx.x >> 3) * 4 + wv;
    if (sid >= s_hi) { return; }
    int b = sid / a.sps;
    int w = (sid - b * a.sps) * 4;
    const T *xb = X + (long long)b * a.ldx;
    T *yb = Y + (long long)b * a.W * F;

    // position of the wave's next step
    int nsid, nw;
    const T *nxb;
    T *nyb;
    bool more;
    auto advance = [&]() {
        nsid = sid + sstride;
        more = nsid < s_hi;
        if (more) {
            const int nb = nsid / a.sps;
            nw = (nsid - nb * a.sps) * 4;
            nxb = X + (long long)nb * a.ldx;
            nyb = Y + (long long)nb * a.W * F;
        }
    };

#if R512F_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
#endif
    // prologue: first span to LDS, second span in flight (table form: the starts of its frames in
    // flight; its span is fetched in the middle of the step, when they have arrived)
    int tvn = 0;
    if constexpr (TBL) {
        const int tv = tbl_fetch(w);
        xrow = tbl_xrow(tv);
        load_span(xb, __builtin_amdgcn_readfirstlane(tv));
    } else {
        load_span(xb, a.ss0 + w * a.stp);
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) { reinterpret_cast<TV *>(xs)[q * 64 + lane] = pre[q]; }
    wave_sync();
    advance();
    if constexpr (TBL) { if (more) { tvn = tbl_fetch(nw); } }
    else { if (more) { load_span(nxb, nw * a.stp + a.ss0); } }

    // Steady state per step: transform the staged span and store its rows while the next span is in
    // flight; then commit that span to LDS and issue the fetch of the one after.  The stores of a
    // step are unconditional and fixed in number, so the wait for the in-flight span leaves them
    // outstanding instead of draining the store queue.
#if R512F_LOCK_EVERY
    int it = 0;
#endif
    while (true) {
#if R512F_LOCK_EVERY
        if ((it++ & (R512F_LOCK_EVERY - 1)) == 0) { __builtin_amdgcn_s_barrier(); }
#if R512F_ALT_PRIO
        // The two workgroups of a CU (blockIdx b and b + 256) share its SIMDs, and of two waves on a
        // SIMD the older one issues first: left alone, one workgroup runs 5 % ahead of the other and
        // the CU is half empty at the end.  Swapping the priority every step keeps them level.
        if ((((int)blockIdx.x >> 8) ^ it) & 1) { __builtin_amdgcn_s_setprio(1); } else { __builtin_amdgcn_s_setprio(0); }
#endif
#endif
        // ---- pass 1: lane (frame f, column n2 = j): 16-point FFT over the rows n1 ----
        C zz[16];
#if R512F_ABLATE != 1
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            if (n1 < NR) {
                C v;
                if constexpr (WIDE) {
                    // odd hops put the rows of frames 1 and 3 at odd element offsets: pair reads at element alignment
                    using T2e = typename vec2e<T>::type;
                    const T2e t = *reinterpret_cast<const T2e *>(xrow + 32 * n1);
                    v = C{t.x, t.y};
                } else {
                    const T2 t = *reinterpret_cast<const T2 *>(xrow + 32 * n1);
                    v = C{t.x, t.y};
                }
                if (NR == 16 || n1 == NR - 1) {          // rows that may reach past the window
                    const int l = 32 * n1 + 2 * j;
                    v.x = (l < L) ? v.x : T(0);
                    v.y = (l + 1 < L) ? v.y : T(0);
                }
                C h;
                if (WINREG) { h = hw[n1]; }
                else { const T2 hh = *reinterpret_cast<const T2 *>(wtab + 32 * n1 + 2 * j); h = C{hh.x, hh.y}; }
                zz[n1] = v * h;
            } else {
                zz[n1] = C{T(0), T(0)};
            }
        }
        if (MN0) {
            C acc = zz[0];
#pragma unroll
            for (int n1 = 1; n1 < NR; ++n1) { acc += zz[n1]; }
            T sum = acc.x + acc.y;
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
            const T mean = sum / T(L);
#pragma unroll
            for (int n1 = 0; n1 < NR; ++n1) {
                if (NR == 16 || n1 == NR - 1) {          // rows that may reach past the window
                    const int l = 32 * n1 + 2 * j;
                    zz[n1] -= C{(l < L) ? mean : T(0), (l + 1 < L) ? mean : T(0)};
                } else {
                    zz[n1] -= C{mean, mean};
                }
            }
        }
#if R512F_ABLATE != 3
        dft16c<T, NR>(zz);
#endif
        const T *nxrow = xrow;
        if constexpr (TBL) {
            if (more) { nxrow = tbl_xrow(tvn); load_span(nxb, __builtin_amdgcn_readfirstlane(tvn)); }
        }
        R512F_STAMP(0);
        wave_sync();                                       // all lanes have read the span: the tile may overwrite it
#if R512F_ABLATE != 4
        {
            T2 *exf = ex + f * R512_EXF + j;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                C v = zz[slot16(k1)];
                if (k1 > 0) {
                    if (TWREG) { v = cx_mul<T>(v, twd[k1]); }
                    else { const T2 t = ttab[k1 * 16 + j]; v = cx_mul<T>(v, t.x, t.y); }
                }
                T2 o; o.x = v.x; o.y = v.y;
                exf[k1 * 17] = o;
            }
        }
        wave_sync();
        R512F_STAMP(1);

        // ---- pass 2: lane (frame f, output residue k1 = j): 16-point FFT over n2 ----
        {
            const T2 *exf = ex + f * R512_EXF + j * 17;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) { const T2 t = exf[n2]; zz[n2] = C{t.x, t.y}; }
        }
#else
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) { zz[k1] = cx_mul<T>(zz[k1], twd[k1]); }
#endif
#if R512F_ABLATE != 3
        dft16c<T, 16>(zz);
#endif
        R512F_STAMP(2);

        // ---- real-FFT split fused with power, in registers.  The lane holds Z[j + 16 k2] (halved);
        // its partner Z[256 - k] for k = j + 16 k2, k2 < 8, is register 15 - k2 of lane (16 - j) % 16
        // (register (16 - k2) % 16 of lane 0 itself when j == 0): one row_mirror + row_ror:1 DPP move.
        // With A = Z_k + conj Z_{256-k}, B = Z_k - conj Z_{256-k}, T = -i W512^k B:
        // |X_k|^2 = |A + T|^2 and |X_{256-k}|^2 = |A - T|^2.
        T pw[16];                                          // pw[k2] = P[k], pw[8 + k2] = P[256 - k]
        T p128 = T(4) * cx_norm<T>(zz[slot16(8)]);         // lane j == 0 only
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const C z = zz[slot16(k2)];
            // (the empty asm keeps the compiler from turning the select of two array elements into a
            // select of the index, which it would lower to a 16-way chain of conditional moves)
            T x0 = zz[slot16((16 - k2) & 15)].x, x1 = zz[slot16(15 - k2)].x;
            T y0 = zz[slot16((16 - k2) & 15)].y, y1 = zz[slot16(15 - k2)].y;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));
            const T sx = (j == 0) ? x0 : x1;
            const T sy = (j == 0) ? y0 : y1;
            const C m = C{dpp_partner16(sx), dpp_partner16(sy)};
            const C A = cx_add_conj<T>(z, m), B = cx_sub_conj<T>(z, m);
            const C Tw = cx_mul<T>(B, ow[k2]);
            pw[k2] = cx_norm<T>(A + Tw);
            pw[8 + k2] = cx_norm<T>(A - Tw);
        }
        if (EPI) {
            power_epilogue_n(pw, a.epi);
            p128 = power_epilogue(p128, a.epi);
        }
        wave_sync();                                       // every lane has read its tile row: reuse the region
        {
            T *pf = xs + f * F + j;                        // 4 rows of 257 values, as they lie in Y
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                pf[16 * k2] = pw[k2];                      // bin j + 16 k2
                pf[M - 2 * j - 16 * k2] = pw[8 + k2];      // bin 256 - (j + 16 k2)
            }
            if (j == 0) { pf[128] = p128; }
        }
        wave_sync();
        R512F_STAMP(3);
#endif

        // ---- store the rows of the step: 1028 consecutive values of Y (fewer at the end of a signal:
        // the index is clamped, so the surplus lanes rewrite the last value with itself and the number
        // of store instructions per step stays fixed)
        {
            T *yblk = yb + (unsigned)(w * F);
#if R512F_ST4
            if (a.W - w >= 4 && (reinterpret_cast<uintptr_t>(yblk) & 15) == 0) {
                // 1028 values = 257 16-byte vectors (float) / 514 (double): whole-vector wave stores
                constexpr int NVEC = 1028 / VEC, FULLI = NVEC / 64;
#pragma unroll
                for (int i = 0; i <= FULLI; ++i) {
                    const int idx = (i < FULLI) ? lane + 64 * i : FULLI * 64 + (lane & (NVEC - FULLI * 64 - 1));
#if R512F_NOSTORE
                    const TV v4 = reinterpret_cast<const TV *>(xs)[idx];
                    asm volatile("" :: "v"(v4.x), "v"(v4.y));
                    if (a.N == -12345) { reinterpret_cast<TV *>(yblk)[idx] = v4; }
#else
#if R512F_NT_STORE
                    {
                        typedef T vecnt __attribute__((ext_vector_type(VEC)));
                        __builtin_nontemporal_store(reinterpret_cast<const vecnt *>(xs)[idx], reinterpret_cast<vecnt *>(yblk) + idx);
                    }
#else
                    reinterpret_cast<TV *>(yblk)[idx] = reinterpret_cast<const TV *>(xs)[idx];
#endif
#endif
                }
            } else
#endif
            if (a.W - w >= 4) {
#pragma unroll
                for (int i = 0; i < 17; ++i) {
                    // the 17th instruction carries the last 4 values; its other lanes repeat value 1027
                    const int idx = (i < 16) ? lane + 64 * i : 1024 + (lane & 3);
#if R512F_ABLATE == 2
                    const T v = xs[idx];
                    if (a.N == -12345) { yblk[idx] = v; }
                    asm volatile("" :: "v"(v));
#else
                    R512F_ST(yblk + idx, xs[idx]);
#endif
                }
            } else {
                const int last = (a.W - w) * F - 1;
#pragma unroll
                for (int i = 0; i < 17; ++i) {
                    const int idx = min(lane + 64 * i, last);
                    R512F_ST(yblk + idx, xs[idx]);
                }
            }
        }
        R512F_STAMP(4);
        if (!more) { break; }
        wave_sync();                                       // tile reads are done before the next span lands

        // ---- commit the span that was in flight, step forward, start the next fetch ----
#pragma unroll
        for (int q = 0; q < NV; ++q) { reinterpret_cast<TV *>(xs)[q * 64 + lane] = pre[q]; }
        wave_sync();
        R512F_STAMP(5);
        sid = nsid; w = nw; xb = nxb; yb = nyb;
        advance();
        if constexpr (TBL) { xrow = nxrow; if (more) { tvn = tbl_fetch(nw); } }
        else { if (more) { load_span(nxb, nw * a.stp + a.ss0); } }
        R512F_STAMP(6);
    }
#if R512F_STAMPS
    if (lane == 0) {
        const int gw = blockIdx.x * 4 + wv;
        if (gw < 8192) { for (int i = 0; i < 8; ++i) { g_r512f_stamps[gw * 8 + i] = st_acc[i]; } }
    }
#endif
}
#if R512F_STAMPS
extern "C" int dsp_debug_read_stamps(unsigned long long *dst, size_t n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_r512f_stamps), n * sizeof(unsigned long long));
}
#endif

// Conditions of the fast path: frames at ss0 + w*hop with 3*hop + 512 <= 2048, or a table of frame starts
// that never decrease, are read where they start and put frames w and w + 3 at most 1536 samples apart;
// and every index of one signal's input and output fits 31 bits.
bool r512_fast_supported(const FrameDesc &d)
{
    const bool placed = d.tbl ? (d.tbl_reach >= 0 && d.tbl_reach + 512 <= 2048) : (3ll * d.stp + 512 <= 2048);
    return placed && d.nfft == 512 && d.L >= 1 && d.L <= 512 &&
           d.N < (1ll << 31) - 4096 && (long long)d.W * 257 < (1ll << 30) && (long long)d.B * ((d.W + 3) / 4) < (1ll << 30) && d.ss0 > -(1ll << 30) && d.ss0 < (1ll << 30) &&
           d.rd_shift >= 0 && d.rd_shift < (1ll << 30);
}

template <typename T, int NR, bool MN0, bool EPI, int WIDE, bool TBL = false>
static hipError_t launch_fast_inst(T *Y, const T *X, const T *win, const T *tw, const R512FastArgs &a, hipStream_t s)
{
    const size_t lds = (4 * (size_t)R512F_WAVE_ELEMS + 512 + 512) * sizeof(T);
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_r512_fast_kernel<T, NR, MN0, EPI, WIDE, TBL>), lds, grant); e != hipSuccess) { return e; }
    static const int wgs_per_cu = [] { const char *e = getenv("DSP_R512F_WGS_PER_CU"); return e ? atoi(e) : 2; }();
    long long grid = ((long long)a.steps + 3) / 4;
    if (grid > 256ll * wgs_per_cu) { grid = 256ll * wgs_per_cu; }
    grid = (grid + 7) & ~7ll;                           // whole groups of 8 workgroups (one per XCD)
    hipLaunchKernelGGL((stft_r512_fast_kernel<T, NR, MN0, EPI, WIDE, TBL>), dim3((unsigned)grid), dim3(256), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_stft_r512_fast(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    R512FastArgs a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + 3) / 4;
    a.steps = d.B * a.sps;
    a.tbl = d.tbl;
    if (d.tbl) { a.stp = d.tbl_reach; a.ss0 = 0; a.rd_shift = 0; }
    const bool epi = d.epi.any();
    const bool wide = 3 * d.stp + 512 > 1024 || (d.stp & 1);   // spans of more than 1024 samples (20 staged elements per lane), odd hops
    const bool nr13 = d.L > 384 && d.L <= 416 && !wide;
#define DSP_R512F(NR_, MN0_, EPI_) launch_fast_inst<T, NR_, MN0_, EPI_, 0>(Y, X, win, tw, a, s)
#define DSP_R512W(MN0_, EPI_) launch_fast_inst<T, 16, MN0_, EPI_, 1>(Y, X, win, tw, a, s)
#define DSP_R512X(MN0_, EPI_) launch_fast_inst<T, 16, MN0_, EPI_, 2>(Y, X, win, tw, a, s)
#define DSP_R512T(MN0_, EPI_) launch_fast_inst<T, 16, MN0_, EPI_, 1, true>(Y, X, win, tw, a, s)
#define DSP_R512U(MN0_, EPI_) launch_fast_inst<T, 16, MN0_, EPI_, 2, true>(Y, X, win, tw, a, s)
    if (d.tbl && d.tbl_reach + 512 > 1280) {
        if (d.mn0) { return epi ? DSP_R512U(true, true) : DSP_R512U(true, false); }
        return epi ? DSP_R512U(false, true) : DSP_R512U(false, false);
    }
    if (d.tbl) {
        if (d.mn0) { return epi ? DSP_R512T(true, true) : DSP_R512T(true, false); }
        return epi ? DSP_R512T(false, true) : DSP_R512T(false, false);
    }
    if (3 * d.stp + 512 > 1280) {
        if (d.mn0) { return epi ? DSP_R512X(true, true) : DSP_R512X(true, false); }
        return epi ? DSP_R512X(false, true) : DSP_R512X(false, false);
    }
    if (wide) {
        if (d.mn0) { return epi ? DSP_R512W(true, true) : DSP_R512W(true, false); }
        return epi ? DSP_R512W(false, true) : DSP_R512W(false, false);
    }
    if (nr13) {
        if (d.mn0) { return epi ? DSP_R512F(13, true, true) : DSP_R512F(13, true, false); }
        return epi ? DSP_R512F(13, false, true) : DSP_R512F(13, false, false);
    }
    if (d.mn0) { return epi ? DSP_R512F(16, true, true) : DSP_R512F(16, true, false); }
    return epi ? DSP_R512F(16, false, true) : DSP_R512F(16, false, false);
#undef DSP_R512F
#undef DSP_R512W
#undef DSP_R512X
#undef DSP_R512T
#undef DSP_R512U
}

template hipError_t launch_stft_r512_fast<float>(float *, const float *, const float *, const float *, const FrameDesc &, hipStream_t);
template hipError_t launch_stft_r512_fast<double>(double *, const double *, const double *, const double *, const FrameDesc &, hipStream_t);

}  // namespace dsp
