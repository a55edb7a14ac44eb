// Fast path of the 512-point fused STFT kernel for gfx950: integer hop with 3*hop + 512 <= 2048 (hop <= 512),
// frames addressed as start = ss0 + w*hop (the stft framing, and the stft_flt framing when
// its hop is integral), or (TBL) frames whose starts come from a table, never decrease and lie at most 1536
// samples apart for frames w and w + 3 (the stft_flt framing with a non-integral hop).  Same arithmetic as the
// basic kernel in r512_kernel.hip (16 lanes per frame, 4 frames = one step per wavefront); what differs is the
// schedule around it:
//   * one workgroup of 8 waves per CU (two per SIMD: the registers set that).  The steps of the batch are cut
//     into 8 contiguous ranges, one per group of workgroups that share an XCD, a workgroup owns 8 neighbouring
//     steps of every round of its range, and its waves DRAW those steps from a counter in LDS.  (The two
//     waves of a SIMD do not run at one pace, the older one issues first: dealt statically, one per
//     wave and round, the slower half of the waves ended 6 % after the faster half.)
//   * the span of the next step is fetched while the current one is transformed.  float, equally spaced
//     frames (RING): by LDS-DMA (global_load_lds_dwordx4) straight into a span slot of the wave's own, issued as
//     soon as pass 1 has read the slot; no registers, no LDS write pass, and the counted wait for it leaves the
//     step's row stores in flight.  Otherwise (double: no LDS left for slots; table form): into registers
//     (16-byte loads), committed to LDS at the top of the next step, where the span shares the exchange tile;
//   * rows leave as 16-byte non-temporal stores of the step's 4 x 257 values, staged through LDS in
//     the order they have in the output;
//   * the compile-time shape (rows of the window that hold data, mean removal, sqrt/log epilogue)
//     removes the per-element selects and branches of the general kernel.
#include "stft_common.h"
#include "fft16.h"
#include "span_ring.h"
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
    FastDiv spd;          // ... as a divisor
    int steps;            // B * sps
    const int *tbl;       // table form: start of frame w at tbl[2 w] (frames need not be equally spaced)
    int all16;            // every step has 4 frames and its block of rows is 16-byte aligned
    int Wy;               // frames per signal in the output array
    double cscale;        // complex form: factor on every output value (1, or 1 / sqrt(2 nfft))
    int gap;              // hops above 512: samples between the end of a frame and the start of the next (hop - 512) that are
                          // never read; the frames of a step then lie 512 apart in the staged span (0 otherwise)
};

constexpr int R512F_WAVE_ELEMS = 4 * R512_EXF * 2;   // T elements of a wave's exchange tile (>= 2048: the longest span)
#ifndef R512F_NWAVES
#define R512F_NWAVES 8          // waves per workgroup (4 to 8; a workgroup's round is 8 steps whatever their number)
#endif
constexpr int R512F_WAVES = R512F_NWAVES;

// value of lane (16 - i) % 16 of the same 16-lane row (row_mirror, then row_ror:1).  Every lane of the row has a
// source, so the "old" operand is never used: bound_ctrl tells the compiler so (otherwise it zeroes a register
// in front of every move: 32 more instructions per step)
__device__ __forceinline__ int dpp_partner16_i(int v)
{
    const int m = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    return __builtin_amdgcn_update_dpp(0, m, 0x121, 0xf, 0xf, true);
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
#ifndef R512F_SPLIT_PAIRS
#define R512F_SPLIT_PAIRS 1     // 1: the real-FFT split handles two bins at a time with packed norms (see there)
#endif
#ifndef R512F_GLDS_SPLIT
#define R512F_GLDS_SPLIT 1      // 1: the span's LDS-DMA loads take their addresses as scalar base + 32-bit lane offset (ring_glds16_at)
#endif
#define R512F_TW_LDS 0          // 1: pass-1 twiddles come from an LDS table instead of 30 registers
#endif
#ifndef R512F_WIN_LDS
#define R512F_WIN_LDS 0         // 1: float window taps come from LDS too (double always does)
#endif
#ifndef R512F_NT_STORE
#define R512F_NT_STORE 1        // 1: spectrogram rows are stored with the non-temporal hint
#endif
#if R512F_NT_STORE
#define R512F_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define R512F_ST(ptr, val) (*(ptr) = (val))
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
#ifndef R512F_TWD_AHEAD_F32
#define R512F_TWD_AHEAD_F32 4   // twiddles read from LDS ahead of their use (0: where they are used), float instantiations with an LDS table
#endif
#ifndef R512F_TWD_AHEAD_F64
#define R512F_TWD_AHEAD_F64 2   // the same for double (registers are short there)
#endif
#ifndef R512F_CYCLIC
#define R512F_CYCLIC 0          // 1: block-cyclic deal of the steps over the 8 groups of workgroups (measured: the same on the 1.7 GB benchmark batch, 2.5 % slower on 13.7 GB)
#endif
#ifndef R512F_NOLOAD
#define R512F_NOLOAD 0          // diagnostic builds: 1 = the span loads are compiled out
#endif
#ifndef R512F_NOSTORE
#define R512F_NOSTORE 0         // diagnostic builds: 1 = the row stores are compiled out
#endif
#ifndef R512F_ST4
#define R512F_ST4 1             // 1: rows leave as 16-byte vector stores when the step's block is 16-byte aligned
#endif
#ifndef R512F_RING_F32
#define R512F_RING_F32 1        // 1: float, equally spaced frames: the next span comes by LDS-DMA into a slot of its own
#endif
#ifndef R512F_TWLDS_SEL
#define R512F_TWLDS_SEL (WIDE == 2)   // float instantiations whose pass-1 twiddles come from LDS: 2048-sample spans (registers run out: measured +7 % from LDS, and 1.4x for the table form, which spilled)
#endif
#ifndef R512F_MINIMAL
#define R512F_MINIMAL 0         // diagnostic builds: 1 = only the benchmark's instantiation is compiled (seconds instead of minutes)
#endif


// CPLX: the rows are the complex spectra themselves (257 interleaved (re, im) pairs per frame, imag(DC) = imag(Nyquist) = 0:
// what window_univar followed by fft_fftw_r2hc gives, reference c/window_univar.c:41-89, c/fft.fftw.r2hc.c:69-78) instead
// of their powers; everything up to the split is the same.
// ILV: the signals are the channels of a [N][C] array (sample n of channel c at X[n pitch + c], pitch >= C, both
// multiples of 4 elements, X 16-byte aligned), read in place: a step is ONE frame of 4 adjacent channels, its span the
// frame's 512 16-byte groups (sample n of the 4 channels), fetched by LDS-DMA like any other span; the slot then holds
// [n][4] and lane (f, j) of pass 1 reads channel f.  The steps run with the channel groups fastest, so that the 8 waves
// of a workgroup take the groups that share the 128-byte lines of a frame.  Measured against gathering the channels
// into a dense batch first (64 / 32 / 16 / 8 / 4 channels x 2^22 samples, Hann 400, hop 160): 1.005 / 0.485 / 0.177 /
// 0.088 / 0.048 ms against 0.936 / 0.470 / 0.626 / 0.183 / 0.147 ms (and no temporary); with the frames of a channel
// group fastest instead, 64 channels took 1.73 ms.
template <typename T, int NR, bool MN0, bool EPI, int WIDE, bool TBL, bool RING, bool CPLX = false, bool ILV = false>
__global__ __launch_bounds__(64 * R512F_WAVES, 2) void stft_r512_fast_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                                             const T *__restrict__ win, const T *__restrict__ tw,
                                                                             const R512FastArgs a)
{
    using T2 = typename vec2<T>::type;
    using TV = typename vec16<T>::type;
    // staged elements per lane: 16 (spans of up to 1024 samples: hop <= 170), WIDE 1: 20 (1280: hop <= 256),
    // WIDE 2: 32 (2048: hop <= 512)
    constexpr int VEC = vec16<T>::n, SPAN = WIDE == 2 ? 2048 : WIDE ? 1280 : 1024, NV = SPAN / 64 / VEC;
    static_assert(SPAN <= R512F_WAVE_ELEMS, "the staged span lives in the wave's exchange region");
    constexpr int WAVE_LDS = R512F_WAVE_ELEMS + (RING ? SPAN : 0);
    constexpr bool WINREG = (sizeof(T) == 4) && !R512F_WIN_LDS;
    // pass-1 twiddles in registers, except where they do not fit beside the other state (double; spans of 2048
    // samples): there they are read from an LDS table
    constexpr bool TWREG = !R512F_TW_LDS && !(R512F_TW_LDS_F64 && sizeof(T) == 8) && !(R512F_TWLDS_SEL);
    constexpr int M = 256, F = 257, NWV = R512F_WAVES;
    constexpr int ROWF = CPLX ? 2 * F : F;                 // T elements of an output row
    constexpr int BLK = 4 * ROWF;                          // ... of a step's block of rows
    static_assert(BLK <= R512F_WAVE_ELEMS, "the staged rows of a step fit the wave's tile");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane >> 4, j = lane & 15;
    const int L = a.L;

    T *xs = reinterpret_cast<T *>(smem_raw) + wv * WAVE_LDS;           // exchange tile and row staging; !RING: the staged span too
    T2 *ex = reinterpret_cast<T2 *>(xs);
    T *slot = xs + R512F_WAVE_ELEMS;                                   // RING: the span slot
    T *wtab = reinterpret_cast<T *>(smem_raw) + NWV * WAVE_LDS;        // halved window, 512 T
    T2 *ttab = reinterpret_cast<T2 *>(wtab + 512);                     // W256^(j k1) at [k1][j], 256 T2
    T *dummy = reinterpret_cast<T *>(ttab + 256);                      // RING: 1 KiB that placeholder LDS-DMA loads write
    unsigned *ticket = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(dummy) + 1024);   // the next draw

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
    } else if (tid < 256) {
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
    constexpr int NRW = NR > 16 ? 16 : NR;                 // rows of 32 that hold data (NR = 17: all 16, and the window covers them: L = 512)
    C hw[NRW];                                             // halved window taps of this lane's columns
    if (WINREG) {
#pragma unroll
        for (int n1 = 0; n1 < NRW; ++n1) {
            const int l = 32 * n1 + 2 * j;
            hw[n1] = C{(l < L) ? T(0.5) * win[l] : T(0), (l + 1 < L) ? T(0.5) * win[l + 1] : T(0)};
        }
    } else {
        for (int l = tid; l < 512; l += 64 * NWV) { wtab[l] = (l < L) ? T(0.5) * win[l] : T(0); }
    }
    if (tid == 0) { *ticket = NWV; }                         // the first 8 steps go to the waves in order
    __syncthreads();
    static_assert(!TBL || WIDE >= 1, "the table form uses the wide instantiations");
    static_assert(!ILV || (WIDE == 2 && !TBL && RING && !CPLX && sizeof(T) == 4), "the interleaved form: float, 2048-element slot, LDS-DMA");
    const int lstp = a.stp - ((WIDE == 2 && !TBL) ? a.gap : 0);   // distance of a step's frames in the staged span
    const int reach = TBL ? a.stp : 3 * lstp;                // start of the step's last frame relative to its first
    const int need = reach + 512;
    const int nvec = (reach + 32 * NRW + VEC - 1) / VEC;      // 16-byte vectors of a span that pass 1 reads
    const T *xrow = ILV ? slot + 8 * j + f : (RING ? slot : xs) + f * lstp + 2 * j;   // (table form: set per step)
    // table form: lane i holds the start of frame w + (i & 3) of a step (clamped to the last frame)
    auto tbl_fetch = [&](const int w_) { return a.tbl[2 * min(w_ + (lane & 3), a.W - 1)]; };
    auto tbl_xrow = [&](const int tv) { return xs + (__shfl(tv, f) - __builtin_amdgcn_readfirstlane(tv)) + 2 * j; };

    // RING, table form: the starts of the 4 frames of the step at frame w_ (wave-uniform) by scalar loads -- a vector
    // load would put the compiler's own wait for ALL outstanding memory operations, the row stores included, in
    // front of the span fetch.  Sets xrow for that step and returns the start of its span.
    auto tbl_step = [&](const int w_) {
        const __attribute__((address_space(4))) int *tp = (const __attribute__((address_space(4))) int *)(uintptr_t)a.tbl;
        const int last = a.W - 1;
        const int t0 = tp[2 * w_], t1 = tp[2 * min(w_ + 1, last)], t2 = tp[2 * min(w_ + 2, last)], t3 = tp[2 * min(w_ + 3, last)];
        const int off = (f == 0) ? 0 : (f == 1) ? t1 - t0 : (f == 2) ? t2 - t0 : t3 - t0;
        xrow = slot + off + 2 * j;
        return t0;
    };

    // ---- fetch of one span (frames w..w+3 of the signal at xb): into registers, or (RING) into the slot.
    // A span that crosses an edge of its signal (reflection or zeros) is gathered sample by sample.
    TV pre[RING ? 1 : NV];
    // sample offset (from the start of the span) of staged element e: with a gap, frame e / 512 lies that many gaps further on
    auto span_off = [&](const int e) { return (WIDE == 2 && !TBL) ? e + (e >> 9) * a.gap : e; };
    auto gather = [&](const T *xb, const int s0, const int q, TV &out) {
        FrameDesc d{};
        d.N = a.N; d.edge = a.edge;
        T t[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int i = (q * 64 + lane) * VEC + v;
            t[v] = (i < need) ? fetch_sample(xb, d, (long long)s0, (long long)a.rd_shift, span_off(i)) : T(0);
        }
        if constexpr (VEC == 4) { out = TV{t[0], t[1], t[2], t[3]}; }
        else { out = TV{t[0], t[1]}; }
    };
    auto load_span = [&](const T *xb, const int s0) {
        if constexpr (ILV) {
            // sample n of the 4 channels is one 16-byte group at xb + n C: 8 loads of 64 groups, each from a line of its own
            const bool in_ilv = s0 >= 0 && s0 + 512 <= a.N;
            if (in_ilv) {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    ring_glds16(xb + (long long)(s0 + q * 64 + lane) * a.ldx, __builtin_amdgcn_readfirstlane(ring_lds_addr(slot + q * 64 * VEC)));
                }
            } else {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    long long n = (long long)s0 + q * 64 + lane;
                    while (n < 0 || n >= a.N) { n = (n < 0) ? -n - 1 : 2ll * a.N - 1 - n; }
                    reinterpret_cast<TV *>(slot)[q * 64 + lane] = *reinterpret_cast<const TV *>(xb + n * a.ldx);
                }
#pragma unroll
                for (int q = 0; q < NV; ++q) { ring_glds16(tw + lane * VEC, __builtin_amdgcn_readfirstlane(ring_lds_addr(dummy))); }
            }
            return;
        }
        const T *src = xb + (s0 - (a.edge ? a.rd_shift : 0));
        const bool inside = s0 >= 0 && s0 + SPAN + ((WIDE == 2 && !TBL) ? 3 * a.gap : 0) <= a.N && (a.edge == 0 || s0 >= a.rd_shift);
        if constexpr (RING) {
            [[maybe_unused]] const unsigned lds_slot = __builtin_amdgcn_readfirstlane(ring_lds_addr(slot));
            // NV LDS-DMA loads per span whatever the case (the counted wait relies on that): behind a gathered
            // span NV placeholder loads follow, from the twiddle table into `dummy`
            if (inside) {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    // only 3*hop + 32*NR samples are ever read by pass 1: lanes past that re-read the last needed vector
#if R512F_NOLOAD
                    ring_glds16(tw + lane * VEC, __builtin_amdgcn_readfirstlane(ring_lds_addr(dummy)));
#else
#if R512F_GLDS_SPLIT
                    ring_glds16_at(ring_uniform(src), (unsigned)sizeof(T) * (unsigned)span_off(min(q * 64 + lane, nvec - 1) * VEC), lds_slot, 1024 * q);
#else
                    ring_glds16(src + span_off(min(q * 64 + lane, nvec - 1) * VEC), __builtin_amdgcn_readfirstlane(ring_lds_addr(slot + q * 64 * VEC)));
#endif
#endif
                }
            } else {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    TV v;
                    gather(xb, s0, q, v);
                    reinterpret_cast<TV *>(slot)[q * 64 + lane] = v;
                }
#pragma unroll
                for (int q = 0; q < NV; ++q) { ring_glds16(tw + lane * VEC, __builtin_amdgcn_readfirstlane(ring_lds_addr(dummy))); }
            }
        } else if (inside && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (!(WIDE == 2 && !TBL) || (a.gap * (int)sizeof(T)) % 16 == 0)) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                // lanes past the last needed vector re-read it (same cache line, no extra traffic): the number of loads stays fixed
#if R512F_NOLOAD
                if (a.N == -12345)
#endif
                pre[RING ? 0 : q] = *reinterpret_cast<const TV *>(src + span_off(min(q * 64 + lane, nvec - 1) * VEC));
            }
        } else if (inside) {
            // a span that does not start on a 16-byte boundary (odd ss0, a hop that is no multiple of 4
            // elements, an unaligned signal): the same loads at element alignment
            using TVu = typename vec16u<T>::type;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const TVu v = *reinterpret_cast<const TVu *>(src + span_off(min(q * 64 + lane, nvec - 1) * VEC));
                if constexpr (VEC == 4) { pre[RING ? 0 : q] = TV{v.x, v.y, v.z, v.w}; }
                else { pre[RING ? 0 : q] = TV{v.x, v.y}; }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) { gather(xb, s0, q, pre[RING ? 0 : q]); }
        }
    };

    // Work distribution: the steps of the whole batch are cut into 8 contiguous ranges, one per group
    // of workgroups that share an XCD (blockIdx % 8); a round of a range is 8 steps for each of its workgroups,
    // neighbouring steps to one workgroup.  The waves of a range therefore read and write one moving window of
    // adjacent addresses (good DRAM page locality, and the samples two neighbouring steps share come from
    // the same L1 or L2) instead of thousands of separate streams.  Draw t of a workgroup is step t % 8 of
    // its round t / 8: the wave that is ready first takes it.
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
#if R512F_CYCLIC
    // (experiment: the groups take turns, block by block, through ONE moving window of the batch instead of 8 ranges)
    const int s_hi = a.steps;
    const int round = 8 * nwg_x * 8, first = xcd * nwg_x * 8 + (blockIdx.x >> 3) * 8;
#else
    const int s_lo = (int)((long long)a.steps * xcd / 8), s_hi = (int)((long long)a.steps * (xcd + 1) / 8);
    const int round = nwg_x * 8, first = s_lo + (blockIdx.x >> 3) * 8;
#endif
    int sid = first + (wv >> 3) * round + (wv & 7);      // draw number wv
    if (sid >= s_hi) { return; }
    // (ILV: sps is the number of channel groups, which run fastest; b is the group, w the frame)
    const int q0 = (int)a.spd.div((unsigned)sid);
    int b = ILV ? sid - q0 * a.sps : q0;
    int w = ILV ? q0 : (sid - b * a.sps) * 4;
    const T *xb = ILV ? X + 4 * b : X + (long long)b * a.ldx;
    T *yb = Y + (long long)(ILV ? 4 * b : b) * a.Wy * ROWF;

    // the wave's next step, from its latest draw
    const unsigned cnt_addr = ring_lds_addr(ticket);
    unsigned drawn = 0;
    int nsid, nw;
    const T *nxb;
    T *nyb;
    bool more;
    auto advance = [&]() {
        const int t = ring_drawn(drawn);
        nsid = first + (t >> 3) * round + (t & 7);
        more = nsid < s_hi;
        if (more) {
            const unsigned nq = a.spd.div((unsigned)nsid);                 // (unsigned: the products below stay short)
            const unsigned nb = ILV ? (unsigned)nsid - nq * (unsigned)a.sps : nq;
            nw = ILV ? (int)nq : (nsid - (int)nb * a.sps) * 4;
            nxb = ILV ? X + 4 * nb : X + (unsigned long long)nb * (unsigned long long)a.ldx;
            nyb = Y + (unsigned long long)(ILV ? 4 * nb : nb) * ((unsigned long long)a.Wy * ROWF);
        }
    };

#if R512F_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
#endif
    // prologue.  RING: the first span in flight.  Otherwise: first span to LDS, second span in flight (table
    // form: the starts of its frames in flight; its span is fetched in the middle of the step, when they
    // have arrived)
    int tvn = 0;
    bool first_step = true;
    if constexpr (RING) {
        if constexpr (TBL) { load_span(xb, tbl_step(w)); }
        else { load_span(xb, a.ss0 + w * a.stp); }
    } else {
        ring_draw(drawn, cnt_addr);
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
    }

    // Steady state per step: transform the staged span and store its rows while the next span is in
    // flight.  The stores of a step are unconditional and fixed in number, so the wait for the in-flight span
    // leaves them outstanding instead of draining the store queue.
    while (true) {
        ring_draw(drawn, cnt_addr);                       // the step after this one: asked for now, needed further down
        if constexpr (RING) {
            // The span of this step was asked for in the middle of the last one.  Memory operations complete in
            // issue order: behind its NV loads lie that step's row stores, so waiting until that many are
            // outstanding leaves the stores in flight.  (First step: nothing lies behind the span.)
            constexpr int K16 = BLK / VEC / 64 + 1;
            if (first_step) { ring_wait_vm<0>(); }
            else if (!ILV && a.all16) { ring_wait_vm<K16>(); }          // (ILV: the rows of a step go to 4 places, element by element)
            else { ring_wait_vm<BLK / 64 + 1>(); }
            first_step = false;
            wave_sync();
            R512F_STAMP(7);
        }
        // ---- pass 1: lane (frame f, column n2 = j): 16-point FFT over the rows n1 ----
        C zz[16];
        int lim = L - 2 * j;                              // element c of the pair in row n1 is inside the window iff 32 n1 + c < lim
        if constexpr (DSP_MASK_RECOMPUTE && NR == 16) { asm volatile("" : "+v"(lim)); }
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            if (n1 < NRW) {
                C v;
                if constexpr (ILV) {
                    v = C{xrow[128 * n1], xrow[128 * n1 + 4]};
                } else if constexpr (WIDE) {
                    // odd hops put the rows of frames 1 and 3 at odd element offsets: pair reads at element alignment
                    using T2e = typename vec2e<T>::type;
                    const T2e t = *reinterpret_cast<const T2e *>(xrow + 32 * n1);
                    v = C{t.x, t.y};
                } else {
                    const T2 t = *reinterpret_cast<const T2 *>(xrow + 32 * n1);
                    v = C{t.x, t.y};
                }
                if (NR == 16 || (NR < 16 && n1 == NR - 1)) {   // rows that may reach past the window
                    v.x = (32 * n1 < lim) ? v.x : T(0);
                    v.y = (32 * n1 + 1 < lim) ? v.y : T(0);
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
            for (int n1 = 1; n1 < NRW; ++n1) { acc += zz[n1]; }
            T sum = acc.x + acc.y;
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
            const T mean = sum / T(L);
#pragma unroll
            for (int n1 = 0; n1 < NRW; ++n1) {
                if (NR == 16 || (NR < 16 && n1 == NR - 1)) {   // rows that may reach past the window
                    zz[n1] -= C{(32 * n1 < lim) ? mean : T(0), (32 * n1 + 1 < lim) ? mean : T(0)};
                } else {
                    zz[n1] -= C{mean, mean};
                }
            }
        }
        dft16x<T, NRW>(zz);
        const T *nxrow = xrow;
        if constexpr (TBL && !RING) {
            if (more) { nxrow = tbl_xrow(tvn); load_span(nxb, __builtin_amdgcn_readfirstlane(tvn)); }
        }
        R512F_STAMP(0);
        wave_sync();                                       // all lanes have read the span: the tile may overwrite it
        if constexpr (RING) {
            // ... and the slot takes the next step's span.  The reads of pass 1 must have RETURNED before the
            // first load into the slot is issued: the compiler is free to sink the arithmetic that consumes them (and
            // with it its own wait) below the assembly statements, and a read that is still queued when the first
            // 1 KiB of the new span lands returns the new span's samples (seen as one wrong frame in a million).
            // ring_drawn waits for every LDS operation of the wave.  Past the end of the workgroup's share the
            // current span is fetched again: the number of loads per step stays fixed.
            advance();
            if constexpr (TBL) { load_span(more ? nxb : xb, tbl_step(more ? nw : w)); }      // (pass 1 is done with xrow)
            else { load_span(more ? nxb : xb, a.ss0 + (more ? nw : w) * a.stp); }
        }
        {
            T2 *exf = ex + f * R512_EXF + j;
            // Twiddles from the LDS table are read TWD twiddles ahead of their use.  (Read where it is used, each
            // follows the exchange write before it -- the compiler cannot tell the table from the tile -- and is
            // waited for on the spot: 15 LDS round trips in a row.)
            constexpr int TWD = TWREG ? 0 : (sizeof(T) == 4 ? R512F_TWD_AHEAD_F32 : R512F_TWD_AHEAD_F64);
            T2 tq[16];
            if constexpr (TWD > 0) {
#pragma unroll
                for (int k1 = 1; k1 <= (TWD < 15 ? TWD : 15); ++k1) { tq[k1] = ttab[k1 * 16 + j]; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                C v = zz[slot16(k1)];
                if constexpr (TWD > 0) {
                    if (k1 >= 1 && k1 + TWD < 16) { tq[k1 + TWD] = ttab[(k1 + TWD) * 16 + j]; __builtin_amdgcn_sched_barrier(0); }
                }
                if (k1 > 0) {
                    if (TWREG) { v = cx_mul<T>(v, twd[k1]); }
                    else if constexpr (TWD > 0) { v = cx_mul<T>(v, tq[k1].x, tq[k1].y); }
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
        dft16x<T, 16>(zz);
        R512F_STAMP(2);

        // ---- real-FFT split fused with power, in registers.  The lane holds Z[j + 16 k2] (halved);
        // its partner Z[256 - k] for k = j + 16 k2, k2 < 8, is register 15 - k2 of lane (16 - j) % 16
        // (register (16 - k2) % 16 of lane 0 itself when j == 0): one row_mirror + row_ror:1 DPP move.
        // With A = Z_k + conj Z_{256-k}, B = Z_k - conj Z_{256-k}, T = -i W512^k B:
        // |X_k|^2 = |A + T|^2 and |X_{256-k}|^2 = |A - T|^2.
        if constexpr (CPLX) {
            // X_k = A + T, X_{256-k} = conj(A - T); bin 128 is 2 conj Z_128 (lane j == 0), bins 0 and 256 come out real
            const T sc = (T)a.cscale;
            const C z8 = zz[slot16(8)];
            wave_sync();                                   // every lane has read its tile row (into zz): reuse the region
            T2 *pf = reinterpret_cast<T2 *>(xs) + f * F + j;       // 4 rows of 257 complex values, as they lie in Y
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const C z = zz[slot16(k2)];
                const C own = zz[slot16((16 - k2) & 15)], oth = zz[slot16(15 - k2)];
                const T px = dpp_partner16(oth.x), py = dpp_partner16(oth.y);
                const C m = C{(j == 0) ? own.x : px, (j == 0) ? own.y : py};
                const C A = cx_add_conj<T>(z, m), B = cx_sub_conj<T>(z, m);
                const C Tw = cx_mul<T>(B, ow[k2]);
                const C u = A + Tw, v = A - Tw;
                // (each value goes to the staging area as soon as it exists: held back, the 32 of them do not fit the registers)
                T2 o; o.x = sc * u.x; o.y = (j == 0 && k2 == 0) ? T(0) : sc * u.y;           // imag(DC) = 0
                pf[16 * k2] = o;
                T2 q; q.x = sc * v.x; q.y = (j == 0 && k2 == 0) ? T(0) : -sc * v.y;          // imag(Nyquist) = 0
                pf[M - 2 * j - 16 * k2] = q;
            }
            if (j == 0) { T2 o; o.x = T(2) * sc * z8.x; o.y = T(-2) * sc * z8.y; pf[128] = o; }
            wave_sync();
        } else {
            T pw[16];                                          // pw[k2] = P[k], pw[8 + k2] = P[256 - k]
            T p128 = T(4) * cx_norm<T>(zz[slot16(8)]);         // lane j == 0 only
            // (lane 0 is its own partner: the move hands it its own register 15 - k2, which the select replaces; a
            // select in front of the move would be turned into a select of the register INDEX, a 16-way chain)
            auto partner = [&](const int k2) __attribute__((always_inline)) {
                const C own = zz[slot16((16 - k2) & 15)], oth = zz[slot16(15 - k2)];
                const T px = dpp_partner16(oth.x), py = dpp_partner16(oth.y);
                return C{(j == 0) ? own.x : px, (j == 0) ? own.y : py};
            };
#if R512F_SPLIT_PAIRS
            // two bins at a time, stage by stage (a packed instruction that consumes the result of the one issued just before it
            // costs a wait state, and the stages of one bin are such a chain); (|A + T|^2, |A - T|^2) as one pair: the real parts
            // of both sums in one packed register, the imaginary parts in another (x * 1 + y is x + y exactly)
            const C pm = C{T(1), T(-1)};
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int ka = h, kb = h + 4;
                const C za = zz[slot16(ka)], zb = zz[slot16(kb)];
                const C ma = partner(ka), mb = partner(kb);
                const C Aa = cx_add_conj<T>(za, ma), Ab = cx_add_conj<T>(zb, mb);
                const C Ba = cx_sub_conj<T>(za, ma), Bb = cx_sub_conj<T>(zb, mb);
                const C ta = cx_mul_first<T>(Ba, ow[ka]), tb = cx_mul_first<T>(Bb, ow[kb]);
                const C Ta = cx_mul_second<T>(Ba, ow[ka], ta), Tb = cx_mul_second<T>(Bb, ow[kb], tb);
                const C rea = cx_fma(C{Ta.x, Ta.x}, pm, C{Aa.x, Aa.x}), reb = cx_fma(C{Tb.x, Tb.x}, pm, C{Ab.x, Ab.x});
                const C ima = cx_fma(C{Ta.y, Ta.y}, pm, C{Aa.y, Aa.y}), imb = cx_fma(C{Tb.y, Tb.y}, pm, C{Ab.y, Ab.y});
                const C qa = rea * rea, qb = reb * reb;
                const C sa = ima * ima, sb = imb * imb;
                const C pa = qa + sa, pb = qb + sb;
                pw[ka] = pa.x; pw[8 + ka] = pa.y;
                pw[kb] = pb.x; pw[8 + kb] = pb.y;
            }
#else
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const C z = zz[slot16(k2)];
                const C m = partner(k2);
                const C A = cx_add_conj<T>(z, m), B = cx_sub_conj<T>(z, m);
                const C Tw = cx_mul<T>(B, ow[k2]);
                pw[k2] = cx_norm<T>(A + Tw);
                pw[8 + k2] = cx_norm<T>(A - Tw);
            }
#endif
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
        }
        R512F_STAMP(3);

        // ---- store the rows of the step: 1028 consecutive values of Y (fewer at the end of a signal:
        // the index is clamped, so the surplus lanes rewrite the last value with itself and the number
        // of store instructions per step stays fixed)
        {
            T *yblk = yb + (unsigned)(w * ROWF);
            if constexpr (ILV) {
                // the 4 rows of the step belong to 4 channels: value idx of the staged block is bin idx % 257 of channel idx / 257
                constexpr int NS = BLK / 64;
#pragma unroll
                for (int i = 0; i <= NS; ++i) {
                    const int idx = min(lane + 64 * i, BLK - 1);
                    const int r = (idx >= ROWF) + (idx >= 2 * ROWF) + (idx >= 3 * ROWF);
                    R512F_ST(yblk + (long long)r * a.Wy * ROWF + (idx - r * ROWF), xs[idx]);
                }
            } else
#if R512F_ST4
            if (RING ? a.all16 != 0 : (a.W - w >= 4 && (reinterpret_cast<uintptr_t>(yblk) & 15) == 0)) {
                // 1028 values = 257 16-byte vectors (float) / 514 (double), twice that for complex rows: whole-vector wave stores
                constexpr int NVEC = BLK / VEC, FULLI = NVEC / 64;
                static_assert(((NVEC - FULLI * 64) & (NVEC - FULLI * 64 - 1)) == 0, "the last store's lanes wrap by a mask");
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
                constexpr int NS = BLK / 64, TAIL = BLK - NS * 64;     // 16 + 4 values (complex rows: 32 + 8)
#pragma unroll
                for (int i = 0; i <= NS; ++i) {
                    // the last instruction carries the last TAIL values; its other lanes repeat them
                    const int idx = (i < NS) ? lane + 64 * i : NS * 64 + (lane & (TAIL - 1));
                    R512F_ST(yblk + idx, xs[idx]);
                }
            } else {
                const int last = (a.W - w) * ROWF - 1;
#pragma unroll
                for (int i = 0; i <= BLK / 64; ++i) {
                    const int idx = min(lane + 64 * i, last);
                    R512F_ST(yblk + idx, xs[idx]);
                }
            }
        }
        R512F_STAMP(4);
        if (!more) { break; }
        wave_sync();                                       // tile reads are done before the next step writes it

        if constexpr (RING) {
            sid = nsid; w = nw; xb = nxb; yb = nyb;
        } else {
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
    }
    // no LDS-DMA load may land, and no draw return, after the wave has ended (the LDS would belong to another
    // workgroup by then)
    if constexpr (RING) { ring_wait_vm<0>(); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    (void)dummy;
#if R512F_STAMPS
    if (lane == 0) {
        const int gw = blockIdx.x * NWV + wv;
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

#ifndef R512F_ROWS_TU
#define R512F_ROWS_TU 0         // 1: this file compiled as r512_fast_rows.hip, the translation unit of the row-count instantiations
#endif
#if !R512F_ROWS_TU
// Conditions of the fast path: frames at ss0 + w*hop with 3*hop + 512 <= 2048, or a table of frame starts
// that never decrease, are read where they start and put frames w and w + 3 at most 1536 samples apart;
// and every index of one signal's input and output fits 31 bits.
bool r512_fast_supported(const FrameDesc &d)
{
    // equally spaced frames at any hop: above 512 the samples between two frames are skipped when a span is fetched
    const bool placed = d.tbl ? (d.tbl_reach >= 0 && d.tbl_reach + 512 <= 2048) : (d.stp >= 1 && d.stp < (1 << 28));
    return placed && d.nfft == 512 && d.L >= 1 && d.L <= 512 &&
           d.N < (1ll << 31) - 4096 && (long long)d.wy() * 257 < (1ll << 30) && (long long)d.B * ((d.W + 3) / 4) < (1ll << 30) && d.ss0 > -(1ll << 30) && d.ss0 < (1ll << 30) &&
           d.rd_shift >= 0 && d.rd_shift < (1ll << 30);
}

#endif
template <typename T, int NR, bool MN0, bool EPI, int WIDE, bool TBL, bool RING, bool CPLX = false, bool ILV = false>
static hipError_t launch_fast_inst(T *Y, const T *X, const T *win, const T *tw, const R512FastArgs &a, hipStream_t s)
{
    constexpr int SPAN = WIDE == 2 ? 2048 : WIDE ? 1280 : 1024;
    const size_t lds = (R512F_WAVES * ((size_t)R512F_WAVE_ELEMS + (RING ? SPAN : 0)) + 512 + 512) * sizeof(T) + 1024 + 16;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_r512_fast_kernel<T, NR, MN0, EPI, WIDE, TBL, RING, CPLX, ILV>), lds, grant); e != hipSuccess) { return e; }
    long long grid = ((long long)a.steps + 7) / 8;
    if (grid > 256) { grid = 256; }                     // one 8-wave workgroup per CU
    grid = (grid + 7) & ~7ll;                           // whole groups of 8 workgroups (one per XCD)
    hipLaunchKernelGGL((stft_r512_fast_kernel<T, NR, MN0, EPI, WIDE, TBL, RING, CPLX, ILV>), dim3((unsigned)grid), dim3(64 * R512F_WAVES), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

// Windows of 8 to 12, 14 and 15 rows of 32 samples (225 to 384 and 417 to 480 samples; 320 = 20 ms and 480 = 30 ms at 16 kHz): only the last row of
// a lane that holds data is masked, and the first stage of the 16-point transform skips the empty rows.  Windows of 512 samples (rows = 17):
// no row is masked.  These instantiations live in a
// translation unit of their own (r512_fast_rows.hip: this file again with R512F_ROWS_TU = 1), hence in a code object of their own, which
// the runtime brings to the device when one of them is first launched: with them in this one the first call of a process took 38 ms
// instead of 24.  (8, 10, 15 and 17 with every flag; the counts in between without mean removal: they return hipErrorNotSupported then.)
hipError_t launch_stft_r512_fast_rows(float *Y, const float *X, const float *win, const float *tw, const R512FastArgs &a, int rows, bool wide, bool mn0, bool epi, hipStream_t s);
hipError_t launch_stft_r512_fast_rows(double *Y, const double *X, const double *win, const double *tw, const R512FastArgs &a, int rows, bool wide, bool mn0, bool epi, hipStream_t s);
constexpr bool r512_fast_has_rows(int rows, bool mn0) { return rows == 8 || rows == 10 || rows == 15 || rows == 17 || (!mn0 && (rows == 9 || rows == 11 || rows == 12 || rows == 14)); }

#if R512F_ROWS_TU
template <typename T>
static hipError_t launch_rows_any(T *Y, const T *X, const T *win, const T *tw, const R512FastArgs &a, int rows, bool wide, bool mn0, bool epi, hipStream_t s)
{
    constexpr bool RING = sizeof(T) == 4 && R512F_RING_F32;
#define DSP_R512F(NR_, MN0_, EPI_) launch_fast_inst<T, NR_, MN0_, EPI_, 0, false, RING>(Y, X, win, tw, a, s)
#define DSP_R512W(NR_, MN0_, EPI_) launch_fast_inst<T, NR_, MN0_, EPI_, 1, false, RING>(Y, X, win, tw, a, s)
#define DSP_R512ROWS(NR_)                                                                                       \
    if (rows == NR_ && wide) {                                                                                  \
        if (mn0) { return epi ? DSP_R512W(NR_, true, true) : DSP_R512W(NR_, true, false); }                     \
        return epi ? DSP_R512W(NR_, false, true) : DSP_R512W(NR_, false, false);                                \
    }                                                                                                           \
    if (rows == NR_) {                                                                                          \
        if (mn0) { return epi ? DSP_R512F(NR_, true, true) : DSP_R512F(NR_, true, false); }                     \
        return epi ? DSP_R512F(NR_, false, true) : DSP_R512F(NR_, false, false);                                \
    }
    DSP_R512ROWS(8)
    DSP_R512ROWS(10)
    DSP_R512ROWS(15)
    DSP_R512ROWS(17)
#undef DSP_R512ROWS
#define DSP_R512ROWS(NR_)                                                                                       \
    if (rows == NR_ && !mn0) {                                                                                  \
        if (wide) { return epi ? DSP_R512W(NR_, false, true) : DSP_R512W(NR_, false, false); }                  \
        return epi ? DSP_R512F(NR_, false, true) : DSP_R512F(NR_, false, false);                                \
    }
    DSP_R512ROWS(9)
    DSP_R512ROWS(11)
    DSP_R512ROWS(12)
    DSP_R512ROWS(14)
#undef DSP_R512ROWS
#undef DSP_R512F
#undef DSP_R512W
    return hipErrorNotSupported;
}
hipError_t launch_stft_r512_fast_rows(float *Y, const float *X, const float *win, const float *tw, const R512FastArgs &a, int rows, bool wide, bool mn0, bool epi, hipStream_t s)
{
    return launch_rows_any<float>(Y, X, win, tw, a, rows, wide, mn0, epi, s);
}
hipError_t launch_stft_r512_fast_rows(double *Y, const double *X, const double *win, const double *tw, const R512FastArgs &a, int rows, bool wide, bool mn0, bool epi, hipStream_t s)
{
    return launch_rows_any<double>(Y, X, win, tw, a, rows, wide, mn0, epi, s);
}
#else
template <typename T>
static hipError_t launch_stft_r512_fast_one(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s);

// Very large batches go as several launches of whole signals, one behind the other in the stream: measured on 64 float64
// channels of one hour (77 GB of input and output) one launch takes 15.4 ms, launches of 8 channels 14.6 ms and of 4
// channels 14.4 ms.  Below R512F_SPLIT_ABOVE bytes of input and output a launch stays whole: 8192 ten-second utterances
// (13.7 GB) measured the same either way (2.62 ms), the 1.7 GB benchmark batch loses 3 % when halved.
#ifndef R512F_SPLIT_ABOVE
#define R512F_SPLIT_ABOVE (16ull << 30)
#endif
#ifndef R512F_SPLIT_BYTES
#define R512F_SPLIT_BYTES (5ull << 30)
#endif
template <typename T>
hipError_t launch_stft_r512_fast(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    const unsigned long long per_signal = ((unsigned long long)d.N + (unsigned long long)d.W * 257ull) * sizeof(T);
    const unsigned long long total = per_signal * (unsigned long long)d.B;
    if (d.B < 2 || total <= R512F_SPLIT_ABOVE) { return launch_stft_r512_fast_one<T>(Y, X, win, tw, d, s); }
    unsigned long long nb = R512F_SPLIT_BYTES / per_signal;
    if (nb < 1) { nb = 1; }
    for (long long b0 = 0; b0 < d.B; b0 += (long long)nb) {
        FrameDesc part = d;
        part.B = (int)(((long long)nb < d.B - b0) ? (long long)nb : d.B - b0);
        if (hipError_t e = launch_stft_r512_fast_one<T>(Y + (size_t)b0 * d.wy() * 257, X + (size_t)b0 * d.ldx, win, tw, part, s); e != hipSuccess) { return e; }
    }
    return hipSuccess;
}

template <typename T>
static hipError_t launch_stft_r512_fast_one(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    R512FastArgs a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + 3) / 4;
    a.spd = FastDiv::of((unsigned)a.sps);
    a.steps = d.B * a.sps;
    a.tbl = d.tbl;
    a.gap = (!d.tbl && d.stp > 512) ? d.stp - 512 : 0;
    if (d.tbl) { a.stp = d.tbl_reach; a.ss0 = 0; a.rd_shift = 0; }
    a.all16 = (d.W % 4 == 0) && (d.wy() % 4 == 0) && (reinterpret_cast<uintptr_t>(Y) % 16 == 0);
    a.Wy = d.wy();
    const bool epi = d.epi.any();
    // float, equally spaced frames: spans by LDS-DMA (the double kernel's exchange tiles leave no LDS for span slots)
    constexpr bool RING = sizeof(T) == 4 && R512F_RING_F32;
    [[maybe_unused]] const bool wide = 3 * d.stp + 512 > 1024 || (d.stp & 1);   // spans of more than 1024 samples (20 staged elements per lane), odd hops
    const bool nr13 = d.L > 384 && d.L <= 416;               // windows that leave the last three rows of pass 1 empty (400: 25 ms at 16 kHz)
#define DSP_R512F(NR_, MN0_, EPI_) launch_fast_inst<T, NR_, MN0_, EPI_, 0, false, RING>(Y, X, win, tw, a, s)
#define DSP_R512W(NR_, MN0_, EPI_) launch_fast_inst<T, NR_, MN0_, EPI_, 1, false, RING>(Y, X, win, tw, a, s)
#define DSP_R512X(MN0_, EPI_) launch_fast_inst<T, 16, MN0_, EPI_, 2, false, RING>(Y, X, win, tw, a, s)
// table form: spans of up to 1280 samples through registers (by LDS-DMA they measured 44.5 % of the HBM peak against
// 50.1 %: the frames of such a table start at odd and even offsets alike, the unaligned LDS reads of pass 1 bound
// it, and the DMA form spills), longer ones by LDS-DMA (61.0 % against 59.3 %)
#define DSP_R512T(NR_, MN0_, EPI_) launch_fast_inst<T, NR_, MN0_, EPI_, 1, true, false>(Y, X, win, tw, a, s)
#define DSP_R512U(MN0_, EPI_) launch_fast_inst<T, 16, MN0_, EPI_, 2, true, RING>(Y, X, win, tw, a, s)
#if R512F_MINIMAL
    if constexpr (sizeof(T) == 4) { if (nr13 && !d.mn0 && !epi && !d.tbl) { return DSP_R512F(13, false, false); } }
    return hipErrorInvalidValue;
#else
    if (d.tbl && d.tbl_reach + 512 > 1280) {
        if (d.mn0) { return epi ? DSP_R512U(true, true) : DSP_R512U(true, false); }
        return epi ? DSP_R512U(false, true) : DSP_R512U(false, false);
    }
    if (d.tbl && nr13) {
        if (d.mn0) { return epi ? DSP_R512T(13, true, true) : DSP_R512T(13, true, false); }
        return epi ? DSP_R512T(13, false, true) : DSP_R512T(13, false, false);
    }
    if (d.tbl) {
        if (d.mn0) { return epi ? DSP_R512T(16, true, true) : DSP_R512T(16, true, false); }
        return epi ? DSP_R512T(16, false, true) : DSP_R512T(16, false, false);
    }
    if (3 * d.stp + 512 > 1280) {
        if (d.mn0) { return epi ? DSP_R512X(true, true) : DSP_R512X(true, false); }
        return epi ? DSP_R512X(false, true) : DSP_R512X(false, false);
    }
    // other windows whose last rows are empty, and L = 512, whose rows are never masked (17 stands for that: the 32 lane masks of the general
    // form live in scalar registers that spill into the lanes of a vector register, about 90 instructions per step): the instantiations of
    // r512_fast_rows.hip
    const int rows = d.L == 512 ? 17 : (d.L + 31) / 32;
    if (r512_fast_has_rows(rows, d.mn0)) { return launch_stft_r512_fast_rows(Y, X, win, tw, a, rows, wide, d.mn0, epi, s); }
    if (wide && nr13) {
        if (d.mn0) { return epi ? DSP_R512W(13, true, true) : DSP_R512W(13, true, false); }
        return epi ? DSP_R512W(13, false, true) : DSP_R512W(13, false, false);
    }
    if (wide) {
        if (d.mn0) { return epi ? DSP_R512W(16, true, true) : DSP_R512W(16, true, false); }
        return epi ? DSP_R512W(16, false, true) : DSP_R512W(16, false, false);
    }
    if (nr13) {
        if (d.mn0) { return epi ? DSP_R512F(13, true, true) : DSP_R512F(13, true, false); }
        return epi ? DSP_R512F(13, false, true) : DSP_R512F(13, false, false);
    }
    if (d.mn0) { return epi ? DSP_R512F(16, true, true) : DSP_R512F(16, true, false); }
    return epi ? DSP_R512F(16, false, true) : DSP_R512F(16, false, false);
#endif
#undef DSP_R512F
#undef DSP_R512W
#undef DSP_R512X
#undef DSP_R512T
#undef DSP_R512U
}

// The interleaved form (float32, reflected edges): d.B channels of a [N][pitch] array, read in place.
bool r512_fast_ilv_supported(const FrameDesc &d, const float *X, size_t pitch)
{
#if R512F_MINIMAL || !R512F_RING_F32
    return false;
#else
    return r512_fast_supported(d) && d.tbl == nullptr && d.edge == 0 && d.B % 4 == 0 && pitch % 4 == 0 && pitch >= (size_t)d.B &&
           reinterpret_cast<uintptr_t>(X) % 16 == 0 && (long long)d.N * (long long)pitch < (1ll << 40) &&
           (long long)(d.B / 4) * d.W < (1ll << 30) && (long long)d.B * d.wy() * 257 < (1ll << 40);
#endif
}

hipError_t launch_stft_r512_fast_ilv(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, size_t pitch, hipStream_t s)
{
#if R512F_MINIMAL || !R512F_RING_F32
    return hipErrorInvalidValue;
#else
    R512FastArgs a{};
    a.ldx = (long long)pitch; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = 0; a.edge = 0; a.epi = d.epi;
    a.sps = d.B / 4;                                       // channel groups
    a.spd = FastDiv::of((unsigned)a.sps);
    a.steps = a.sps * d.W;
    a.Wy = d.wy();
    const bool epi = d.epi.any();
#define DSP_R512I(MN0_, EPI_) launch_fast_inst<float, 16, MN0_, EPI_, 2, false, true, false, true>(Y, X, win, tw, a, s)
    if (d.mn0) { return epi ? DSP_R512I(true, true) : DSP_R512I(true, false); }
    return epi ? DSP_R512I(false, true) : DSP_R512I(false, false);
#undef DSP_R512I
#endif
}

// The complex form: Y[b][w][257] interleaved (re, im) pairs times `scale`; equally spaced frames, no mean removal.
bool r512_fast_cplx_supported(const FrameDesc &d)
{
    return r512_fast_supported(d) && d.tbl == nullptr && !d.mn0 && !d.epi.any() && (long long)d.wy() * 514 < (1ll << 30);
}

template <typename T>
hipError_t launch_stft_r512_fast_cplx(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, double scale, hipStream_t s)
{
#if R512F_MINIMAL
    return hipErrorInvalidValue;
#else
    R512FastArgs a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + 3) / 4;
    a.spd = FastDiv::of((unsigned)a.sps);
    a.steps = d.B * a.sps;
    a.tbl = nullptr;
    a.gap = d.stp > 512 ? d.stp - 512 : 0;
    a.cscale = scale;
    a.all16 = (d.W % 4 == 0) && (d.wy() % 4 == 0) && (reinterpret_cast<uintptr_t>(Y) % 16 == 0);
    a.Wy = d.wy();
    constexpr bool RING = sizeof(T) == 4 && R512F_RING_F32;
    [[maybe_unused]] const bool wide = 3 * d.stp + 512 > 1024 || (d.stp & 1);
    const bool nr13 = d.L > 384 && d.L <= 416;
    if (3 * d.stp + 512 > 1280) { return launch_fast_inst<T, 16, false, false, 2, false, RING, true>(Y, X, win, tw, a, s); }
    if (wide) {
        return nr13 ? launch_fast_inst<T, 13, false, false, 1, false, RING, true>(Y, X, win, tw, a, s)
                    : launch_fast_inst<T, 16, false, false, 1, false, RING, true>(Y, X, win, tw, a, s);
    }
    return nr13 ? launch_fast_inst<T, 13, false, false, 0, false, RING, true>(Y, X, win, tw, a, s)
                : launch_fast_inst<T, 16, false, false, 0, false, RING, true>(Y, X, win, tw, a, s);
#endif
}
template hipError_t launch_stft_r512_fast_cplx<float>(float *, const float *, const float *, const float *, const FrameDesc &, double, hipStream_t);
template hipError_t launch_stft_r512_fast_cplx<double>(double *, const double *, const double *, const double *, const FrameDesc &, double, hipStream_t);

template hipError_t launch_stft_r512_fast<float>(float *, const float *, const float *, const float *, const FrameDesc &, hipStream_t);
template hipError_t launch_stft_r512_fast<double>(double *, const double *, const double *, const double *, const FrameDesc &, hipStream_t);
#endif   // !R512F_ROWS_TU

}  // namespace dsp
