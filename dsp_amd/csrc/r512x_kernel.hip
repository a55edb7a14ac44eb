// Fused STFT kernel for nfft = 1024 and 2048, float32 and float64 (gfx950): the 256-point complex
// engine of the 512-point kernel (16 lanes, 16 values each: a 16 x 16 four-step transform with one exchange
// through LDS) under one decimation-in-frequency stage of radix Q = nfft / 512.
//
// A frame's nfft reals are M = 256 Q complex points z_n = x_2n + i x_2n+1.  With n = m + 256 q:
//     Z[Q k + r] = sum_m W256^(m k) * ( W_M^(m r) * sum_q W_Q^(q r) z_(m + 256 q) ),
// i.e. Q independent 256-point transforms, one per residue r of the bin index, each on 16 lanes (a "group";
// Q groups per frame, 4 / Q frames per wavefront and step).  The radix-Q stage is done where the samples are
// read: group g reads the rows n1 = Q i + g of the 16 x 16 arrangement m = j + 16 n1 in all Q quarters of the
// windowed frame (16 LDS reads per lane, window taps in registers), runs the Q-point butterflies of those
// rows and multiplies by W_M^(m r); then the Q groups trade results with v_permlane16_swap (and
// v_permlane32_swap for Q = 4) so that each holds ONE residue, r = Q - 1 - g, for all 16 rows.  On the way out
// lane (r, j) holds Z[Q (j + 16 k2) + r]; the partner of bin K in the real-FFT split, Z[M - K], lies in the
// group of residue (Q - r) % Q at the mirrored lane: a DPP row move for r = 0 and r = Q / 2, one swap of the
// wave's halves more for the odd residues of Q = 4.
//
// Around the arithmetic, the schedule of the 512-point kernel: one workgroup of 8 waves per CU whose waves
// draw steps from a counter in LDS; the span of the next step comes by LDS-DMA into a slot of the wave's own
// while the current one is transformed; rows are staged through LDS in output order and leave as 16-byte
// non-temporal stores.  A step's block of rows (1026 or 1025 values) starts at any multiple of 4 bytes, so
// the staging is shifted to put the 16-byte boundaries of the OUTPUT on 16-byte boundaries of the LDS, the
// body leaves as four vector stores and the 0 to 3 values before and after it as one more instruction: five
// stores per step whatever the alignment, which keeps the counted wait for the next span exact.
// float64: the same source with 4 waves per workgroup (twice the bytes per wave in the LDS; one wave per SIMD has the
// registers for every per-lane table in double), lane moves as two 32-bit moves per value, rows stored element-wise (17
// store instructions per step).
#include "stft_common.h"
#include "fft16.h"
#include "span_ring.h"

namespace dsp {

struct R512XArgs {
    long long ldx;        // elements between signals
    int N;                // samples per signal
    int L;                // window length
    int W;                // frames per signal
    int Wy;               // frames per signal in the output array
    int stp;              // hop
    int ss0;              // start of frame 0
    int rd_shift;         // edge == 1: samples are read at n - rd_shift
    int edge;             // 0 reflect, 1 zero
    Epilogue epi;
    int sps;              // steps per signal
    FastDiv spd;          // ... as a divisor
    int steps;            // B * sps
    const int *tbl;       // table form: start of frame w at tbl[2 w] (frames need not be equally spaced); else nullptr
    int reach;            // table form: bound on the distance between the starts of the first and last frame of a step
    double cscale;        // complex form: factor on every output value
};

constexpr int R512X_TILE = 4 * R512_EXF * 2;          // floats of a wave's exchange tile / row staging
template <typename T> __host__ __device__ constexpr int r512x_waves() { return sizeof(T) == 4 ? 8 : 4; }   // per workgroup

#ifndef R512X_TWD_LDS
#define R512X_TWD_LDS (sizeof(T) == 4 && (Q == 4 || (SPAN == 2048 && !RQ)))   // where the pass-1 twiddles of the 256-point transform come from an LDS table instead of 30 registers (these instantiations spill otherwise; measured: 1024 / hop 512 0.480 -> 0.449 ms, and 6 % slower where nothing spills)
#endif
#ifndef R512X_ZERO_TAIL
#define R512X_ZERO_TAIL 1       // 1: nfft 2048 (one frame per step, the slot its own) with a window shorter than the transform reads the slot
                                // unmasked: what lies past the window in the slot is made harmless instead (see UNMASKED in the kernel)
#endif
#ifndef R512X_WIN_LDS
#define R512X_WIN_LDS (sizeof(T) == 8 && Q == 2)   // the halved window taps of the input stage come from an LDS table instead of 16 register pairs: in float64
                                         // they are 64 registers of the 300 the kernel wants, and what does not fit the 256 vector registers lives in
                                         // accumulation registers at one copy per use (93 copies per step of the 1024-point kernel; nfft 1024 float64
                                         // 0.926 -> 0.898 ms; with the input stage's twiddles in the table as well 0.913; nfft 2048, Q = 4, measured 0.7 % slower with
                                         // it and keeps its registers)
#endif
#ifndef R512X_TWD_AHEAD
#define R512X_TWD_AHEAD 1       // 1: the LDS twiddles of a step are read in one go before the pass-1 transform
#endif
#ifndef R512X_LOAD_SPREAD
#define R512X_LOAD_SPREAD 0     // where the LDS-DMA loads of the next span are issued: 0 all behind pass 1's transform, 1 one after every second
                                // exchange write, 2 all behind the exchange writes, 3 all behind pass 2's transform, 4 one per pair of the split (power form)
#endif
#ifndef R512X_NOLOAD
#define R512X_NOLOAD 0          // diagnostic builds: 1 = the span loads fetch the twiddle table instead (no HBM reads)
#endif
#ifndef R512X_NOSTORE
#define R512X_NOSTORE 0         // diagnostic builds: 1 = the row stores are compiled out
#endif
#ifndef R512X_STAMPS
#define R512X_STAMPS 0          // diagnostic builds: 1 = per-phase s_memtime sums into g_r512x_stamps
#endif
#if R512X_STAMPS
__device__ unsigned long long g_r512x_stamps[2048 * 8];
#define R512X_STAMP(i)                                                                            \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        st_acc[i] += t_ - st_last; st_last = t_;                                                  \
    } while (0)
#else
#define R512X_STAMP(i) do {} while (0)
#endif
#ifndef R512X_MINIMAL
#define R512X_MINIMAL 0         // diagnostic builds: 1 = only the plain full-window instantiations
#endif

__device__ __forceinline__ float r512x_mirror(float v)       // lane 15 - i of the same row of 16
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
}
__device__ __forceinline__ float r512x_ror1(float v)         // lane (i - 1) % 16 of the same row
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, true));
}
// ... in the rows of 16 lanes that ROWS names (bit i = row i of the wave) only; the other rows keep their own value
template <int ROWS> __device__ __forceinline__ float r512x_ror1_rows(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x121, ROWS, 0xf, false));
}
template <int ROWS> __device__ __forceinline__ double r512x_ror1_rows(double v)
{
    return __hiloint2double(__float_as_int(r512x_ror1_rows<ROWS>(__int_as_float(__double2hiint(v)))), __float_as_int(r512x_ror1_rows<ROWS>(__int_as_float(__double2loint(v)))));
}
typedef unsigned r512x_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float r512x_xor32(float v, bool low_half)   // lane i ^ 32
{
    const r512x_u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(low_half ? t.y : t.x);
}

// any lane's value in one instruction: ds_bpermute_b32 (through the LDS crossbar, no LDS memory); addr = 4 * source lane
__device__ __forceinline__ float r512x_from_lane(float v, int addr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
__device__ __forceinline__ double r512x_from_lane(double v, int addr)
{
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}
#ifndef R512X_DIRECT_STORE
#define R512X_DIRECT_STORE 1    // 1: float64, power form: rows stored straight from the registers
#endif
#ifndef R512X_BPERM_AHEAD
#define R512X_BPERM_AHEAD 1     // 1: (power form) the 16 ds_bpermute of a step are issued before the first result is used
#endif
#ifndef R512X_BPERM
#define R512X_BPERM (Q == 4)    // the split fetches its partner values with one ds_bpermute each instead of DPP moves, half-wave swaps and selects
                                // (measured: nfft 2048 -2.4 % float32, -6.6 % float64; nfft 1024, whose partners need no half-wave swap, +4 % / +1.5 %)
#endif

// v_permlane16_swap: the odd rows of 16 lanes of a trade places with the even rows of b;
// v_permlane32_swap: the upper half of a with the lower half of b
__device__ __forceinline__ void r512x_swap16(cx<float> &a, cx<float> &b)
{
    const r512x_u2 tx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const r512x_u2 ty = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = cx<float>{__uint_as_float(tx.x), __uint_as_float(ty.x)};
    b = cx<float>{__uint_as_float(tx.y), __uint_as_float(ty.y)};
}
__device__ __forceinline__ void r512x_swap32(cx<float> &a, cx<float> &b)
{
    const r512x_u2 tx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const r512x_u2 ty = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = cx<float>{__uint_as_float(tx.x), __uint_as_float(ty.x)};
    b = cx<float>{__uint_as_float(tx.y), __uint_as_float(ty.y)};
}

// the same moves on doubles: the two 32-bit halves separately
__device__ __forceinline__ double r512x_mirror(double v)
{
    return __hiloint2double(__float_as_int(r512x_mirror(__int_as_float(__double2hiint(v)))), __float_as_int(r512x_mirror(__int_as_float(__double2loint(v)))));
}
__device__ __forceinline__ double r512x_ror1(double v)
{
    return __hiloint2double(__float_as_int(r512x_ror1(__int_as_float(__double2hiint(v)))), __float_as_int(r512x_ror1(__int_as_float(__double2loint(v)))));
}
__device__ __forceinline__ double r512x_xor32(double v, bool low_half)
{
    return __hiloint2double(__float_as_int(r512x_xor32(__int_as_float(__double2hiint(v)), low_half)), __float_as_int(r512x_xor32(__int_as_float(__double2loint(v)), low_half)));
}
__device__ __forceinline__ void r512x_swap16(double &a, double &b)
{
    const r512x_u2 tl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const r512x_u2 th = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)th.x, (int)tl.x);
    b = __hiloint2double((int)th.y, (int)tl.y);
}
__device__ __forceinline__ void r512x_swap32(double &a, double &b)
{
    const r512x_u2 tl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const r512x_u2 th = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)th.x, (int)tl.x);
    b = __hiloint2double((int)th.y, (int)tl.y);
}
__device__ __forceinline__ void r512x_swap16(cx<double> &a, cx<double> &b) { r512x_swap16(a.x, b.x); r512x_swap16(a.y, b.y); }
__device__ __forceinline__ void r512x_swap32(cx<double> &a, cx<double> &b) { r512x_swap32(a.x, b.x); r512x_swap32(a.y, b.y); }

// CPLX: the rows are the complex spectra (M + 1 interleaved (re, im) pairs per frame, the imaginary parts of the first and
// last zero: window_univar followed by fft_fftw_r2hc, reference c/window_univar.c:41-89, c/fft.fftw.r2hc.c:69-78) instead of
// their powers.
// RQ (nfft 2048, full window, hop 512 = a quarter of the frame; hop 1024 likewise, two quarters at a time): the slot is a ring of four quarters.  A wave draws UNITS of
// R512X_UNIT consecutive frames of one signal instead of single steps; frame t of a unit is the quarters t .. t + 3 of the unit's
// samples, quarter n lives at slot position n % 4, and once pass 1 of frame t has read the slot only quarter t + 4 is fetched, into
// the position of quarter t (2 loads of 1 KiB per frame instead of 8; a unit's first frame takes all 8).  Issuing an LDS-DMA load
// costs the wave about 160 cycles where these go out, and a 2048-point step spent 1.3 k of its 7.6 k cycles on them.  Same
// arithmetic on the same samples as the step form: the two return the same bits.
#ifndef R512X_UNIT_FRAMES
#define R512X_UNIT_FRAMES 6       // measured on 64 x 60 s @ 44.1 kHz, float32 / float64 ms: 4: 0.537 / 1.196, 6: 0.532 / 1.181, 8: 0.541 / 1.188, 12: 0.537 / 1.180,
                                 // 16: 0.545, 32: 0.557, 64: 0.583 (single steps: 0.562 / 1.329): longer units spread the rows a workgroup writes
#endif
constexpr int R512X_UNIT = R512X_UNIT_FRAMES;   // steps per unit
#ifndef R512X_RING_1024
#define R512X_RING_1024 0       // 1: nfft 1024 with a full window and hop 256 takes the ring form as well (two frames per step, blocks t .. t + 2 of a
                                // ring of four): same bits, but measured SLOWER than the 1280-sample span of the step form, 0.468 -> 0.481 ms in
                                // float32 and 1.055 -> 1.08 ms in float64 (units of 4, 6 or 12 steps alike): the step form fetches 5 loads per step
                                // there, the ring 3 on average, and pays for them with a per-lane choice of block and a slot of 2048 samples
#endif
template <typename T, int Q, bool FULL, bool MN0, bool EPI, int SPAN, bool ALN, bool CPLX = false, bool RQ = false, int NK = 16>
__global__ __launch_bounds__(64 * r512x_waves<T>(), sizeof(T) == 4 ? 2 : 1) void stft_r512x_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                                                               const T *__restrict__ win, const T *__restrict__ tw,
                                                                                               const R512XArgs a)
{
    static_assert(Q == 2 || Q == 4, "nfft 1024 or 2048");
    static_assert(!RQ || (FULL && SPAN == 2048 && ALN), "the ring form: full window, a 2048-sample slot");
    constexpr int FPU = RQ ? R512X_UNIT * (4 / Q) : 4 / Q;   // frames per drawn unit of work (RQ: R512X_UNIT steps)
    using C = cx<T>;
    typedef T pair_u __attribute__((ext_vector_type(2), aligned(sizeof(T))));        // a pair of samples at sample alignment
    typedef T pair_a __attribute__((ext_vector_type(2), aligned(2 * sizeof(T))));    // ... at its own alignment
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int NFFT = 512 * Q, M = 256 * Q, F = M + 1, FPS = 4 / Q, NWV = r512x_waves<T>(), NV = SPAN * (int)sizeof(T) / 1024;
    constexpr int WAVE_LDS = R512X_TILE + SPAN;
    constexpr int ROWF = CPLX ? 2 * F : F;                 // values of an output row
    constexpr int NVI = (FPS * ROWF + 255) / 256;          // float32: whole-vector wave stores of a step's block of rows
    // float64, power form: the rows leave straight from the registers (the lanes of a frame hold consecutive bins of every k2:
    // K = 16 Q k2 + Q j + r), 17 store instructions; otherwise they are staged in LDS in output order.  (Measured on 64 x 60 s @
    // 44.1 kHz, nfft 2048 / hop 512: float64 1.179 -> 1.059 ms; float32, whose staged rows leave as 6 stores of 16 bytes per lane,
    // 0.529 -> 0.545 ms with it: float64 only; nfft 1024 / hop 256 float64 1.10 -> 1.04 ms.)
    constexpr bool DIRECT = R512X_DIRECT_STORE && !CPLX && !F32;
    constexpr int NSTORE = DIRECT ? 17 : F32 ? NVI + 1 : (FPS * ROWF + 63) / 64;   // store instructions per step
    static_assert(FPS * ROWF + 3 <= R512X_TILE, "row staging fits the tile");
    // Windows shorter than the transform: the rows of a lane past the window must not reach the sums even as 0 x NaN.  The general way is a lane
    // mask per row and element (32 of them, in scalar registers that spill).  With one frame per step the slot belongs to that frame alone, and
    // it is cheaper to see to it that the slot holds nothing past the window that is not a sample of the window: the lanes past the last
    // vector of the span copy the vector before it (ring_load_span<.., SPARE>), and the up to three samples behind the window inside the
    // last vector are overwritten with zeros once the span has landed.  (Needs two vectors of window: r512x_supported.)
    constexpr bool UNMASKED = R512X_ZERO_TAIL && Q == 4 && !FULL && !MN0 && !RQ;
    // nfft 1024 (two frames share a slot: nothing to be made harmless there) has instantiations by the number NK of the lane's rows of 64 samples
    // that hold data instead (NK = ceil(L / 64) < 16: row k of a lane is the samples 64 k + 2 j + 32 g + {0, 1} of the frame, so rows below NK - 1 lie
    // inside the window for every lane, row NK - 1 is masked, the rows above are zero and are not read).  NK = 16: the general form.
    static_assert(NK == 16 || (Q == 2 && !FULL && !MN0 && !RQ), "row-count instantiations: nfft 1024, a window shorter than the transform, no mean removal");
    constexpr bool ROWS = NK < 16;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane >> 4, j = lane & 15;
    const int f = grp / Q, g = grp % Q;                    // frame of the step; group inside the frame
    const int r = Q - 1 - g;                               // residue mod Q of the bins the group ends up with
    constexpr int R = 16 / Q;                              // rows of the input stage per group
    const int L = a.L;

    T *xs = reinterpret_cast<T *>(smem_raw) + wv * WAVE_LDS;           // exchange tile, then row staging
    pair_a *ex = reinterpret_cast<pair_a *>(xs);
    T *slot = xs + R512X_TILE;                                         // the span
    pair_a *ttab = reinterpret_cast<pair_a *>(reinterpret_cast<T *>(smem_raw) + NWV * WAVE_LDS);   // W256^(j k1) at [k1][j]
    constexpr int WLN = 16 * Q;                                         // lanes of a frame
    pair_a *wtab = ttab + (R512X_TWD_LDS ? 256 : 0);                    // window taps of lane l of a frame, pair e: [e][l]
    constexpr int WTE = 16;                                             // its pairs per lane
    unsigned char *dummy = reinterpret_cast<unsigned char *>(wtab + (R512X_WIN_LDS ? WTE * WLN : 0));   // 1 KiB that placeholder LDS-DMA loads write
    unsigned *ticket = reinterpret_cast<unsigned *>(dummy + 1024);

    // ---- per-lane constants ----
    auto twn = [&](int idx) {                              // W_nfft^idx, idx < nfft (the table holds idx < M)
        T sg = T(1);
        if (idx >= M) { idx -= M; sg = T(-1); }
        return C{sg * tw[2 * idx], sg * tw[2 * idx + 1]};
    };
    C twd[16];                                             // W256^(j k1)
    if constexpr (!R512X_TWD_LDS) {
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) { twd[k1] = twn(2 * Q * j * k1); }
    } else if (tid < 256) {
        const C t = twn(2 * Q * (tid & 15) * (tid >> 4));
        ttab[tid] = pair_a{t.x, t.y};
    }
    // input stage: row i of the group is n1 = Q i + g, i.e. m = j + 16 n1; halved window taps of its Q quarters
    // and the twiddles W_M^(m rr), rr = 1 .. Q - 1
    C hw[16], dt[(Q - 1) * R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int m = j + 16 * (Q * i + g);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int l = 2 * m + 512 * q;
            hw[Q * i + q] = C{(l < L) ? T(0.5) * win[l] : T(0), (l + 1 < L) ? T(0.5) * win[l + 1] : T(0)};
        }
#pragma unroll
        for (int rr = 1; rr < Q; ++rr) { dt[(Q - 1) * i + rr - 1] = twn((2 * m * rr) & (NFFT - 1)); }
    }
    if constexpr (R512X_WIN_LDS) {
        if (tid < WLN) {                                   // (the lanes of frame 0 of wave 0)
#pragma unroll
            for (int e = 0; e < 16; ++e) { wtab[e * WLN + tid] = pair_a{hw[e].x, hw[e].y}; }
        }
    }
    C ow[8];                                               // -i W_nfft^K, K = Q (j + 16 k2) + r
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) { const int K = Q * (j + 16 * k2) + r; ow[k2] = C{tw[2 * K + 1], -tw[2 * K]}; }
    if (tid == 0) { *ticket = NWV; }
    __syncthreads();

    const int reach = a.tbl ? a.reach : (FPS - 1) * a.stp;
    const int need = (RQ && Q == 2) ? 1536 : reach + L;    // samples of a span that pass 1 uses (RQ, nfft 1024: three whole blocks)
    const T *xrow = slot + f * a.stp + 2 * j + 32 * g;   // row i, quarter q at xrow[32 Q i + 512 q]
    // Start of the span of the step at frame w_ (wave-uniform).  Table form: the starts of the step's frames by
    // scalar loads (a vector load would put the compiler's own wait for ALL outstanding memory operations, the row
    // stores included, in front of the span fetch), and xrow for that step: called when pass 1 is done with the last one.
    auto step_start = [&](const int w_) {
        if (a.tbl == nullptr) { return a.ss0 + w_ * a.stp; }
        const __attribute__((address_space(4))) int *tp = (const __attribute__((address_space(4))) int *)(uintptr_t)a.tbl;
        const int t0 = tp[2 * w_];
        if constexpr (FPS == 2) {
            const int t1 = tp[2 * min(w_ + 1, a.W - 1)];
            xrow = slot + (f ? t1 - t0 : 0) + 2 * j + 32 * g;
        }
        return t0;
    };

    // ---- fetch of one span into the slot: NV LDS-DMA loads whatever the case (the counted wait relies on
    // that); a span that crosses an edge of its signal is gathered sample by sample and NV placeholder loads follow
    auto load_span = [&](const T *xb, const int s0) {
        ring_load_span<NV, UNMASKED>(slot, reinterpret_cast<T *>(dummy), tw, R512X_NOLOAD ? tw - (s0 - (a.edge ? a.rd_shift : 0)) : xb, s0, need, a.N, a.edge, a.rd_shift, lane,
                           [&](const T *sig, const int first, const int i) {
                               FrameDesc d{};
                               d.N = a.N; d.edge = a.edge;
                               return fetch_sample(sig, d, (long long)first, (long long)a.rd_shift, i);
                           });
    };

    // ... and the same one load at a time (R512X_LOAD_SPREAD: where in the step the loads of the next span are issued)
    auto span_fetch = [&](const T *sig, const int first, const int i) {
        FrameDesc d{};
        d.N = a.N; d.edge = a.edge;
        return fetch_sample(sig, d, (long long)first, (long long)a.rd_shift, i);
    };
    RingSpan<T> nspan{};
    auto span_piece = [&](const int q) __attribute__((always_inline)) {
        if (q < NV) { ring_span_piece<T>(nspan, q, slot, reinterpret_cast<T *>(dummy), tw, need, lane, span_fetch); }
    };

    // Work distribution: 8 contiguous ranges of steps, one per group of workgroups that share an XCD; a round
    // of a range is 8 neighbouring steps per workgroup, drawn by its waves from the counter.
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int s_lo = (int)((long long)a.steps * xcd / 8), s_hi = (int)((long long)a.steps * (xcd + 1) / 8);
    const int round = nwg_x * NWV, first = s_lo + (blockIdx.x >> 3) * NWV;
    int sid = first + wv;
    if (sid >= s_hi) { return; }
    int b = (int)a.spd.div((unsigned)sid);
    int w = (sid - b * a.sps) * FPU;
    [[maybe_unused]] int t = 0, ulen = RQ ? (min(FPU, a.W - w) + FPS - 1) / FPS : 1;      // RQ: the step inside the unit, the unit's steps
    // RQ: 512-sample blocks between two steps (nfft 2048: hop 512: 1, hop 1024: 2; nfft 1024, two frames per step at hop 256: 1)
    [[maybe_unused]] const int qstep = RQ ? (FPS * a.stp) >> 9 : 0;
    const T *xb = X + (long long)b * a.ldx;
    T *yb = Y + (long long)b * a.Wy * ROWF;

    const unsigned cnt_addr = ring_lds_addr(ticket);
    unsigned drawn = 0;
    int nsid = 0, nw = 0;
    const T *nxb = xb;
    T *nyb = yb;
    bool more = false;
    auto advance = [&]() {
        const unsigned t = (unsigned)ring_drawn(drawn);          // (unsigned: the division and the products below stay short)
        nsid = first + (int)(t / NWV) * round + (int)(t % NWV);
        more = nsid < s_hi;
        if (more) {
            const unsigned nb = a.spd.div((unsigned)nsid);
            nw = (nsid - (int)nb * a.sps) * FPU;
            nxb = X + (unsigned long long)nb * (unsigned long long)a.ldx;
            nyb = Y + (unsigned long long)nb * ((unsigned long long)a.Wy * ROWF);
        }
    };

    bool first_step = true;
    load_span(xb, step_start(w));

#if R512X_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
#endif
    while (true) {
        // the step after this one: asked for now, needed further down (RQ: when this is the unit's last frame)
        if (!RQ || t + 1 >= ulen) { ring_draw(drawn, cnt_addr); }
        // The span of this step was asked for in the middle of the last one; memory operations complete in issue
        // order and behind its NV loads lie that step's 5 row stores, which stay in flight.
        // (window taps from the LDS table: asked for ahead of the wait)
        [[maybe_unused]] C hwl[WTE];
        if constexpr (R512X_WIN_LDS) {
#pragma unroll
            for (int e = 0; e < WTE; ++e) { const pair_a t = wtab[e * WLN + (lane & (WLN - 1))]; hwl[e] = C{t.x, t.y}; }
        }
        if (first_step) { ring_wait_vm<0>(); } else { ring_wait_vm<NSTORE>(); }
        first_step = false;
        if constexpr (UNMASKED) {
            constexpr int VEC = 16 / (int)sizeof(T);
            if (lane < ((-L) & (VEC - 1))) { slot[L + lane] = T(0); }
        }
        wave_sync();
        R512X_STAMP(0);                                    // wait for the span

        // ---- input stage: the group's rows in all Q quarters, windowed; Q-point butterflies; twiddles ----
        C zz[16];
        // RQ, nfft 1024: step t of a unit is the blocks t, t + 1, t + 2 of the unit's samples, block n at ring position n % 4.  Half q of
        // frame 0 is block q; frame 1 starts 256 samples further on, so its rows i >= 4 lie one block further (at 256 samples less).
        [[maybe_unused]] const T *rq_lo[2] = {xrow, xrow}, *rq_hi[2] = {xrow, xrow};
        if constexpr (RQ && Q == 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                rq_lo[q] = xrow + 512 * ((q + t) & 3);
                rq_hi[q] = f ? xrow - 512 + 512 * ((q + 1 + t) & 3) : rq_lo[q];
            }
        }
        {
            C A[16];                                       // A[Q i + q]
            [[maybe_unused]] int lim = L - 2 * j - 32 * g;  // element c of the pair (i, q) is inside the window iff 32 Q i + 512 q + c < lim
            if constexpr (DSP_MASK_RECOMPUTE && !FULL) { asm volatile("" : "+v"(lim)); }
#pragma unroll
            for (int i = 0; i < R; ++i) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    if constexpr (ROWS) { if (i + 8 * q >= NK) { A[Q * i + q] = C{T(0), T(0)}; continue; } }
                    const T *p;
                    if constexpr (RQ && Q == 2) { p = (i < 4 ? rq_lo[q] : rq_hi[q]) + 32 * Q * i; }
                    else { p = xrow + 32 * Q * i + 512 * (RQ ? ((q + qstep * t) & 3) : q); }
                    C v;
                    if constexpr (ALN) { const pair_a t = *reinterpret_cast<const pair_a *>(p); v = C{t.x, t.y}; }
                    else { const pair_u t = *reinterpret_cast<const pair_u *>(p); v = C{t.x, t.y}; }
                    if (!FULL && !UNMASKED && (!ROWS || i + 8 * q == NK - 1)) {   // keep data past the window out (0 * NaN)
                        v.x = (32 * Q * i + 512 * q < lim) ? v.x : T(0);
                        v.y = (32 * Q * i + 512 * q + 1 < lim) ? v.y : T(0);
                    }
                    if constexpr (R512X_WIN_LDS) { A[Q * i + q] = v * hwl[Q * i + q]; }
                    else { A[Q * i + q] = v * hw[Q * i + q]; }
                }
            }
            if (MN0) {
                C acc = A[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) { acc += A[i]; }
                T sum = acc.x + acc.y;
#pragma unroll
                for (int o = 1; o < 16 * Q; o <<= 1) { sum += __shfl_xor(sum, o); }
                const T mean = sum / (T)L;
#pragma unroll
                for (int i = 0; i < R; ++i) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        A[Q * i + q] -= C{(FULL || 32 * Q * i + 512 * q < lim) ? mean : T(0), (FULL || 32 * Q * i + 512 * q + 1 < lim) ? mean : T(0)};
                    }
                }
            }
            if constexpr (Q == 2) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    C o0 = A[2 * i] + A[2 * i + 1];
                    C o1 = cx_mul<T>(A[2 * i] - A[2 * i + 1], dt[i]);
                    // the even rows of lanes (g = 0) keep the odd bins of their own rows and get those of the other
                    // group's rows; the odd rows of lanes the even bins
                    r512x_swap16(o1, o0);
                    zz[2 * i] = o1;                        // n1 = 2 i + 0: the row that group 0 read
                    zz[2 * i + 1] = o0;                    // n1 = 2 i + 1
                }
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    dft4c<T>(A[4 * i], A[4 * i + 1], A[4 * i + 2], A[4 * i + 3]);
#pragma unroll
                    for (int rr = 1; rr < 4; ++rr) { A[4 * i + rr] = cx_mul<T>(A[4 * i + rr], dt[3 * i + rr - 1]); }
                    // neighbouring rows of lanes trade the low bit of the residue ...
                    r512x_swap16(A[4 * i + 1], A[4 * i + 0]);
                    r512x_swap16(A[4 * i + 3], A[4 * i + 2]);
                    // (A[4 i + 2 r1 + 1 - s0] now holds residue bit r1 of the row that the group with low bit s0 read)
                    // ... and the halves of the wave the high bit
                    r512x_swap32(A[4 * i + 2], A[4 * i + 0]);
                    r512x_swap32(A[4 * i + 3], A[4 * i + 1]);
                    // the first operand of each swap now holds a row read by the lower half (s1 = 0), the second one
                    // of the upper half: row n1 = 4 i + s, s = 2 s1 + s0, is in A[4 i + 3 - s]
#pragma unroll
                    for (int sgrp = 0; sgrp < 4; ++sgrp) { zz[4 * i + sgrp] = A[4 * i + 3 - sgrp]; }
                }
            }
        }
        R512X_STAMP(1);                                    // input stage
        // pass-1 twiddles from the LDS table: asked for here, ahead of the transform that precedes their use.  (Read
        // inside the loop of the exchange writes, each read follows the write before it -- the compiler cannot tell
        // the table from the tile -- and is waited for on the spot: 15 LDS round trips in a row.)
        C twl[16];
        if constexpr (R512X_TWD_LDS && R512X_TWD_AHEAD) {
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) { const pair_a t = ttab[k1 * 16 + j]; twl[k1] = C{t.x, t.y}; }
            __builtin_amdgcn_sched_barrier(0);             // (the scheduler would sink the reads below the transform)
        }
        dft16x<T, 16>(zz);
        wave_sync();                                       // all lanes have read the span
        // The slot takes the next step's span.  The reads of pass 1 must have RETURNED before the first load into
        // the slot is issued (the compiler may sink their consumers, and with them its own wait, below the
        // assembly statements): ring_drawn waits for every LDS operation of the wave.  Past the end of the
        // workgroup's share the current span is fetched again: the number of loads per step stays fixed.
        [[maybe_unused]] bool same_unit = false;
        if constexpr (RQ) {
            same_unit = t + 1 < ulen;
            if (same_unit) {
                // the next frame of the unit: its last quarter into the position of this frame's first
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                more = true;
                // (hop 1024: its last two quarters, which lie side by side in the ring as well)
                // (nfft 1024: block t + 3, which this step does not touch)
                T *const pos = slot + 512 * ((Q == 2 ? t + 3 : qstep * t) & 3);
                const int s_new = a.ss0 + (w + FPS) * a.stp + (Q == 2 ? 1024 : 2048 - a.stp);
                if (qstep == 1) { ring_load_span<NV / 4>(pos, reinterpret_cast<T *>(dummy), tw, xb, s_new, 512, a.N, a.edge, a.rd_shift, lane, span_fetch); }
                else { ring_load_span<NV / 2>(pos, reinterpret_cast<T *>(dummy), tw, xb, s_new, 1024, a.N, a.edge, a.rd_shift, lane, span_fetch); }
            } else {
                advance();
                load_span(more ? nxb : xb, step_start(more ? nw : w));
            }
        } else {
            advance();
            if constexpr (R512X_LOAD_SPREAD == 0) { load_span(more ? nxb : xb, step_start(more ? nw : w)); }
            else { nspan = ring_span_open<T>(more ? nxb : xb, step_start(more ? nw : w), need, a.N, a.edge, a.rd_shift); }
        }
        R512X_STAMP(2);                                    // pass-1 transform, draw, span fetch issued
        {
            pair_a *exf = ex + grp * R512_EXF + j;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                C v = zz[slot16(k1)];
                if (k1 > 0) {
                    if constexpr (R512X_TWD_LDS && R512X_TWD_AHEAD) { v = cx_mul<T>(v, twl[k1]); }
                    else if constexpr (R512X_TWD_LDS) { const pair_a t = ttab[k1 * 16 + j]; v = cx_mul<T>(v, t.x, t.y); }
                    else { v = cx_mul<T>(v, twd[k1]); }
                }
                exf[k1 * 17] = pair_a{v.x, v.y};
                if constexpr (R512X_LOAD_SPREAD == 1) { if ((k1 & 1) == 0) { span_piece(k1 >> 1); } }
            }
        }
        wave_sync();
        if constexpr (R512X_LOAD_SPREAD == 2) {
#pragma unroll
            for (int q = 0; q < NV; ++q) { span_piece(q); }
        }
        R512X_STAMP(3);                                    // exchange writes

        // ---- pass 2: lane (group, k1 = j): 16-point FFT over the columns ----
        {
            const pair_a *exf = ex + grp * R512_EXF + j * 17;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) { const pair_a t = exf[n2]; zz[n2] = C{t.x, t.y}; }
        }
        dft16x<T, 16>(zz);
        if constexpr (R512X_LOAD_SPREAD == 3) {
#pragma unroll
            for (int q = 0; q < NV; ++q) { span_piece(q); }
        }

        // ---- real-FFT split fused with power.  The lane holds Z[K] (halved), K = Q (j + 16 k2) + r, in
        // register k2.  Z[M - K] is, for r = 0: register 15 - k2 of lane (16 - j) % 16 (register (16 - k2) % 16
        // of lane 0 itself); for r > 0: register 15 - k2 of lane 15 - j of group Q - r.
        // With A = Z_K + conj Z_(M-K), B = Z_K - conj Z_(M-K), T = -i W_nfft^K B:
        // |X_K|^2 = |A + T|^2 and |X_(M-K)|^2 = |A - T|^2.
        R512X_STAMP(4);                                    // pass-2 reads and transform
        // the step's block of rows in the output, and how it lies against 16-byte boundaries
        T *yblk = yb + (size_t)w * ROWF;
        const int nout = __builtin_amdgcn_readfirstlane(min(FPS, a.W - w) * ROWF);
        int head = 0, sh = 0;                              // float32: value e of the block is staged at xs[sh + e]
        if constexpr (F32) { head = ring_block_head(yblk); sh = ring_stage_shift(head); }
        const bool own0 = (r == 0) && (j == 0);
        // where Z[M - K] sits: residue (Q - r) % Q of the same frame, lane (16 - j) % 16 of its group for r = 0, lane 15 - j otherwise
        [[maybe_unused]] const int paddr = 4 * (16 * ((grp - g) + (Q - 1 - (Q - r) % Q)) + (r == 0 ? ((16 - j) & 15) : 15 - j));   // (residue r' is held by group Q - 1 - r' of the frame)
        if constexpr (CPLX) {
            // X_K = A + T, X_(M-K) = conj(A - T), each written to the staging area as soon as it exists; K = M / 2 is
            // 2 conj Z_(M/2) (lane 0 of residue 0)
            const T sc = (T)a.cscale;
            const C zmid = zz[slot16(8)];
            wave_sync();                                   // every lane has read its tile row (into zz): reuse the region
            T *pf = xs + sh + f * ROWF;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const C z = zz[slot16(k2)];
                const C own = zz[slot16((16 - k2) & 15)], oth = zz[slot16(15 - k2)];
                C m;
                if constexpr (R512X_BPERM) {
                m = C{r512x_from_lane(oth.x, paddr), r512x_from_lane(oth.y, paddr)};
                } else {
                const T mx1 = r512x_mirror(oth.x), my1 = r512x_mirror(oth.y);
                if constexpr (Q == 2) {
                    // residue 0 (the odd rows of 16 lanes) takes lane (16 - j) % 16, residue 1 lane 15 - j: the rotation by one lane in
                    // the odd rows only, which the move itself can do
                    m = C{r512x_ror1_rows<0xA>(mx1), r512x_ror1_rows<0xA>(my1)};
                } else {
                    const T mx2 = r512x_ror1(mx1), my2 = r512x_ror1(my1);
                    const T mx3 = r512x_xor32(mx1, lane < 32), my3 = r512x_xor32(my1, lane < 32);
                    m = C{r == 0 ? mx2 : (r == 2 ? mx1 : mx3), r == 0 ? my2 : (r == 2 ? my1 : my3)};
                }
                }
                m = C{own0 ? own.x : m.x, own0 ? own.y : m.y};
                const C A = cx_add_conj<T>(z, m), B = cx_sub_conj<T>(z, m);
                const C Tw = cx_mul<T>(B, ow[k2]);
                const C u = A + Tw, v = A - Tw;
                const int K = Q * (j + 16 * k2) + r;
                const bool edge = own0 && k2 == 0;         // bins 0 and M: imaginary parts zero
                pf[2 * K] = sc * u.x; pf[2 * K + 1] = edge ? T(0) : sc * u.y;
                pf[2 * (M - K)] = sc * v.x; pf[2 * (M - K) + 1] = edge ? T(0) : -sc * v.y;
            }
            if (own0) { pf[M] = T(2) * sc * zmid.x; pf[M + 1] = T(-2) * sc * zmid.y; }
            wave_sync();
            R512X_STAMP(5);
        } else {
            T pw[16];                                          // pw[k2] = P[K], pw[8 + k2] = P[M - K]
            T pmid = T(4) * cx_norm<T>(zz[slot16(8)]);         // K = M / 2: lane j == 0 of group 0 only
            // the partner values by ds_bpermute: all of them asked for before the first is used (fetched pair by pair inside
            // the loop, each pair was waited for on the spot: 8 LDS round trips in a row)
            [[maybe_unused]] C mm[8];
            if constexpr (R512X_BPERM && R512X_BPERM_AHEAD) {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const C oth = zz[slot16(15 - k2)];
                    mm[k2] = C{r512x_from_lane(oth.x, paddr), r512x_from_lane(oth.y, paddr)};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (lane 0 of residue 0 is its own partner: the moves hand it its own register 15 - k2, which the last
            // select replaces by register (16 - k2) % 16)
            auto partner = [&](const int k2) __attribute__((always_inline)) {
                const C own = zz[slot16((16 - k2) & 15)], oth = zz[slot16(15 - k2)];
                C m;
                if constexpr (R512X_BPERM && R512X_BPERM_AHEAD) {
                    m = mm[k2];
                } else if constexpr (R512X_BPERM) {
                    m = C{r512x_from_lane(oth.x, paddr), r512x_from_lane(oth.y, paddr)};
                } else {
                    const T mx1 = r512x_mirror(oth.x), my1 = r512x_mirror(oth.y);
                    if constexpr (Q == 2) {
                        // residue 0 (the odd rows of 16 lanes) takes lane (16 - j) % 16, residue 1 lane 15 - j: the rotation by one lane in
                        // the odd rows only, which the move itself can do
                        m = C{r512x_ror1_rows<0xA>(mx1), r512x_ror1_rows<0xA>(my1)};
                    } else {
                        const T mx2 = r512x_ror1(mx1), my2 = r512x_ror1(my1);
                        const T mx3 = r512x_xor32(mx1, lane < 32), my3 = r512x_xor32(my1, lane < 32);
                        m = C{r == 0 ? mx2 : (r == 2 ? mx1 : mx3), r == 0 ? my2 : (r == 2 ? my1 : my3)};
                    }
                }
                return C{own0 ? own.x : m.x, own0 ? own.y : m.y};
            };
            // two bins at a time, stage by stage: a packed instruction that consumes the result of the one issued just before it
            // costs a wait state, and the stages of one bin are such a chain
            const C pm = C{T(1), T(-1)};
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if constexpr (R512X_LOAD_SPREAD == 4) { span_piece(2 * h); span_piece(2 * h + 1); }
                const int ka = h, kb = h + 4;
                const C ma = partner(ka), mb = partner(kb);
                const C za = zz[slot16(ka)], zb = zz[slot16(kb)];
                const C Aa = cx_add_conj<T>(za, ma), Ab = cx_add_conj<T>(zb, mb);
                const C Ba = cx_sub_conj<T>(za, ma), Bb = cx_sub_conj<T>(zb, mb);
                const C ta = cx_mul_first<T>(Ba, ow[ka]), tb = cx_mul_first<T>(Bb, ow[kb]);
                const C Ta = cx_mul_second<T>(Ba, ow[ka], ta), Tb = cx_mul_second<T>(Bb, ow[kb], tb);
                // (|A + T|^2, |A - T|^2) as one pair: the real parts of both sums in one packed register, the imaginary parts in
                // another (x * 1 + y is x + y exactly: the bits of cx_norm(A + T), cx_norm(A - T))
                const C rea = cx_fma(C{Ta.x, Ta.x}, pm, C{Aa.x, Aa.x}), reb = cx_fma(C{Tb.x, Tb.x}, pm, C{Ab.x, Ab.x});
                const C ima = cx_fma(C{Ta.y, Ta.y}, pm, C{Aa.y, Aa.y}), imb = cx_fma(C{Tb.y, Tb.y}, pm, C{Ab.y, Ab.y});
                const C qa = rea * rea, qb = reb * reb;
                const C ra = ima * ima, rb = imb * imb;
                const C pa = qa + ra, pb = qb + rb;
                pw[ka] = pa.x; pw[8 + ka] = pa.y;
                pw[kb] = pb.x; pw[8 + kb] = pb.y;
            }
            if (EPI) {
                power_epilogue_n(pw, a.epi);
                pmid = power_epilogue(pmid, a.epi);
            }
            wave_sync();                                       // every lane has read its tile row: reuse the region
            R512X_STAMP(5);                                    // split
            if constexpr (DIRECT) {
                // bin M / 2 of a frame is held by lane j = 0 of its group of residue 0; every lane of the frame stores it
                auto bcast = [&](const int src) {
                    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint((double)pmid), src), __builtin_amdgcn_readlane(__double2loint((double)pmid), src));
                };
                T mid = (T)bcast(16 * (Q - 1));
                if constexpr (FPS == 2) { const T mid1 = (T)bcast(16 * (2 * Q - 1)); mid = f ? mid1 : mid; }
                T *const yk = yblk + f * ROWF + Q * j + r;         // K = 16 Q k2 + Q j + r
                // (a last step with one frame: the lanes of the second do not store; the instructions are issued all the same, which is
                // what the counted wait of the next span relies on)
                if (FPS == 1 || f * ROWF < nout) {
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) {
                        if constexpr (R512X_NOSTORE == 0) {
                            __builtin_nontemporal_store(pw[k2], yk + 16 * Q * k2);
                            __builtin_nontemporal_store(pw[8 + k2], yblk + f * ROWF + M - Q * j - r - 16 * Q * k2);
                        }
                    }
                    if constexpr (R512X_NOSTORE == 0) { __builtin_nontemporal_store(mid, yblk + f * ROWF + M / 2); }
                }
            } else {
                T *pf = xs + sh + f * F;
                // (K = 16 Q k2 + Q j + r: both runs of addresses as one base and constant offsets that are not negative)
                T *const pk = pf + (Q * j + r), *const pm = pf + (M - 16 * Q * 7) - (Q * j + r);
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    pk[16 * Q * k2] = pw[k2];
                    pm[16 * Q * (7 - k2)] = pw[8 + k2];
                }
                if (own0) { pf[M / 2] = pmid; }
            }
            wave_sync();
        }

        // ---- five stores: four of whole 16-byte vectors, one for the values before and after them.  Surplus
        // lanes repeat a valid store (same value to the same address).
        R512X_STAMP(6);                                    // staging
        if constexpr (DIRECT) {
        } else if constexpr (F32) {
            ring_store_rows<NVI, R512X_NOSTORE != 0>(yblk, xs + sh, head, nout, lane);
        } else {
            // float64: element-wise, a wave stores 512 consecutive bytes per instruction, at any 8-byte alignment
#pragma unroll
            for (int i = 0; i < NSTORE; ++i) {
                const int e = min(lane + 64 * i, nout - 1);
                __builtin_nontemporal_store(xs[e], yblk + e);
            }
        }
        R512X_STAMP(7);                                    // stores issued
        if (!more) { break; }
        wave_sync();                                       // staging reads are done before the next step writes the tile
        if (RQ && same_unit) { ++t; w += FPS; }
        else {
            sid = nsid; w = nw; xb = nxb; yb = nyb;
            if constexpr (RQ) { t = 0; ulen = (min(FPU, a.W - w) + FPS - 1) / FPS; }
        }
    }
    // no LDS-DMA load may land, and no draw return, after the wave has ended
    ring_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if R512X_STAMPS
    if (lane == 0) {
        const int gw = blockIdx.x * NWV + wv;
        if (gw < 2048) { for (int i = 0; i < 8; ++i) { g_r512x_stamps[gw * 8 + i] = st_acc[i]; } }
    }
#endif
}
#if R512X_STAMPS
extern "C" int dsp_debug_read_stamps_x(unsigned long long *dst, size_t n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_r512x_stamps), n * sizeof(unsigned long long));
}
#endif

// nfft 1024 or 2048; frames at ss0 + w * hop, or a table of frame starts that never decrease and are read where they
// start; the frames of a step inside 2048 samples; every index of one signal's input and output inside 31 bits
bool r512x_supported(const FrameDesc &d)
{
    if (d.nfft != 1024 && d.nfft != 2048) { return false; }
    const int fps = 4 / (d.nfft / 512);
    // (tbl_reach bounds the distance of frames w and w + 3, hence that of the fps <= 2 frames of a step)
    const bool placed = d.tbl ? (d.tbl_reach >= 0 && (fps == 1 || d.tbl_reach + d.nfft <= 2048))
                              : (d.stp >= 1 && (long long)(fps - 1) * d.stp + d.nfft <= 2048);
    if (R512X_ZERO_TAIL && d.nfft == 2048 && d.L < 8) { return false; }      // (the unmasked form needs two 16-byte vectors of window)
    return placed && d.L >= 1 && d.L <= d.nfft &&
           d.N < (1ll << 31) - 4096 && (long long)d.wy() * (d.nfft / 2 + 1) < (1ll << 30) && (long long)d.B * d.W < (1ll << 30) &&
           d.ss0 > -(1ll << 30) && d.ss0 < (1ll << 30) && d.rd_shift >= 0 && d.rd_shift < (1ll << 30);
}

template <typename T, int Q, bool FULL, bool MN0, bool EPI, int SPAN, bool ALN, bool CPLX = false, bool RQ = false, int NK = 16>
static hipError_t launch_r512x_inst(T *Y, const T *X, const T *win, const T *tw, const R512XArgs &a, hipStream_t s)
{
    constexpr int NWV = r512x_waves<T>();
    const size_t lds = ((size_t)NWV * (R512X_TILE + SPAN) + (R512X_TWD_LDS ? 512 : 0) + (R512X_WIN_LDS ? 2 * 16 * 16 * Q : 0)) * sizeof(T) + 1024 + 16;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_r512x_kernel<T, Q, FULL, MN0, EPI, SPAN, ALN, CPLX, RQ, NK>), lds, grant); e != hipSuccess) { return e; }
    long long grid = ((long long)a.steps + NWV - 1) / NWV;
    if (grid > 256) { grid = 256; }                     // one workgroup per CU
    grid = (grid + 7) & ~7ll;                           // whole groups of 8 workgroups (one per XCD)
    hipLaunchKernelGGL((stft_r512x_kernel<T, Q, FULL, MN0, EPI, SPAN, ALN, CPLX, RQ, NK>), dim3((unsigned)grid), dim3(64 * NWV), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

// Whether a launch takes the ring-of-quarters form: nfft 2048, full window, frames at ss0 + 512 w or ss0 + 1024 w, and enough units of
// R512X_UNIT frames: at least DSP_R512X_MIN_UNITS (default: two per wave of a full grid; 0 = never)
template <typename T>
static bool r512x_ring(const FrameDesc &d)
{
#if R512X_MINIMAL
    return false;
#else
    const int fpu = R512X_UNIT * (d.nfft == 2048 ? 1 : 2);      // frames per unit
    long long min_units = 2 * 256 * r512x_waves<T>();
    if (const char *e = getenv("DSP_R512X_MIN_UNITS")) { min_units = atoll(e) > 0 ? atoll(e) : (1ll << 62); }
    const long long units = (long long)d.B * ((d.W + fpu - 1) / fpu);
    if (d.L != d.nfft || d.tbl != nullptr || units < min_units) { return false; }
    // (nfft 2048, hop 1024 measured 4 % faster in float64, 0.2827 -> 0.2715 ms on 8 x 2^23 samples, and 1.5 % slower in float32: float64 only)
    if (d.nfft == 2048) { return d.stp == 512 || (d.stp == 1024 && sizeof(T) == 8); }
    return d.nfft == 1024 && d.stp == 256 && R512X_RING_1024;
#endif
}

template <typename T, int Q>
static hipError_t launch_r512x_ring(T *Y, const T *X, const T *win, const T *tw, R512XArgs a, const FrameDesc &d, hipStream_t s)
{
    constexpr int fpu = R512X_UNIT * (4 / Q);
    a.sps = (d.W + fpu - 1) / fpu;                          // units per signal
    a.spd = FastDiv::of((unsigned)a.sps);
    a.steps = d.B * a.sps;
    const bool epi = d.epi.any();
#define DSP_R512Q(M_, E_) launch_r512x_inst<T, Q, true, M_, E_, 2048, true, false, true>(Y, X, win, tw, a, s)
    if (d.mn0) { return epi ? DSP_R512Q(true, true) : DSP_R512Q(true, false); }
    return epi ? DSP_R512Q(false, true) : DSP_R512Q(false, false);
#undef DSP_R512Q
}

template <typename T, int Q, int SPAN, bool ALN>
static hipError_t launch_r512x_shape(T *Y, const T *X, const T *win, const T *tw, const R512XArgs &a, const FrameDesc &d, hipStream_t s)
{
    const bool full = d.L == d.nfft, epi = d.epi.any();
#define DSP_R512X(F_, M_, E_) launch_r512x_inst<T, Q, F_, M_, E_, SPAN, ALN>(Y, X, win, tw, a, s)
#if R512X_MINIMAL
    if (full && !d.mn0 && !epi) { return DSP_R512X(true, false, false); }
    return hipErrorInvalidValue;
#else
    if (full) {
        if (d.mn0) { return epi ? DSP_R512X(true, true, true) : DSP_R512X(true, true, false); }
        return epi ? DSP_R512X(true, false, true) : DSP_R512X(true, false, false);
    }
    if (d.mn0) { return epi ? DSP_R512X(false, true, true) : DSP_R512X(false, true, false); }
    if constexpr (Q == 2 && ALN) {
        // windows of 8 to 15 rows of 64 samples (449 to 960 samples; 513-576: 25 ms at 22.05 kHz; 769-832: 25 ms at 32 kHz; 833-896: 20 ms at
        // 44.1 kHz; 897-960: 20 ms at 48 kHz): one masked row instead of 32 lane masks
        const int nk = (d.L + 63) / 64;
#define DSP_R512XK(NK_) if (nk == NK_) { return epi ? launch_r512x_inst<T, Q, false, false, true, SPAN, ALN, false, false, NK_>(Y, X, win, tw, a, s)  \
                                                    : launch_r512x_inst<T, Q, false, false, false, SPAN, ALN, false, false, NK_>(Y, X, win, tw, a, s); }
        DSP_R512XK(8) DSP_R512XK(9) DSP_R512XK(10) DSP_R512XK(11) DSP_R512XK(12) DSP_R512XK(13) DSP_R512XK(14) DSP_R512XK(15)
#undef DSP_R512XK
    }
    return epi ? DSP_R512X(false, false, true) : DSP_R512X(false, false, false);
#endif
#undef DSP_R512X
}

template <typename T>
static hipError_t launch_r512x_any(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s)
{
    const int Q = d.nfft / 512, fps = 4 / Q;
    R512XArgs a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.Wy = d.wy(); a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + fps - 1) / fps;
    a.spd = FastDiv::of((unsigned)a.sps);
    a.steps = d.B * a.sps;
    a.tbl = d.tbl; a.reach = 0;
    if (d.tbl) { a.stp = 0; a.ss0 = 0; a.rd_shift = 0; a.reach = (fps == 1) ? 0 : d.tbl_reach; }
    if (r512x_ring<T>(d)) {
#if R512X_RING_1024
        if (Q == 2) { return launch_r512x_ring<T, 2>(Y, X, win, tw, a, d, s); }
#endif
        return launch_r512x_ring<T, 4>(Y, X, win, tw, a, d, s);
    }
    if (Q == 4) { return launch_r512x_shape<T, 4, 2048, true>(Y, X, win, tw, a, d, s); }
    if (d.tbl || (d.stp & 1)) { return launch_r512x_shape<T, 2, 2048, false>(Y, X, win, tw, a, d, s); }
    if (d.stp + 1024 <= 1280) { return launch_r512x_shape<T, 2, 1280, true>(Y, X, win, tw, a, d, s); }
    return launch_r512x_shape<T, 2, 2048, true>(Y, X, win, tw, a, d, s);
}
hipError_t launch_stft_r512x(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s)
{
    return launch_r512x_any<float>(Y, X, win, tw, d, s);
}
hipError_t launch_stft_r512x(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, hipStream_t s)
{
    return launch_r512x_any<double>(Y, X, win, tw, d, s);
}

// The complex form: Y[b][w][nfft/2+1] interleaved (re, im) pairs times `scale`; equally spaced frames, no mean removal.
bool r512x_cplx_supported(const FrameDesc &d)
{
    return r512x_supported(d) && d.tbl == nullptr && !d.mn0 && !d.epi.any() && (long long)d.wy() * (d.nfft + 2) < (1ll << 30);
}

template <typename T, int Q, int SPAN, bool ALN>
static hipError_t launch_r512x_cplx_shape(T *Y, const T *X, const T *win, const T *tw, const R512XArgs &a, const FrameDesc &d, hipStream_t s)
{
#if R512X_MINIMAL
    return hipErrorInvalidValue;
#else
    if (d.L == d.nfft) { return launch_r512x_inst<T, Q, true, false, false, SPAN, ALN, true>(Y, X, win, tw, a, s); }
    return launch_r512x_inst<T, Q, false, false, false, SPAN, ALN, true>(Y, X, win, tw, a, s);
#endif
}

template <typename T>
static hipError_t launch_r512x_cplx_any(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    const int Q = d.nfft / 512, fps = 4 / Q;
    R512XArgs a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.Wy = d.wy(); a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + fps - 1) / fps;
    a.spd = FastDiv::of((unsigned)a.sps);
    a.steps = d.B * a.sps;
    a.tbl = nullptr; a.reach = 0;
    a.cscale = scale;
    if (r512x_ring<T>(d)) {
        const int fpu = R512X_UNIT * fps;
        a.sps = (d.W + fpu - 1) / fpu;
        a.spd = FastDiv::of((unsigned)a.sps);
        a.steps = d.B * a.sps;
#if R512X_RING_1024
        if (Q == 2) { return launch_r512x_inst<T, 2, true, false, false, 2048, true, true, true>(Y, X, win, tw, a, s); }
#endif
        return launch_r512x_inst<T, 4, true, false, false, 2048, true, true, true>(Y, X, win, tw, a, s);
    }
    if (Q == 4) { return launch_r512x_cplx_shape<T, 4, 2048, true>(Y, X, win, tw, a, d, s); }
    if (d.stp & 1) { return launch_r512x_cplx_shape<T, 2, 2048, false>(Y, X, win, tw, a, d, s); }
    if (d.stp + 1024 <= 1280) { return launch_r512x_cplx_shape<T, 2, 1280, true>(Y, X, win, tw, a, d, s); }
    return launch_r512x_cplx_shape<T, 2, 2048, true>(Y, X, win, tw, a, d, s);
}
hipError_t launch_stft_r512x_cplx(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    return launch_r512x_cplx_any<float>(Y, X, win, tw, d, scale, s);
}
hipError_t launch_stft_r512x_cplx(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    return launch_r512x_cplx_any<double>(Y, X, win, tw, d, scale, s);
}

}  // namespace dsp
