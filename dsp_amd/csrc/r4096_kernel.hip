// Register-resident 4096-point fused STFT kernel for gfx950 (float32): framing + window +
// [mean removal] + real FFT + power + [sqrt][log], one wavefront per frame, eight waves per workgroup
// sharing the halved window in LDS, one workgroup per CU (151.5 KB of LDS: a 16.9 KB tile per wave).
//
// The 4096 real samples of a frame are packed as 2048 complex points z_n = x_2n + i x_2n+1 and
// transformed as 32 x 64, every complex value one (re, im) register pair so that the butterflies
// run on the packed two-component float instructions:
//   loads   lane l is the complex column n2 = l and takes its 32 rows n1: thirty-two 8-byte loads per lane, each
//           wave load 512 contiguous bytes (the run form below loads only a quarter of them per frame);
//   pass 1  window (registers or LDS), one 32-point FFT over the rows in registers (two 16-point transforms
//           and 16 butterflies, the constant twiddles folded into the additions: fft16.h); twiddle
//           W2048^(n2 k1) as two per-lane factors (k1 = 4 q + b);
//   radix-2 the first (decimation-in-frequency) stage of the 64-point transforms over n2 pairs column
//           n2 with n2 + 32, 32 lanes apart: one v_permlane32_swap of rows k1 and k1 + 16 gives every
//           lane both inputs of its butterfly; u = x + y feeds the even outputs, v = (x - y) W64^n2
//           the odd ones;
//   exchange through LDS: 64 rows r = k1 + 32 parity of 32 values, pitch 33 (conflict-free both ways);
//   pass 2  lane r runs one 32-point FFT and holds Z[r + 64 k'];
//   split   of the pair Z[k], Z[2048 - k] that the real-FFT split needs, lane r holds Z[k] for k = r + 64 a,
//           a < 16, itself and fetches the other from the upper 16 values of lane 64 - r, which go
//           through LDS once (8.5 KB per frame instead of the whole spectrum both ways); split and power
//           in registers; a store instruction writes 64 consecutive bins, one per lane.
// Two forms of the frame loop (template parameter RUN): with a full window and a hop of 1024 a wave
// works through runs of 32 consecutive frames, loads every sample once and keeps the frame rotated in
// its registers; otherwise frames are dealt round-robin inside 8 contiguous ranges (one per group of
// workgroups that share an XCD) and the waves of a workgroup stay in step, so that the 3/4 of a frame
// that overlaps its neighbours is served by that XCD's L2.  Both are described where the loop starts.
#include <mutex>
#include <type_traits>
#include "stft_common.h"
#include "fft16.h"
#include <cstdlib>

namespace dsp {

struct R4096Args {
    long long ldx;
    long long N;
    long long ss0;
    long long rd_shift;
    const int *tbl;
    int L, W, stp, edge;
    int Wy;               // frames per signal in the output array
    Epilogue epi;
    int frames;           // B * W
    int B;                // signals
    unsigned *take;       // run form: 8 counters (one per blockIdx % 8), zero at launch
    double cscale;        // complex form: factor on every output value
};

// Counters of the run form: R4096_SLOTS sets of 8, used in turn by successive launches on a device (how a
// set is handed out: launch_r4096_inst).
constexpr int R4096_SLOTS = 64;
__device__ unsigned g_r4096_take[R4096_SLOTS * 8];

constexpr int R4096_PITCH = 33;                         // complex slots per exchange row
constexpr int R4096_WAVE_C = 64 * R4096_PITCH;          // complex slots per wave (>= 2048)
#ifndef R4096_NWAVES
#define R4096_NWAVES 8
#endif
// waves per workgroup (they share the window table): float64 takes twice the LDS and twice the registers per wave
template <typename T> constexpr int r4096_waves() { return sizeof(T) == 4 ? R4096_NWAVES : R4096_NWAVES / 2; }
// a pair of samples accessed at the alignment of one (frame starts are only aligned to a sample)
typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));
template <typename T> struct pair_u_of;
template <> struct pair_u_of<float> { using type = float2_u; };
template <> struct pair_u_of<double> { using type = double2_u; };
typedef unsigned u2v __attribute__((ext_vector_type(2)));

#ifndef R4096_ABLATE
#define R4096_ABLATE 0          // diagnostic builds: bit 0 = no row stores, bit 1 = no sample loads
#endif
#ifndef R4096_LOCKSTEP
#define R4096_LOCKSTEP 1        // 1: the waves of a workgroup meet at a barrier every R4096_LOCK_EVERY frames
#endif
#ifndef R4096_LOCK_EVERY
#define R4096_LOCK_EVERY 2      // power of two
#endif
#ifndef R4096_INTERLEAVE
#define R4096_INTERLEAVE 1         // 1: twiddle products of four rows are issued before the fmas that consume them
#endif
#ifndef R4096_ALT_PRIO
#define R4096_ALT_PRIO 0
#endif
#ifndef R4096_PHASE_PRIO
#define R4096_PHASE_PRIO 0      // bit 0: raised priority from the twiddle stage to the reads of pass 2, bit 1: from the Z writes to the end of the split's reads, bit 2: during the window step
#endif
#define R4096_PRIO(bit, v) do { if (R4096_PHASE_PRIO & (bit)) { __builtin_amdgcn_s_setprio(v); } } while (0)
#ifndef R4096_RUN
#define R4096_RUN 32            // frames per run of the run form
#endif
#ifndef R4096_NT_LOAD
#define R4096_NT_LOAD 1         // 1: the run form's sample loads carry the non-temporal hint (every sample is read once; measured: 0.8 %)
#endif
#ifndef R4096_NT_STORE
#define R4096_NT_STORE 0        // 1: rows are stored with the non-temporal hint
#endif
#ifndef R4096_TA_REG
#define R4096_TA_REG 0          // 1: run form keeps the twiddle stage's per-lane factors in registers instead of LDS
#endif
#ifndef R4096_LOADPOS
#define R4096_LOADPOS 0         // run form: where the next group's loads are issued: 0 = right after the window step, 1 = after the first transform, 2 = after the exchange
#endif
#ifndef R4096_TA_AHEAD
#define R4096_TA_AHEAD 0        // 1: run form reads the twiddle stage's per-lane factors right after the window step (measured: 1.7 % slower than reading them where they are used, which frees 14 registers)
#endif
#ifndef R4096_WREG_F64
#define R4096_WREG_F64 1        // 1: the plain run form keeps the window in registers in float64 as well (128 of the wave's 512)
#endif
#ifndef R4096_TW_FULL
#define R4096_TW_FULL 0         // 1: the plain float32 run form (window in registers, so 16 KB of LDS are free) reads the factor W2048^(n2 k1) of
                                // every row from a table in LDS: 31 products per frame instead of the 52 of the two-factor form; 2: the
                                // same with the reads of a group of four issued a group ahead.  Measured: 3.721 / 3.722 ms against 3.728 ms
                                // (42 fewer packed instructions per frame for 0.2 %: the arithmetic is no longer what bounds the kernel)
#endif
#ifndef R4096_HACK
#define R4096_HACK 0          // timing experiments only (wrong results): bit 1 = twiddle stage without the second factor, bit 3 = rows at a pitch of 2048 values (every store instruction on a 256-byte boundary)
#endif
#ifndef R4096_STAMPS
#define R4096_STAMPS 0          // diagnostic builds: 1 = per-phase s_memtime sums into g_r4096_stamps
#endif
#if R4096_STAMPS
__device__ unsigned long long g_r4096_stamps[2048 * 10];   // per wave: 8 phase sums, then the wave's whole life in shader cycles (s_memtime) and in ticks of the constant 100 MHz clock (s_memrealtime)
#define R4096_STAMP(i)                                                                            \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        st_acc[i] += t_ - st_last; st_last = t_;                                                  \
    } while (0)
#else
#define R4096_STAMP(i) do {} while (0)
#endif


constexpr int R4096_X2P = 17;                           // second exchange: complex slots per lane (16 values + 1)
// which instantiations keep the window in registers instead of LDS: the run form without mean removal, epilogue or
// complex rows (those need the registers themselves)
constexpr bool r4096_wreg(bool mn0, bool epi, bool run, bool cplx) { return run && !mn0 && !epi && !cplx; }
constexpr bool r4096_twfull(size_t elem, bool wreg) { return R4096_TW_FULL && wreg && elem == 4; }
constexpr size_t r4096_lds(size_t elem, int waves, bool wtab, bool run, bool twfull) { return 2 * elem * ((size_t)waves * R4096_WAVE_C + (twfull ? 64 * 32 : run ? 64 * 8 : 0)) + (wtab ? 4096 * elem : 0); }

// |a|^2 as a plain multiply and a plain fma: 2 + 2 issue cycles, where the packed square and the add of its halves take 4 + 2
__device__ __forceinline__ float norm2(const cxf a) { return fmaf(a.y, a.y, a.x * a.x); }
__device__ __forceinline__ double norm2(const cxd a) { return fma(a.y, a.y, a.x * a.x); }
// x = (lower half-wave: lo of its own lane | upper: lo of the lane 32 below), y = (lower: hi of the lane 32 above | upper: its own hi):
// v_permlane32_swap on every 32-bit half of the value
__device__ __forceinline__ void half_swap(const float lo, const float hi, float &x, float &y)
{
    const u2v s = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    x = __uint_as_float(s.x); y = __uint_as_float(s.y);
}
__device__ __forceinline__ void half_swap(const double lo, const double hi, double &x, double &y)
{
    const u2v a = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(lo), (unsigned)__double2loint(hi), false, false);
    const u2v b = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(lo), (unsigned)__double2hiint(hi), false, false);
    x = __hiloint2double((int)b.x, (int)a.x); y = __hiloint2double((int)b.y, (int)a.y);
}
// keeps a value in its registers across this point (no rematerialisation, nothing moved across)
__device__ __forceinline__ void pin(cxf &a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(cxd &a) { asm volatile("" : "+v"(a.x), "+v"(a.y)); }

template <int R> using rot = std::integral_constant<int, R>;

// CPLX: the rows are the complex spectra (2049 interleaved (re, im) pairs per frame, the imaginary parts of the first and
// last zero: window_univar followed by fft_fftw_r2hc, reference c/window_univar.c:41-89, c/fft.fftw.r2hc.c:69-78) instead of
// their powers.  In the run form the frame in the registers is the true one rotated by ROT quarters of its length, which
// multiplies bin k by (-i)^(ROT k): the complex form multiplies back by i^(ROT k) -- a swap and a sign, chosen at compile
// time, since a lane's bins are 4 lane + c + 256 m and 2048 minus those.
template <typename T, bool FULL, bool MN0, bool EPI, bool RUN, bool CPLX = false>
__global__ __launch_bounds__(64 * r4096_waves<T>(), r4096_waves<T>() / 4) void stft_r4096_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                                                                   const T *__restrict__ win, const T *__restrict__ tw,
                                                                                                   const R4096Args a)
{
    using C2 = cx<T>;
    using pair_u = typename pair_u_of<T>::type;
    constexpr int R4096_WAVES = r4096_waves<T>();
    constexpr int M = 2048, F = 2049;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    // The halved window, 4096 floats, in front of the tiles -- except in the plain run form (WREG), which has the
    // registers to keep a lane's 64 window values for the whole launch.
    // float64 has no LDS left for the table (4 tiles of 33.8 KB): its other instantiations read the window from memory
    // (32 KB that the L2 keeps) frame by frame.
    constexpr bool WREG = r4096_wreg(MN0, EPI, RUN, CPLX) && (sizeof(T) == 4 || R4096_WREG_F64);
    constexpr bool WTAB = !WREG && sizeof(T) == 4;
    constexpr bool TWFULL = r4096_twfull(sizeof(T), WREG);
    T *wtab = reinterpret_cast<T *>(smem_raw);
    C2 *tiles = reinterpret_cast<C2 *>(smem_raw) + (WTAB ? 2048 : 0);
    C2 *ex = tiles + wv * R4096_WAVE_C;                                               // this wave's tile

    if constexpr (WTAB) {
        for (int l = tid; l < 4096; l += 64 * R4096_WAVES) { wtab[l] = (l < L) ? T(0.5) * win[l] : T(0); }
    }

    // Lane roles.  Pass 1: the lane is the complex column n2 = lane and holds its 32 rows n1 (z[n2 + 64 n1] = samples
    // 128 n1 + 2 lane, + 1): an 8-byte load per row, 512 contiguous bytes per wave load.  Pass 2 and the split: the lane is
    // the row r = lane of the exchange tile (r = k1 + 32 parity) and holds Z[r + 64 k'], k' < 32.
    const int n2 = lane, jj = lane & 31, hh2 = lane >> 5;
    // per-lane twiddles: W2048^(n2 b), b = 1..3, and W2048^(4 n2 a), a = 1..7
    auto tw2048 = [&](int m) {                            // W2048^m = W4096^(2m), table holds k < 2048
        int idx = (2 * m) & 4095;
        T sg = T(1);
        if (idx >= 2048) { idx -= 2048; sg = T(-1); }
        return C2{sg * tw[2 * idx], sg * tw[2 * idx + 1]};
    };
    C2 tb[4], ta[8];
#pragma unroll
    for (int b = 1; b < 4; ++b) { tb[b] = tw2048(n2 * b); }
#pragma unroll
    for (int q = 1; q < 8; ++q) { ta[q] = tw2048(4 * n2 * q); }
    const C2 wl = C2{tw[2 * 64 * jj], tw[2 * 64 * jj + 1]};                   // W64^(n2 % 32)
    // Second exchange: lane l leaves Z[l + 64 (16 + i)] at slot 17 l + i.  The split pairs Z[r + 64 a] with
    // Z[2048 - r - 64 a] = Z[(64 - r) + 64 (31 - a)]: slot 15 - a of lane 64 - r; for r = 32 that is the lane itself; for
    // r = 0 the partner is Z[64 (32 - a)], slot 16 - a of the lane itself (a = 0, bin 0, pairs with itself: selected in the split).
    const int pslot = ((64 - lane) & 63) * R4096_X2P + (lane == 0 ? 1 : 0);
    const C2 ow = C2{tw[2 * lane + 1], -tw[2 * lane]};                        // -i W4096^r, for the split
    // RUN keeps a frame's raw samples (and, WREG, the window) in registers; the twiddles that depend on the lane alone and
    // are used once per frame (ta) then live in LDS, behind the tiles, and are read ahead of their use
    C2 *twl = tiles + R4096_WAVES * R4096_WAVE_C;                             // ta[q] at [64 q + lane]
    if constexpr (TWFULL) {
        // ... or, where the LDS has room, the whole factor of every row: W2048^(n2 k1) at [64 k1 + lane]
        for (int k1 = 1 + wv; k1 < 32; k1 += R4096_WAVES) { twl[64 * k1 + lane] = tw2048(n2 * k1); }
    } else if (RUN && wv == 0) {
#pragma unroll
        for (int q = 1; q < 8; ++q) { twl[64 * q + lane] = ta[q]; }
    }
    __syncthreads();

    // Work distribution, two forms.
    // Runs (RUN: full window, uniform hop of 1024 = a quarter of the frame): a wave takes runs of
    // R4096_RUN consecutive frames of one signal and reads it as a stream of groups of 1024 samples:
    // frame t of the run is groups t .. t + 3, and group g lives in the registers pre[8 (g % 4) ..].
    // Once the window step of frame t has read them, the 8 wave loads of group t + 4 go out into the
    // registers of group t, a whole frame before they are waited for.  The registers thus hold frame t
    // rotated by t % 4 quarters; the window is applied with the same rotation and the transform is taken
    // of the rotated frame, whose power spectrum is that of the frame itself (a circular shift by a
    // quarter of the length multiplies bin k by i^k).  Every sample is loaded once, no register moves,
    // and nothing depends on the L2 holding a neighbour's lines.
    // The runs are cut into 8 contiguous ranges by blockIdx % 8 and dealt round-robin inside a range.
    // Frames (otherwise): 8 ranges of frames, static round-robin; the 8 waves of a workgroup take 8
    // neighbouring frames per round and meet at a barrier every other round: without it they drift apart
    // by more rounds than the L2 holds, the 3/4 of its samples that a frame shares with its neighbours
    // miss, and 2.6x the samples come from HBM (measured with FETCH_SIZE).
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    constexpr int RL = R4096_RUN;                          // frames per run
    const int rps = (a.W + RL - 1) / RL;                   // runs per signal
    const int units = RUN ? a.B * rps : a.frames;
    const int f_lo = (int)((long long)units * xcd / 8), f_hi = (int)((long long)units * (xcd + 1) / 8);
    const int fstride = nwg_x * R4096_WAVES;
    int fid = f_lo + (blockIdx.x >> 3) * R4096_WAVES + wv;   // the wave's current frame, or (first) run
    int nfid = fid + fstride;
    if (fid >= f_hi) { return; }

    // where frame f lies: output row, first sample, and whether all 4096 samples can be read in place
    struct Where { long long row; const T *xb; long long ss, rdoff; bool inside; };
    auto locate = [&](int f) {
        Where q;
        const int b = f / a.W;
        const int w = f - b * a.W;
        q.row = (long long)b * a.Wy + w;
        q.xb = X + (long long)b * a.ldx;
        if (a.tbl) { q.ss = a.tbl[2 * w]; q.rdoff = q.ss - (long long)a.tbl[2 * w + 1]; }
        else { q.ss = a.ss0 + (long long)w * a.stp; q.rdoff = a.rd_shift; }
        if (!a.edge) { q.rdoff = 0; }
        q.inside = q.ss >= 0 && q.ss + 4096 <= a.N && q.ss >= q.rdoff;
        return q;
    };
    C2 ze[16], zo[16];      // even rows (ze) and odd rows (zo) of the column; later: the two halves of a 32-point transform
    // A frame's 32 loads go into registers of their own (the transform registers are busy to the end of the split):
    // frame form: issued under pass 2 of the frame before; run form: the four groups of 1024 samples of the frame
    C2 pre[32];
    const int loff = 2 * lane;                              // first sample of the lane inside a frame; row i adds 128
    // raw samples of a frame that lies inside its signal: 32 coalesced 512-byte wave loads
    auto load_inside = [&](const Where &q) {
        const T *src = q.xb + (q.ss - q.rdoff) + loff;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            pair_u v = {T(0), T(0)};
#if R4096_ABLATE & 2
            if (a.N == -12345)
#endif
            if (FULL || 128 * i < L) { v = *reinterpret_cast<const pair_u *>(src + 128 * i); }
            pre[i] = C2{v.x, v.y};
        }
    };
    // a frame that crosses an edge of its signal: gather sample by sample into the (idle) tile first
    auto load_edge = [&](const Where &q) {
        FrameDesc d{};
        d.N = a.N; d.edge = a.edge;
        T *stage = reinterpret_cast<T *>(ex);
        wave_sync();
        for (int l = lane; l < 4096; l += 64) { stage[l] = (l < L) ? fetch_sample(q.xb, d, q.ss, q.rdoff, l) : T(0); }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 32; ++i) { pre[i] = *reinterpret_cast<const C2 *>(stage + loff + 128 * i); }
        asm volatile("" ::: "memory");                 // keeps these LDS reads from being merged with load_inside's into flat loads
    };

    // RUN: the current run (signal, first frame, frames) and the frame inside it
    int run_b = 0, run_w0 = 0, run_len = 0, tpos = 0;
    auto open_run = [&](int r) {
        run_b = r / rps;
        run_w0 = (r - run_b * rps) * RL;
        run_len = min(RL, a.W - run_w0);
    };
    // Group g of the run (b, w0): where its 1024 samples lie.  A group that crosses an edge of the
    // signal is gathered sample by sample into the tile first, which is idle at the call sites.
    struct Group { const T *src; bool inside; };
    auto find_group = [&](int b, int w0, int g) {
        const T *xb = X + (long long)b * a.ldx;
        const long long ssg = a.ss0 + (long long)w0 * a.stp + 1024LL * g;
        const long long rdoff = a.edge ? a.rd_shift : 0;
        Group q;
        q.inside = ssg >= 0 && ssg + 1024 <= a.N && ssg >= rdoff;
        q.src = xb + (ssg - rdoff) + loff;
        if (!q.inside) {
            FrameDesc d{};
            d.N = a.N; d.edge = a.edge;
            T *stage = reinterpret_cast<T *>(ex);
            wave_sync();
            for (int l = lane; l < 1024; l += 64) { stage[l] = fetch_sample(xb, d, ssg, rdoff, l); }
            wave_sync();
        }
        return q;
    };
    // ... and into the registers of group slot j: 8 coalesced 512-byte wave loads (or the staged copy)
    auto fetch_group = [&](const Group &q, const int j) {
        if (q.inside) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#if R4096_NT_LOAD
                const pair_u v = __builtin_nontemporal_load(reinterpret_cast<const pair_u *>(q.src + 128 * r));
#else
                const pair_u v = *reinterpret_cast<const pair_u *>(q.src + 128 * r);
#endif
                pre[8 * j + r] = C2{v.x, v.y};
            }
        } else {
            const T *stage = reinterpret_cast<const T *>(ex);
#pragma unroll
            for (int r = 0; r < 8; ++r) { pre[8 * j + r] = *reinterpret_cast<const C2 *>(stage + loff + 128 * r); }
            asm volatile("" ::: "memory");
        }
    };
    // the first frame of run r
    auto load_run_head = [&](int r) {
        const int b = r / rps, w0 = (r - b * rps) * RL;
#pragma unroll
        for (int g = 0; g < 4; ++g) { fetch_group(find_group(b, w0, g), g); }
    };

    // WREG: the lane's 64 values of the halved window stay in registers for the whole launch (the window is full:
    // L = 4096); a frame rotated by ROT quarters takes them in the same rotation, a renaming at compile time
    C2 wreg[WREG ? 32 : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const pair_u w2 = *reinterpret_cast<const pair_u *>(win + loff + 128 * i);
            wreg[i] = C2{T(0.5) * w2.x, T(0.5) * w2.y};
        }
    }
    Where cur{};
    if constexpr (!RUN) {
        cur = locate(fid);
        if (cur.inside) { load_inside(cur); } else { load_edge(cur); }
    }
#if R4096_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0) :: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
    const unsigned long long st_t0 = st_last;
#endif

    // forward 32-point DFT of x[n], n < 32, given as even part e[m] = x[2m] and odd part o[m] = x[2m+1]; on return X[k]
    // is in e[slot16(k)] for k < 16 and in o[slot16(k - 16)] for k >= 16.  Twiddles folded into the additions (fft16.h).
    auto dft32f = [](C2 (&e)[16], C2 (&o)[16]) __attribute__((always_inline)) {
        dft16f<T, 16>(e);
        dft16f<T, 16>(o);
#define R4096_B(K) bfly_tw<T, 32, K>(e[slot16(K)], o[slot16(K)]);
        R4096_B(0) R4096_B(1) R4096_B(2) R4096_B(3) R4096_B(4) R4096_B(5) R4096_B(6) R4096_B(7)
        R4096_B(8) R4096_B(9) R4096_B(10) R4096_B(11) R4096_B(12) R4096_B(13) R4096_B(14) R4096_B(15)
#undef R4096_B
    };

    // One frame: transform the samples in the registers, start the next frame's loads, write the row.
    // Returns false after the last frame (RUN: of the run).  The first frame runs in a copy of its own, so that inside
    // the loop every wait on the prefetched samples sees the same sequence of loads followed by this
    // frame's stores and lets the stores drain in the background.
    int iter = 0;
    auto frame = [&](auto rot_c) __attribute__((always_inline)) -> bool {
        constexpr int ROT = decltype(rot_c)::value;          // RUN: tpos % 4, the rotation of the frame in the registers
#if R4096_ALT_PRIO
        if constexpr (RUN) {
            // The two waves of a SIMD are arbitrated oldest first: left alone, waves 0-3 of the workgroup
            // run 40 % faster than waves 4-7 and finish their (equal) share early.  Alternating the
            // priority frame by frame gives both the same pace.
            if (wv >> 2) { __builtin_amdgcn_s_setprio(1 - (ROT & 1)); } else { __builtin_amdgcn_s_setprio(ROT & 1); }
        }
#endif
        // ---------------- pass 1: window, 32-point FFT over n1 ----------------
        R4096_PRIO(4, 1);
        {
            C2 acc = C2{T(0), T(0)};
            int lim = L - loff;                              // sample l = 128 i + loff + c is inside the window iff 128 i + c < lim
            asm volatile("" : "+v"(lim));                    // recomputed per frame: hoisted lane masks would not fit in SGPRs
            // RUN: register group j (rows 8 j .. 8 j + 7) holds the quarter (j - tpos) % 4 of the frame
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int wi = 8 * (((i >> 3) - ROT) & 3) + (i & 7);      // the row's place in the frame
                C2 hw;
                if constexpr (WREG) { hw = wreg[wi]; }
                else if constexpr (WTAB) { hw = *reinterpret_cast<const C2 *>(wtab + loff + 128 * wi); }
                else {
                    const T *wp = win + loff + 128 * wi;
                    if constexpr (FULL) { const pair_u w2 = *reinterpret_cast<const pair_u *>(wp); hw = C2{T(0.5) * w2.x, T(0.5) * w2.y}; }
                    else { hw = C2{(128 * wi < lim) ? T(0.5) * wp[0] : T(0), (128 * wi + 1 < lim) ? T(0.5) * wp[1] : T(0)}; }
                }
                C2 v = pre[i];
                if (!FULL) {                                 // keep data past the window out (0 * NaN)
                    v.x = (128 * i < lim) ? v.x : T(0);
                    v.y = (128 * i + 1 < lim) ? v.y : T(0);
                }
                v = v * hw;
                if (MN0) { acc += v; }
                if (i & 1) { zo[i >> 1] = v; } else { ze[i >> 1] = v; }
            }
            if (MN0) {
                T sum = acc.x + acc.y;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { sum += __shfl_xor(sum, o); }
                const T mean = sum / (T)L;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const C2 mm = C2{(FULL || 128 * i < lim) ? mean : T(0), (FULL || 128 * i + 1 < lim) ? mean : T(0)};
                    if (i & 1) { zo[i >> 1] -= mm; } else { ze[i >> 1] -= mm; }
                }
            }
        }
        if constexpr (RUN) {
            // the windowed values are complete here: without this the products sink below the loads that
            // follow, the old and the new samples are both live, and the frame no longer fits the registers
#pragma unroll
            for (int i = 0; i < 16; ++i) { pin(ze[i]); pin(zo[i]); }
        }
        // RUN: the per-lane factors of the twiddle stage come from LDS; asked for here, two transforms ahead
        // of their use (read where they are used, each of the 7 reads was waited for on the spot)
        C2 taq_r[8];
        if constexpr (RUN && R4096_TA_AHEAD) {
#pragma unroll
            for (int q = 1; q < 8; ++q) { taq_r[q] = twl[64 * q + lane]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        R4096_STAMP(6);
        R4096_PRIO(4, 0);
        // RUN: the stream moves on by one group, whose loads go out into the registers of the group that the next
        // frame no longer holds
        auto next_group = [&]() __attribute__((always_inline)) {
            if constexpr (RUN) {
                if (tpos + 1 < run_len) {
                    const Group q = find_group(run_b, run_w0, tpos + 4);
                    fetch_group(q, ROT);
                }
            }
        };
        if (R4096_LOADPOS == 0) { next_group(); }
        R4096_STAMP(7);
        // the column's 32-point transform, in registers: X[k1] in ze[slot16(k1)], X[k1 + 16] in zo[slot16(k1)]
        dft32f(ze, zo);
        if (R4096_LOADPOS == 1) { next_group(); }
        R4096_STAMP(0);
        R4096_PRIO(1, 1);

        // twiddle W2048^(n2 k1), k1 = 4 q + b
#if R4096_INTERLEAVE
        // four rows at a time, the products of a group before the fmas that consume them
        [[maybe_unused]] C2 wnext[4];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            C2 *zp[4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) { const int k1 = 4 * q + bb; zp[bb] = (k1 < 16) ? &ze[slot16(k1)] : &zo[slot16(k1 - 16)]; }
            C2 t[4];
            if constexpr (TWFULL) {
                // the factors of group q + 1 are asked for before the products of group q are formed
#if R4096_TW_FULL == 2
                if (q == 0) {
#pragma unroll
                    for (int bb = 1; bb < 4; ++bb) { wnext[bb] = twl[64 * bb + lane]; }
                }
                C2 wq[4];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) { wq[bb] = wnext[bb]; }
                if (q < 7) {
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) { wnext[bb] = twl[64 * (4 * q + 4 + bb) + lane]; }
                }
#else
                C2 wq[4];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) { if (4 * q + bb) { wq[bb] = twl[64 * (4 * q + bb) + lane]; } }
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) { if (4 * q + bb) { t[bb] = cx_mul_first<T>(*zp[bb], wq[bb]); } }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) { if (4 * q + bb) { *zp[bb] = cx_mul_second<T>(*zp[bb], wq[bb], t[bb]); } }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
#pragma unroll
            for (int bb = 1; bb < 4; ++bb) { t[bb] = cx_mul_first<T>(*zp[bb], tb[bb]); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bb = 1; bb < 4; ++bb) { *zp[bb] = cx_mul_second<T>(*zp[bb], tb[bb], t[bb]); }
            if (q && !(R4096_HACK & 2)) {
                const C2 taq = (RUN && !R4096_TA_REG) ? (R4096_TA_AHEAD ? taq_r[q] : twl[64 * q + lane]) : ta[q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) { t[bb] = cx_mul_first<T>(*zp[bb], taq); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) { *zp[bb] = cx_mul_second<T>(*zp[bb], taq, t[bb]); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            C2 &z = (k1 < 16) ? ze[slot16(k1)] : zo[slot16(k1 - 16)];
            const int q = k1 >> 2, bb = k1 & 3;
            if (bb) { z = cx_mul<T>(z, tb[bb]); }
            if (q) { z = cx_mul<T>(z, RUN ? twl[64 * q + lane] : ta[q]); }
        }
#endif
        // first radix-2 stage of the 64-point transforms over n2: columns n2 and n2 + 32 sit 32 lanes
        // apart; one swap of rows k1 and k1 + 16 leaves the lower half-wave with row k1
        // (x = column n2, y = column n2 + 32) and the upper one with row k1 + 16
        wave_sync();                                       // the previous frame's reads of the tile are done
        {
            // tile row r = k1 + 32 parity, column n2 % 32 at its own position
            C2 *dst = ex + (16 * hh2) * R4096_PITCH + jj;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                const C2 lo = ze[slot16(k1)], hi = zo[slot16(k1)];
                T xr, yr_, xi, yi;
                half_swap(lo.x, hi.x, xr, yr_);
                half_swap(lo.y, hi.y, xi, yi);
                const C2 x = C2{xr, xi}, y = C2{yr_, yi};
                dst[k1 * R4096_PITCH] = x + y;
                dst[(32 + k1) * R4096_PITCH] = cx_mul<T>(x - y, wl);
            }
        }
        wave_sync();
        if (R4096_LOADPOS == 2) { next_group(); }
        // where the next frame lies; its loads go out now, under pass 2
        bool more;
        Where nxt;
        if constexpr (RUN) {
            more = false; nxt = cur;
        } else {
            more = nfid < f_hi;
            nxt = locate(more ? nfid : fid);
#if R4096_LOCKSTEP
            if ((iter++ & (R4096_LOCK_EVERY - 1)) == 0) { __builtin_amdgcn_s_barrier(); }   // the workgroup's 8 waves request neighbouring frames together
#endif
            if (nxt.inside) { load_inside(nxt); }
        }
        R4096_STAMP(1);

        // ---------------- pass 2: the lane's row r = lane: 32-point FFT over the 32 columns ----------------
        {
            const C2 *row = ex + lane * R4096_PITCH;
#pragma unroll
            for (int m = 0; m < 16; ++m) { ze[m] = row[2 * m]; zo[m] = row[2 * m + 1]; }   // even / odd columns
        }
        R4096_PRIO(1, 0);
        dft32f(ze, zo);
        R4096_STAMP(2);
        R4096_PRIO(2, 1);
        wave_sync();                                       // every lane has its row: the tile is free
        // second exchange: the upper half of the lane's values, Z[r + 64 (16 + i)], for its partner
        {
            C2 *x2w = ex + lane * R4096_X2P;
#pragma unroll
            for (int i = 0; i < 16; ++i) { x2w[i] = zo[slot16(i)]; }
        }
        wave_sync();
        R4096_STAMP(3);
        R4096_STAMP(4);
        // ---------------- split + power + store: the lane's bins k = r + 64 a, a < 16, and 2048 - k ----------------
#if R4096_HACK & 8
        // timing only: rows at a pitch of 2048 values, every store instruction on a 256-byte boundary
        T *y = Y + (RUN ? (long long)run_b * a.Wy + run_w0 + tpos : cur.row) * 2048;
#else
        T *y = Y + (RUN ? (long long)run_b * a.Wy + run_w0 + tpos : cur.row) * (CPLX ? 2 * F : F);
#endif
        const C2 *zpart = ex + pslot;                                      // Z[2048 - k] at zpart[15 - a]
        T *yf = y + (CPLX ? 2 : 1) * lane, *yr = y + (CPLX ? 2 : 1) * (M - ((R4096_HACK & 8) ? 1 : 0) - lane);
        [[maybe_unused]] C2 cu = C2{T(0), T(0)};
        if constexpr (CPLX) {
            // the rotation of the run form undone: bin k times i^(ROT k) = i^(ROT r), bin 2048 - k times its conjugate
            constexpr int RC = RUN ? ROT : 0;
            const int e = (RC * lane) & 3;
            const T sc = (T)a.cscale;
            cu = C2{e == 0 ? sc : (e == 2 ? -sc : T(0)), e == 1 ? sc : (e == 3 ? -sc : T(0))};
        }
        // Four pairs at a time, stage by stage (no instruction feeds the one right behind it).  The factor of the
        // split, -i W4096^k = (-i W4096^r) W64^a, is taken in two parts: the constant W64^a in folded form (fft16.h:
        // a sum before the product with the lane's factor, a real factor and a power of i inside the last additions).
        auto pairs4 = [&](auto g_c) __attribute__((always_inline)) {
            constexpr int G = decltype(g_c)::value;
            constexpr TwF f[4] = {twf_n(4 * G, 64), twf_n(4 * G + 1, 64), twf_n(4 * G + 2, 64), twf_n(4 * G + 3, 64)};
            C2 A[4], B[4], t[4], P[4], Q[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int aa = 4 * G + c;
                const C2 z = ze[slot16(aa)];
                C2 zm = zpart[15 - aa];
                if (aa == 0) { zm = (lane == 0) ? z : zm; }                // bin 0 pairs with itself
                A[c] = cx_add_conj<T>(z, zm);
                B[c] = cx_sub_conj<T>(z, zm);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { if (f[c].t != 0.0) { B[c] = cx_pre<T>(B[c], (T)f[c].t); } }
#pragma unroll
            for (int c = 0; c < 4; ++c) { t[c] = cx_mul_first<T>(B[c], ow); }
#pragma unroll
            for (int c = 0; c < 4; ++c) { B[c] = cx_mul_second<T>(B[c], ow, t[c]); }
            // X_k = A + T, X_(2048-k) = conj(A - T), T = sig i^m B
#define R4096_PQ(c) P[c] = cx_axpy_i<T, f[c].m>(A[c], (T)f[c].sig, B[c]); Q[c] = cx_axpy_i<T, f[c].m + 2>(A[c], (T)f[c].sig, B[c]);
            R4096_PQ(0) R4096_PQ(1) R4096_PQ(2) R4096_PQ(3)
#undef R4096_PQ
            if constexpr (CPLX) {
                constexpr int RC = RUN ? ROT : 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int aa = 4 * G + c;
                    C2 u, v;
                    if constexpr (RC == 0) { u = P[c] * C2{cu.x, cu.x}; v = Q[c] * C2{cu.x, -cu.x}; }
                    else { u = cx_mul<T>(P[c], cu); v = cx_mul<T>(Q[c], cu) * C2{T(1), T(-1)}; }
                    if (aa == 0) { u.y = (lane == 0) ? T(0) : u.y; v.y = (lane == 0) ? T(0) : v.y; }   // bins 0 and 2048
                    *reinterpret_cast<pair_u *>(yf + 128 * aa) = pair_u{u.x, u.y};
                    *reinterpret_cast<pair_u *>(yr - 128 * aa) = pair_u{v.x, v.y};
                }
            } else {
                T p[4], q[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { p[c] = norm2(P[c]); q[c] = norm2(Q[c]); }
                if (EPI) { power_epilogue_n(p, a.epi); power_epilogue_n(q, a.epi); }
#if R4096_ABLATE & 1
                asm volatile("" :: "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
                if (a.N == -12345)
#endif
                {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#if R4096_NT_STORE
                        __builtin_nontemporal_store(p[c], yf + 64 * (4 * G + c));
                        __builtin_nontemporal_store(q[c], yr - 64 * (4 * G + c));
#else
                        yf[64 * (4 * G + c)] = p[c];
                        yr[-64 * (4 * G + c)] = q[c];
#endif
                    }
                }
            }
        };
        pairs4(rot<0>{}); pairs4(rot<1>{}); pairs4(rot<2>{}); pairs4(rot<3>{});
        {
            // bin 1024 = row 0, k' = 16: the first value lane 0 left in the second exchange.  It pairs with itself (Z
            // holds half values).  Every lane stores the same value to the same address: a store under
            // `if (lane == 0)` sits behind a branch, the count of stores after the prefetch loads is then
            // unknown at compile time, and the wait for the loads also waits for the first store of this row.
            const C2 zmid = ex[0];
            if constexpr (CPLX) {
                // bin 1024 is 2 conj Z_1024 (times i^(ROT 1024) = 1)
                *reinterpret_cast<pair_u *>(y + 2048) = pair_u{T(2) * (T)a.cscale * zmid.x, T(-2) * (T)a.cscale * zmid.y};
            }
            T p = T(4) * norm2(zmid);
            if (EPI) { p = power_epilogue(p, a.epi); }
#if R4096_ABLATE & 1
            asm volatile("" :: "v"(p));
            if (a.N == -12345)
#endif
            if constexpr (!CPLX) { y[1024] = p; }
        }
        R4096_STAMP(5);
        R4096_PRIO(2, 0);
        if constexpr (RUN) {
            if (tpos + 1 >= run_len) { return false; }
            ++tpos;
        } else {
            if (!more) { return false; }
            if (!nxt.inside) { load_edge(nxt); }
            cur = nxt;
            fid = nfid;
            nfid += fstride;
        }
        return true;
    };
#if R4096_STAMPS
    if constexpr (RUN) {
        for (; fid < f_hi;) {
            // The wave's next run is the next one nobody has taken yet in this range (the waves of a SIMD
            // do not run at the same pace: a static deal leaves the faster ones idle at the end).  Asked
            // for before this run's loads, so that the wait for those covers the answer as well.
            unsigned took = 0;
            if (lane == 0) { took = __hip_atomic_fetch_add(a.take + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            open_run(fid);
            load_run_head(fid);
            asm volatile("" : "+v"(took) : "v"(pre[7].y), "v"(pre[15].y), "v"(pre[23].y), "v"(pre[31].y));
            const int nrun = f_lo + fstride + (int)__builtin_amdgcn_readfirstlane(took);
            tpos = 0;
            while (frame(rot<0>{}) && frame(rot<1>{}) && frame(rot<2>{}) && frame(rot<3>{})) { }
            fid = nrun;
        }
    } else {
        if (frame(rot<0>{})) { while (frame(rot<0>{})) { } }
    }
    unsigned long long st_t1, st_rt1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t1) :: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1) :: "memory");
    if (lane == 0) {
        const int gw = blockIdx.x * R4096_WAVES + wv;
        if (gw < 2048) {
            for (int i = 0; i < 8; ++i) { g_r4096_stamps[gw * 10 + i] = st_acc[i]; }
            g_r4096_stamps[gw * 10 + 8] = st_t1 - st_t0;
            g_r4096_stamps[gw * 10 + 9] = st_rt1 - st_rt0;
        }
    }
#else
    if constexpr (RUN) {
        // Run by run: the first frame's samples (16 wave loads, waited for at once, so that the waits
        // inside the loop count only the loop's own loads and stores), then its frames, four to a round:
        // the rotation of a frame is then a constant of its copy of the code.
        for (; fid < f_hi;) {
            // The wave's next run is the next one nobody has taken yet in this range (the waves of a SIMD
            // do not run at the same pace: a static deal leaves the faster ones idle at the end).  Asked
            // for before this run's loads, so that the wait for those covers the answer as well.
            unsigned took = 0;
            if (lane == 0) { took = __hip_atomic_fetch_add(a.take + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            open_run(fid);
            load_run_head(fid);
            asm volatile("" : "+v"(took) : "v"(pre[7].y), "v"(pre[15].y), "v"(pre[23].y), "v"(pre[31].y));
            const int nrun = f_lo + fstride + (int)__builtin_amdgcn_readfirstlane(took);
            tpos = 0;
            while (frame(rot<0>{}) && frame(rot<1>{}) && frame(rot<2>{}) && frame(rot<3>{})) { }
            fid = nrun;
        }
    } else {
        if (!frame(rot<0>{})) { return; }
        while (frame(rot<0>{})) { }
    }
#endif
}
#if R4096_STAMPS
extern "C" int dsp_debug_read_stamps4096(unsigned long long *dst, size_t n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_r4096_stamps), n * sizeof(unsigned long long));
}
#endif


bool r4096_supported(const FrameDesc &d)
{
    return d.nfft == 4096 && d.L >= 1 && d.L <= 4096 && (long long)d.B * d.wy() < (1ll << 31) - 4096;
}

// Counters of the run form, one state for every instantiation: the set a launch uses, cleared in the
// launch's stream just before it.  A set is handed out again only behind an event recorded after the
// launch that last used it (the new launch's stream waits for that event: no host synchronisation), so two
// launches never share counters, however many are in flight and whatever their flags.
namespace {
std::mutex g_take_mu;
unsigned *g_take_base[64];
unsigned g_take_turn[64];
hipEvent_t g_take_ev[64][R4096_SLOTS];
bool g_take_busy[64][R4096_SLOTS];
}  // namespace

template <typename T, bool FULL, bool MN0, bool EPI, bool RUN, bool CPLX = false>
static hipError_t launch_r4096_inst(T *Y, const T *X, const T *win, const T *tw, const R4096Args &a, hipStream_t s, unsigned *take)
{
    constexpr int WAVES = r4096_waves<T>();
    const size_t lds = r4096_lds(sizeof(T), WAVES, sizeof(T) == 4 && !r4096_wreg(MN0, EPI, RUN, CPLX), RUN, r4096_twfull(sizeof(T), r4096_wreg(MN0, EPI, RUN, CPLX)));
    const auto kernel = &stft_r4096_kernel<T, FULL, MN0, EPI, RUN, CPLX>;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(kernel), lds, grant); e != hipSuccess) { return e; }
    long long grid = ((long long)a.frames + WAVES - 1) / WAVES;
    if (grid > 256) { grid = 256; }                      // one workgroup per CU (135 to 152 KB of LDS each)
    grid = (grid + 7) & ~7ll;
    R4096Args b = a;
    if (RUN && take) {
        // the caller's counters (it answers for launches that overlap)
        b.take = take;
        if (hipError_t e = hipMemsetAsync(b.take, 0, 8 * sizeof(unsigned), s); e != hipSuccess) { return e; }
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * WAVES), lds, s, Y, X, win, tw, b);
        return hipGetLastError();
    }
    if (RUN) {
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) { return e; }
        if (dev < 0 || dev >= 64) { return hipErrorInvalidDevice; }
        std::lock_guard<std::mutex> lock(g_take_mu);
        if (!g_take_base[dev]) {
            if (hipError_t e = hipGetSymbolAddress(reinterpret_cast<void **>(&g_take_base[dev]), HIP_SYMBOL(g_r4096_take)); e != hipSuccess) { return e; }
        }
        const int slot = (int)(g_take_turn[dev]++ % R4096_SLOTS);
        hipEvent_t &ev = g_take_ev[dev][slot];
        if (!ev) {
            if (hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); e != hipSuccess) { return e; }
        } else if (g_take_busy[dev][slot]) {
            if (hipError_t e = hipStreamWaitEvent(s, ev, 0); e != hipSuccess) { return e; }
        }
        b.take = g_take_base[dev] + 8 * slot;
        if (hipError_t e = hipMemsetAsync(b.take, 0, 8 * sizeof(unsigned), s); e != hipSuccess) { return e; }
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * WAVES), lds, s, Y, X, win, tw, b);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) { return e; }
        g_take_busy[dev][slot] = true;
        return hipEventRecord(ev, s);
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * WAVES), lds, s, Y, X, win, tw, b);
    return hipGetLastError();
}

// the run form: full window, frames at ss0 + w * 1024, where there are at least DSP_R4096_MIN_RUNS runs (default: two
// per wave of a full grid; 0 = never)
template <typename T>
static bool r4096_runs(const FrameDesc &d)
{
    long long min_runs = 2 * 256 * r4096_waves<T>();
    if (const char *e = getenv("DSP_R4096_MIN_RUNS")) { min_runs = atoll(e) > 0 ? atoll(e) : (1ll << 62); }
    const long long runs = (long long)d.B * ((d.W + R4096_RUN - 1) / R4096_RUN);
    return d.L == 4096 && d.tbl == nullptr && d.stp == 1024 && runs >= min_runs;
}

template <typename T>
static hipError_t launch_r4096_any(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, hipStream_t s, unsigned *take)
{
    R4096Args a{};
    a.ldx = d.ldx; a.N = d.N; a.ss0 = d.ss0; a.rd_shift = d.rd_shift; a.tbl = d.tbl;
    a.L = d.L; a.W = d.W; a.Wy = d.wy(); a.stp = d.stp; a.edge = d.edge; a.epi = d.epi;
    a.frames = d.B * d.W;
    a.B = d.B;
    const bool full = d.L == 4096, epi = d.epi.any();
#define DSP_R4096(F_, M_, E_, R_) launch_r4096_inst<T, F_, M_, E_, R_>(Y, X, win, tw, a, s, take)
    if (r4096_runs<T>(d)) {
        if (d.mn0) { return epi ? DSP_R4096(true, true, true, true) : DSP_R4096(true, true, false, true); }
        return epi ? DSP_R4096(true, false, true, true) : DSP_R4096(true, false, false, true);
    }
    if (full) {
        if (d.mn0) { return epi ? DSP_R4096(true, true, true, false) : DSP_R4096(true, true, false, false); }
        return epi ? DSP_R4096(true, false, true, false) : DSP_R4096(true, false, false, false);
    }
    if (d.mn0) { return epi ? DSP_R4096(false, true, true, false) : DSP_R4096(false, true, false, false); }
    return epi ? DSP_R4096(false, false, true, false) : DSP_R4096(false, false, false, false);
#undef DSP_R4096
}

hipError_t launch_stft_r4096(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s, unsigned *take)
{
    return launch_r4096_any<float>(Y, X, win, tw, d, s, take);
}
hipError_t launch_stft_r4096(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, hipStream_t s, unsigned *take)
{
    return launch_r4096_any<double>(Y, X, win, tw, d, s, take);
}

// The complex form: Y[b][w][2049] interleaved (re, im) pairs times `scale`; equally spaced frames, no mean removal.
bool r4096_cplx_supported(const FrameDesc &d)
{
    return r4096_supported(d) && d.tbl == nullptr && !d.mn0 && !d.epi.any() && (long long)d.B * d.wy() < (1ll << 30);
}

template <typename T>
static hipError_t launch_r4096_cplx_any(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    R4096Args a{};
    a.ldx = d.ldx; a.N = d.N; a.ss0 = d.ss0; a.rd_shift = d.rd_shift; a.tbl = nullptr;
    a.L = d.L; a.W = d.W; a.Wy = d.wy(); a.stp = d.stp; a.edge = d.edge; a.epi = d.epi;
    a.frames = d.B * d.W;
    a.B = d.B;
    a.cscale = scale;
    if (r4096_runs<T>(d)) { return launch_r4096_inst<T, true, false, false, true, true>(Y, X, win, tw, a, s, nullptr); }
    if (d.L == 4096) { return launch_r4096_inst<T, true, false, false, false, true>(Y, X, win, tw, a, s, nullptr); }
    return launch_r4096_inst<T, false, false, false, false, true>(Y, X, win, tw, a, s, nullptr);
}

hipError_t launch_stft_r4096_cplx(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    return launch_r4096_cplx_any<float>(Y, X, win, tw, d, scale, s);
}
hipError_t launch_stft_r4096_cplx(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    return launch_r4096_cplx_any<double>(Y, X, win, tw, d, scale, s);
}

}  // namespace dsp
