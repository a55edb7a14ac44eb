// Fused STFT kernel for nfft = 256, 128 and 64, float32 and float64 (gfx950).
//
// The kernel is written for CL lanes per frame, nfft = 32 CL: CL = 8 (nfft 256, described below), 4 (nfft 128: 64 complex
// points as 16 x 4 on 4 lanes, 16 frames per wave and step, pass 2 four 4-point FFTs per lane) and 2 (nfft 64: 16 x 2 on
// 2 lanes, 32 frames per step, pass 2 eight 2-point butterflies per lane).
//
// A frame's 256 reals are 128 complex points z_n = x_2n + i x_2n+1, transformed as 16 x 8 on 8 lanes:
//   pass 1  lane (frame f, column n2 = j < 8) holds the rows n1 < 16, z_(j + 8 n1): window (taps in registers),
//           a register-resident 16-point FFT over n1, twiddle W128^(j k1);
//   exchange through LDS, 16 rows k1 of 8 values per frame (row pitch 9, frame pitch 152: conflict-free);
//   pass 2  the lane takes the two rows k1 = j and k1 = j + 8, an 8-point FFT over n2 each, and holds
//           Z[k1 + 16 k2], k2 < 8;
//   split   the partner Z[128 - K] of K = k1 + 16 k2 is register 7 - k2 of the OTHER row of lane (8 - j) % 8
//           (lane 0: its own rows): a DPP row_half_mirror and a row_shr:1, fused with the power step.
// A wavefront transforms 8 frames per step.  The schedule around the arithmetic is that of the 512-point
// kernel: one workgroup of 8 waves per CU whose waves draw steps from a counter in LDS; the span of the next
// step (7 hop + 256 samples at most 2048) comes by LDS-DMA into a slot of the wave's own; rows are staged
// through LDS in output order, shifted so that the 16-byte boundaries of the output fall on 16-byte boundaries of
// the LDS, and leave as 16-byte non-temporal stores: six store instructions per step whatever the alignment.
// float64: the same kernel with 4 waves per workgroup (a wave's tile and span slot take twice the bytes; one wave per
// SIMD also has the registers for the 16 complex values, the twiddles and the window taps in double) and rows stored
// element-wise, 17 store instructions per step.
#include "stft_common.h"
#include "fft16.h"
#include "span_ring.h"

namespace dsp {

struct R256Args {
    long long ldx;        // elements between signals
    int N;                // samples per signal
    int L;                // window length
    int W;                // frames per signal
    int stp;              // hop
    int ss0;              // start of frame 0
    int Wy;               // frames per signal in the output array
    int rd_shift;         // edge == 1: samples are read at n - rd_shift
    int edge;             // 0 reflect, 1 zero
    Epilogue epi;
    int sps;              // steps (groups of 8 consecutive frames) per signal
    FastDiv spd;          // ... as a divisor
    int steps;            // B * sps
    const int *tbl;       // table form: start of frame w at tbl[2 w] (frames need not be equally spaced); else nullptr
    int reach;            // table form: bound on the distance between the starts of the first and last frame of a step
    double cscale;        // complex form: factor on every output value
};

// Exchange tile of a wave: per frame 16 rows k1 of CL values at row pitch CL + 1; the frame pitch (complex slots) makes
// both the writes of pass 1 (lane (f, j) at f FP + j + k1 (CL + 1)) and the reads of pass 2 (f FP + (j + CL r)(CL + 1) + n2)
// fall on 32 different 8-byte banks over a half-wave.
__host__ __device__ constexpr int r256_fp(int CL) { return CL == 8 ? 152 : CL == 4 ? 84 : 50; }
__host__ __device__ constexpr int r256_tile(int CL) { return (64 / CL) * r256_fp(CL) * 2; }   // floats: exchange tile / row staging
template <typename T> __host__ __device__ constexpr int r256_waves() { return sizeof(T) == 4 ? 8 : 4; }   // per workgroup

#ifndef R256_MINIMAL
#define R256_MINIMAL 0          // diagnostic builds: 1 = only the plain instantiations
#endif

__device__ __forceinline__ float r256_half_mirror(float v)   // lane 7 - i of the same 8 lanes
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
}
__device__ __forceinline__ float r256_shr1(float v)          // lane i - 1 of the same row of 16 (lane 0 of a row: 0)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}

// lane (CL - j) % CL of the lane's group of CL (a frame's lanes); the value that lane j == 0 gets is not used for CL == 8
template <int CL> __device__ __forceinline__ float r256_grp_neg(float v)
{
    if constexpr (CL == 8) { return r256_shr1(r256_half_mirror(v)); }
    else if constexpr (CL == 4) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x6C, 0xf, 0xf, true)); }   // quad_perm:[0,3,2,1]
    else { return v; }
}

template <int CL> __device__ __forceinline__ double r256_grp_neg(double v)   // the two halves of a double move separately
{
    const float lo = r256_grp_neg<CL>(__int_as_float(__double2loint(v))), hi = r256_grp_neg<CL>(__int_as_float(__double2hiint(v)));
    return __hiloint2double(__float_as_int(hi), __float_as_int(lo));
}

// forward 8-point DFT; X[k] is left in a[slot8(k)]
__host__ __device__ constexpr int slot8(int k) { return k < 4 ? 2 * k : 2 * (k - 4) + 1; }
template <typename T>
__device__ __forceinline__ void dft8c(cx<T> (&a)[8])
{
    constexpr T R2 = T(0.70710678118654752440084436210484904L);
    dft4c<T>(a[0], a[2], a[4], a[6]);                // E[k] in a[2 k]
    dft4c<T>(a[1], a[3], a[5], a[7]);                // O[k] in a[2 k + 1]
    a[3] = cx_mul<T>(a[3], R2, -R2);                 // W8^1
    a[5] = cx_swap(a[5]) * cx<T>{T(1), T(-1)};       // W8^2 = -i
    a[7] = cx_mul<T>(a[7], -R2, -R2);                // W8^3
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const cx<T> e = a[2 * k], o = a[2 * k + 1];
        a[2 * k] = e + o;
        a[2 * k + 1] = e - o;
    }
}

// forward CL-point DFT of a row of pass 2; X[k] is left in a[r256_slot<CL>(k)]
template <int CL> __host__ __device__ constexpr int r256_slot(int k) { return CL == 8 ? slot8(k) : k; }
template <typename T, int CL> __device__ __forceinline__ void r256_dft_row(cx<T> (&a)[CL])
{
    if constexpr (CL == 8) { dft8c<T>(a); }
    else if constexpr (CL == 4) { dft4c<T>(a[0], a[1], a[2], a[3]); }
    else { const cx<T> t = a[0]; a[0] = t + a[1]; a[1] = t - a[1]; }
}

// CPLX: the rows are the complex spectra (nfft/2 + 1 interleaved (re, im) pairs per frame, the imaginary parts of the first and
// last zero: window_univar followed by fft_fftw_r2hc, reference c/window_univar.c:41-89, c/fft.fftw.r2hc.c:69-78) instead of
// their powers.
template <typename T, int CL, int NR, bool MN0, bool EPI, int SPAN, bool ALN, bool CPLX = false>
__global__ __launch_bounds__(64 * r256_waves<T>(), sizeof(T) == 4 ? 2 : 1) void stft_r256_kernel(T *__restrict__ Y, const T *__restrict__ X,
                                                                                             const T *__restrict__ win, const T *__restrict__ tw,
                                                                                             const R256Args a)
{
    using C = cx<T>;
    typedef T pair_u __attribute__((ext_vector_type(2), aligned(sizeof(T))));        // a pair of samples at sample alignment
    typedef T pair_a __attribute__((ext_vector_type(2), aligned(2 * sizeof(T))));    // ... at its own alignment
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int M = 16 * CL, F = M + 1, FPS = 64 / CL, RW = 16 / CL, NWV = r256_waves<T>(), NV = SPAN * (int)sizeof(T) / 1024;
    constexpr int RP = CL + 1, FP = r256_fp(CL), TILE = r256_tile(CL);
    constexpr int WAVE_LDS = TILE + SPAN;
    static_assert(CL == 8 || CL == 4 || CL == 2, "8, 4 or 2 lanes per frame");
    constexpr int ROWF = CPLX ? 2 * F : F;                 // values of an output row
    constexpr int NVI = (FPS * ROWF + 255) / 256;          // float32: whole-vector wave stores of a step's block of rows
    constexpr int NSTORE = F32 ? NVI + 1 : (FPS * ROWF + 63) / 64;   // store instructions per step
    static_assert(FPS * ROWF + 3 <= TILE && 16 * RP <= FP, "row staging and a frame's rows fit the tile");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane / CL, j = lane % CL;
    const int L = a.L;

    T *xs = reinterpret_cast<T *>(smem_raw) + wv * WAVE_LDS;      // exchange tile, then row staging
    pair_a *ex = reinterpret_cast<pair_a *>(xs);
    T *slot = xs + TILE;                                          // the span
    unsigned char *dummy = smem_raw + (size_t)NWV * WAVE_LDS * sizeof(T);   // 1 KiB that placeholder LDS-DMA loads write
    unsigned *ticket = reinterpret_cast<unsigned *>(dummy + 1024);

    // ---- per-lane constants ----
    C twd[16];                                             // W_M^(j k1) = W_nfft^(2 j k1), 2 j k1 < nfft
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        int idx = 2 * j * k1;
        T sg = T(1);
        if (idx >= M) { idx -= M; sg = T(-1); }
        twd[k1] = C{sg * tw[2 * idx], sg * tw[2 * idx + 1]};
    }
    constexpr int NRW = NR > 16 ? 16 : NR;                 // rows that hold data (NR = 17: all 16, and the window covers them: L = nfft)
    C hw[NRW];                                             // halved window taps of the lane's column
#pragma unroll
    for (int n1 = 0; n1 < NRW; ++n1) {
        const int l = 2 * CL * n1 + 2 * j;
        hw[n1] = C{(l < L) ? T(0.5) * win[l] : T(0), (l + 1 < L) ? T(0.5) * win[l + 1] : T(0)};
    }
    C ow[RW][CL / 2];                                      // -i W_nfft^K for K = j + CL r + 16 k2, k2 < CL/2
#pragma unroll
    for (int r = 0; r < RW; ++r) {
#pragma unroll
        for (int k2 = 0; k2 < CL / 2; ++k2) {
            const int K = j + CL * r + 16 * k2;
            ow[r][k2] = C{tw[2 * K + 1], -tw[2 * K]};
        }
    }
    if (tid == 0) { *ticket = NWV; }
    __syncthreads();

    const int reach = a.tbl ? a.reach : (FPS - 1) * a.stp;
    const int need = reach + min(L, 2 * CL * NRW);             // samples of a span that pass 1 uses
    const T *xrow = slot + f * a.stp + 2 * j;
    // Start of the span of the step at frame w_ (wave-uniform).  Table form: the starts of the step's frames by
    // scalar loads (a vector load would put the compiler's own wait for ALL outstanding memory operations, the row
    // stores included, in front of the span fetch), and xrow for that step: called when pass 1 is done with the last one.
    auto step_start = [&](const int w_) {
        if (a.tbl == nullptr) { return a.ss0 + w_ * a.stp; }
        const __attribute__((address_space(4))) int *tp = (const __attribute__((address_space(4))) int *)(uintptr_t)a.tbl;
        const int last = a.W - 1;
        const int t0 = tp[2 * w_];
        int off = 0;
#pragma unroll
        for (int i = 1; i < FPS; ++i) {
            const int ti = tp[2 * min(w_ + i, last)];
            off = (f == i) ? ti - t0 : off;
        }
        xrow = slot + off + 2 * j;
        return t0;
    };

    // ---- fetch of one span into the slot: NV LDS-DMA loads whatever the case (the counted wait relies on
    // that); a span that crosses an edge of its signal is gathered sample by sample and NV placeholder loads follow
    auto load_span = [&](const T *xb, const int s0) {
        ring_load_span<NV>(slot, reinterpret_cast<T *>(dummy), tw, xb, s0, need, a.N, a.edge, a.rd_shift, lane,
                           [&](const T *sig, const int first, const int i) {
                               FrameDesc d{};
                               d.N = a.N; d.edge = a.edge;
                               return fetch_sample(sig, d, (long long)first, (long long)a.rd_shift, i);
                           });
    };

    // Work distribution: 8 contiguous ranges of steps, one per group of workgroups that share an XCD; a round
    // of a range is NWV neighbouring steps per workgroup, drawn by its waves from the counter.
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int s_lo = (int)((long long)a.steps * xcd / 8), s_hi = (int)((long long)a.steps * (xcd + 1) / 8);
    const int round = nwg_x * NWV, first = s_lo + (blockIdx.x >> 3) * NWV;
    int sid = first + wv;
    if (sid >= s_hi) { return; }
    int b = (int)a.spd.div((unsigned)sid);
    int w = (sid - b * a.sps) * FPS;
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
            nw = (nsid - (int)nb * a.sps) * FPS;
            nxb = X + (unsigned long long)nb * (unsigned long long)a.ldx;
            nyb = Y + (unsigned long long)nb * ((unsigned long long)a.Wy * ROWF);
        }
    };

    bool first_step = true;
    load_span(xb, step_start(w));

    while (true) {
        ring_draw(drawn, cnt_addr);                        // the step after this one: asked for now, needed further down
        // The span of this step was asked for in the middle of the last one; memory operations complete in issue
        // order and behind its NV loads lie that step's row stores, which stay in flight.
        if (first_step) { ring_wait_vm<0>(); } else { ring_wait_vm<NSTORE>(); }
        first_step = false;
        wave_sync();

        // ---- pass 1: lane (frame f, column j): 16-point FFT over the rows ----
        C zz[16];
        int lim = L - 2 * j;                              // element c of the pair in row n1 is inside the window iff 2 CL n1 + c < lim
        if constexpr (DSP_MASK_RECOMPUTE && NR == 16) { asm volatile("" : "+v"(lim)); }
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            if (n1 < NRW) {
                C v;
                if constexpr (ALN) { const pair_a t = *reinterpret_cast<const pair_a *>(xrow + 2 * CL * n1); v = C{t.x, t.y}; }
                else { const pair_u t = *reinterpret_cast<const pair_u *>(xrow + 2 * CL * n1); v = C{t.x, t.y}; }
                if (NR == 16 || (NR < 16 && n1 == NR - 1)) {   // rows that may reach past the window: keep that data out (0 * NaN)
                    v.x = (2 * CL * n1 < lim) ? v.x : T(0);
                    v.y = (2 * CL * n1 + 1 < lim) ? v.y : T(0);
                }
                zz[n1] = v * hw[n1];
            } else {
                zz[n1] = C{T(0), T(0)};
            }
        }
        if (MN0) {
            C acc = zz[0];
#pragma unroll
            for (int n1 = 1; n1 < NRW; ++n1) { acc += zz[n1]; }
            T sum = acc.x + acc.y;
#pragma unroll
            for (int m = 1; m < CL; m *= 2) { sum += __shfl_xor(sum, m); }
            const T mean = sum / (T)L;
#pragma unroll
            for (int n1 = 0; n1 < NRW; ++n1) {
                if (NR == 16 || (NR < 16 && n1 == NR - 1)) {
                    zz[n1] -= C{(2 * CL * n1 < lim) ? mean : T(0), (2 * CL * n1 + 1 < lim) ? mean : T(0)};
                } else {
                    zz[n1] -= C{mean, mean};
                }
            }
        }
        dft16x<T, NRW>(zz);
        wave_sync();                                       // all lanes have read the span
        // The slot takes the next step's span.  The reads of pass 1 must have RETURNED before the first load into
        // the slot is issued (the compiler may sink their consumers, and with them its own wait, below the
        // assembly statements): ring_drawn waits for every LDS operation of the wave.  Past the end of the
        // workgroup's share the current span is fetched again: the number of loads per step stays fixed.
        advance();
        load_span(more ? nxb : xb, step_start(more ? nw : w));
        {
            pair_a *exf = ex + f * FP + j;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                C v = zz[slot16(k1)];
                if (k1 > 0) { v = cx_mul<T>(v, twd[k1]); }
                exf[k1 * RP] = pair_a{v.x, v.y};
            }
        }
        wave_sync();

        // ---- pass 2: the lane's RW rows k1 = j + CL r: a CL-point FFT over the columns each ----
        C zr[RW][CL];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const pair_a *ra = ex + f * FP + (j + CL * r) * RP;
#pragma unroll
            for (int n2 = 0; n2 < CL; ++n2) { const pair_a t = ra[n2]; zr[r][n2] = C{t.x, t.y}; }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) { r256_dft_row<T, CL>(zr[r]); }

        // ---- real-FFT split fused with power.  zr[r][slot(k2)] = Z[K], K = j + CL r + 16 k2 (halved).
        // Partner Z[M - K]: row 16 - k1 = (CL - j) + CL (RW - 1 - r), register CL - 1 - k2: row RW - 1 - r of lane CL - j;
        // for j = 0: own row RW - r, register CL - 1 - k2 (r > 0), own row 0, register (CL - k2) % CL (r = 0).
        // With A = Z_K + conj Z_(M-K), B = Z_K - conj Z_(M-K), T = -i W_nfft^K B:
        // |X_K|^2 = |A + T|^2 and |X_(M-K)|^2 = |A - T|^2.  K < M/2 (k2 < CL/2) covers every pair once; K = M/2 pairs with itself.
        // the step's block of rows in the output, and how it lies against 16-byte boundaries
        T *yblk = yb + (size_t)w * ROWF;
        const int nout = __builtin_amdgcn_readfirstlane(min(FPS, a.W - w) * ROWF);
        int head = 0, sh = 0;                              // float32: value e of the block is staged at xs[sh + e]
        if constexpr (F32) { head = ring_block_head(yblk); sh = ring_stage_shift(head); }
        auto partner = [&](const int r, const int k2) {
            const C oth = zr[RW - 1 - r][r256_slot<CL>(CL - 1 - k2)];
            const C own = (r == 0) ? zr[0][r256_slot<CL>((CL - k2) % CL)] : zr[RW - r][r256_slot<CL>(CL - 1 - k2)];
            const T px = r256_grp_neg<CL>(oth.x), py = r256_grp_neg<CL>(oth.y);
            return C{(j == 0) ? own.x : px, (j == 0) ? own.y : py};
        };
        if constexpr (CPLX) {
            // X_K = A + T, X_(M-K) = conj(A - T), each written to the staging area as soon as it exists; K = M/2 is
            // 2 conj Z_(M/2) (lane j == 0)
            const T sc = (T)a.cscale;
            const C zmid = zr[0][r256_slot<CL>(CL / 2)];
            wave_sync();                                   // every lane has read its tile rows (into zr): reuse the region
            T *pf = xs + sh + f * ROWF;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
#pragma unroll
                for (int k2 = 0; k2 < CL / 2; ++k2) {
                    const int K = j + CL * r + 16 * k2;
                    const C z = zr[r][r256_slot<CL>(k2)];
                    const C m = partner(r, k2);
                    const C A = cx_add_conj<T>(z, m), B = cx_sub_conj<T>(z, m);
                    const C Tw = cx_mul<T>(B, ow[r][k2]);
                    const C u = A + Tw, v = A - Tw;
                    const bool edge = (j == 0) && r == 0 && k2 == 0;     // bins 0 and M: imaginary parts zero
                    pf[2 * K] = sc * u.x; pf[2 * K + 1] = edge ? T(0) : sc * u.y;
                    pf[2 * (M - K)] = sc * v.x; pf[2 * (M - K) + 1] = edge ? T(0) : -sc * v.y;
                }
            }
            if (j == 0) { pf[M] = T(2) * sc * zmid.x; pf[M + 1] = T(-2) * sc * zmid.y; }
            wave_sync();
        } else {
            T pw[16];                                          // [(r CL/2 + k2) 2]: P[K], [.. + 1]: P[M - K]
            T pmid = T(4) * cx_norm<T>(zr[0][r256_slot<CL>(CL / 2)]);  // K = M/2: lane j == 0 only
            // two bins at a time, stage by stage (a packed instruction that consumes the result of the one issued just before it
            // costs a wait state, and the stages of one bin are such a chain); (|A + T|^2, |A - T|^2) as one pair: the real parts
            // of both sums in one packed register, the imaginary parts in another (x * 1 + y is x + y exactly)
            const C pm = C{T(1), T(-1)};
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                constexpr int H = CL / 2;
                const int ra = h / H, ka = h % H, rb = (h + 4) / H, kb = (h + 4) % H;
                const C za = zr[ra][r256_slot<CL>(ka)], zb = zr[rb][r256_slot<CL>(kb)];
                const C ma = partner(ra, ka), mb = partner(rb, kb);
                const C Aa = cx_add_conj<T>(za, ma), Ab = cx_add_conj<T>(zb, mb);
                const C Ba = cx_sub_conj<T>(za, ma), Bb = cx_sub_conj<T>(zb, mb);
                const C ta = cx_mul_first<T>(Ba, ow[ra][ka]), tb = cx_mul_first<T>(Bb, ow[rb][kb]);
                const C Ta = cx_mul_second<T>(Ba, ow[ra][ka], ta), Tb = cx_mul_second<T>(Bb, ow[rb][kb], tb);
                const C rea = cx_fma(C{Ta.x, Ta.x}, pm, C{Aa.x, Aa.x}), reb = cx_fma(C{Tb.x, Tb.x}, pm, C{Ab.x, Ab.x});
                const C ima = cx_fma(C{Ta.y, Ta.y}, pm, C{Aa.y, Aa.y}), imb = cx_fma(C{Tb.y, Tb.y}, pm, C{Ab.y, Ab.y});
                const C qa = rea * rea, qb = reb * reb;
                const C sa = ima * ima, sb = imb * imb;
                const C pa = qa + sa, pb = qb + sb;
                pw[2 * h] = pa.x; pw[2 * h + 1] = pa.y;
                pw[2 * (h + 4)] = pb.x; pw[2 * (h + 4) + 1] = pb.y;
            }
            if (EPI) {
                power_epilogue_n(pw, a.epi);
                pmid = power_epilogue(pmid, a.epi);
            }
            wave_sync();                                       // every lane has read its tile rows: reuse the region
            {
                T *pf = xs + sh + f * F;
#pragma unroll
                for (int r = 0; r < RW; ++r) {
#pragma unroll
                    for (int k2 = 0; k2 < CL / 2; ++k2) {
                        const int K = j + CL * r + 16 * k2;
                        pf[K] = pw[(r * (CL / 2) + k2) * 2];
                        pf[M - K] = pw[(r * (CL / 2) + k2) * 2 + 1];
                    }
                }
                if (j == 0) { pf[M / 2] = pmid; }
            }
            wave_sync();
        }

        // ---- float32: NVI + 1 stores, NVI of whole 16-byte vectors, one for the values before and after them; float64:
        // NSTORE element-wise stores (a wave stores 512 consecutive bytes).  Surplus lanes repeat a valid store (same value
        // to the same address): the number of store instructions per step is fixed.
        if constexpr (F32) {
            ring_store_rows<NVI>(yblk, xs + sh, head, nout, lane);
        } else {
#pragma unroll
            for (int i = 0; i < NSTORE; ++i) {
                const int e = min(lane + 64 * i, nout - 1);
                __builtin_nontemporal_store(xs[e], yblk + e);
            }
        }
        if (!more) { break; }
        wave_sync();                                       // staging reads are done before the next step writes the tile
        sid = nsid; w = nw; xb = nxb; yb = nyb;
    }
    // no LDS-DMA load may land, and no draw return, after the wave has ended
    ring_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Largest span a wave's LDS share holds beside its tile: waves x (tile + span) elements + 1 KiB inside 160 KiB (the same
// element counts in both precisions: float64 has half the waves)
__host__ __device__ constexpr int r256_max_span(int CL) { return CL == 2 ? 1536 : 2048; }
// frames of a step, and how many times tbl_reach (a bound on the distance of frames w and w + 3) bounds the distance of
// the first and last frame of a step
static int r256_fps(int nfft) { return 64 / (nfft / 32); }
static int r256_reach_mult(int nfft) { return (r256_fps(nfft) - 1 + 2) / 3; }

// nfft 256, 128 or 64; frames at ss0 + w * hop, or a table of frame starts that never decrease and are read where they
// start; the frames of a step inside the largest span; every index of one signal's input and output inside 31 bits
bool r256_supported(const FrameDesc &d)
{
    if (d.nfft != 256 && d.nfft != 128 && d.nfft != 64) { return false; }
    const int fps = r256_fps(d.nfft), span = r256_max_span(d.nfft / 32);
    const bool placed = d.tbl ? (d.tbl_reach >= 0 && (long long)r256_reach_mult(d.nfft) * d.tbl_reach + d.nfft <= span)
                              : (d.stp >= 1 && (long long)(fps - 1) * d.stp + d.nfft <= span);
    return placed && d.L >= 1 && d.L <= d.nfft &&
           d.N < (1ll << 31) - 4096 && (long long)d.wy() * (d.nfft / 2 + 1) < (1ll << 30) && (long long)d.B * d.W < (1ll << 30) &&
           d.ss0 > -(1ll << 30) && d.ss0 < (1ll << 30) && d.rd_shift >= 0 && d.rd_shift < (1ll << 30);
}

template <typename T, int CL, int NR, bool MN0, bool EPI, int SPAN, bool ALN, bool CPLX = false>
static hipError_t launch_r256_inst(T *Y, const T *X, const T *win, const T *tw, const R256Args &a, hipStream_t s)
{
    static_assert(SPAN <= r256_max_span(CL), "span too large for the LDS");
    constexpr int NWV = r256_waves<T>();
    const size_t lds = (size_t)NWV * (r256_tile(CL) + SPAN) * sizeof(T) + 1024 + 16;
    static_assert((size_t)NWV * (r256_tile(CL) + SPAN) * sizeof(T) + 1024 + 16 <= 160 * 1024, "LDS of a workgroup");
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_r256_kernel<T, CL, NR, MN0, EPI, SPAN, ALN, CPLX>), lds, grant); e != hipSuccess) { return e; }
    long long grid = ((long long)a.steps + NWV - 1) / NWV;
    if (grid > 256) { grid = 256; }                     // one workgroup per CU
    grid = (grid + 7) & ~7ll;                           // whole groups of 8 workgroups (one per XCD)
    hipLaunchKernelGGL((stft_r256_kernel<T, CL, NR, MN0, EPI, SPAN, ALN, CPLX>), dim3((unsigned)grid), dim3(64 * NWV), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

template <typename T, int CL, int NR, int SPAN, bool ALN>
static hipError_t launch_r256_shape(T *Y, const T *X, const T *win, const T *tw, const R256Args &a, const FrameDesc &d, bool cplx, hipStream_t s)
{
    const bool epi = d.epi.any();
    if (cplx) { return launch_r256_inst<T, CL, NR, false, false, SPAN, ALN, true>(Y, X, win, tw, a, s); }
#if R256_MINIMAL
    if (!d.mn0 && !epi) { return launch_r256_inst<T, CL, NR, false, false, SPAN, ALN>(Y, X, win, tw, a, s); }
    return hipErrorInvalidValue;
#else
    if (d.mn0) { return epi ? launch_r256_inst<T, CL, NR, true, true, SPAN, ALN>(Y, X, win, tw, a, s) : launch_r256_inst<T, CL, NR, true, false, SPAN, ALN>(Y, X, win, tw, a, s); }
    return epi ? launch_r256_inst<T, CL, NR, false, true, SPAN, ALN>(Y, X, win, tw, a, s) : launch_r256_inst<T, CL, NR, false, false, SPAN, ALN>(Y, X, win, tw, a, s);
#endif
}

template <typename T, int CL>
static hipError_t launch_r256_cl(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, bool cplx, double scale, hipStream_t s)
{
    constexpr int FPS = 64 / CL, NFFT = 32 * CL, MAXS = r256_max_span(CL);
    R256Args a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.Wy = d.wy(); a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + FPS - 1) / FPS;
    a.spd = FastDiv::of((unsigned)a.sps);
    a.steps = d.B * a.sps;
    a.tbl = d.tbl; a.reach = 0;
    a.cscale = scale;
    if (d.tbl) {
        a.stp = 0; a.ss0 = 0; a.rd_shift = 0; a.reach = r256_reach_mult(NFFT) * d.tbl_reach;
        return launch_r256_shape<T, CL, 16, MAXS, false>(Y, X, win, tw, a, d, cplx, s);
    }
    const int reach = (FPS - 1) * d.stp + NFFT;
    if (d.stp & 1) { return launch_r256_shape<T, CL, 16, MAXS, false>(Y, X, win, tw, a, d, cplx, s); }
    // L = nfft (the window covers all 16 rows: no row of a lane is ever masked; the 32 lane masks of the general form live in scalar
    // registers that spill into the lanes of a vector register)
    if (d.L == NFFT) {
        if (reach <= 1024) { return launch_r256_shape<T, CL, 17, 1024, true>(Y, X, win, tw, a, d, cplx, s); }
        if constexpr (sizeof(T) == 4) {
            if (reach <= 1280) { return launch_r256_shape<T, CL, 17, 1280, true>(Y, X, win, tw, a, d, cplx, s); }
        }
        return launch_r256_shape<T, CL, 17, MAXS, true>(Y, X, win, tw, a, d, cplx, s);
    }
    if constexpr (sizeof(T) == 4) {
        // windows that leave the last three rows of pass 1 empty (nfft 256: 193 to 208 samples, 200 = 25 ms at 8 kHz)
        const bool nr13 = d.L > 12 * 2 * CL && d.L <= 13 * 2 * CL;
        if (reach <= 1024) {
            if constexpr (CL == 8) {
                // nfft 256: 10 rows (145 to 160 samples: 20 ms at 8 kHz) and 15 rows (225 to 240: 30 ms) likewise
                const int rows = (d.L + 2 * CL - 1) / (2 * CL);
                if (rows == 10) { return launch_r256_shape<T, CL, 10, 1024, true>(Y, X, win, tw, a, d, cplx, s); }
                if (rows == 15) { return launch_r256_shape<T, CL, 15, 1024, true>(Y, X, win, tw, a, d, cplx, s); }
            }
            return nr13 ? launch_r256_shape<T, CL, 13, 1024, true>(Y, X, win, tw, a, d, cplx, s)
                        : launch_r256_shape<T, CL, 16, 1024, true>(Y, X, win, tw, a, d, cplx, s);
        }
        if (reach <= 1280) { return launch_r256_shape<T, CL, 16, 1280, true>(Y, X, win, tw, a, d, cplx, s); }
    } else {
        if (reach <= 1024) {
            if constexpr (CL == 8) {
                const int rows = (d.L + 2 * CL - 1) / (2 * CL);
                if (rows == 10) { return launch_r256_shape<T, CL, 10, 1024, true>(Y, X, win, tw, a, d, cplx, s); }
                if (rows == 13) { return launch_r256_shape<T, CL, 13, 1024, true>(Y, X, win, tw, a, d, cplx, s); }
                if (rows == 15) { return launch_r256_shape<T, CL, 15, 1024, true>(Y, X, win, tw, a, d, cplx, s); }
            }
            return launch_r256_shape<T, CL, 16, 1024, true>(Y, X, win, tw, a, d, cplx, s);
        }
    }
    return launch_r256_shape<T, CL, 16, MAXS, true>(Y, X, win, tw, a, d, cplx, s);
}

template <typename T>
static hipError_t launch_r256_any(T *Y, const T *X, const T *win, const T *tw, const FrameDesc &d, bool cplx, double scale, hipStream_t s)
{
    if (d.nfft == 256) { return launch_r256_cl<T, 8>(Y, X, win, tw, d, cplx, scale, s); }
    if (d.nfft == 128) { return launch_r256_cl<T, 4>(Y, X, win, tw, d, cplx, scale, s); }
    return launch_r256_cl<T, 2>(Y, X, win, tw, d, cplx, scale, s);
}

hipError_t launch_stft_r256(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s)
{
    return launch_r256_any<float>(Y, X, win, tw, d, false, 0.0, s);
}
hipError_t launch_stft_r256(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, hipStream_t s)
{
    return launch_r256_any<double>(Y, X, win, tw, d, false, 0.0, s);
}

// The complex form: Y[b][w][nfft/2 + 1] interleaved (re, im) pairs times `scale`; equally spaced frames, no mean removal.
bool r256_cplx_supported(const FrameDesc &d)
{
    return r256_supported(d) && d.tbl == nullptr && !d.mn0 && !d.epi.any() && (long long)d.wy() * (d.nfft + 2) < (1ll << 30);
}

hipError_t launch_stft_r256_cplx(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    return launch_r256_any<float>(Y, X, win, tw, d, true, scale, s);
}
hipError_t launch_stft_r256_cplx(double *Y, const double *X, const double *win, const double *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    return launch_r256_any<double>(Y, X, win, tw, d, true, scale, s);
}

}  // namespace dsp
