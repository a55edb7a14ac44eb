// Fused STFT kernel for nfft = 256, float32, equally spaced frames (gfx950).
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
    int rd_shift;         // edge == 1: samples are read at n - rd_shift
    int edge;             // 0 reflect, 1 zero
    Epilogue epi;
    int sps;              // steps (groups of 8 consecutive frames) per signal
    int steps;            // B * sps
    const int *tbl;       // table form: start of frame w at tbl[2 w] (frames need not be equally spaced); else nullptr
    int reach;            // table form: bound on the distance between the starts of the first and last frame of a step
    float cscale;         // complex form: factor on every output value
};

constexpr int R256_FP = 152;                         // complex slots per frame in the exchange tile (16 rows, pitch 9, + 8)
constexpr int R256_TILE = 8 * R256_FP * 2;           // floats of a wave's exchange tile / row staging
constexpr int R256_WAVES = 8;

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

// forward 8-point DFT; X[k] is left in a[slot8(k)]
__host__ __device__ constexpr int slot8(int k) { return k < 4 ? 2 * k : 2 * (k - 4) + 1; }
__device__ __forceinline__ void dft8c(cx<float> (&a)[8])
{
    constexpr float R2 = 0.70710678118654752440084436210484904f;
    dft4c<float>(a[0], a[2], a[4], a[6]);            // E[k] in a[2 k]
    dft4c<float>(a[1], a[3], a[5], a[7]);            // O[k] in a[2 k + 1]
    a[3] = cx_mul<float>(a[3], R2, -R2);             // W8^1
    a[5] = cx_swap(a[5]) * cx<float>{1.f, -1.f};     // W8^2 = -i
    a[7] = cx_mul<float>(a[7], -R2, -R2);            // W8^3
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const cx<float> e = a[2 * k], o = a[2 * k + 1];
        a[2 * k] = e + o;
        a[2 * k + 1] = e - o;
    }
}

// CPLX: the rows are the complex spectra (129 interleaved (re, im) pairs per frame, the imaginary parts of the first and
// last zero: window_univar followed by fft_fftw_r2hc, reference c/window_univar.c:41-89, c/fft.fftw.r2hc.c:69-78) instead of
// their powers.
template <int NR, bool MN0, bool EPI, int SPAN, bool ALN, bool CPLX = false>
__global__ __launch_bounds__(64 * R256_WAVES, 2) void stft_r256_kernel(float *__restrict__ Y, const float *__restrict__ X,
                                                                       const float *__restrict__ win, const float *__restrict__ tw,
                                                                       const R256Args a)
{
    using C = cx<float>;
    typedef float f2e __attribute__((ext_vector_type(2), aligned(4)));
    constexpr int M = 128, F = 129, FPS = 8, NWV = R256_WAVES, NV = SPAN / 256;
    constexpr int WAVE_LDS = R256_TILE + SPAN;
    constexpr int ROWF = CPLX ? 2 * F : F;                 // floats of an output row
    constexpr int NVI = (FPS * ROWF + 255) / 256;          // whole-vector wave stores of a step's block of rows
    static_assert(FPS * ROWF + 3 <= R256_TILE, "row staging fits the tile");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane >> 3, j = lane & 7;
    const int L = a.L;

    float *xs = reinterpret_cast<float *>(smem_raw) + wv * WAVE_LDS;   // exchange tile, then row staging
    float2 *ex = reinterpret_cast<float2 *>(xs);
    float *slot = xs + R256_TILE;                                      // the span
    float *dummy = reinterpret_cast<float *>(smem_raw) + NWV * WAVE_LDS;   // 1 KiB that placeholder LDS-DMA loads write
    unsigned *ticket = reinterpret_cast<unsigned *>(dummy + 256);

    // ---- per-lane constants ----
    C twd[16];                                             // W128^(j k1) = W256^(2 j k1), j k1 <= 105
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        int idx = 2 * j * k1;
        float sg = 1.0f;
        if (idx >= M) { idx -= M; sg = -1.0f; }
        twd[k1] = C{sg * tw[2 * idx], sg * tw[2 * idx + 1]};
    }
    C hw[NR];                                              // halved window taps of the lane's column
#pragma unroll
    for (int n1 = 0; n1 < NR; ++n1) {
        const int l = 16 * n1 + 2 * j;
        hw[n1] = C{(l < L) ? 0.5f * win[l] : 0.f, (l + 1 < L) ? 0.5f * win[l + 1] : 0.f};
    }
    C owa[4], owb[4];                                      // -i W256^K for K = j + 16 k2 and K = j + 8 + 16 k2, k2 < 4
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        const int Ka = j + 16 * k2, Kb = Ka + 8;
        owa[k2] = C{tw[2 * Ka + 1], -tw[2 * Ka]};
        owb[k2] = C{tw[2 * Kb + 1], -tw[2 * Kb]};
    }
    if (tid == 0) { *ticket = NWV; }
    __syncthreads();

    const int reach = a.tbl ? a.reach : (FPS - 1) * a.stp;
    const int need = reach + min(L, 16 * NR);              // samples of a span that pass 1 uses
    const float *xrow = slot + f * a.stp + 2 * j;
    // Start of the span of the step at frame w_ (wave-uniform).  Table form: the starts of the step's 8 frames by
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
    auto load_span = [&](const float *xb, const int s0) {
        ring_load_span<NV>(slot, dummy, tw, xb, s0, need, a.N, a.edge, a.rd_shift, lane,
                           [&](const float *sig, const int first, const int i) {
                               FrameDesc d{};
                               d.N = a.N; d.edge = a.edge;
                               return fetch_sample(sig, d, (long long)first, (long long)a.rd_shift, i);
                           });
    };

    // Work distribution: 8 contiguous ranges of steps, one per group of workgroups that share an XCD; a round
    // of a range is 8 neighbouring steps per workgroup, drawn by its waves from the counter.
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int s_lo = (int)((long long)a.steps * xcd / 8), s_hi = (int)((long long)a.steps * (xcd + 1) / 8);
    const int round = nwg_x * NWV, first = s_lo + (blockIdx.x >> 3) * NWV;
    int sid = first + wv;
    if (sid >= s_hi) { return; }
    int b = sid / a.sps;
    int w = (sid - b * a.sps) * FPS;
    const float *xb = X + (long long)b * a.ldx;
    float *yb = Y + (long long)b * a.W * ROWF;

    const unsigned cnt_addr = ring_lds_addr(ticket);
    unsigned drawn = 0;
    int nsid = 0, nw = 0;
    const float *nxb = xb;
    float *nyb = yb;
    bool more = false;
    auto advance = [&]() {
        const int t = ring_drawn(drawn);
        nsid = first + (t >> 3) * round + (t & 7);
        more = nsid < s_hi;
        if (more) {
            const int nb = nsid / a.sps;
            nw = (nsid - nb * a.sps) * FPS;
            nxb = X + (long long)nb * a.ldx;
            nyb = Y + (long long)nb * a.W * ROWF;
        }
    };

    bool first_step = true;
    load_span(xb, step_start(w));

    while (true) {
        ring_draw(drawn, cnt_addr);                        // the step after this one: asked for now, needed further down
        // The span of this step was asked for in the middle of the last one; memory operations complete in issue
        // order and behind its NV loads lie that step's 6 row stores, which stay in flight.
        if (first_step) { ring_wait_vm<0>(); } else { ring_wait_vm<NVI + 1>(); }
        first_step = false;
        wave_sync();

        // ---- pass 1: lane (frame f, column j): 16-point FFT over the rows ----
        C zz[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            if (n1 < NR) {
                C v;
                if constexpr (ALN) { const float2 t = *reinterpret_cast<const float2 *>(xrow + 16 * n1); v = C{t.x, t.y}; }
                else { const f2e t = *reinterpret_cast<const f2e *>(xrow + 16 * n1); v = C{t.x, t.y}; }
                if (NR == 16 || n1 == NR - 1) {            // rows that may reach past the window: keep that data out (0 * NaN)
                    const int l = 16 * n1 + 2 * j;
                    v.x = (l < L) ? v.x : 0.f;
                    v.y = (l + 1 < L) ? v.y : 0.f;
                }
                zz[n1] = v * hw[n1];
            } else {
                zz[n1] = C{0.f, 0.f};
            }
        }
        if (MN0) {
            C acc = zz[0];
#pragma unroll
            for (int n1 = 1; n1 < NR; ++n1) { acc += zz[n1]; }
            float sum = acc.x + acc.y;
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
            const float mean = sum / (float)L;
#pragma unroll
            for (int n1 = 0; n1 < NR; ++n1) {
                if (NR == 16 || n1 == NR - 1) {
                    const int l = 16 * n1 + 2 * j;
                    zz[n1] -= C{(l < L) ? mean : 0.f, (l + 1 < L) ? mean : 0.f};
                } else {
                    zz[n1] -= C{mean, mean};
                }
            }
        }
        dft16c<float, NR>(zz);
        wave_sync();                                       // all lanes have read the span
        // The slot takes the next step's span.  The reads of pass 1 must have RETURNED before the first load into
        // the slot is issued (the compiler may sink their consumers, and with them its own wait, below the
        // assembly statements): ring_drawn waits for every LDS operation of the wave.  Past the end of the
        // workgroup's share the current span is fetched again: the number of loads per step stays fixed.
        advance();
        load_span(more ? nxb : xb, step_start(more ? nw : w));
        {
            float2 *exf = ex + f * R256_FP + j;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                C v = zz[slot16(k1)];
                if (k1 > 0) { v = cx_mul<float>(v, twd[k1]); }
                exf[k1 * 9] = make_float2(v.x, v.y);
            }
        }
        wave_sync();

        // ---- pass 2: the lane's two rows k1 = j (za) and k1 = j + 8 (zb): 8-point FFTs over the columns ----
        C za[8], zb[8];
        {
            const float2 *ra = ex + f * R256_FP + j * 9, *rb = ra + 72;
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) { const float2 t = ra[n2]; za[n2] = C{t.x, t.y}; }
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) { const float2 t = rb[n2]; zb[n2] = C{t.x, t.y}; }
        }
        dft8c(za);
        dft8c(zb);

        // ---- real-FFT split fused with power.  za[slot8(k2)] = Z[j + 16 k2], zb[slot8(k2)] = Z[j + 8 + 16 k2] (halved).
        // Partner of K = j + 16 k2:     Z[128 - K] = row b, register 7 - k2, of lane 8 - j   (j = 0: own za[(8 - k2) % 8]);
        // partner of K = j + 8 + 16 k2: Z[128 - K] = row a, register 7 - k2, of lane 8 - j   (j = 0: own zb[7 - k2]).
        // With A = Z_K + conj Z_(128-K), B = Z_K - conj Z_(128-K), T = -i W256^K B:
        // |X_K|^2 = |A + T|^2 and |X_(128-K)|^2 = |A - T|^2.  K < 64 (k2 < 4) covers every pair once; K = 64 pairs with itself.
        // the step's block of rows in the output, and how it lies against 16-byte boundaries
        float *yblk = yb + (size_t)w * ROWF;
        const int nout = __builtin_amdgcn_readfirstlane(min(FPS, a.W - w) * ROWF);
        const int head = ring_block_head(yblk);
        const int sh = ring_stage_shift(head);             // value e of the block is staged at xs[sh + e]
        if constexpr (CPLX) {
            // X_K = A + T, X_(128-K) = conj(A - T), each written to the staging area as soon as it exists; K = 64 is
            // 2 conj Z_64 (lane j == 0)
            const float sc = a.cscale;
            const C zmid = za[slot8(4)];
            wave_sync();                                   // every lane has read its tile rows (into za, zb): reuse the region
            float *pf = xs + sh + f * ROWF;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                auto partner = [&](const C oth, const C own) {
                    const float px = r256_shr1(r256_half_mirror(oth.x)), py = r256_shr1(r256_half_mirror(oth.y));
                    return C{(j == 0) ? own.x : px, (j == 0) ? own.y : py};
                };
                const int Ka = j + 16 * k2, Kb = Ka + 8;
                {
                    const C z = za[slot8(k2)];
                    const C m = partner(zb[slot8(7 - k2)], za[slot8((8 - k2) & 7)]);
                    const C A = cx_add_conj<float>(z, m), B = cx_sub_conj<float>(z, m);
                    const C Tw = cx_mul<float>(B, owa[k2]);
                    const C u = A + Tw, v = A - Tw;
                    const bool edge = (j == 0) && k2 == 0;     // bins 0 and 128: imaginary parts zero
                    pf[2 * Ka] = sc * u.x; pf[2 * Ka + 1] = edge ? 0.f : sc * u.y;
                    pf[2 * (M - Ka)] = sc * v.x; pf[2 * (M - Ka) + 1] = edge ? 0.f : -sc * v.y;
                }
                {
                    const C z = zb[slot8(k2)];
                    const C m = partner(za[slot8(7 - k2)], zb[slot8(7 - k2)]);
                    const C A = cx_add_conj<float>(z, m), B = cx_sub_conj<float>(z, m);
                    const C Tw = cx_mul<float>(B, owb[k2]);
                    const C u = A + Tw, v = A - Tw;
                    pf[2 * Kb] = sc * u.x; pf[2 * Kb + 1] = sc * u.y;
                    pf[2 * (M - Kb)] = sc * v.x; pf[2 * (M - Kb) + 1] = -sc * v.y;
                }
            }
            if (j == 0) { pf[M] = 2.0f * sc * zmid.x; pf[M + 1] = -2.0f * sc * zmid.y; }
            wave_sync();
        } else {
            float pw[16];                                      // [k2]: P[Ka], [4 + k2]: P[128 - Ka], [8 + k2]: P[Kb], [12 + k2]: P[128 - Kb]
            float pmid = 4.0f * cx_norm<float>(za[slot8(4)]);  // K = 64: lane j == 0 only
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                auto partner = [&](const C oth, const C own) {
                    const float px = r256_shr1(r256_half_mirror(oth.x)), py = r256_shr1(r256_half_mirror(oth.y));
                    return C{(j == 0) ? own.x : px, (j == 0) ? own.y : py};
                };
                {
                    const C z = za[slot8(k2)];
                    const C m = partner(zb[slot8(7 - k2)], za[slot8((8 - k2) & 7)]);
                    const C A = cx_add_conj<float>(z, m), B = cx_sub_conj<float>(z, m);
                    const C Tw = cx_mul<float>(B, owa[k2]);
                    pw[k2] = cx_norm<float>(A + Tw);
                    pw[4 + k2] = cx_norm<float>(A - Tw);
                }
                {
                    const C z = zb[slot8(k2)];
                    const C m = partner(za[slot8(7 - k2)], zb[slot8(7 - k2)]);
                    const C A = cx_add_conj<float>(z, m), B = cx_sub_conj<float>(z, m);
                    const C Tw = cx_mul<float>(B, owb[k2]);
                    pw[8 + k2] = cx_norm<float>(A + Tw);
                    pw[12 + k2] = cx_norm<float>(A - Tw);
                }
            }
            if (EPI) {
                power_epilogue_n(pw, a.epi);
                pmid = power_epilogue(pmid, a.epi);
            }
            wave_sync();                                       // every lane has read its tile rows: reuse the region
            {
                float *pf = xs + sh + f * F;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const int Ka = j + 16 * k2, Kb = Ka + 8;
                    pf[Ka] = pw[k2];
                    pf[M - Ka] = pw[4 + k2];
                    pf[Kb] = pw[8 + k2];
                    pf[M - Kb] = pw[12 + k2];
                }
                if (j == 0) { pf[M / 2] = pmid; }
            }
            wave_sync();
        }

        // ---- six stores: five of whole 16-byte vectors, one for the values before and after them.  Surplus
        // lanes repeat a valid store (same value to the same address).
        ring_store_rows<NVI>(yblk, xs + sh, head, nout, lane);
        if (!more) { break; }
        wave_sync();                                       // staging reads are done before the next step writes the tile
        sid = nsid; w = nw; xb = nxb; yb = nyb;
    }
    // no LDS-DMA load may land, and no draw return, after the wave has ended
    ring_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// nfft 256; frames at ss0 + w * hop, or a table of frame starts that never decrease and are read where they start;
// the 8 frames of a step inside 2048 samples; every index of one signal's input and output inside 31 bits
bool r256_supported(const FrameDesc &d)
{
    // (tbl_reach bounds the distance of frames w and w + 3: three times that bounds the distance of frames w and w + 7)
    const bool placed = d.tbl ? (d.tbl_reach >= 0 && 3ll * d.tbl_reach + 256 <= 2048) : (d.stp >= 1 && 7ll * d.stp + 256 <= 2048);
    return d.nfft == 256 && placed && d.L >= 1 && d.L <= 256 &&
           d.N < (1ll << 31) - 4096 && (long long)d.W * 129 < (1ll << 30) && (long long)d.B * d.W < (1ll << 30) &&
           d.ss0 > -(1ll << 30) && d.ss0 < (1ll << 30) && d.rd_shift >= 0 && d.rd_shift < (1ll << 30);
}

template <int NR, bool MN0, bool EPI, int SPAN, bool ALN, bool CPLX = false>
static hipError_t launch_r256_inst(float *Y, const float *X, const float *win, const float *tw, const R256Args &a, hipStream_t s)
{
    const size_t lds = ((size_t)R256_WAVES * (R256_TILE + SPAN) + 256) * sizeof(float) + 16;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_r256_kernel<NR, MN0, EPI, SPAN, ALN, CPLX>), lds, grant); e != hipSuccess) { return e; }
    long long grid = ((long long)a.steps + R256_WAVES - 1) / R256_WAVES;
    if (grid > 256) { grid = 256; }                     // one 8-wave workgroup per CU
    grid = (grid + 7) & ~7ll;                           // whole groups of 8 workgroups (one per XCD)
    hipLaunchKernelGGL((stft_r256_kernel<NR, MN0, EPI, SPAN, ALN, CPLX>), dim3((unsigned)grid), dim3(64 * R256_WAVES), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

template <int NR, int SPAN, bool ALN>
static hipError_t launch_r256_shape(float *Y, const float *X, const float *win, const float *tw, const R256Args &a, const FrameDesc &d, hipStream_t s)
{
    const bool epi = d.epi.any();
#if R256_MINIMAL
    if (!d.mn0 && !epi) { return launch_r256_inst<NR, false, false, SPAN, ALN>(Y, X, win, tw, a, s); }
    return hipErrorInvalidValue;
#else
    if (d.mn0) { return epi ? launch_r256_inst<NR, true, true, SPAN, ALN>(Y, X, win, tw, a, s) : launch_r256_inst<NR, true, false, SPAN, ALN>(Y, X, win, tw, a, s); }
    return epi ? launch_r256_inst<NR, false, true, SPAN, ALN>(Y, X, win, tw, a, s) : launch_r256_inst<NR, false, false, SPAN, ALN>(Y, X, win, tw, a, s);
#endif
}

hipError_t launch_stft_r256(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s)
{
    R256Args a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + 7) / 8;
    a.steps = d.B * a.sps;
    a.tbl = d.tbl; a.reach = 0;
    if (d.tbl) {
        a.stp = 0; a.ss0 = 0; a.rd_shift = 0; a.reach = 3 * d.tbl_reach;
        return launch_r256_shape<16, 2048, false>(Y, X, win, tw, a, d, s);
    }
    const int reach = 7 * d.stp + 256;
    // windows of 193 to 208 samples (200: 25 ms at 8 kHz) leave the last three rows of pass 1 empty
    const bool nr13 = d.L > 192 && d.L <= 208;
    if (d.stp & 1) { return launch_r256_shape<16, 2048, false>(Y, X, win, tw, a, d, s); }
    if (reach <= 1024) { return nr13 ? launch_r256_shape<13, 1024, true>(Y, X, win, tw, a, d, s) : launch_r256_shape<16, 1024, true>(Y, X, win, tw, a, d, s); }
    if (reach <= 1280) { return launch_r256_shape<16, 1280, true>(Y, X, win, tw, a, d, s); }
    return launch_r256_shape<16, 2048, true>(Y, X, win, tw, a, d, s);
}

// The complex form: Y[b][w][129] interleaved (re, im) pairs times `scale`; equally spaced frames, no mean removal.
bool r256_cplx_supported(const FrameDesc &d)
{
    return r256_supported(d) && d.tbl == nullptr && !d.mn0 && !d.epi.any() && (long long)d.W * 258 < (1ll << 30);
}

hipError_t launch_stft_r256_cplx(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, double scale, hipStream_t s)
{
    R256Args a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + 7) / 8;
    a.steps = d.B * a.sps;
    a.tbl = nullptr; a.reach = 0;
    a.cscale = (float)scale;
    const int reach = 7 * d.stp + 256;
    const bool nr13 = d.L > 192 && d.L <= 208;
    if (d.stp & 1) { return launch_r256_inst<16, false, false, 2048, false, true>(Y, X, win, tw, a, s); }
    if (reach <= 1024) {
        return nr13 ? launch_r256_inst<13, false, false, 1024, true, true>(Y, X, win, tw, a, s)
                    : launch_r256_inst<16, false, false, 1024, true, true>(Y, X, win, tw, a, s);
    }
    if (reach <= 1280) { return launch_r256_inst<16, false, false, 1280, true, true>(Y, X, win, tw, a, s); }
    return launch_r256_inst<16, false, false, 2048, true, true>(Y, X, win, tw, a, s);
}

}  // namespace dsp
