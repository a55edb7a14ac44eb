// Fused STFT kernel for nfft = 1024 and 2048, float32, equally spaced frames (gfx950): the 256-point complex
// engine of the 512-point kernel (16 lanes, 16 values each: a 16 x 16 four-step transform with one exchange
// through LDS) under one decimation-in-frequency stage of radix Q = nfft / 512.
//
// A frame's nfft reals are M = 256 Q complex points z_n = x_2n + i x_2n+1.  With n = m + 256 g:
//     Z[Q k + r] = sum_m W256^(m k) * ( W_M^(m r) * sum_g W_Q^(g r) z_(m + 256 g) ),
// so group r of a frame (16 lanes; Q groups per frame, 4 / Q frames per wavefront and step) reads ALL Q
// quarters of the windowed frame from the staged span, forms its own combination (signs folded into the
// window table it reads, the factor (-i)^r and the twiddle W_M^(m r) as per-lane constants) and runs the
// 256-point transform of that: no exchange between groups on the way in.  On the way out lane (r, j)
// holds Z[Q (j + 16 k2) + r]; the partner of bin K in the real-FFT split, Z[M - K], lies in group
// (Q - r) % Q at the mirrored lane: a DPP row move for r = 0 and r = Q / 2, one swap of the wave's halves
// more for the odd groups of Q = 4.
//
// Around the arithmetic, the schedule of the 512-point kernel: one workgroup of 8 waves per CU whose waves
// draw steps from a counter in LDS; the span of the next step comes by LDS-DMA into a slot of the wave's own
// while the current one is transformed; rows are staged through LDS in output order and leave as 16-byte
// non-temporal stores.  A step's block of rows (1026 or 1025 values) starts at any multiple of 4 bytes, so
// the staging is shifted to put the 16-byte boundaries of the OUTPUT on 16-byte boundaries of the LDS, the
// body leaves as four vector stores and the 0 to 3 values before and after it as one more instruction: five
// stores per step whatever the alignment, which keeps the counted wait for the next span exact.
#include "stft_common.h"
#include "fft16.h"

namespace dsp {

struct R512XArgs {
    long long ldx;        // elements between signals
    int N;                // samples per signal
    int L;                // window length
    int W;                // frames per signal
    int stp;              // hop
    int ss0;              // start of frame 0
    int rd_shift;         // edge == 1: samples are read at n - rd_shift
    int edge;             // 0 reflect, 1 zero
    Epilogue epi;
    int sps;              // steps per signal
    int steps;            // B * sps
};

constexpr int R512X_TILE = 4 * R512_EXF * 2;          // floats of a wave's exchange tile / row staging
constexpr int R512X_WAVES = 8;

#ifndef R512X_DTW_LDS
#define R512X_DTW_LDS 1         // 1: the twiddles of the radix-Q stage come from an LDS table instead of 32 registers
#endif
#ifndef R512X_TWD_LDS
#define R512X_TWD_LDS 0         // 1: the pass-1 twiddles of the 256-point transform too (30 registers)
#endif
#ifndef R512X_CHUNK
#define R512X_CHUNK 4           // rows of pass 1's input that are read and combined together
#endif
#ifndef R512X_MINIMAL
#define R512X_MINIMAL 0         // diagnostic builds: 1 = only the plain full-window instantiations
#endif

__device__ __forceinline__ void r512x_glds16(const void *g, const unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ unsigned r512x_lds_addr(const void *p)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
template <int N> __device__ __forceinline__ void r512x_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// one lane adds 1 to the workgroup's counter; the old value is in lane 0 of t once r512x_drawn has waited
__device__ __forceinline__ void r512x_draw(unsigned &t, const unsigned cnt)
{
    unsigned long long keep;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %2, %3\n\ts_mov_b64 exec, %1"
                 : "=&v"(t), "=&s"(keep) : "v"(cnt), "v"(1u) : "memory");
}
__device__ __forceinline__ int r512x_drawn(unsigned &t)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t) :: "memory");
    return (int)__builtin_amdgcn_readfirstlane(t);
}
__device__ __forceinline__ float r512x_mirror(float v)       // lane 15 - i of the same row of 16
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));
}
__device__ __forceinline__ float r512x_ror1(float v)         // lane (i - 1) % 16 of the same row
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
}
typedef unsigned r512x_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float r512x_xor32(float v, bool low_half)   // lane i ^ 32
{
    const r512x_u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(low_half ? t.y : t.x);
}

template <int Q, bool FULL, bool MN0, bool EPI, int SPAN, bool ALN>
__global__ __launch_bounds__(64 * R512X_WAVES, 2) void stft_r512x_kernel(float *__restrict__ Y, const float *__restrict__ X,
                                                                         const float *__restrict__ win, const float *__restrict__ tw,
                                                                         const R512XArgs a)
{
    static_assert(Q == 2 || Q == 4, "nfft 1024 or 2048");
    using C = cx<float>;
    typedef float f2e __attribute__((ext_vector_type(2), aligned(4)));
    typedef float f4nt __attribute__((ext_vector_type(4)));
    constexpr int NFFT = 512 * Q, M = 256 * Q, F = M + 1, FPS = 4 / Q, NWV = R512X_WAVES, NV = SPAN / 256;
    constexpr int WAVE_LDS = R512X_TILE + SPAN;
    static_assert(FPS * F + 3 <= R512X_TILE, "row staging fits the tile");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane >> 4, j = lane & 15;
    const int f = grp / Q, r = grp % Q;                    // frame of the step, residue of the bins mod Q
    const int L = a.L;

    float *xs = reinterpret_cast<float *>(smem_raw) + wv * WAVE_LDS;   // exchange tile, then row staging
    float2 *ex = reinterpret_cast<float2 *>(xs);
    float *slot = xs + R512X_TILE;                                     // the span
    float *wtab = reinterpret_cast<float *>(smem_raw) + NWV * WAVE_LDS;   // halved window, signed: [parity of r][NFFT]
    float2 *dtab = reinterpret_cast<float2 *>(wtab + 2 * NFFT);        // W_M^((j + 16 n1) r) at [r][n1][j]
    float2 *ttab = dtab + (R512X_DTW_LDS ? 256 * Q : 0);               // W256^(j k1) at [k1][j]
    float *dummy = reinterpret_cast<float *>(ttab + (R512X_TWD_LDS ? 256 : 0));   // 1 KiB that placeholder LDS-DMA loads write
    unsigned *ticket = reinterpret_cast<unsigned *>(dummy + 256);

    // ---- tables and per-lane constants ----
    // window table: entry l of copy p is h_l / 2 times the sign group r (parity p) gives quarter g = l / 512:
    // W_Q^(g r) = (-i)^(g r); its real sign part (-1)^(g r / 2 ...) is folded here for the quarters g >= Q / 2,
    // i.e. the factor (-1)^r of z_(m + M/2) against z_m
    for (int l = tid; l < 2 * NFFT; l += 64 * NWV) {
        const int p = l / NFFT, ll = l - p * NFFT;
        const float h = (ll < L) ? 0.5f * win[ll] : 0.0f;
        wtab[l] = (p && ll >= NFFT / 2) ? -h : h;
    }
    auto twn = [&](int idx) {                              // W_nfft^idx, idx < nfft (the table holds idx < M)
        float sg = 1.0f;
        if (idx >= M) { idx -= M; sg = -1.0f; }
        return C{sg * tw[2 * idx], sg * tw[2 * idx + 1]};
    };
    C twd[16];                                             // W256^(j k1)
    if constexpr (!R512X_TWD_LDS) {
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) { twd[k1] = twn(2 * Q * j * k1); }
    } else if (tid < 256) {
        const C t = twn(2 * Q * (tid & 15) * (tid >> 4));
        ttab[tid] = make_float2(t.x, t.y);
    }
    C dtw[16];                                             // W_M^((j + 16 n1) r)
    if constexpr (!R512X_DTW_LDS) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) { dtw[n1] = twn((2 * (j + 16 * n1) * r) & (NFFT - 1)); }
    } else {
        for (int i = tid; i < 256 * Q; i += 64 * NWV) {
            const C t = twn((2 * (i & 255) * (i >> 8)) & (NFFT - 1));      // [r][m], m = j + 16 n1
            dtab[i] = make_float2(t.x, t.y);
        }
    }
    const float2 *drow = dtab + 256 * r + j;
    C ow[8];                                               // -i W_nfft^K, K = Q (j + 16 k2) + r
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) { const int K = Q * (j + 16 * k2) + r; ow[k2] = C{tw[2 * K + 1], -tw[2 * K]}; }
    // Q = 4: (-i)^r, the factor of the odd quarters against the even ones
    const C tq = (r == 0) ? C{1.f, 0.f} : (r == 1) ? C{0.f, -1.f} : (r == 2) ? C{-1.f, 0.f} : C{0.f, 1.f};
    if (tid == 0) { *ticket = NWV; }
    __syncthreads();

    const int reach = (FPS - 1) * a.stp;
    const int need = reach + L;                            // samples of a span that pass 1 uses
    const int nvec = (need + 3) / 4;
    const float *xrow = slot + f * a.stp + 2 * j;
    const float *wrow = wtab + (r & 1) * NFFT + 2 * j;

    // ---- fetch of one span into the slot: NV LDS-DMA loads whatever the case (the counted wait relies on
    // that); a span that crosses an edge of its signal is gathered sample by sample and NV placeholder loads follow
    auto load_span = [&](const float *xb, const int s0) {
        const float *src = xb + (s0 - (a.edge ? a.rd_shift : 0));
        const bool inside = s0 >= 0 && s0 + 4 * nvec <= a.N && (a.edge == 0 || s0 >= a.rd_shift);
        if (inside) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                // lanes past the last vector that pass 1 uses read that vector again
                r512x_glds16(src + min(q * 64 + lane, nvec - 1) * 4, __builtin_amdgcn_readfirstlane(r512x_lds_addr(slot + q * 256)));
            }
        } else {
            FrameDesc d{};
            d.N = a.N; d.edge = a.edge;
#pragma clang loop unroll(disable)
            for (int q = 0; q < NV; ++q) {
                float t[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = (q * 64 + lane) * 4 + v;
                    t[v] = (i < need) ? fetch_sample(xb, d, (long long)s0, (long long)a.rd_shift, i) : 0.f;
                }
                reinterpret_cast<float4 *>(slot)[q * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) { r512x_glds16(tw + lane * 4, __builtin_amdgcn_readfirstlane(r512x_lds_addr(dummy))); }
        }
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
    float *yb = Y + (long long)b * a.W * F;

    const unsigned cnt_addr = r512x_lds_addr(ticket);
    unsigned drawn = 0;
    int nsid = 0, nw = 0;
    const float *nxb = xb;
    float *nyb = yb;
    bool more = false;
    auto advance = [&]() {
        const int t = r512x_drawn(drawn);
        nsid = first + (t >> 3) * round + (t & 7);
        more = nsid < s_hi;
        if (more) {
            const int nb = nsid / a.sps;
            nw = (nsid - nb * a.sps) * FPS;
            nxb = X + (long long)nb * a.ldx;
            nyb = Y + (long long)nb * a.W * F;
        }
    };

    bool first_step = true;
    load_span(xb, a.ss0 + w * a.stp);

    while (true) {
        r512x_draw(drawn, cnt_addr);                       // the step after this one: asked for now, needed further down
        // The span of this step was asked for in the middle of the last one; memory operations complete in issue
        // order and behind its NV loads lie that step's 5 row stores, which stay in flight.
        if (first_step) { r512x_wait_vm<0>(); } else { r512x_wait_vm<5>(); }
        first_step = false;
        wave_sync();

        // ---- pass 1 input: the lane's combination of the Q quarters, rows n1 of column j ----
        C zz[16];
        auto lds2 = [&](const float *p) {
            if constexpr (ALN) { const float2 t = *reinterpret_cast<const float2 *>(p); return C{t.x, t.y}; }
            else { const f2e t = *reinterpret_cast<const f2e *>(p); return C{t.x, t.y}; }
        };
        auto sample = [&](const int n1, const int g) {
            C v = lds2(xrow + 32 * n1 + 512 * g);
            if (!FULL) {
                const int l = 32 * n1 + 2 * j + 512 * g;
                v.x = (l < L) ? v.x : 0.f;
                v.y = (l + 1 < L) ? v.y : 0.f;
            }
            return v;
        };
        auto tap = [&](const int n1, const int g) {
            const float2 t = *reinterpret_cast<const float2 *>(wrow + 32 * n1 + 512 * g);
            return C{t.x, t.y};
        };
        auto twiddle_in = [&](const int n1) {
            if constexpr (R512X_DTW_LDS) { const float2 t = drow[16 * n1]; return C{t.x, t.y}; }
            else { return dtw[n1]; }
        };
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            if constexpr (Q == 2) {
                zz[n1] = cx_fma(sample(n1, 1), tap(n1, 1), sample(n1, 0) * tap(n1, 0));
            } else {
                const C p = cx_fma(sample(n1, 2), tap(n1, 2), sample(n1, 0) * tap(n1, 0));
                const C q = cx_fma(sample(n1, 3), tap(n1, 3), sample(n1, 1) * tap(n1, 1));
                // p + (-i)^r q
                const C t = cx_fma(q, C{tq.x, tq.x}, p);
                zz[n1] = cx_fma(cx_swap(q), C{-tq.y, tq.y}, t);
            }
            if constexpr (!MN0) { zz[n1] = cx_mul<float>(zz[n1], twiddle_in(n1)); }
            // the rows of a chunk are complete before the reads of the next chunk go out (left alone the
            // compiler issues the reads of all 16 rows first and runs out of registers)
            if (n1 % R512X_CHUNK == R512X_CHUNK - 1) {
#pragma unroll
                for (int i = n1 + 1 - R512X_CHUNK; i <= n1; ++i) { asm volatile("" : "+v"(zz[i]) :: "memory"); }
            }
        }
        if (MN0) {
            // the mean of the windowed frame is the sum over its group r = 0 (every sign +) divided by L; taken
            // off every windowed sample inside the window, it changes the lane's combination by mean times the
            // same combination of the 0/1 masks
            C acc = zz[0];
#pragma unroll
            for (int n1 = 1; n1 < 16; ++n1) { acc += zz[n1]; }
            float sum = acc.x + acc.y;
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
            const float mean = __shfl(sum, f * 16 * Q) / (float)L;
            const float sg = (r & 1) ? -1.f : 1.f;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                auto mask = [&](const int g) {
                    const int l = 32 * n1 + 2 * j + 512 * g;
                    return C{(FULL || l < L) ? 1.f : 0.f, (FULL || l + 1 < L) ? 1.f : 0.f};
                };
                C cm;
                if constexpr (Q == 2) {
                    cm = mask(0) + mask(1) * C{sg, sg};
                } else {
                    const C p = mask(0) + mask(2) * C{sg, sg};
                    const C q = mask(1) + mask(3) * C{sg, sg};
                    const C t = cx_fma(q, C{tq.x, tq.x}, p);
                    cm = cx_fma(cx_swap(q), C{-tq.y, tq.y}, t);
                }
                zz[n1] -= cm * C{mean, mean};
            }
        }
        if constexpr (MN0) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) { zz[n1] = cx_mul<float>(zz[n1], twiddle_in(n1)); }
        }
        dft16c<float, 16>(zz);
        wave_sync();                                       // all lanes have read the span
        // The slot takes the next step's span.  The reads of pass 1 must have RETURNED before the first load into
        // the slot is issued (the compiler may sink their consumers, and with them its own wait, below the
        // assembly statements): r512x_drawn waits for every LDS operation of the wave.  Past the end of the
        // workgroup's share the current span is fetched again: the number of loads per step stays fixed.
        advance();
        load_span(more ? nxb : xb, a.ss0 + (more ? nw : w) * a.stp);
        {
            float2 *exf = ex + grp * R512_EXF + j;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                C v = zz[slot16(k1)];
                if (k1 > 0) {
                    if constexpr (R512X_TWD_LDS) { const float2 t = ttab[k1 * 16 + j]; v = cx_mul<float>(v, t.x, t.y); }
                    else { v = cx_mul<float>(v, twd[k1]); }
                }
                exf[k1 * 17] = make_float2(v.x, v.y);
            }
        }
        wave_sync();

        // ---- pass 2: lane (group, k1 = j): 16-point FFT over the columns ----
        {
            const float2 *exf = ex + grp * R512_EXF + j * 17;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) { const float2 t = exf[n2]; zz[n2] = C{t.x, t.y}; }
        }
        dft16c<float, 16>(zz);

        // ---- real-FFT split fused with power.  The lane holds Z[K] (halved), K = Q (j + 16 k2) + r, in
        // register k2.  Z[M - K] is, for r = 0: register 15 - k2 of lane (16 - j) % 16 (register (16 - k2) % 16
        // of lane 0 itself); for r > 0: register 15 - k2 of lane 15 - j of group Q - r.
        // With A = Z_K + conj Z_(M-K), B = Z_K - conj Z_(M-K), T = -i W_nfft^K B:
        // |X_K|^2 = |A + T|^2 and |X_(M-K)|^2 = |A - T|^2.
        float pw[16];                                      // pw[k2] = P[K], pw[8 + k2] = P[M - K]
        float pmid = 4.0f * cx_norm<float>(zz[slot16(8)]); // K = M / 2: lane j == 0 of group 0 only
        const bool own0 = (r == 0) && (j == 0);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const C z = zz[slot16(k2)];
            float x0 = zz[slot16((16 - k2) & 15)].x, x1 = zz[slot16(15 - k2)].x;
            float y0 = zz[slot16((16 - k2) & 15)].y, y1 = zz[slot16(15 - k2)].y;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));   // (keeps the select on the values, not on the index)
            const float sx = own0 ? x0 : x1;
            const float sy = own0 ? y0 : y1;
            const float mx1 = r512x_mirror(sx), my1 = r512x_mirror(sy);
            const float mx2 = r512x_ror1(mx1), my2 = r512x_ror1(my1);
            C m;
            if constexpr (Q == 2) {
                m = C{r == 0 ? mx2 : mx1, r == 0 ? my2 : my1};
            } else {
                const float mx3 = r512x_xor32(mx1, lane < 32), my3 = r512x_xor32(my1, lane < 32);
                m = C{r == 0 ? mx2 : (r == 2 ? mx1 : mx3), r == 0 ? my2 : (r == 2 ? my1 : my3)};
            }
            const C A = cx_add_conj<float>(z, m), B = cx_sub_conj<float>(z, m);
            const C Tw = cx_mul<float>(B, ow[k2]);
            pw[k2] = cx_norm<float>(A + Tw);
            pw[8 + k2] = cx_norm<float>(A - Tw);
        }
        if (EPI) {
            power_epilogue_n(pw, a.epi);
            pmid = power_epilogue(pmid, a.epi);
        }
        // the step's block of rows in the output, and how it lies against 16-byte boundaries
        float *yblk = yb + (size_t)w * F;
        const int nout = __builtin_amdgcn_readfirstlane(min(FPS, a.W - w) * F);
        const int head = __builtin_amdgcn_readfirstlane((int)(((16u - ((unsigned)reinterpret_cast<uintptr_t>(yblk) & 15u)) & 15u) >> 2));
        const int sh = (4 - head) & 3;                     // value e of the block is staged at xs[sh + e]
        wave_sync();                                       // every lane has read its tile row: reuse the region
        {
            float *pf = xs + sh + f * F;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const int K = Q * (j + 16 * k2) + r;
                pf[K] = pw[k2];
                pf[M - K] = pw[8 + k2];
            }
            if (own0) { pf[M / 2] = pmid; }
        }
        wave_sync();

        // ---- five stores: four of whole 16-byte vectors, one for the values before and after them.  Surplus
        // lanes repeat a valid store (same value to the same address).
        {
            const int nb = (nout - head) >> 2, tail = nout - head - 4 * nb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = min(lane + 64 * i, nb - 1);
                const f4nt val = *reinterpret_cast<const f4nt *>(xs + sh + head + 4 * v);
                __builtin_nontemporal_store(val, reinterpret_cast<f4nt *>(yblk + head + 4 * v));
            }
            int e = nout - 1;
            if (lane < head) { e = lane; }
            else if (lane >= 4 && lane - 4 < tail) { e = head + 4 * nb + lane - 4; }
            __builtin_nontemporal_store(xs[sh + e], yblk + e);
        }
        if (!more) { break; }
        wave_sync();                                       // staging reads are done before the next step writes the tile
        sid = nsid; w = nw; xb = nxb; yb = nyb;
    }
    // no LDS-DMA load may land, and no draw return, after the wave has ended
    r512x_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// nfft 1024 or 2048, frames at ss0 + w * hop with the frames of a step inside 2048 samples, and every index
// of one signal's input and output inside 31 bits
bool r512x_supported(const FrameDesc &d)
{
    if (d.nfft != 1024 && d.nfft != 2048) { return false; }
    const int fps = 4 / (d.nfft / 512);
    return d.tbl == nullptr && d.stp >= 1 && (long long)(fps - 1) * d.stp + d.nfft <= 2048 && d.L >= 1 && d.L <= d.nfft &&
           d.N < (1ll << 31) - 4096 && (long long)d.W * (d.nfft / 2 + 1) < (1ll << 30) && (long long)d.B * d.W < (1ll << 30) &&
           d.ss0 > -(1ll << 30) && d.ss0 < (1ll << 30) && d.rd_shift >= 0 && d.rd_shift < (1ll << 30);
}

template <int Q, bool FULL, bool MN0, bool EPI, int SPAN, bool ALN>
static hipError_t launch_r512x_inst(float *Y, const float *X, const float *win, const float *tw, const R512XArgs &a, hipStream_t s)
{
    const size_t lds = ((size_t)R512X_WAVES * (R512X_TILE + SPAN) + 2 * 512 * Q + (R512X_DTW_LDS ? 512 * Q : 0) + (R512X_TWD_LDS ? 512 : 0) + 256) * sizeof(float) + 16;
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_r512x_kernel<Q, FULL, MN0, EPI, SPAN, ALN>), lds, grant); e != hipSuccess) { return e; }
    long long grid = ((long long)a.steps + R512X_WAVES - 1) / R512X_WAVES;
    if (grid > 256) { grid = 256; }                     // one 8-wave workgroup per CU
    grid = (grid + 7) & ~7ll;                           // whole groups of 8 workgroups (one per XCD)
    hipLaunchKernelGGL((stft_r512x_kernel<Q, FULL, MN0, EPI, SPAN, ALN>), dim3((unsigned)grid), dim3(64 * R512X_WAVES), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

template <int Q, int SPAN, bool ALN>
static hipError_t launch_r512x_shape(float *Y, const float *X, const float *win, const float *tw, const R512XArgs &a, const FrameDesc &d, hipStream_t s)
{
    const bool full = d.L == d.nfft, epi = d.epi.any();
#define DSP_R512X(F_, M_, E_) launch_r512x_inst<Q, F_, M_, E_, SPAN, ALN>(Y, X, win, tw, a, s)
#if R512X_MINIMAL
    if (full && !d.mn0 && !epi) { return DSP_R512X(true, false, false); }
    return hipErrorInvalidValue;
#else
    if (full) {
        if (d.mn0) { return epi ? DSP_R512X(true, true, true) : DSP_R512X(true, true, false); }
        return epi ? DSP_R512X(true, false, true) : DSP_R512X(true, false, false);
    }
    if (d.mn0) { return epi ? DSP_R512X(false, true, true) : DSP_R512X(false, true, false); }
    return epi ? DSP_R512X(false, false, true) : DSP_R512X(false, false, false);
#endif
#undef DSP_R512X
}

hipError_t launch_stft_r512x(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s)
{
    const int Q = d.nfft / 512, fps = 4 / Q;
    R512XArgs a{};
    a.ldx = d.ldx; a.N = (int)d.N; a.L = d.L; a.W = d.W; a.stp = d.stp; a.ss0 = (int)d.ss0;
    a.rd_shift = (int)d.rd_shift; a.edge = d.edge; a.epi = d.epi;
    a.sps = (d.W + fps - 1) / fps;
    a.steps = d.B * a.sps;
    if (Q == 4) { return launch_r512x_shape<4, 2048, true>(Y, X, win, tw, a, d, s); }
    if (d.stp & 1) { return launch_r512x_shape<2, 2048, false>(Y, X, win, tw, a, d, s); }
    if (d.stp + 1024 <= 1280) { return launch_r512x_shape<2, 1280, true>(Y, X, win, tw, a, d, s); }
    return launch_r512x_shape<2, 2048, true>(Y, X, win, tw, a, d, s);
}

}  // namespace dsp
