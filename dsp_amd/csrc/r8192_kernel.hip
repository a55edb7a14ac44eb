// Register-resident 8192-point fused STFT kernel for gfx950 (float32): framing + window + [mean removal] + real FFT +
// power + [sqrt][log], one wavefront per frame, four waves per workgroup (one per SIMD, 512 registers each), one
// workgroup per CU (a 33.3 KB tile per wave).
//
// The 8192 real samples of a frame are packed as 4096 complex points z_n = x_2n + i x_2n+1 and transformed as 64 x 64,
// every complex value one (re, im) register pair (packed two-component float instructions):
//   loads   lane l is the complex column n2 = l and takes its 64 rows n1: sixty-four 8-byte loads per lane, each wave
//           load 512 contiguous bytes; the next frame's loads are issued under pass 2 of the current one;
//   pass 1  window (a full window: the lane's 128 values stay in registers for the whole launch; a shorter one is read
//           from memory frame by frame, 32 KB that the L2 keeps: the LDS is full), one 64-point FFT over the rows in
//           registers (four 16-point transforms and 16 radix-4 butterflies, the constant twiddles folded into the
//           additions: fft16.h); twiddle W4096^(n2 k1) as two per-lane factors (k1 = 8 q + b);
//   exchange through LDS: 64 rows k1 of 64 values, pitch 65 (conflict-free both ways);
//   pass 2  lane r = k1 runs one 64-point FFT over n2 and holds Z[r + 64 k2];
//   split   of the pair Z[k], Z[4096 - k] that the real-FFT split needs, lane r holds Z[k] for k = r + 64 a, a < 32,
//           itself and fetches the other from the upper 32 values of lane 64 - r, which go through LDS once; split and
//           power in registers; a store instruction writes 64 consecutive bins, one per lane.
// Frames are dealt round-robin inside 8 contiguous ranges (one per group of workgroups that share an XCD) and the
// waves of a workgroup stay in step (a barrier every other round), so that the part of a frame that overlaps its
// neighbours is served by that XCD's L2.
#include <type_traits>
#include "stft_common.h"
#include "fft16.h"

namespace dsp {

struct R8192Args {
    long long ldx, N, ss0, rd_shift;
    const int *tbl;
    int L, W, stp, edge;
    int Wy;               // frames per signal in the output array
    Epilogue epi;
    int frames;           // B * W
};

constexpr int R8192_PITCH = 65;                         // complex slots per exchange row
constexpr int R8192_WAVE_C = 64 * R8192_PITCH;          // complex slots per wave (>= 4096)
constexpr int R8192_WAVES = 4;
constexpr int R8192_X2P = 33;                           // second exchange: complex slots per lane (32 values + 1)
typedef float r8_pair_u __attribute__((ext_vector_type(2), aligned(4)));   // 8-byte access at 4-byte alignment (frame starts are only 4-byte aligned)

template <bool FULL, bool MN0, bool EPI>
__global__ __launch_bounds__(64 * R8192_WAVES, 1) void stft_r8192_kernel(float *__restrict__ Y, const float *__restrict__ X,
                                                                         const float *__restrict__ win, const float *__restrict__ tw,
                                                                         const R8192Args a)
{
    using C2 = cx<float>;
    constexpr int M = 4096, F = 4097;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    C2 *ex = reinterpret_cast<C2 *>(smem_raw) + wv * R8192_WAVE_C;            // this wave's tile

    // per-lane twiddles: W4096^(n2 b), b = 1..7, and W4096^(8 n2 q), q = 1..7; the table holds W8192^k, k < 4096
    auto tw4096 = [&](int m) {
        int idx = (2 * m) & 8191;
        float sg = 1.0f;
        if (idx >= 4096) { idx -= 4096; sg = -1.0f; }
        return C2{sg * tw[2 * idx], sg * tw[2 * idx + 1]};
    };
    C2 tb[8], ta[8];
#pragma unroll
    for (int b = 1; b < 8; ++b) { tb[b] = tw4096(lane * b); }
#pragma unroll
    for (int q = 1; q < 8; ++q) { ta[q] = tw4096(8 * lane * q); }
    // Second exchange: lane l leaves Z[l + 64 (32 + i)] at slot 33 l + i.  The split pairs Z[r + 64 a] with
    // Z[4096 - r - 64 a] = Z[(64 - r) + 64 (63 - a)]: slot 31 - a of lane 64 - r; for r = 32 that is the lane itself; for
    // r = 0 the partner is Z[64 (64 - a)], slot 32 - a of the lane itself (a = 0, bin 0, pairs with itself).
    const int pslot = ((64 - lane) & 63) * R8192_X2P + (lane == 0 ? 1 : 0);
    const C2 ow = C2{tw[2 * lane + 1], -tw[2 * lane]};                        // -i W8192^r, for the split

    // 8 ranges of frames (blockIdx % 8 = XCD), static round-robin inside a range; the 4 waves of a workgroup take 4
    // neighbouring frames per round
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int f_lo = (int)((long long)a.frames * xcd / 8), f_hi = (int)((long long)a.frames * (xcd + 1) / 8);
    const int fstride = nwg_x * R8192_WAVES;
    int fid = f_lo + (blockIdx.x >> 3) * R8192_WAVES + wv;
    int nfid = fid + fstride;
    if (fid >= f_hi) { return; }

    struct Where { long long row; const float *xb; long long ss, rdoff; bool inside; };
    auto locate = [&](int f) {
        Where q;
        const int b = f / a.W;
        const int w = f - b * a.W;
        q.row = (long long)b * a.Wy + w;
        q.xb = X + (long long)b * a.ldx;
        if (a.tbl) { q.ss = a.tbl[2 * w]; q.rdoff = q.ss - (long long)a.tbl[2 * w + 1]; }
        else { q.ss = a.ss0 + (long long)w * a.stp; q.rdoff = a.rd_shift; }
        if (!a.edge) { q.rdoff = 0; }
        q.inside = q.ss >= 0 && q.ss + 8192 <= a.N && q.ss >= q.rdoff;
        return q;
    };
    C2 z0[16], z1[16], z2[16], z3[16];   // rows n1 = 4 m + j of the column in zj[m]; later the four quarters of a 64-point transform
    C2 pre[64];                          // the next frame's samples
    const int loff = 2 * lane;           // first sample of the lane inside a frame; row i adds 128
    auto load_inside = [&](const Where &q) {
        const float *src = q.xb + (q.ss - q.rdoff) + loff;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            r8_pair_u v = {0.f, 0.f};
            if (FULL || 128 * i < L) { v = *reinterpret_cast<const r8_pair_u *>(src + 128 * i); }
            pre[i] = C2{v.x, v.y};
        }
    };
    // a frame that crosses an edge of its signal: gather sample by sample into the (idle) tile first
    auto load_edge = [&](const Where &q) {
        FrameDesc d{};
        d.N = a.N; d.edge = a.edge;
        float *stage = reinterpret_cast<float *>(ex);
        wave_sync();
        for (int l = lane; l < 8192; l += 64) { stage[l] = (l < L) ? fetch_sample(q.xb, d, q.ss, q.rdoff, l) : 0.f; }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 64; ++i) { pre[i] = *reinterpret_cast<const C2 *>(stage + loff + 128 * i); }
        asm volatile("" ::: "memory");
    };

    // a full window: the lane's 128 values of the halved window, for the whole launch
    C2 wreg[FULL ? 64 : 1];
    if constexpr (FULL) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const r8_pair_u w2 = *reinterpret_cast<const r8_pair_u *>(win + loff + 128 * i);
            wreg[i] = C2{0.5f * w2.x, 0.5f * w2.y};
        }
    }
    Where cur = locate(fid);
    if (cur.inside) { load_inside(cur); } else { load_edge(cur); }

    // forward 64-point DFT of x[n], n = 4 m + j in qj[m]; on return X[k] is in q(k >> 4)[slot16(k & 15)]
    auto dft64f = [](C2 (&q0)[16], C2 (&q1)[16], C2 (&q2)[16], C2 (&q3)[16]) __attribute__((always_inline)) {
        dft16f<float, 16>(q0);
        dft16f<float, 16>(q1);
        dft16f<float, 16>(q2);
        dft16f<float, 16>(q3);
#define R8192_B(K) dft4c_tw<float, 64, K, 2 * K, 3 * K>(q0[slot16(K)], q1[slot16(K)], q2[slot16(K)], q3[slot16(K)]);
        R8192_B(0) R8192_B(1) R8192_B(2) R8192_B(3) R8192_B(4) R8192_B(5) R8192_B(6) R8192_B(7)
        R8192_B(8) R8192_B(9) R8192_B(10) R8192_B(11) R8192_B(12) R8192_B(13) R8192_B(14) R8192_B(15)
#undef R8192_B
    };
    // the value X[k] of a transform left by dft64f
    auto at = [&](int k) __attribute__((always_inline)) -> C2 & {
        const int s = slot16(k & 15);
        return (k >> 4) == 0 ? z0[s] : (k >> 4) == 1 ? z1[s] : (k >> 4) == 2 ? z2[s] : z3[s];
    };
    // ... and where input n of a transform goes
    auto in = [&](int n) __attribute__((always_inline)) -> C2 & { return (n & 3) == 0 ? z0[n >> 2] : (n & 3) == 1 ? z1[n >> 2] : (n & 3) == 2 ? z2[n >> 2] : z3[n >> 2]; };

    int iter = 0;
    for (;;) {
        // ---------------- pass 1: window, 64-point FFT over n1 ----------------
        {
            C2 acc = C2{0.f, 0.f};
            int lim = L - loff;                              // sample l = 128 i + loff + c is inside the window iff 128 i + c < lim
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                C2 hw;
                if constexpr (FULL) { hw = wreg[i]; }
                else {
                    const float *wp = win + loff + 128 * i;
                    hw = C2{(128 * i < lim) ? 0.5f * wp[0] : 0.f, (128 * i + 1 < lim) ? 0.5f * wp[1] : 0.f};
                }
                C2 v = pre[i];
                if (!FULL) {                                 // keep data past the window out (0 * NaN)
                    v.x = (128 * i < lim) ? v.x : 0.f;
                    v.y = (128 * i + 1 < lim) ? v.y : 0.f;
                }
                v = v * hw;
                if (MN0) { acc += v; }
                in(i) = v;
            }
            if (MN0) {
                float sum = acc.x + acc.y;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { sum += __shfl_xor(sum, o); }
                const float mean = sum / (float)L;
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    in(i) -= C2{(FULL || 128 * i < lim) ? mean : 0.f, (FULL || 128 * i + 1 < lim) ? mean : 0.f};
                }
            }
        }
        dft64f(z0, z1, z2, z3);
        // twiddle W4096^(n2 k1), k1 = 8 q + b
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            C2 t[8];
#pragma unroll
            for (int bb = 1; bb < 8; ++bb) { t[bb] = cx_mul_first<float>(at(8 * q + bb), tb[bb]); }
#pragma unroll
            for (int bb = 1; bb < 8; ++bb) { at(8 * q + bb) = cx_mul_second<float>(at(8 * q + bb), tb[bb], t[bb]); }
            if (q) {
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) { t[bb] = cx_mul_first<float>(at(8 * q + bb), ta[q]); }
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) { at(8 * q + bb) = cx_mul_second<float>(at(8 * q + bb), ta[q], t[bb]); }
            }
        }
        // exchange: row k1, column n2 = lane
        wave_sync();                                       // the previous frame's reads of the tile are done
#pragma unroll
        for (int k1 = 0; k1 < 64; ++k1) { ex[k1 * R8192_PITCH + lane] = at(k1); }
        wave_sync();
        // where the next frame lies; its loads go out now, under pass 2
        const bool more = nfid < f_hi;
        const Where nxt = locate(more ? nfid : fid);
        if ((iter++ & 1) == 0) { __builtin_amdgcn_s_barrier(); }   // the workgroup's waves request neighbouring frames together
        if (nxt.inside) { load_inside(nxt); }

        // ---------------- pass 2: the lane's row r = lane: 64-point FFT over the 64 columns ----------------
        {
            const C2 *row = ex + lane * R8192_PITCH;
#pragma unroll
            for (int n = 0; n < 64; ++n) { in(n) = row[n]; }
        }
        dft64f(z0, z1, z2, z3);
        wave_sync();                                       // every lane has its row: the tile is free
        // second exchange: the upper half of the lane's values, Z[r + 64 (32 + i)], for its partner
        {
            C2 *x2w = ex + lane * R8192_X2P;
#pragma unroll
            for (int i = 0; i < 32; ++i) { x2w[i] = at(32 + i); }
        }
        wave_sync();
        // ---------------- split + power + store: the lane's bins k = r + 64 a, a < 32, and 4096 - k ----------------
        float *y = Y + cur.row * F;
        const C2 *zpart = ex + pslot;                                      // Z[4096 - k] at zpart[31 - a]
        float *yf = y + lane, *yr = y + M - lane;
        // Four pairs at a time, stage by stage.  The factor of the split, -i W8192^k = (-i W8192^r) W128^a, is taken in two
        // parts: the constant W128^a in folded form (fft16.h), then the lane's factor.
        auto pairs4 = [&](auto g_c) __attribute__((always_inline)) {
            constexpr int G = decltype(g_c)::value;
            constexpr TwF f[4] = {twf_ct(4 * G, 128), twf_ct(4 * G + 1, 128), twf_ct(4 * G + 2, 128), twf_ct(4 * G + 3, 128)};
            C2 A[4], B[4], t[4], P[4], Q[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int aa = 4 * G + c;
                const C2 z = at(aa);
                C2 zm = zpart[31 - aa];
                if (aa == 0) { zm = (lane == 0) ? z : zm; }                // bin 0 pairs with itself
                A[c] = cx_add_conj<float>(z, zm);
                B[c] = cx_sub_conj<float>(z, zm);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { if (f[c].t != 0.0) { B[c] = cx_pre<float>(B[c], (float)f[c].t); } }
#pragma unroll
            for (int c = 0; c < 4; ++c) { t[c] = cx_mul_first<float>(B[c], ow); }
#pragma unroll
            for (int c = 0; c < 4; ++c) { B[c] = cx_mul_second<float>(B[c], ow, t[c]); }
            // X_k = A + T, X_(4096-k) = conj(A - T), T = sig i^m B
#define R8192_PQ(c) P[c] = cx_axpy_i<float, f[c].m>(A[c], (float)f[c].sig, B[c]); Q[c] = cx_axpy_i<float, f[c].m + 2>(A[c], (float)f[c].sig, B[c]);
            R8192_PQ(0) R8192_PQ(1) R8192_PQ(2) R8192_PQ(3)
#undef R8192_PQ
            float p[4], q[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { p[c] = fmaf(P[c].y, P[c].y, P[c].x * P[c].x); q[c] = fmaf(Q[c].y, Q[c].y, Q[c].x * Q[c].x); }
            if (EPI) { power_epilogue_n(p, a.epi); power_epilogue_n(q, a.epi); }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                yf[64 * (4 * G + c)] = p[c];
                yr[-64 * (4 * G + c)] = q[c];
            }
        };
        using std::integral_constant;
        pairs4(integral_constant<int, 0>{}); pairs4(integral_constant<int, 1>{}); pairs4(integral_constant<int, 2>{}); pairs4(integral_constant<int, 3>{});
        pairs4(integral_constant<int, 4>{}); pairs4(integral_constant<int, 5>{}); pairs4(integral_constant<int, 6>{}); pairs4(integral_constant<int, 7>{});
        {
            // bin 2048 = row 0, k2 = 32: the first value lane 0 left in the second exchange; it pairs with itself (Z holds
            // half values).  Every lane stores the same value to the same address (no branch around a store).
            const C2 zmid = ex[0];
            float p = 4.0f * fmaf(zmid.y, zmid.y, zmid.x * zmid.x);
            if (EPI) { p = power_epilogue(p, a.epi); }
            y[2048] = p;
        }
        if (!more) { break; }
        if (!nxt.inside) { load_edge(nxt); }
        cur = nxt;
        fid = nfid;
        nfid += fstride;
    }
}

bool r8192_supported(const FrameDesc &d)
{
    return d.nfft == 8192 && d.L >= 1 && d.L <= 8192 && (long long)d.B * d.wy() < (1ll << 31) - 8192 && (long long)d.B * d.W < (1ll << 31) - 8192;
}

template <bool FULL, bool MN0, bool EPI>
static hipError_t launch_r8192_inst(float *Y, const float *X, const float *win, const float *tw, const R8192Args &a, hipStream_t s)
{
    const size_t lds = (size_t)R8192_WAVES * R8192_WAVE_C * sizeof(float2);
    static LdsGrant grant;
    if (hipError_t e = ensure_lds(reinterpret_cast<const void *>(&stft_r8192_kernel<FULL, MN0, EPI>), lds, grant); e != hipSuccess) { return e; }
    long long grid = ((long long)a.frames + R8192_WAVES - 1) / R8192_WAVES;
    if (grid > 256) { grid = 256; }                      // one workgroup per CU (133 KB of LDS)
    grid = (grid + 7) & ~7ll;
    hipLaunchKernelGGL((stft_r8192_kernel<FULL, MN0, EPI>), dim3((unsigned)grid), dim3(64 * R8192_WAVES), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

hipError_t launch_stft_r8192(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s)
{
    R8192Args a{};
    a.ldx = d.ldx; a.N = d.N; a.ss0 = d.ss0; a.rd_shift = d.rd_shift; a.tbl = d.tbl;
    a.L = d.L; a.W = d.W; a.Wy = d.wy(); a.stp = d.stp; a.edge = d.edge; a.epi = d.epi;
    a.frames = d.B * d.W;
    const bool full = d.L == 8192, epi = d.epi.any();
#define DSP_R8192(F_, M_, E_) launch_r8192_inst<F_, M_, E_>(Y, X, win, tw, a, s)
    if (full) {
        if (d.mn0) { return epi ? DSP_R8192(true, true, true) : DSP_R8192(true, true, false); }
        return epi ? DSP_R8192(true, false, true) : DSP_R8192(true, false, false);
    }
    if (d.mn0) { return epi ? DSP_R8192(false, true, true) : DSP_R8192(false, true, false); }
    return epi ? DSP_R8192(false, false, true) : DSP_R8192(false, false, false);
#undef DSP_R8192
}

}  // namespace dsp
