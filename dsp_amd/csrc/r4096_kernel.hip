// Register-resident 4096-point fused STFT kernel for gfx950 (float32): framing + window +
// [mean removal] + real FFT + power + [sqrt][log], one wavefront per frame.
//
// The 4096 real samples of a frame are packed as 2048 complex points z_n = x_2n + i x_2n+1 and
// transformed as 32 x 64:
//   pass 1  lane n2 (0..63) loads its 32 points z[64 n1 + n2] straight from global memory (for a
//           fixed n1 the wave reads 512 contiguous bytes), multiplies by the window (LDS table),
//           runs a 32-point FFT over n1 in registers and multiplies by W2048^(n2 k1);
//   exchange through LDS (32 rows of pitch 66 complex: conflict-free 8-byte writes and reads);
//   pass 2  lane (k1, h) runs a 32-point FFT over the n2 of parity h; the two halves of the
//           64-point transform are combined with the neighbouring lane by one DPP quad swap;
//   split   Z goes to LDS once, each lane reads bins k and 2048-k, and the real-FFT split is fused
//           with the power step; rows are stored as whole 256-byte wave stores.
// Frames are dealt round-robin inside 8 contiguous ranges (one per group of workgroups that share an
// XCD), so the 3/4 of a frame that overlaps its neighbours is served by that XCD's L2 and every
// sample is fetched from HBM once.
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
    int amp, lg;
    int frames;           // B * W
};

constexpr int R4096_PITCH = 66;                         // complex slots per exchange row
constexpr int R4096_WAVE_C = 32 * R4096_PITCH;          // complex slots per wave (>= 2048 + 8)
constexpr int R4096_WAVES = 8;                          // waves per workgroup (share the window table)

// W32^k, k < 16, and W64^k, k < 32, as (cos, -sin)(2 pi k / n) pairs; indexed with compile-time
// constants only, so they fold into the instructions as literals
__device__ constexpr float W32C[32] = {1.000000000e+00f, -0.000000000e+00f, 9.807852507e-01f, -1.950903237e-01f, 9.238795042e-01f, -3.826834261e-01f, 8.314695954e-01f, -5.555702448e-01f, 7.071067691e-01f, -7.071067691e-01f, 5.555702448e-01f, -8.314695954e-01f, 3.826834261e-01f, -9.238795042e-01f, 1.950903237e-01f, -9.807852507e-01f, 6.123234263e-17f, -1.000000000e+00f, -1.950903237e-01f, -9.807852507e-01f, -3.826834261e-01f, -9.238795042e-01f, -5.555702448e-01f, -8.314695954e-01f, -7.071067691e-01f, -7.071067691e-01f, -8.314695954e-01f, -5.555702448e-01f, -9.238795042e-01f, -3.826834261e-01f, -9.807852507e-01f, -1.950903237e-01f};
__device__ constexpr float W64C[64] = {1.000000000e+00f, -0.000000000e+00f, 9.951847196e-01f, -9.801714122e-02f, 9.807852507e-01f, -1.950903237e-01f, 9.569403529e-01f, -2.902846634e-01f, 9.238795042e-01f, -3.826834261e-01f, 8.819212914e-01f, -4.713967443e-01f, 8.314695954e-01f, -5.555702448e-01f, 7.730104327e-01f, -6.343932748e-01f, 7.071067691e-01f, -7.071067691e-01f, 6.343932748e-01f, -7.730104327e-01f, 5.555702448e-01f, -8.314695954e-01f, 4.713967443e-01f, -8.819212914e-01f, 3.826834261e-01f, -9.238795042e-01f, 2.902846634e-01f, -9.569403529e-01f, 1.950903237e-01f, -9.807852507e-01f, 9.801714122e-02f, -9.951847196e-01f, 6.123234263e-17f, -1.000000000e+00f, -9.801714122e-02f, -9.951847196e-01f, -1.950903237e-01f, -9.807852507e-01f, -2.902846634e-01f, -9.569403529e-01f, -3.826834261e-01f, -9.238795042e-01f, -4.713967443e-01f, -8.819212914e-01f, -5.555702448e-01f, -8.314695954e-01f, -6.343932748e-01f, -7.730104327e-01f, -7.071067691e-01f, -7.071067691e-01f, -7.730104327e-01f, -6.343932748e-01f, -8.314695954e-01f, -5.555702448e-01f, -8.819212914e-01f, -4.713967443e-01f, -9.238795042e-01f, -3.826834261e-01f, -9.569403529e-01f, -2.902846634e-01f, -9.807852507e-01f, -1.950903237e-01f, -9.951847196e-01f, -9.801714122e-02f};

// forward 32-point DFT of x[n], n = 0..31, given as even part e[m] = x[2m] and odd part o[m] = x[2m+1].
// On return X[k] is in e[slot16(k)] for k < 16 and in o[slot16(k - 16)] for k >= 16.
__device__ __forceinline__ void dft32(float (&er)[16], float (&ei)[16], float (&orr)[16], float (&oi)[16])
{
    dft16<float, 16>(er, ei);
    dft16<float, 16>(orr, oi);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int s = slot16(k);
        float tr = orr[s], ti = oi[s];
        if (k == 8) { const float t = tr; tr = ti; ti = -t; }
        else if (k > 0) { cmul(tr, ti, W32C[2 * k], W32C[2 * k + 1]); }
        orr[s] = er[s] - tr; oi[s] = ei[s] - ti;
        er[s] += tr; ei[s] += ti;
    }
}

template <bool FULL, bool MN0, bool EPI>
__global__ __launch_bounds__(64 * R4096_WAVES, 2) void stft_r4096_kernel(float *__restrict__ Y, const float *__restrict__ X,
                                                                         const float *__restrict__ win, const float *__restrict__ tw,
                                                                         const R4096Args a)
{
    constexpr int M = 2048, F = 2049;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    float *wtab = reinterpret_cast<float *>(smem_raw);                               // halved window, 4096 floats
    float2 *ex = reinterpret_cast<float2 *>(wtab + 4096) + wv * R4096_WAVE_C;         // this wave's tile

    for (int l = tid; l < 4096; l += 64 * R4096_WAVES) { wtab[l] = (l < L) ? 0.5f * win[l] : 0.0f; }

    // per-lane twiddles: W2048^(lane b), b = 1..3, and W2048^(4 lane a), a = 1..7; W4096^lane for the split
    auto tw2048 = [&](int m, float &c, float &s) {       // W2048^m = W4096^(2m), table holds k < 2048
        int idx = (2 * m) & 4095;
        float sg = 1.0f;
        if (idx >= 2048) { idx -= 2048; sg = -1.0f; }
        c = sg * tw[2 * idx]; s = sg * tw[2 * idx + 1];
    };
    float tbr[4], tbi[4], tar[8], tai[8];
#pragma unroll
    for (int b = 1; b < 4; ++b) { tw2048(lane * b, tbr[b], tbi[b]); }
#pragma unroll
    for (int q = 1; q < 8; ++q) { tw2048(4 * lane * q, tar[q], tai[q]); }
    const float owr = tw[2 * lane], owi = tw[2 * lane + 1];            // W4096^lane
    const int k1p = lane >> 1, hp = lane & 1;                          // pass-2 role of the lane
    const float sgn = hp ? -1.0f : 1.0f;
    __syncthreads();

    // frames: 8 contiguous ranges by blockIdx % 8, round-robin over the waves inside a range
    const int xcd = blockIdx.x & 7, nwg_x = gridDim.x >> 3;
    const int f_lo = (int)((long long)a.frames * xcd / 8), f_hi = (int)((long long)a.frames * (xcd + 1) / 8);
    const int fstride = nwg_x * R4096_WAVES;

    for (int fid = f_lo + (blockIdx.x >> 3) * R4096_WAVES + wv; fid < f_hi; fid += fstride) {
        const int b = fid / a.W;
        const int w = fid - b * a.W;
        const float *xb = X + (long long)b * a.ldx;
        long long ss, rdoff;
        if (a.tbl) { ss = a.tbl[2 * w]; rdoff = ss - (long long)a.tbl[2 * w + 1]; }
        else { ss = a.ss0 + (long long)w * a.stp; rdoff = a.rd_shift; }

        // ---------------- pass 1: load, window, 32-point FFT over n1 ----------------
        float er[16], ei[16], orr[16], oi[16];          // even / odd n1
        {
            const float *src = xb + (ss - (a.edge ? rdoff : 0)) + 2 * lane;
            const bool inside = ss >= 0 && ss + 4096 <= a.N && (a.edge == 0 || ss >= rdoff) &&
                                ((reinterpret_cast<uintptr_t>(src) & 7) == 0);
            if (inside) {
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    float2 v = make_float2(0.f, 0.f);
                    if (FULL || 128 * n1 < L) { v = *reinterpret_cast<const float2 *>(src + 128 * n1); }
                    if (n1 & 1) { orr[n1 >> 1] = v.x; oi[n1 >> 1] = v.y; } else { er[n1 >> 1] = v.x; ei[n1 >> 1] = v.y; }
                }
            } else {
                FrameDesc d{};
                d.N = a.N; d.edge = a.edge;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    const int l = 128 * n1 + 2 * lane;
                    const float p = (l < L) ? fetch_sample(xb, d, ss, rdoff, l) : 0.f;
                    const float c = (l + 1 < L) ? fetch_sample(xb, d, ss, rdoff, l + 1) : 0.f;
                    if (n1 & 1) { orr[n1 >> 1] = p; oi[n1 >> 1] = c; } else { er[n1 >> 1] = p; ei[n1 >> 1] = c; }
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const float2 h = *reinterpret_cast<const float2 *>(wtab + 128 * n1 + 2 * lane);
                float &p = (n1 & 1) ? orr[n1 >> 1] : er[n1 >> 1];
                float &c = (n1 & 1) ? oi[n1 >> 1] : ei[n1 >> 1];
                if (!FULL) {                                 // keep data past the window out (0 * NaN)
                    const int l = 128 * n1 + 2 * lane;
                    p = (l < L) ? p : 0.f;
                    c = (l + 1 < L) ? c : 0.f;
                }
                p *= h.x; c *= h.y;
                if (MN0) { sum += p + c; }
            }
            if (MN0) {
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { sum += __shfl_xor(sum, o); }
                const float mean = sum / (float)L;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    const int l = 128 * n1 + 2 * lane;
                    float &p = (n1 & 1) ? orr[n1 >> 1] : er[n1 >> 1];
                    float &c = (n1 & 1) ? oi[n1 >> 1] : ei[n1 >> 1];
                    p -= (FULL || l < L) ? mean : 0.f;
                    c -= (FULL || l + 1 < L) ? mean : 0.f;
                }
            }
        }
        dft32(er, ei, orr, oi);

        // twiddle W2048^(lane k1), k1 = 4 q + b, and write row k1 of the tile
        wave_sync();                                       // the previous frame's reads of the tile are done
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) {
            float r = (k1 < 16) ? er[slot16(k1)] : orr[slot16(k1 - 16)];
            float i = (k1 < 16) ? ei[slot16(k1)] : oi[slot16(k1 - 16)];
            const int q = k1 >> 2, bb = k1 & 3;
            if (bb) { cmul(r, i, tbr[bb], tbi[bb]); }
            if (q) { cmul(r, i, tar[q], tai[q]); }
            ex[k1 * R4096_PITCH + lane] = make_float2(r, i);
        }
        wave_sync();

        // ---------------- pass 2: 32-point FFT over the n2 of parity hp, then the 64-point combine ----------------
        {
            const float2 *row = ex + k1p * R4096_PITCH + hp;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const float2 v = row[2 * m];
                if (m & 1) { orr[m >> 1] = v.x; oi[m >> 1] = v.y; } else { er[m >> 1] = v.x; ei[m >> 1] = v.y; }
            }
        }
        dft32(er, ei, orr, oi);
        wave_sync();                                       // every lane has its row: the tile becomes Z
        {
            float2 *zrow = ex + k1p + hp * (1024 + 8);     // Z[k] sits at k + 8 (k >= 1024): conflict-free halves
#pragma unroll
            for (int k2 = 0; k2 < 32; ++k2) {
                float gr = (k2 < 16) ? er[slot16(k2)] : orr[slot16(k2 - 16)];
                float gi = (k2 < 16) ? ei[slot16(k2)] : oi[slot16(k2 - 16)];
                // odd-parity lanes carry G1 and multiply it by W64^k2; then X[k2] = G0 + W G1 (even lane),
                // X[k2 + 32] = G0 - W G1 (odd lane) after one swap with the neighbouring lane
                float tr = gr, ti = gi;
                if (k2 == 16) { const float t = tr; tr = ti; ti = -t; }
                else if (k2 > 0) { cmul(tr, ti, W64C[2 * k2], W64C[2 * k2 + 1]); }
                gr = hp ? tr : gr; gi = hp ? ti : gi;
                const float pr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(gr), 0xB1, 0xf, 0xf, false));
                const float pi = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(gi), 0xB1, 0xf, 0xf, false));
                zrow[32 * k2] = make_float2(fmaf(gr, sgn, pr), fmaf(gi, sgn, pi));
            }
        }
        wave_sync();

        // ---------------- split + power + store: lane handles bins k = lane + 64 m and 2048 - k ----------------
        float *y = Y + ((long long)b * a.W + w) * F;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int k = lane + 64 * m;
            const float2 z = ex[k];
            const int km = (m == 0 && lane == 0) ? 0 : (M + 8 - k);
            const float2 zm = ex[km];
            const float ar = z.x + zm.x, ai = z.y - zm.y, br = z.x - zm.x, bi = z.y + zm.y;
            // -i W4096^k B with W4096^k = W4096^lane W64^m
            float ur = br, ui = bi;
            if (m > 0) { cmul(ur, ui, W64C[2 * m], W64C[2 * m + 1]); }
            const float tr = ur * owi + ui * owr, ti = ui * owi - ur * owr;
            const float pr = ar + tr, pi = ai + ti, qr = ar - tr, qi = ai - ti;
            float p = pr * pr + pi * pi, q = qr * qr + qi * qi;
            if (EPI) { p = power_epilogue(p, a.amp, a.lg); q = power_epilogue(q, a.amp, a.lg); }
            y[k] = p;
            y[M - k] = q;
        }
        if (lane == 0) {
            const float2 z = ex[1024 + 8];                 // bin 1024 pairs with itself (Z holds half values)
            float p = 4.0f * (z.x * z.x + z.y * z.y);
            if (EPI) { p = power_epilogue(p, a.amp, a.lg); }
            y[1024] = p;
        }
    }
}


bool r4096_supported(const FrameDesc &d)
{
    return d.nfft == 4096 && d.L >= 1 && d.L <= 4096 && (long long)d.B * d.W < (1ll << 31) - 4096;
}

template <bool FULL, bool MN0, bool EPI>
static hipError_t launch_r4096_inst(float *Y, const float *X, const float *win, const float *tw, const R4096Args &a, hipStream_t s)
{
    const size_t lds = 4096 * sizeof(float) + (size_t)R4096_WAVES * R4096_WAVE_C * sizeof(float2);
    static bool granted = false;
    (void)hipGetLastError();
    if (!granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&stft_r4096_kernel<FULL, MN0, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { return e; }
        granted = true;
    }
    long long grid = ((long long)a.frames + R4096_WAVES - 1) / R4096_WAVES;
    if (grid > 256) { grid = 256; }                      // one 8-wave workgroup per CU (151 KB of LDS each)
    grid = (grid + 7) & ~7ll;
    hipLaunchKernelGGL((stft_r4096_kernel<FULL, MN0, EPI>), dim3((unsigned)grid), dim3(64 * R4096_WAVES), lds, s, Y, X, win, tw, a);
    return hipGetLastError();
}

hipError_t launch_stft_r4096(float *Y, const float *X, const float *win, const float *tw, const FrameDesc &d, hipStream_t s)
{
    R4096Args a{};
    a.ldx = d.ldx; a.N = d.N; a.ss0 = d.ss0; a.rd_shift = d.rd_shift; a.tbl = d.tbl;
    a.L = d.L; a.W = d.W; a.stp = d.stp; a.edge = d.edge; a.amp = d.amp; a.lg = d.lg;
    a.frames = d.B * d.W;
    const bool full = d.L == 4096, epi = d.amp || d.lg;
#define DSP_R4096(F_, M_, E_) launch_r4096_inst<F_, M_, E_>(Y, X, win, tw, a, s)
    if (full) {
        if (d.mn0) { return epi ? DSP_R4096(true, true, true) : DSP_R4096(true, true, false); }
        return epi ? DSP_R4096(true, false, true) : DSP_R4096(true, false, false);
    }
    if (d.mn0) { return epi ? DSP_R4096(false, true, true) : DSP_R4096(false, true, false); }
    return epi ? DSP_R4096(false, false, true) : DSP_R4096(false, false, false);
#undef DSP_R4096
}

}  // namespace dsp
