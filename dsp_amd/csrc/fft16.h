// Register-resident 16-point FFT building blocks shared by the 512-point kernels.
#pragma once
#include "stft_common.h"

namespace dsp {

// forward 4-point DFT in place, natural order.  Z3: input 3 is known to be zero.
template <typename T, bool Z3 = false>
__device__ __forceinline__ void dft4(T &r0, T &i0, T &r1, T &i1, T &r2, T &i2, T &r3, T &i3)
{
    const T ar = r0 + r2, ai = i0 + i2, br = r0 - r2, bi = i0 - i2;
    const T cr = Z3 ? r1 : r1 + r3, ci = Z3 ? i1 : i1 + i3, dr = Z3 ? r1 : r1 - r3, di = Z3 ? i1 : i1 - i3;
    r0 = ar + cr; i0 = ai + ci;
    r2 = ar - cr; i2 = ai - ci;
    r1 = br + di; i1 = bi - dr;      // b - i d
    r3 = br - di; i3 = bi + dr;      // b + i d
}

// (r + i im) *= (c + i s)
template <typename T>
__device__ __forceinline__ void cmul(T &r, T &i, const T c, const T s)
{
    const T t = r * c - i * s;
    i = r * s + i * c;
    r = t;
}

// Forward 16-point DFT of (re[n], im[n]), n = 4 n1 + n2, as 4 x 4.  On return X[k] sits in slot
// 4 (k & 3) + (k >> 2).  Inputs n >= NZ are known to be zero (NZ in 13..16).
template <typename T, int NZ = 16>
__device__ __forceinline__ void dft16(T (&re)[16], T (&im)[16])
{
    static_assert(NZ >= 13 && NZ <= 16, "only the last row of each column may be known-zero");
    constexpr T C1 = T(0.92387953251128675612818318939678829L);   // cos(pi/8)
    constexpr T S1 = T(0.38268343236508977172845998403039887L);   // sin(pi/8)
    constexpr T R2 = T(0.70710678118654752440084436210484904L);   // sqrt(1/2)
    dft4<T, (NZ <= 12)>(re[0], im[0], re[4], im[4], re[8], im[8], re[12], im[12]);
    dft4<T, (NZ <= 13)>(re[1], im[1], re[5], im[5], re[9], im[9], re[13], im[13]);
    dft4<T, (NZ <= 14)>(re[2], im[2], re[6], im[6], re[10], im[10], re[14], im[14]);
    dft4<T, (NZ <= 15)>(re[3], im[3], re[7], im[7], re[11], im[11], re[15], im[15]);
    // slot 4 k1 + n2 now holds a[n2][k1]; multiply by W16^(n2 k1) = (cos, -sin)(pi n2 k1 / 8)
    cmul(re[5], im[5], C1, -S1);     // 1
    cmul(re[6], im[6], R2, -R2);     // 2
    cmul(re[7], im[7], S1, -C1);     // 3
    cmul(re[9], im[9], R2, -R2);     // 2
    { const T t = re[10]; re[10] = im[10]; im[10] = -t; }   // 4: times -i
    cmul(re[11], im[11], -R2, -R2);  // 6
    cmul(re[13], im[13], S1, -C1);   // 3
    cmul(re[14], im[14], -R2, -R2);  // 6
    cmul(re[15], im[15], -C1, S1);   // 9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        dft4(re[4 * k1], im[4 * k1], re[4 * k1 + 1], im[4 * k1 + 1], re[4 * k1 + 2], im[4 * k1 + 2],
             re[4 * k1 + 3], im[4 * k1 + 3]);
    }
}
__host__ __device__ constexpr int slot16(int k) { return 4 * (k & 3) + (k >> 2); }

// ---- the same transforms on (re, im) pairs ------------------------------------------------------
// A complex value is one 2-element vector, so that float arithmetic maps to the packed two-lane
// instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: one issue slot for both components; the
// swaps and sign flips below become their op_sel / neg modifiers).  A plain float instruction costs
// the same issue slot for one component, so this halves the instruction count of the butterflies.
// For double the same source compiles to scalar f64 instructions.
typedef float cxf __attribute__((ext_vector_type(2)));
struct cxd { double x, y; };
__device__ __forceinline__ cxd operator+(const cxd a, const cxd b) { return cxd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cxd operator-(const cxd a, const cxd b) { return cxd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cxd operator*(const cxd a, const cxd b) { return cxd{a.x * b.x, a.y * b.y}; }   // element-wise
__device__ __forceinline__ cxd &operator+=(cxd &a, const cxd b) { a.x += b.x; a.y += b.y; return a; }
__device__ __forceinline__ cxd &operator-=(cxd &a, const cxd b) { a.x -= b.x; a.y -= b.y; return a; }
template <typename T> struct cx_of;
template <> struct cx_of<float> { using type = cxf; };
template <> struct cx_of<double> { using type = cxd; };
template <typename T> using cx = typename cx_of<T>::type;

__device__ __forceinline__ cxf cx_swap(const cxf a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ cxd cx_swap(const cxd a) { return cxd{a.y, a.x}; }
__device__ __forceinline__ cxf cx_fma(const cxf a, const cxf b, const cxf c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ cxd cx_fma(const cxd a, const cxd b, const cxd c) { return cxd{fma(a.x, b.x, c.x), fma(a.y, b.y, c.y)}; }
// a - i b and a + i b
template <typename T> __device__ __forceinline__ cx<T> cx_sub_i(const cx<T> a, const cx<T> b) { return cx_fma(cx_swap(b), cx<T>{T(1), T(-1)}, a); }
template <typename T> __device__ __forceinline__ cx<T> cx_add_i(const cx<T> a, const cx<T> b) { return cx_fma(cx_swap(b), cx<T>{T(-1), T(1)}, a); }
// a + conj(b) and a - conj(b)
template <typename T> __device__ __forceinline__ cx<T> cx_add_conj(const cx<T> a, const cx<T> b) { return cx_fma(b, cx<T>{T(1), T(-1)}, a); }
template <typename T> __device__ __forceinline__ cx<T> cx_sub_conj(const cx<T> a, const cx<T> b) { return cx_fma(b, cx<T>{T(-1), T(1)}, a); }
// a * (c + i s)
template <typename T> __device__ __forceinline__ cx<T> cx_mul(const cx<T> a, const T c, const T s)
{
    const cx<T> t = a * cx<T>{c, c};
    return cx_fma(cx_swap(a), cx<T>{-s, s}, t);
}
template <typename T> __device__ __forceinline__ cx<T> cx_mul(const cx<T> a, const cx<T> w) { return cx_mul<T>(a, w.x, w.y); }
// the two halves of cx_mul, for callers that interleave several independent products: on gfx950 a packed
// fma that consumes the packed product issued just before it costs an extra s_nop
template <typename T> __device__ __forceinline__ cx<T> cx_mul_first(const cx<T> a, const cx<T> w) { return a * cx<T>{w.x, w.x}; }
template <typename T> __device__ __forceinline__ cx<T> cx_mul_second(const cx<T> a, const cx<T> w, const cx<T> t) { return cx_fma(cx_swap(a), cx<T>{-w.y, w.y}, t); }
// |a|^2: the packed square and the add of its halves.  (DSP_NORM_FMA 1: a plain multiply and a plain fma instead, which the
// 4096-point kernel uses (norm2 there, measured faster); in the 512-, 1024- and 2048-point kernels it measured 2.5 / 4 / 2.5 %
// slower.)
#ifndef DSP_NORM_FMA
#define DSP_NORM_FMA 0
#endif
template <typename T> __device__ __forceinline__ T cx_norm(const cx<T> a)
{
#if DSP_NORM_FMA
    return __builtin_fma(a.y, a.y, a.x * a.x);
#else
    const cx<T> q = a * a; return q.x + q.y;
#endif
}

template <typename T, bool Z3 = false>
__device__ __forceinline__ void dft4c(cx<T> &a0, cx<T> &a1, cx<T> &a2, cx<T> &a3)
{
    const cx<T> s0 = a0 + a2, s1 = a0 - a2;
    const cx<T> s2 = Z3 ? a1 : a1 + a3, t = Z3 ? a1 : a1 - a3;
    a0 = s0 + s2;
    a2 = s0 - s2;
    a1 = cx_sub_i<T>(s1, t);
    a3 = cx_add_i<T>(s1, t);
}

// ... with only the first NIN inputs other than zero (NIN = 1 ... 4)
template <typename T, int NIN>
__device__ __forceinline__ void dft4c_in(cx<T> &a0, cx<T> &a1, cx<T> &a2, cx<T> &a3)
{
    static_assert(NIN >= 1 && NIN <= 4, "one to four inputs");
    if constexpr (NIN >= 3) { dft4c<T, NIN == 3>(a0, a1, a2, a3); }
    else if constexpr (NIN == 2) {
        const cx<T> e = a0, o = a1;
        a0 = e + o;
        a2 = e - o;
        a1 = cx_sub_i<T>(e, o);
        a3 = cx_add_i<T>(e, o);
    } else {
        a1 = a0; a2 = a0; a3 = a0;
    }
}
// the inputs of column k of a 16-point transform's first stage (rows k, k + 4, k + 8, k + 12) that lie below NZ
constexpr int dft16_nin(int NZ, int k) { return NZ >= 12 + k + 1 ? 4 : NZ >= 8 + k + 1 ? 3 : NZ >= 4 + k + 1 ? 2 : 1; }

// dft16 on pairs: same index map, same output slots, same known-zero rows
template <typename T, int NZ = 16>
__device__ __forceinline__ void dft16c(cx<T> (&z)[16])
{
    static_assert(NZ >= 13 && NZ <= 16, "only the last row of each column may be known-zero");
    constexpr T C1 = T(0.92387953251128675612818318939678829L);   // cos(pi/8)
    constexpr T S1 = T(0.38268343236508977172845998403039887L);   // sin(pi/8)
    constexpr T R2 = T(0.70710678118654752440084436210484904L);   // sqrt(1/2)
    dft4c<T, (NZ <= 12)>(z[0], z[4], z[8], z[12]);
    dft4c<T, (NZ <= 13)>(z[1], z[5], z[9], z[13]);
    dft4c<T, (NZ <= 14)>(z[2], z[6], z[10], z[14]);
    dft4c<T, (NZ <= 15)>(z[3], z[7], z[11], z[15]);
    z[5] = cx_mul<T>(z[5], C1, -S1);       // W16^1
    z[6] = cx_mul<T>(z[6], R2, -R2);       // W16^2
    z[7] = cx_mul<T>(z[7], S1, -C1);       // W16^3
    z[9] = cx_mul<T>(z[9], R2, -R2);       // W16^2
    z[10] = cx_swap(z[10]) * cx<T>{T(1), T(-1)};   // W16^4 = -i
    z[11] = cx_mul<T>(z[11], -R2, -R2);    // W16^6
    z[13] = cx_mul<T>(z[13], S1, -C1);     // W16^3
    z[14] = cx_mul<T>(z[14], -R2, -R2);    // W16^6
    z[15] = cx_mul<T>(z[15], -C1, S1);     // W16^9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) { dft4c<T>(z[4 * k1], z[4 * k1 + 1], z[4 * k1 + 2], z[4 * k1 + 3]); }
}

// ---- twiddled butterflies with the multiplications folded into the additions ------------------------
// A product w z, w = wr + i wi a constant of modulus one, is written as sig R(z + t rot(z)) with rot(z) = i z,
// |t| <= 1 and R the identity (|wr| >= |wi|: t = wi / wr, sig = wr) or rot (t = -wr / wi, sig = wi).  The
// inner sum is one fma; the factor sig R is carried into the additions that follow as the constant of their
// fma: e +- w o costs 3 instructions instead of 4, a radix-4 butterfly with three twiddled inputs 11
// instead of 14, and no multiply feeds the instruction right behind it.
struct TwF { double t, sig; int m; };                      // w z = sig i^m (z + t i z)
constexpr TwF twf(double wr, double wi)
{
    const double ar = wr < 0 ? -wr : wr, ai = wi < 0 ? -wi : wi;
    return (ar >= ai) ? TwF{wi / wr, wr, 0} : TwF{-wr / wi, wi, 1};
}
// z + t i z
template <typename T> __device__ __forceinline__ cx<T> cx_pre(const cx<T> z, const T t) { return cx_fma(cx_swap(z), cx<T>{-t, t}, z); }
// acc + k i^M z
template <typename T, int M> __device__ __forceinline__ cx<T> cx_axpy_i(const cx<T> acc, const T k, const cx<T> z)
{
    if constexpr ((M & 3) == 0) { return cx_fma(z, cx<T>{k, k}, acc); }
    else if constexpr ((M & 3) == 1) { return cx_fma(cx_swap(z), cx<T>{-k, k}, acc); }
    else if constexpr ((M & 3) == 2) { return cx_fma(z, cx<T>{-k, -k}, acc); }
    else { return cx_fma(cx_swap(z), cx<T>{k, -k}, acc); }
}
// (W_N^K)(cos, -sin) as constants
constexpr double cx_pi = 3.14159265358979323846264338327950288;
constexpr double cos_tab(int k, int n)
{
    // cos(2 pi k / n) for n | 64 by exact symmetry reduction to the first octant values below
    constexpr double c64[17] = {1.0, 0.99518472667219688624483695310947992, 0.98078528040323044912618223613423903, 0.95694033573220886493579788698026996,
                                0.92387953251128675612818318939678829, 0.88192126434835502971275686366038619, 0.83146961230254523707878837761790575,
                                0.77301045336273696081090660975846754, 0.70710678118654752440084436210484904, 0.63439328416364549821517161322549085,
                                0.55557023301960222474283081394853287, 0.47139673682599764855638762590525134, 0.38268343236508977172845998403039887,
                                0.29028467725446236763619237581739827, 0.19509032201612826784828486847702224, 0.09801714032956060199419556388864185, 0.0};
    int j = ((k * (64 / n)) % 64 + 64) % 64;               // angle in units of 2 pi / 64
    if (j > 32) { j = 64 - j; }                            // cos is even about pi
    return (j <= 16) ? c64[j] : -c64[32 - j];
}
constexpr double sin_tab(int k, int n) { return cos_tab(k * (64 / n) - 16, 64); }
// the twiddle W_N^K = exp(-2 pi i K / N) in folded form
constexpr TwF twf_n(int k, int n) { return twf(cos_tab(k, n), -sin_tab(k, n)); }

// The same for any N divisible by 8, the constants from a Taylor series evaluated at compile time (argument at most
// pi/4, 11 terms: below 1e-17 absolute)
constexpr double ct_sin_small(double x)
{
    double term = x, sum = x;
    for (int i = 1; i <= 11; ++i) { term *= -x * x / ((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
constexpr double ct_cos_small(double x)
{
    double term = 1.0, sum = 1.0;
    for (int i = 1; i <= 11; ++i) { term *= -x * x / ((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
struct CosSin { double c, s; };
constexpr CosSin cos_sin_ct(int k, int n)                   // (cos, sin)(2 pi k / n)
{
    const int j = ((k % n) + n) % n, q = n / 4;
    const int quad = j / q, rem = j % q;
    const bool low = 2 * rem <= q;
    const double th = 2.0 * cx_pi * (low ? rem : q - rem) / n;
    const double c = low ? ct_cos_small(th) : ct_sin_small(th), s = low ? ct_sin_small(th) : ct_cos_small(th);
    return quad == 0 ? CosSin{c, s} : quad == 1 ? CosSin{-s, c} : quad == 2 ? CosSin{-c, -s} : CosSin{s, -c};
}
constexpr TwF twf_ct(int k, int n) { const CosSin w = cos_sin_ct(k, n); return twf(w.c, -w.s); }

// radix-4 butterfly on a0 and the products w1 a1, w2 a2, w3 a3 (w_j = W_N^(K_j)), outputs in natural order in a0..a3
template <typename T, int N, int K1, int K2, int K3>
__device__ __forceinline__ void dft4c_tw(cx<T> &a0, cx<T> &a1, cx<T> &a2, cx<T> &a3)
{
    constexpr TwF f1 = twf_n(K1, N), f2 = twf_n(K2, N), f3 = twf_n(K3, N);
    const cx<T> p2 = (f2.t == 0.0) ? a2 : cx_pre<T>(a2, T(f2.t));
    const cx<T> p1 = (f1.t == 0.0) ? a1 : cx_pre<T>(a1, T(f1.t));
    const cx<T> p3 = (f3.t == 0.0) ? a3 : cx_pre<T>(a3, T(f3.t));
    const cx<T> s0 = cx_axpy_i<T, f2.m>(a0, T(f2.sig), p2);
    const cx<T> s1 = cx_axpy_i<T, f2.m + 2>(a0, T(f2.sig), p2);
    // w1 a1 +- w3 a3 = sig1 i^m1 (p1 +- (sig3 / sig1) i^(m3 - m1) p3)
    constexpr double r31 = f3.sig / f1.sig;
    const cx<T> s2 = cx_axpy_i<T, f3.m - f1.m + 4>(p1, T(r31), p3);
    const cx<T> tt = cx_axpy_i<T, f3.m - f1.m + 6>(p1, T(r31), p3);
    a0 = cx_axpy_i<T, f1.m>(s0, T(f1.sig), s2);
    a2 = cx_axpy_i<T, f1.m + 2>(s0, T(f1.sig), s2);
    a1 = cx_axpy_i<T, f1.m + 3>(s1, T(f1.sig), tt);         // s1 - i t
    a3 = cx_axpy_i<T, f1.m + 1>(s1, T(f1.sig), tt);         // s1 + i t
}

// dft16c with the twiddles folded: same index map, same output slots, same known-zero rows; 72 instructions for 81
template <typename T, int NZ = 16>
__device__ __forceinline__ void dft16f(cx<T> (&z)[16])
{
    static_assert(NZ >= 4 && NZ <= 16, "rows NZ and above are known to be zero; every column keeps its first row");
    dft4c_in<T, dft16_nin(NZ, 0)>(z[0], z[4], z[8], z[12]);
    dft4c_in<T, dft16_nin(NZ, 1)>(z[1], z[5], z[9], z[13]);
    dft4c_in<T, dft16_nin(NZ, 2)>(z[2], z[6], z[10], z[14]);
    dft4c_in<T, dft16_nin(NZ, 3)>(z[3], z[7], z[11], z[15]);
    // slot 4 k1 + n2 holds a[n2][k1]; row k1 takes the twiddles W16^(n2 k1)
    dft4c<T>(z[0], z[1], z[2], z[3]);
    dft4c_tw<T, 16, 1, 2, 3>(z[4], z[5], z[6], z[7]);
    dft4c_tw<T, 16, 2, 4, 6>(z[8], z[9], z[10], z[11]);
    dft4c_tw<T, 16, 3, 6, 9>(z[12], z[13], z[14], z[15]);
}

// e +- W_N^K o -> (e, o), folded: 3 instructions (2 for K = 0 and for W = -i)
template <typename T, int N, int K>
__device__ __forceinline__ void bfly_tw(cx<T> &e, cx<T> &o)
{
    constexpr TwF f = twf_n(K, N);
    const cx<T> p = (f.t == 0.0) ? o : cx_pre<T>(o, T(f.t));
    const cx<T> lo = cx_axpy_i<T, f.m>(e, T(f.sig), p);
    o = cx_axpy_i<T, f.m + 2>(e, T(f.sig), p);
    e = lo;
}

// the 16-point transform the fused kernels use: folded twiddles unless DSP_FOLDED_DFT16 is 0 (A/B builds)
#ifndef DSP_FOLDED_DFT16
#define DSP_FOLDED_DFT16 1
#endif
template <typename T, int NZ = 16>
__device__ __forceinline__ void dft16x(cx<T> (&z)[16])
{
#if DSP_FOLDED_DFT16
    dft16f<T, NZ>(z);
#else
    dft16c<T, NZ>(z);
#endif
}

constexpr int R512_EXF = 16 * 17;   // complex slots per frame in the exchange region (row pitch 17)

}  // namespace dsp
