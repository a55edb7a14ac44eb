/*
 * CPU oracle for the STFT hot path  --  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's algorithm (c/stft.c, c/stft_flt.c, c/frame_univar.c,
 * c/frame_univar_flt.c, c/apply_win.c, c/fft.fftw.r2hc.c, c/fft_power.c).  Only tests, the smoke
 * check and the benchmark's cpu_baseline leg may load it; the product library never does.
 *
 * Pinning status: the reference ships no tests, golden vectors or fixtures for this path, and its
 * fused entry points (c/stft.c, c/stft_flt.c, c/fft.fftw.r2hc.c) need <fftw3.h>/libfftw3 (FFTW3,
 * version unpinned by the reference, absent from its tree and from this image), so they cannot be
 * built here.  What is pinned: framing, windowing, the transform's sign / scaling / output order and
 * the power step are checked against the reference's own FFTW-free sources compiled as they lie
 * (oracle/Makefile -> oracle/_ref/libdspref.so: frame_univar.c, frame_univar_flt.c, window_univar.c,
 * window_univar_flt.c, apply_win.c, fft.rad2.c, fft_power.c) and frozen as fixtures in tests/golden/.
 * What is NOT pinned by executed reference code ("parity unpinned"): the mn0 / amp / lg branches of
 * c/stft.c and the pointer walk of c/stft_flt.c, which are restated from the source text alone.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <math.h>

#define PI_L 3.141592653589793238462643383279502884L
static int orc_fftw_lock = 0;

#define T float
#define SFX s
#define CFX c
#define ROUND roundf
#define SQRT sqrtf
#define LOG logf
#define FABS fabsf
#define CBRT cbrtf
#define POW powf
#define EPS FLT_EPSILON
#include "stft_oracle_impl.h"
#undef T
#undef SFX
#undef CFX
#undef ROUND
#undef SQRT
#undef LOG
#undef FABS
#undef CBRT
#undef POW
#undef EPS

#define T double
#define SFX d
#define CFX z
#define ROUND round
#define SQRT sqrt
#define LOG log
#define FABS fabs
#define CBRT cbrt
#define POW pow
#define EPS DBL_EPSILON
#include "stft_oracle_impl.h"
#undef T
#undef SFX
#undef CFX
#undef ROUND
#undef SQRT
#undef LOG
#undef FABS
#undef CBRT
#undef POW
#undef EPS

/* number of frames; reference c/stft.c:40 */
size_t orc_stft_num_frames(const size_t N, const size_t L, const size_t stp, const int snip_edges)
{
    if (stp<1u || (snip_edges && L>N)) { return 0u; }
    return (snip_edges) ? 1u+(N-L)/stp : (N+stp/2u)/stp;
}
