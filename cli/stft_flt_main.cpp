// stft_flt: STFT of a univariate signal file X1 with a window file X2 where the frame centres are
// c0 + w*stp for floating-point c0 and stp and frames are zero-padded outside the signal.  Same
// options, defaults, checks and output sizing as the reference tool (reference
// srci/stft_flt.cpp:69-118, src/stft_flt.cpp); the transform is the stft_flt_s / stft_flt_d entry
// point of libdsp_stft_hip.so.
#include "cli_common.hpp"
#include "../include/dsp_stft.h"

#include <cfloat>
#include <new>

using namespace dspcli;

namespace {

template <typename T>
int process(const Diag &dg, std::istream &s1, std::istream &s2, const TensorInfo &i1, const TensorInfo &i2, const TensorInfo &o1, std::ostream *os,
            size_t L, size_t W, size_t nfft, double c0, double stp, int mn0, int amp, int lg,
            int (*fn)(T *, const T *, const T *, size_t, size_t, size_t, size_t, T, T, int, int, int))
{
    T *X1 = new (std::nothrow) T[i1.N()];
    if (!X1) { return dg.fail("problem allocating for input file 1 (X1)"); }
    T *X2 = new (std::nothrow) T[i2.N()];
    if (!X2) { return dg.fail("problem allocating for input file 2 (X2)"); }
    T *Y = new (std::nothrow) T[o1.N() ? o1.N() : 1];
    if (!Y) { return dg.fail("problem allocating for output file (Y)"); }
    if (!read_payload(s1, X1, i1.nbytes())) { return dg.fail("problem reading input file (X)"); }
    if (!read_payload(s2, X2, i2.nbytes())) { return dg.fail("problem reading input file 2 (X2)"); }
    if (fn(Y, X1, X2, i1.N(), L, W, nfft, (T)c0, (T)stp, mn0, amp, lg)) { return dg.fail("problem during function call"); }
    if (os && !write_payload(*os, Y, o1.nbytes())) { return dg.fail("problem writing output file (Y)"); }
    delete[] X1; delete[] X2; delete[] Y;
    return 0;
}

}  // namespace

int main(int argc, char *argv[])
{
    ArgParser ap;
    ap.prog = "stft_flt";
    ap.specs = {
        {'c', "c0", "<dbl>", ArgKind::Dbl, "center of first frame in samps [default=0.0]"},
        {'s', "stp", "<dbl>", ArgKind::Dbl, "step size btwn frames [default=160.0]"},
        {'w', "nframes", "<uint>", ArgKind::Int, "number of frames [default=(N-1)/stp]"},
        {'z', "zero-mean", nullptr, ArgKind::Flag, "include to zero the mean of each frame [default=false]"},
        {'a', "amplitude", nullptr, ArgKind::Flag, "include to output amplitude (sqrt of power) [default=false]"},
        {'l', "log", nullptr, ArgKind::Flag, "include to output log of amplitude or power [default=false]"},
        {'i', "intended", nullptr, ArgKind::Flag, "frame exactly at the documented positions instead of where the reference code reads [default=false]"},
        {'o', "ofile", "<file>", ArgKind::File, "output file (Y)"},
        {'h', "help", nullptr, ArgKind::Flag, "display this help and exit"},
    };
    const Diag dg{ap.prog};
    const bool parsed = ap.parse(argc, argv);
    if (ap["help"].count > 0) {
        ap.print_usage(std::cout, "input files (X1,X2)");
        std::cout << "\nSTFT power spectrogram of the real vector X1 with the window vector X2 (length L), computed on the GPU,\n"
                     "with W frames centred at round(c0 + w*stp), c0 and stp floating-point; samples outside X1 count as zero.\n"
                     "nfft is the next power of two >= L, F = nfft/2+1; the default W is (N-1-c0)/stp.\n"
                     "The output is W x F for a row-major input, F x W for a column-major one.\n"
                     "-z subtracts each windowed frame's mean, -a gives sqrt(power), -l the natural log of the result.\n"
                     "Files are NumPy .npy or Armadillo binary; '-' or a missing name reads stdin (one input only).\n"
                     "Examples:\n  stft_flt -s65 X1 X2 -o Y\n  cat X1 | stft_flt -c200 - X2 > Y\n";
        return 1;
    }
    if (!parsed) {
        for (const auto &e : ap.errors) { dg.fail(e); }
        return 1;
    }

    Inputs in;
    if (int rc = open_inputs(dg, ap.positionals, in)) { return rc; }
    Output out;
    TensorInfo i1, i2, o1;
    std::string err;
    if (!read_header(*in.s1, i1, err)) { return dg.fail("problem reading header for input file 1 (" + err + ")"); }
    if (!read_header(*in.s2, i2, err)) { return dg.fail("problem reading header for input file 2 (" + err + ")"); }

    const double c0 = ap["c0"].count > 0 ? ap["c0"].dval : 0.0;
    if (c0 > double(i1.N() - 1u)) { return dg.fail("c0 (center of first frame) must be <= N-1"); }
    const double stp = ap["stp"].count > 0 ? ap["stp"].dval : 160.0;
    if (stp < double(FLT_EPSILON)) { return dg.fail("stp (step size) must be positive"); }
    size_t W;
    if (ap["nframes"].count == 0) { W = size_t((double(i1.N() - 1u) - c0) / stp); }
    else if (ap["nframes"].ival < 1) { return dg.fail("W (nframes) must be positive"); }
    else { W = size_t(ap["nframes"].ival); }
    const int mn0 = ap["zero-mean"].count > 0, amp = ap["amplitude"].count > 0, lg = ap["log"].count > 0;

    if (i1.iscomplex() || i2.iscomplex()) { return dg.fail("inputs must be real-valued"); }
    if (i1.T != i2.T) { return dg.fail("inputs must have the same data type"); }
    if (!i1.isvec()) { return dg.fail("input 1 (X1) must be a vector"); }
    if (i1.isempty()) { return dg.fail("input 1 (X1) found to be empty"); }
    if (!i2.isvec()) { return dg.fail("input 2 (X2) must be a vector"); }
    if (i2.isempty()) { return dg.fail("input 2 (X2) found to be empty"); }

    const size_t L = i2.N();
    size_t nfft = 1;
    while (nfft < L) { nfft *= 2; }
    const size_t F = nfft / 2 + 1;
    o1.F = i1.F; o1.T = i1.T; o1.colmajor = i1.colmajor; o1.ndim = 2;
    o1.R = i1.isrowmajor() ? W : F;
    o1.C = i1.isrowmajor() ? F : W;
    o1.S = i1.S; o1.H = i1.H;

    if (int rc = open_output(dg, ap["ofile"], out)) { return rc; }
    if (out.s && !write_header(*out.s, o1, err)) { return dg.fail("problem writing header for output file 1 (" + err + ")"); }
    dsp_stft_flt_set_mode(ap["intended"].count > 0);

    if (o1.T == 1) { return process<float>(dg, *in.s1, *in.s2, i1, i2, o1, out.s, L, W, nfft, c0, stp, mn0, amp, lg, stft_flt_s); }
    if (o1.T == 2) { return process<double>(dg, *in.s1, *in.s2, i1, i2, o1, out.s, L, W, nfft, c0, stp, mn0, amp, lg, stft_flt_d); }
    return dg.fail("data type not supported");
}
