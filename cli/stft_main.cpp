// stft: short-term Fourier transform of a univariate signal file X1 with a window file X2, written
// as a W x F (row-major input) or F x W (column-major input) tensor file.  Same options, defaults,
// checks, output sizing and exit codes as the reference tool (reference src/stft.cpp:23-258); the
// transform itself is the stft_s / stft_d entry point of libdsp_stft_hip.so (reference call site
// src/stft.cpp:219,241).
#include "cli_common.hpp"
#include "../include/dsp_stft.h"

#include <cstring>
#include <new>

using namespace dspcli;

namespace {

template <typename T>
int process(const Diag &dg, std::istream &s1, std::istream &s2, const TensorInfo &i1, const TensorInfo &i2, const TensorInfo &o1, std::ostream *os,
            size_t L, size_t nfft, size_t stp, int snip_edges, int mn0, int amp, int lg,
            int (*fn)(T *, const T *, const T *, size_t, size_t, size_t, size_t, int, int, int, int))
{
    T *X1 = new (std::nothrow) T[i1.N()];
    if (!X1) { return dg.fail("problem allocating for input file 1 (X1)"); }
    T *X2 = new (std::nothrow) T[i2.N()];
    if (!X2) { return dg.fail("problem allocating for input file 2 (X2)"); }
    T *Y = new (std::nothrow) T[o1.N() ? o1.N() : 1];
    if (!Y) { return dg.fail("problem allocating for output file (Y)"); }
    if (!read_payload(s1, X1, i1.nbytes())) { return dg.fail("problem reading input file (X)"); }
    if (!read_payload(s2, X2, i2.nbytes())) { return dg.fail("problem reading input file 2 (X2)"); }
    if (fn(Y, X1, X2, i1.N(), L, nfft, stp, snip_edges, mn0, amp, lg)) { return dg.fail("problem during function call"); }
    if (os && !write_payload(*os, Y, o1.nbytes())) { return dg.fail("problem writing output file (Y)"); }
    delete[] X1; delete[] X2; delete[] Y;
    return 0;
}

}  // namespace

int main(int argc, char *argv[])
{
    ArgParser ap;
    ap.prog = "stft";
    ap.specs = {
        {'s', "step", "<uint>", ArgKind::Int, "step in samps between each frame [default=160]"},
        {'e', "snip-edges", nullptr, ArgKind::Flag, "include to snip edges [default=false]"},
        {'z', "zero-mean", nullptr, ArgKind::Flag, "include to zero the mean of each frame [default=false]"},
        {'a', "amplitude", nullptr, ArgKind::Flag, "include to output amplitude (sqrt of power) [default=false]"},
        {'l', "log", nullptr, ArgKind::Flag, "include to output log of amplitude or power [default=false]"},
        {'o', "ofile", "<file>", ArgKind::File, "output file (Y)"},
        {'h', "help", nullptr, ArgKind::Flag, "display this help and exit"},
    };
    const Diag dg{ap.prog};
    const bool parsed = ap.parse(argc, argv);
    if (ap["help"].count > 0) {
        ap.print_usage(std::cout, "input files (X1,X2)");
        std::cout << "\nSTFT power spectrogram of the real vector X1 with the window vector X2 (length L), computed on the GPU.\n"
                     "nfft is the next power of two >= L, F = nfft/2+1, and the number of frames is\n"
                     "  W = 1+(N-L)/step with -e (first frame starts at sample 0), else W = (N+step/2)/step\n"
                     "  (first frame centred at sample step/2, edges reflected).\n"
                     "The frames are contiguous in the output: W x F for a row-major input, F x W for a column-major one.\n"
                     "-z subtracts each windowed frame's mean, -a gives sqrt(power), -l the natural log of the result.\n"
                     "Files are NumPy .npy or Armadillo binary; '-' or a missing name reads stdin (one input only);\n"
                     "the output goes to -o, or to stdout when that is not a terminal.\n"
                     "Examples:\n  stft -s65 X1 X2 -o Y\n  cat X1 | stft -e - X2 > Y\n";
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
    if ((i1.T != 1 && i1.T != 2 && !i1.iscomplex()) || (i2.T != 1 && i2.T != 2 && !i2.iscomplex())) { return dg.fail("input data type must be in {1,2}"); }

    size_t stp = 160;
    if (ap["step"].count > 0) {
        if (ap["step"].ival < 1) { return dg.fail("stp must be positive"); }
        stp = (size_t)ap["step"].ival;
    }
    const int snip_edges = ap["snip-edges"].count > 0, mn0 = ap["zero-mean"].count > 0;
    const int amp = ap["amplitude"].count > 0, lg = ap["log"].count > 0;

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
    // the reference sizes Y with this formula even when L > N with -e, where it wraps around
    // (reference src/stft.cpp:183); here that case gets no buffer and the library reports it
    const size_t W = (snip_edges && L > i1.N()) ? 0 : (snip_edges ? 1 + (i1.N() - L) / stp : (i1.N() + stp / 2) / stp);
    o1.F = i1.F; o1.T = i1.T; o1.colmajor = i1.colmajor; o1.ndim = 2;
    o1.R = i1.isrowmajor() ? W : F;
    o1.C = i1.isrowmajor() ? F : W;
    o1.S = i1.S; o1.H = i1.H;

    if (int rc = open_output(dg, ap["ofile"], out)) { return rc; }
    if (snip_edges && L > i1.N()) {
        // let the library print its own message for the invalid combination before any header is written
        float dummy = 0.f;
        if (i1.T == 1) { (void)stft_s(&dummy, &dummy, &dummy, i1.N(), L, nfft, stp, snip_edges, mn0, amp, lg); }
        else { double dd = 0.0; (void)stft_d(&dd, &dd, &dd, i1.N(), L, nfft, stp, snip_edges, mn0, amp, lg); }
        return dg.fail("problem during function call");
    }
    if (out.s && !write_header(*out.s, o1, err)) { return dg.fail("problem writing header for output file 1 (" + err + ")"); }

    if (o1.T == 1) { return process<float>(dg, *in.s1, *in.s2, i1, i2, o1, out.s, L, nfft, stp, snip_edges, mn0, amp, lg, stft_s); }
    if (o1.T == 2) { return process<double>(dg, *in.s1, *in.s2, i1, i2, o1, out.s, L, nfft, stp, snip_edges, mn0, amp, lg, stft_d); }
    return dg.fail("data type not supported");
}
