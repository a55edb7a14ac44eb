// stft: short-term Fourier transform of a univariate signal file X1 with a window file X2, written
// as a W x F (row-major input) or F x W (column-major input) tensor file.  Same options, defaults,
// checks, output sizing and exit codes as the reference tool (reference src/stft.cpp:23-258); the
// transform itself is the stft_s / stft_d entry point of libdsp_stft_hip.so (reference call site
// src/stft.cpp:219,241).  Opt-in additions (the reference tool has none of them): -d for a matrix of
// signals (the batched entry points), --device, --ngpu.
#include "cli_common.hpp"
#include "../include/dsp_stft.h"

#include <cstring>
#include <new>

using namespace dspcli;

namespace {

template <typename T> struct Calls;
template <> struct Calls<float> {
    static int one(float *Y, const float *X, const float *w, size_t N, size_t L, size_t nfft, size_t stp, int e, int z, int a, int l) { return stft_s(Y, X, w, N, L, nfft, stp, e, z, a, l); }
    static int strided(float *Y, const float *X, const float *w, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp, int e, int z, int a, int l) { return dsp_stft_batch_strided_s(Y, X, w, B, N, ldx, incx, L, nfft, stp, e, z, a, l); }
    static int multi(float *Y, const float *X, const float *w, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp, int e, int z, int a, int l, int ngpu) { return dsp_stft_multi_s(Y, X, w, B, N, ldx, L, nfft, stp, e, z, a, l, nullptr, ngpu); }
};
template <> struct Calls<double> {
    static int one(double *Y, const double *X, const double *w, size_t N, size_t L, size_t nfft, size_t stp, int e, int z, int a, int l) { return stft_d(Y, X, w, N, L, nfft, stp, e, z, a, l); }
    static int strided(double *Y, const double *X, const double *w, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t nfft, size_t stp, int e, int z, int a, int l) { return dsp_stft_batch_strided_d(Y, X, w, B, N, ldx, incx, L, nfft, stp, e, z, a, l); }
    static int multi(double *Y, const double *X, const double *w, size_t B, size_t N, size_t ldx, size_t L, size_t nfft, size_t stp, int e, int z, int a, int l, int ngpu) { return dsp_stft_multi_d(Y, X, w, B, N, ldx, L, nfft, stp, e, z, a, l, nullptr, ngpu); }
};

template <typename T>
int process(const Diag &dg, std::istream &s1, std::istream &s2, const TensorInfo &i1, const TensorInfo &i2, const TensorInfo &o1, std::ostream *os,
            const Signals &sg, size_t L, size_t nfft, size_t stp, int snip_edges, int mn0, int amp, int lg, int ngpu)
{
    T *X1 = new (std::nothrow) T[i1.N()];
    if (!X1) { return dg.fail("problem allocating for input file 1 (X1)"); }
    T *X2 = new (std::nothrow) T[i2.N()];
    if (!X2) { return dg.fail("problem allocating for input file 2 (X2)"); }
    T *Y = new (std::nothrow) T[o1.N() ? o1.N() : 1];
    if (!Y) { return dg.fail("problem allocating for output file (Y)"); }
    if (!read_payload(s1, X1, i1.nbytes())) { return dg.fail("problem reading input file (X)"); }
    if (!read_payload(s2, X2, i2.nbytes())) { return dg.fail("problem reading input file 2 (X2)"); }
    int rc;
    if (!sg.multi && ngpu == 1) { rc = Calls<T>::one(Y, X1, X2, sg.N, L, nfft, stp, snip_edges, mn0, amp, lg); }      // the reference's own call
    else if (ngpu > 1) { rc = Calls<T>::multi(Y, X1, X2, sg.B, sg.N, sg.ldx, L, nfft, stp, snip_edges, mn0, amp, lg, ngpu); }
    else { rc = Calls<T>::strided(Y, X1, X2, sg.B, sg.N, sg.ldx, sg.incx, L, nfft, stp, snip_edges, mn0, amp, lg); }
    if (rc) { return dg.fail("problem during function call"); }
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
        {'d', "dim", "<uint>", ArgKind::Int, "X1 is a matrix: transform every vector along this axis (0 or 1) [default: X1 is a vector]"},
        {0, "device", "<uint>", ArgKind::Int, "GPU to run on [default: the current device]"},
        {0, "ngpu", "<uint>", ArgKind::Int, "split the signals over this many GPUs (needs contiguous signals) [default=1]"},
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
                     "With -d X1 may be a matrix: every vector along axis d (0: columns, 1: rows) is a signal, and the output is one\n"
                     "such block per signal, B x W x F (row-major) or F x W x B (column-major); --ngpu splits the signals over GPUs.\n"
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
    Signals sg;
    if (int rc = locate_signals(dg, i1, ap["dim"], sg)) { return rc; }
    if (i1.isempty()) { return dg.fail("input 1 (X1) found to be empty"); }
    if (!i2.isvec()) { return dg.fail("input 2 (X2) must be a vector"); }
    if (i2.isempty()) { return dg.fail("input 2 (X2) found to be empty"); }

    const size_t L = i2.N();
    size_t nfft = 1;
    while (nfft < L) { nfft *= 2; }
    const size_t F = nfft / 2 + 1;
    // the reference sizes Y with this formula even when L > N with -e, where it wraps around
    // (reference src/stft.cpp:183); here that case gets no buffer and the library reports it
    const size_t W = (snip_edges && L > sg.N) ? 0 : (snip_edges ? 1 + (sg.N - L) / stp : (sg.N + stp / 2) / stp);
    size_output(i1, sg, W, F, o1);
    int device, ngpu;
    if (int rc = device_options(dg, ap["device"], ap["ngpu"], device, ngpu)) { return rc; }
    if (ngpu > 1 && sg.incx != 1) { return dg.fail("with --ngpu the signals must be contiguous (row-major: -d 1, column-major: -d 0)"); }
    if (device >= 0 && dsp_set_device(device)) { return dg.fail("problem selecting the device"); }

    if (int rc = open_output(dg, ap["ofile"], out)) { return rc; }
    if (snip_edges && L > sg.N) {
        // let the library print its own message for the invalid combination before any header is written
        float dummy = 0.f;
        if (i1.T == 1) { (void)stft_s(&dummy, &dummy, &dummy, sg.N, L, nfft, stp, snip_edges, mn0, amp, lg); }
        else { double dd = 0.0; (void)stft_d(&dd, &dd, &dd, sg.N, L, nfft, stp, snip_edges, mn0, amp, lg); }
        return dg.fail("problem during function call");
    }
    if (out.s && !write_header(*out.s, o1, err)) { return dg.fail("problem writing header for output file 1 (" + err + ")"); }

    if (o1.T == 1) { return process<float>(dg, *in.s1, *in.s2, i1, i2, o1, out.s, sg, L, nfft, stp, snip_edges, mn0, amp, lg, ngpu); }
    if (o1.T == 2) { return process<double>(dg, *in.s1, *in.s2, i1, i2, o1, out.s, sg, L, nfft, stp, snip_edges, mn0, amp, lg, ngpu); }
    return dg.fail("data type not supported");
}
