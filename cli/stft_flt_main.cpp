// stft_flt: STFT of a univariate signal file X1 with a window file X2 where the frame centres are
// c0 + w*stp for floating-point c0 and stp and frames are zero-padded outside the signal.  Same
// options, defaults, checks and output sizing as the reference tool (reference
// srci/stft_flt.cpp:69-118, src/stft_flt.cpp); the transform is the stft_flt_s / stft_flt_d entry
// point of libdsp_stft_hip.so.  Opt-in additions (the reference tool has none of them): -d for a matrix of
// signals (the batched entry points), --device, --ngpu.
#include "cli_common.hpp"
#include "../include/dsp_stft.h"

#include <cfloat>
#include <new>

using namespace dspcli;

namespace {

template <typename T> struct Calls;
template <> struct Calls<float> {
    static int one(float *Y, const float *X, const float *w, size_t N, size_t L, size_t W, size_t nfft, double c0, double stp, int z, int a, int l) { return stft_flt_s(Y, X, w, N, L, W, nfft, (float)c0, (float)stp, z, a, l); }
    static int strided(float *Y, const float *X, const float *w, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t W, size_t nfft, double c0, double stp, int z, int a, int l) { return dsp_stft_flt_batch_strided_s(Y, X, w, B, N, ldx, incx, L, W, nfft, (float)c0, (float)stp, z, a, l); }
    static int multi(float *Y, const float *X, const float *w, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft, double c0, double stp, int z, int a, int l, int ngpu) { return dsp_stft_flt_multi_s(Y, X, w, B, N, ldx, L, W, nfft, (float)c0, (float)stp, z, a, l, nullptr, ngpu); }
};
template <> struct Calls<double> {
    static int one(double *Y, const double *X, const double *w, size_t N, size_t L, size_t W, size_t nfft, double c0, double stp, int z, int a, int l) { return stft_flt_d(Y, X, w, N, L, W, nfft, c0, stp, z, a, l); }
    static int strided(double *Y, const double *X, const double *w, size_t B, size_t N, size_t ldx, size_t incx, size_t L, size_t W, size_t nfft, double c0, double stp, int z, int a, int l) { return dsp_stft_flt_batch_strided_d(Y, X, w, B, N, ldx, incx, L, W, nfft, c0, stp, z, a, l); }
    static int multi(double *Y, const double *X, const double *w, size_t B, size_t N, size_t ldx, size_t L, size_t W, size_t nfft, double c0, double stp, int z, int a, int l, int ngpu) { return dsp_stft_flt_multi_d(Y, X, w, B, N, ldx, L, W, nfft, c0, stp, z, a, l, nullptr, ngpu); }
};

template <typename T>
int process(const Diag &dg, std::istream &s1, std::istream &s2, const TensorInfo &i1, const TensorInfo &i2, const TensorInfo &o1, std::ostream *os,
            const Signals &sg, size_t L, size_t W, size_t nfft, double c0, double stp, int mn0, int amp, int lg, int ngpu)
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
    if (!sg.multi && ngpu == 1) { rc = Calls<T>::one(Y, X1, X2, sg.N, L, W, nfft, c0, stp, mn0, amp, lg); }          // the reference's own call
    else if (ngpu > 1) { rc = Calls<T>::multi(Y, X1, X2, sg.B, sg.N, sg.ldx, L, W, nfft, c0, stp, mn0, amp, lg, ngpu); }
    else { rc = Calls<T>::strided(Y, X1, X2, sg.B, sg.N, sg.ldx, sg.incx, L, W, nfft, c0, stp, mn0, amp, lg); }
    if (rc) { return dg.fail("problem during function call"); }
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
        std::cout << "\nSTFT power spectrogram of the real vector X1 with the window vector X2 (length L), computed on the GPU,\n"
                     "with W frames centred at round(c0 + w*stp), c0 and stp floating-point; samples outside X1 count as zero.\n"
                     "nfft is the next power of two >= L, F = nfft/2+1; the default W is (N-1-c0)/stp.\n"
                     "The output is W x F for a row-major input, F x W for a column-major one.\n"
                     "-z subtracts each windowed frame's mean, -a gives sqrt(power), -l the natural log of the result.\n"
                     "With -d X1 may be a matrix: every vector along axis d (0: columns, 1: rows) is a signal (N its length), and the\n"
                     "output is one such block per signal, B x W x F (row-major) or F x W x B (column-major); --ngpu splits them over GPUs.\n"
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

    // (with -d the signal length is the extent of that axis; without it the element count, as in the reference, whose
    // checks of c0 and W come before the check that X1 is a vector)
    Signals sg;
    const bool have_dim = ap["dim"].count > 0;
    if (have_dim) { if (int rc = locate_signals(dg, i1, ap["dim"], sg)) { return rc; } }
    const size_t Nsig = have_dim ? sg.N : i1.N();
    const double c0 = ap["c0"].count > 0 ? ap["c0"].dval : 0.0;
    if (c0 > double(Nsig - 1u)) { return dg.fail("c0 (center of first frame) must be <= N-1"); }
    const double stp = ap["stp"].count > 0 ? ap["stp"].dval : 160.0;
    if (stp < double(FLT_EPSILON)) { return dg.fail("stp (step size) must be positive"); }
    size_t W;
    if (ap["nframes"].count == 0) { W = size_t((double(Nsig - 1u) - c0) / stp); }
    else if (ap["nframes"].ival < 1) { return dg.fail("W (nframes) must be positive"); }
    else { W = size_t(ap["nframes"].ival); }
    const int mn0 = ap["zero-mean"].count > 0, amp = ap["amplitude"].count > 0, lg = ap["log"].count > 0;

    if (i1.iscomplex() || i2.iscomplex()) { return dg.fail("inputs must be real-valued"); }
    if (i1.T != i2.T) { return dg.fail("inputs must have the same data type"); }
    if (!have_dim) { if (int rc = locate_signals(dg, i1, ap["dim"], sg)) { return rc; } }
    if (i1.isempty()) { return dg.fail("input 1 (X1) found to be empty"); }
    if (!i2.isvec()) { return dg.fail("input 2 (X2) must be a vector"); }
    if (i2.isempty()) { return dg.fail("input 2 (X2) found to be empty"); }

    const size_t L = i2.N();
    size_t nfft = 1;
    while (nfft < L) { nfft *= 2; }
    const size_t F = nfft / 2 + 1;
    size_output(i1, sg, W, F, o1);
    int device, ngpu;
    if (int rc = device_options(dg, ap["device"], ap["ngpu"], device, ngpu)) { return rc; }
    if (ngpu > 1 && sg.incx != 1) { return dg.fail("with --ngpu the signals must be contiguous (row-major: -d 1, column-major: -d 0)"); }
    if (device >= 0 && dsp_set_device(device)) { return dg.fail("problem selecting the device"); }

    if (int rc = open_output(dg, ap["ofile"], out)) { return rc; }
    if (out.s && !write_header(*out.s, o1, err)) { return dg.fail("problem writing header for output file 1 (" + err + ")"); }
    dsp_stft_flt_set_mode(ap["intended"].count > 0);

    if (o1.T == 1) { return process<float>(dg, *in.s1, *in.s2, i1, i2, o1, out.s, sg, L, W, nfft, c0, stp, mn0, amp, lg, ngpu); }
    if (o1.T == 2) { return process<double>(dg, *in.s1, *in.s2, i1, i2, o1, out.s, sg, L, W, nfft, c0, stp, mn0, amp, lg, ngpu); }
    return dg.fail("data type not supported");
}
