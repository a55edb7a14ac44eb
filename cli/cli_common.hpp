// Shared plumbing of the stft / stft_flt tools: where the inputs come from and the output goes
// (reference src/stft.cpp:118-135,190-199), coloured diagnostics, whole-payload reads and writes.
#pragma once
#include "args.hpp"
#include "tensor_io.hpp"

#include <fstream>
#include <iostream>
#include <string>
#include <unistd.h>
#include <vector>

namespace dspcli {

struct Diag {
    std::string prog;
    int fail(const std::string &msg) const
    {
        std::cerr << prog << ": \033[1;31merror:\033[0m " << msg << std::endl;
        return 1;
    }
};

// input i (0-based) is stdin when it is missing, empty or "-"
inline bool is_stdin_arg(const std::vector<std::string> &pos, size_t i) { return pos.size() <= i || pos[i].empty() || pos[i] == "-"; }

struct Inputs {
    std::ifstream f1, f2;
    std::istream *s1 = nullptr, *s2 = nullptr;
};

// Opens the two inputs with the reference's rules: at most one of them may be stdin, and stdin
// must not be a terminal.  Returns 0 or the exit code after printing the reason.
inline int open_inputs(const Diag &dg, const std::vector<std::string> &pos, Inputs &in)
{
    const bool std1 = is_stdin_arg(pos, 0), std2 = is_stdin_arg(pos, 1);
    if (std1 && std2) { return dg.fail("can only use stdin for one input"); }
    if ((std1 || std2) && isatty(fileno(stdin))) { return dg.fail("no stdin detected"); }
    if (std1) { in.s1 = &std::cin; } else { in.f1.open(pos[0], std::ios::binary); in.s1 = &in.f1; }
    if (!*in.s1) { return dg.fail("problem opening input file 1"); }
    if (std2) { in.s2 = &std::cin; } else { in.f2.open(pos[1], std::ios::binary); in.s2 = &in.f2; }
    if (!*in.s2) { return dg.fail("problem opening input file 2"); }
    return 0;
}

struct Output {
    std::ofstream f;
    std::ostream *s = nullptr;     // nullptr: nothing is written (stdout is a terminal and no -o)
};

inline int open_output(const Diag &dg, const ArgSpec &ofile, Output &out)
{
    bool to_stdout;
    if (ofile.count > 0) { to_stdout = ofile.sval.empty() || ofile.sval == "-"; }
    else { to_stdout = !isatty(fileno(stdout)); }
    if (!to_stdout && ofile.count == 0) { return 0; }
    if (to_stdout) { out.s = &std::cout; }
    else { out.f.open(ofile.sval, std::ios::binary); out.s = &out.f; }
    if (!*out.s) { return dg.fail("problem opening output file 1"); }
    return 0;
}

// Where the signals of input 1 lie.  Without -d the input must be a vector (the reference's rule).  With -d the
// input may be a matrix (R x C) transformed along axis d, every vector along that axis a signal: B signals of N samples,
// signal b at X + b * ldx, its samples incx elements apart (the tensors' two major orders, reference
// c/fft.fftw.r2hc.c:63-65: contiguous signals when d is the fastest axis, interleaved channels otherwise).
struct Signals {
    bool multi = false;
    size_t B = 1, N = 0, ldx = 0, incx = 1;
};

inline int locate_signals(const Diag &dg, const TensorInfo &i1, const ArgSpec &dimopt, Signals &sg)
{
    if (dimopt.count == 0) {
        if (!i1.isvec()) { return dg.fail("input 1 (X1) must be a vector"); }
        sg.multi = false; sg.B = 1; sg.N = i1.N(); sg.ldx = i1.N(); sg.incx = 1;
        return 0;
    }
    if (dimopt.ival < 0 || dimopt.ival > 1) { return dg.fail("dim must be in {0,1}"); }
    if (!i1.ismat()) { return dg.fail("input 1 (X1) must be a matrix or a vector with -d"); }
    const bool along_rows = dimopt.ival == 1;              // axis 1: every row is a signal
    sg.multi = true;
    sg.N = along_rows ? i1.C : i1.R;
    sg.B = along_rows ? i1.R : i1.C;
    const bool contiguous = along_rows == i1.isrowmajor();  // the transformed axis is the fastest one
    if (contiguous) { sg.ldx = sg.N; sg.incx = 1; }
    else { sg.ldx = 1; sg.incx = sg.B; }
    return 0;
}

// Output header: frames contiguous, W x F for a row-major input and F x W for a column-major one (the same bytes);
// with -d one such block per signal: B x W x F, or F x W x B.
inline void size_output(const TensorInfo &i1, const Signals &sg, size_t W, size_t F, TensorInfo &o1)
{
    o1.F = i1.F; o1.T = i1.T; o1.colmajor = i1.colmajor;
    if (!sg.multi) {
        o1.ndim = 2;
        o1.R = i1.isrowmajor() ? W : F;
        o1.C = i1.isrowmajor() ? F : W;
        o1.S = i1.S; o1.H = i1.H;
    } else {
        o1.ndim = 3;
        o1.R = i1.isrowmajor() ? sg.B : F;
        o1.C = W;
        o1.S = i1.isrowmajor() ? F : sg.B;
        o1.H = 1;
    }
}

// --device / --ngpu: values or the reason they are unusable
inline int device_options(const Diag &dg, const ArgSpec &devopt, const ArgSpec &ngpuopt, int &device, int &ngpu)
{
    device = -1; ngpu = 1;
    if (devopt.count > 0) {
        if (devopt.ival < 0 || devopt.ival > 63) { return dg.fail("device must be in [0 63]"); }
        device = (int)devopt.ival;
    }
    if (ngpuopt.count > 0) {
        if (ngpuopt.ival < 1 || ngpuopt.ival > 64) { return dg.fail("ngpu must be in [1 64]"); }
        ngpu = (int)ngpuopt.ival;
    }
    if (device >= 0 && ngpu > 1) { return dg.fail("--device and --ngpu cannot be combined"); }
    return 0;
}

inline bool read_payload(std::istream &is, void *dst, size_t nbytes)
{
    char *p = static_cast<char *>(dst);
    while (nbytes > 0) {
        const size_t chunk = nbytes < (1u << 30) ? nbytes : (1u << 30);
        is.read(p, (std::streamsize)chunk);
        if ((size_t)is.gcount() != chunk) { return false; }
        p += chunk; nbytes -= chunk;
    }
    return true;
}

inline bool write_payload(std::ostream &os, const void *src, size_t nbytes)
{
    const char *p = static_cast<const char *>(src);
    while (nbytes > 0) {
        const size_t chunk = nbytes < (1u << 30) ? nbytes : (1u << 30);
        os.write(p, (std::streamsize)chunk);
        if (!os) { return false; }
        p += chunk; nbytes -= chunk;
    }
    os.flush();
    return (bool)os;
}

}  // namespace dspcli
