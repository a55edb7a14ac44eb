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
