// dsp_tensor: reads one tensor file (NumPy .npy or Armadillo binary) and prints its header
// (info) or rewrites it in the requested format (copy).  Needs no GPU; it exists so that the
// file layer of the stft tools can be exercised and files converted between the two formats.
//   dsp_tensor info [file]            -> one line: format dtype R C S H major
//   dsp_tensor copy [-f npy|arma] [file] [-o out]
#include "cli_common.hpp"

#include <vector>

using namespace dspcli;

int main(int argc, char *argv[])
{
    const Diag dg{"dsp_tensor"};
    if (argc < 2) { return dg.fail("usage: dsp_tensor info|copy [-f npy|arma] [file] [-o out]"); }
    const std::string cmd = argv[1];
    ArgParser ap;
    ap.prog = "dsp_tensor";
    ap.max_positionals = 1;
    ap.specs = {
        {'f', "format", "<npy|arma>", ArgKind::File, "output file format [default: as the input]"},
        {'o', "ofile", "<file>", ArgKind::File, "output file"},
    };
    if (!ap.parse(argc - 1, argv + 1)) {
        for (const auto &e : ap.errors) { dg.fail(e); }
        return 1;
    }
    std::ifstream f;
    std::istream *is = &std::cin;
    if (!is_stdin_arg(ap.positionals, 0)) { f.open(ap.positionals[0], std::ios::binary); is = &f; }
    if (!*is) { return dg.fail("problem opening input file"); }
    TensorInfo ti;
    std::string err;
    if (!read_header(*is, ti, err)) { return dg.fail("problem reading header (" + err + ")"); }
    if (cmd == "info") {
        std::cout << (ti.F == FMT_NPY ? "npy" : "arma") << " " << ti.T << " " << ti.R << " " << ti.C << " " << ti.S << " " << ti.H << " "
                  << (ti.colmajor ? "colmajor" : "rowmajor") << (ti.isvec() ? " vec" : "") << std::endl;
        return 0;
    }
    if (cmd != "copy") { return dg.fail("unknown command " + cmd); }
    TensorInfo to = ti;
    if (ap["format"].count > 0) {
        if (ap["format"].sval == "npy") { to.F = FMT_NPY; }
        else if (ap["format"].sval == "arma") { to.F = FMT_ARMA; }
        else { return dg.fail("format must be npy or arma"); }
    }
    std::vector<char> buf(ti.nbytes());
    if (!read_payload(*is, buf.data(), buf.size())) { return dg.fail("problem reading payload"); }
    Output out;
    if (int rc = open_output(dg, ap["ofile"], out)) { return rc; }
    if (!out.s) { return dg.fail("no output (use -o or redirect stdout)"); }
    if (!write_header(*out.s, to, err)) { return dg.fail("problem writing header (" + err + ")"); }
    if (!write_payload(*out.s, buf.data(), buf.size())) { return dg.fail("problem writing payload"); }
    return 0;
}
