#include "tensor_io.hpp"

#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <istream>
#include <ostream>
#include <sstream>
#include <vector>

namespace dspcli {
namespace {

bool read_exact(std::istream &is, char *dst, size_t n)
{
    is.read(dst, (std::streamsize)n);
    return (size_t)is.gcount() == n;
}

bool read_line(std::istream &is, std::string &line, size_t limit)
{
    line.clear();
    char c;
    while (line.size() < limit && is.get(c)) {
        if (c == '\n') { return true; }
        line.push_back(c);
    }
    return false;
}

// value text of key in a Python dict literal such as {'descr': '<f4', 'fortran_order': False, 'shape': (3, 4), }
bool dict_value(const std::string &d, const std::string &key, std::string &val)
{
    size_t p = d.find("'" + key + "'");
    if (p == std::string::npos) { p = d.find("\"" + key + "\""); }
    if (p == std::string::npos) { return false; }
    p = d.find(':', p);
    if (p == std::string::npos) { return false; }
    ++p;
    while (p < d.size() && std::isspace((unsigned char)d[p])) { ++p; }
    if (p >= d.size()) { return false; }
    size_t e;
    if (d[p] == '(') { e = d.find(')', p); if (e == std::string::npos) { return false; } ++e; }
    else if (d[p] == '\'' || d[p] == '"') { e = d.find(d[p], p + 1); if (e == std::string::npos) { return false; } ++e; }
    else { e = p; while (e < d.size() && d[e] != ',' && d[e] != '}') { ++e; } }
    val = d.substr(p, e - p);
    while (!val.empty() && std::isspace((unsigned char)val.back())) { val.pop_back(); }
    return true;
}

bool parse_dims(const std::string &text, std::vector<size_t> &dims)
{
    dims.clear();
    size_t p = 0;
    while (p < text.size()) {
        while (p < text.size() && !std::isdigit((unsigned char)text[p])) {
            if (text[p] == '-') { return false; }
            ++p;
        }
        if (p >= text.size()) { break; }
        size_t v = 0;
        while (p < text.size() && std::isdigit((unsigned char)text[p])) { v = v * 10 + (size_t)(text[p] - '0'); ++p; }
        dims.push_back(v);
    }
    return true;
}

void set_dims(TensorInfo &info, const std::vector<size_t> &d)
{
    info.ndim = (int)d.size();
    info.R = d.size() > 0 ? d[0] : 1;
    info.C = d.size() > 1 ? d[1] : 1;
    info.S = d.size() > 2 ? d[2] : 1;
    info.H = d.size() > 3 ? d[3] : 1;
}

bool read_npy(std::istream &is, TensorInfo &info, std::string &err)
{
    // the first byte (0x93) is already consumed
    char rest[7];
    if (!read_exact(is, rest, 7) || std::memcmp(rest, "NUMPY", 5) != 0) { err = "not a NumPy file (bad magic)"; return false; }
    const int major = (unsigned char)rest[5];
    size_t hlen;
    if (major == 1) {
        unsigned char b[2];
        if (!read_exact(is, reinterpret_cast<char *>(b), 2)) { err = "truncated NumPy header"; return false; }
        hlen = (size_t)b[0] | ((size_t)b[1] << 8);
    } else if (major == 2 || major == 3) {
        unsigned char b[4];
        if (!read_exact(is, reinterpret_cast<char *>(b), 4)) { err = "truncated NumPy header"; return false; }
        hlen = (size_t)b[0] | ((size_t)b[1] << 8) | ((size_t)b[2] << 16) | ((size_t)b[3] << 24);
    } else { err = "unsupported NumPy format version"; return false; }
    if (hlen > (1u << 20)) { err = "NumPy header too long"; return false; }
    std::string dict(hlen, '\0');
    if (!read_exact(is, &dict[0], hlen)) { err = "truncated NumPy header"; return false; }
    std::string descr, forder, shape;
    if (!dict_value(dict, "descr", descr) || !dict_value(dict, "fortran_order", forder) || !dict_value(dict, "shape", shape)) {
        err = "NumPy header lacks descr / fortran_order / shape"; return false;
    }
    if (descr.size() >= 2 && (descr[0] == '\'' || descr[0] == '"')) { descr = descr.substr(1, descr.size() - 2); }
    if (descr == "<f4" || descr == "=f4") { info.T = 1; }
    else if (descr == "<f8" || descr == "=f8") { info.T = 2; }
    else if (descr == "<c8" || descr == "=c8") { info.T = 101; }
    else if (descr == "<c16" || descr == "=c16") { info.T = 102; }
    else { err = "NumPy dtype " + descr + " is not one of <f4 <f8 <c8 <c16"; return false; }
    info.colmajor = forder.compare(0, 4, "True") == 0;
    std::vector<size_t> dims;
    if (!parse_dims(shape, dims) || dims.size() > 4) { err = "NumPy shape must have at most 4 dimensions"; return false; }
    set_dims(info, dims);
    info.F = FMT_NPY;
    return true;
}

bool read_arma(std::istream &is, TensorInfo &info, std::string &err)
{
    // the first byte ('A') is already consumed
    std::string line;
    if (!read_line(is, line, 64)) { err = "truncated Armadillo header"; return false; }
    line = "A" + line;
    bool cube;
    if (line.compare(0, 13, "ARMA_MAT_BIN_") == 0) { cube = false; }
    else if (line.compare(0, 13, "ARMA_CUB_BIN_") == 0) { cube = true; }
    else { err = "not an Armadillo binary file (" + line + ")"; return false; }
    const std::string ty = line.substr(13);
    if (ty == "FN004") { info.T = 1; }
    else if (ty == "FN008") { info.T = 2; }
    else if (ty == "FC008") { info.T = 101; }
    else if (ty == "FC016") { info.T = 102; }
    else { err = "Armadillo element type " + ty + " is not one of FN004 FN008 FC008 FC016"; return false; }
    if (!read_line(is, line, 128)) { err = "truncated Armadillo header"; return false; }
    std::vector<size_t> dims;
    if (!parse_dims(line, dims) || dims.size() != (cube ? 3u : 2u)) { err = "bad Armadillo size line"; return false; }
    set_dims(info, dims);
    info.colmajor = true;
    info.F = FMT_ARMA;
    return true;
}

}  // namespace

bool read_header(std::istream &is, TensorInfo &info, std::string &err)
{
    info = TensorInfo{};
    char c;
    if (!is.get(c)) { err = "empty input"; return false; }
    if ((unsigned char)c == 0x93) { return read_npy(is, info, err); }
    if (c == 'A') { return read_arma(is, info, err); }
    err = "unknown file format (neither NumPy .npy nor Armadillo binary)";
    return false;
}

bool write_header(std::ostream &os, const TensorInfo &info, std::string &err)
{
    if (info.elem_bytes() == 0) { err = "unknown element type"; return false; }
    if (info.F == FMT_NPY) {
        const char *descr = info.T == 1 ? "<f4" : info.T == 2 ? "<f8" : info.T == 101 ? "<c8" : "<c16";
        int nd = info.H > 1 ? 4 : info.S > 1 ? 3 : 2;
        if (info.ndim == 1 && info.C == 1 && info.S == 1 && info.H == 1) { nd = 1; }
        std::ostringstream d;
        d << "{'descr': '" << descr << "', 'fortran_order': " << (info.colmajor ? "True" : "False") << ", 'shape': (";
        const size_t dims[4] = {info.R, info.C, info.S, info.H};
        for (int i = 0; i < nd; ++i) { d << dims[i] << (nd == 1 || i + 1 < nd ? "," : "") << (i + 1 < nd ? " " : ""); }
        d << "), }";
        std::string dict = d.str();
        // magic (6) + version (2) + length (2) + dict + '\n' is padded with spaces to a multiple of 64
        size_t total = 10 + dict.size() + 1;
        const size_t pad = (64 - total % 64) % 64;
        dict.append(pad, ' ');
        dict.push_back('\n');
        if (dict.size() > 65535) { err = "NumPy header too long"; return false; }
        const unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(dict.size() & 0xff), (unsigned char)(dict.size() >> 8)};
        os.write(reinterpret_cast<const char *>(pre), 10);
        os.write(dict.data(), (std::streamsize)dict.size());
        return (bool)os;
    }
    if (info.F == FMT_ARMA) {
        if (!info.colmajor) { err = "Armadillo files are column-major"; return false; }
        if (info.H > 1) { err = "Armadillo files hold at most 3 dimensions"; return false; }
        const char *ty = info.T == 1 ? "FN004" : info.T == 2 ? "FN008" : info.T == 101 ? "FC008" : "FC016";
        if (info.S > 1) { os << "ARMA_CUB_BIN_" << ty << "\n" << info.R << " " << info.C << " " << info.S << "\n"; }
        else { os << "ARMA_MAT_BIN_" << ty << "\n" << info.R << " " << info.C << "\n"; }
        return (bool)os;
    }
    err = "no file format set";
    return false;
}

}  // namespace dspcli
