// Tensor file headers of the stft / stft_flt command-line tools: NumPy .npy (v1-v3 headers, C or
// Fortran order) and Armadillo binary matrices and cubes, detected by their magic bytes, read from
// and written to non-seekable streams.  The reference tools delegate this to read_input_header /
// write_output_header and the ioinfo struct of a header that is not part of its tree (reference
// src/stft.cpp:14,37,139-140,199); the fields and predicates below are the ones its tools use.
#pragma once
#include <cstddef>
#include <iosfwd>
#include <string>

namespace dspcli {

enum FileFormat : int { FMT_NONE = 0, FMT_ARMA = 101, FMT_NPY = 102 };

struct TensorInfo {
    int F = FMT_NONE;            // file format
    size_t T = 0;                // element type: 1 float32, 2 float64, 101 complex64, 102 complex128
    size_t R = 0, C = 0, S = 0, H = 0;
    bool colmajor = false;
    int ndim = 2;                // dimensions the file's header listed (informational)

    size_t N() const { return R * C * S * H; }
    size_t elem_bytes() const { return T == 1 ? 4 : T == 2 ? 8 : T == 101 ? 8 : T == 102 ? 16 : 0; }
    size_t nbytes() const { return N() * elem_bytes(); }
    bool isempty() const { return N() == 0; }
    bool isreal() const { return T == 1 || T == 2; }
    bool iscomplex() const { return T == 101 || T == 102; }
    bool iscolmajor() const { return colmajor; }
    bool isrowmajor() const { return !colmajor; }
    bool isvec() const { return (R == N()) || (C == N()) || (S == N()) || (H == N()); }
    bool isrowvec() const { return C == N() && N() > 0; }
    bool ismat() const { return S <= 1 && H <= 1; }
};

// Reads one header; on success the stream is positioned at the first payload byte.
// On failure returns false and puts a one-line reason into err.
bool read_header(std::istream &is, TensorInfo &info, std::string &err);

// Writes the header for info.F (NumPy or Armadillo).  Armadillo files are column-major by
// definition: a row-major tensor cannot be written in that format (returns false).
bool write_header(std::ostream &os, const TensorInfo &info, std::string &err);

}  // namespace dspcli
