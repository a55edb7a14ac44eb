"""ctypes front end of the CPU oracle (oracle/liborc.so) and of the reference's FFTW-free
sources compiled as they lie (oracle/_ref/libdspref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg as the checker.  Nothing under dsp_amd/ imports it.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_sz = C.c_size_t
_int = C.c_int


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _load(path):
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    return C.CDLL(path)


_orc = None
_ref = None


def orc():
    global _orc
    if _orc is None:
        _orc = _load(os.path.join(_HERE, "liborc.so"))
        _orc.orc_stft_num_frames.restype = _sz
        _orc.orc_stft_num_frames.argtypes = [_sz, _sz, _sz, _int]
    return _orc


def ref():
    """The compiled reference pieces; None when oracle/_ref was never built."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libdspref.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
    return _ref


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "s", C.c_float
    if dtype == np.float64:
        return "d", C.c_double
    raise TypeError(dtype)


def _csfx(dtype):
    dtype = np.dtype(dtype)
    return ("c", np.float32) if dtype in (np.complex64, np.float32) else ("z", np.float64)


def num_frames(N, L, stp, snip_edges):
    return int(orc().orc_stft_num_frames(N, L, stp, int(bool(snip_edges))))


def stft(x, win, stp, nfft=None, snip_edges=False, mn0=False, amp=False, lg=False, exact=True):
    """Oracle for stft_s/stft_d (reference c/stft.c).  Returns (rc, Y[W, F])."""
    x = np.ascontiguousarray(x)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    s, _ = _sfx(x.dtype)
    N, L = x.size, win.size
    if nfft is None:
        nfft = 1
        while nfft < L:
            nfft *= 2
    F = nfft // 2 + 1
    W = num_frames(N, L, stp, snip_edges) if (L >= 1 and stp >= 1) else 0
    Y = np.zeros((W, F), dtype=x.dtype)
    f = getattr(orc(), f"orc_stft_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(x), _ptr(win), _sz(N), _sz(L), _sz(nfft), _sz(stp), _int(snip_edges), _int(mn0),
           _int(amp), _int(lg), _int(0 if exact else 1))
    return rc, Y


def stft_flt(x, win, W, c0, stp, nfft=None, mn0=False, amp=False, lg=False, intended=False, exact=True):
    """Oracle for stft_flt_s/_d (reference c/stft_flt.c).  Returns (rc, Y[W, F])."""
    x = np.ascontiguousarray(x)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    s, ct = _sfx(x.dtype)
    N, L = x.size, win.size
    if nfft is None:
        nfft = 1
        while nfft < L:
            nfft *= 2
    F = nfft // 2 + 1
    Y = np.zeros((W, F), dtype=x.dtype)
    f = getattr(orc(), f"orc_stft_flt_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(x), _ptr(win), _sz(N), _sz(L), _sz(W), _sz(nfft), ct(c0), ct(stp), _int(mn0), _int(amp),
           _int(lg), _int(intended), _int(0 if exact else 1))
    return rc, Y


def flt_starts(L, W, c0, stp, dtype):
    s, ct = _sfx(dtype)
    out = np.zeros(W, dtype=np.int32)
    getattr(orc(), f"orc_flt_starts_{s}")(_ptr(out), _sz(L), _sz(W), ct(c0), ct(stp))
    return out


def _frame_generic(lib, prefix, x, L, stp, snip_edges):
    x = np.ascontiguousarray(x)
    cplx = np.iscomplexobj(x)
    s = {"float32": "s", "float64": "d", "complex64": "c", "complex128": "z"}[x.dtype.name]
    N = x.size
    W = num_frames(N, L, stp, snip_edges) if (L >= 1 and stp >= 1) else 0
    Y = np.zeros((W, L), dtype=x.dtype)
    f = getattr(lib, f"{prefix}frame_univar_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(x), _sz(N), _sz(L), _sz(stp), _int(snip_edges))
    del cplx
    return rc, Y


def frame_univar(x, L, stp, snip_edges=False):
    return _frame_generic(orc(), "orc_", x, L, stp, snip_edges)


def ref_frame_univar(x, L, stp, snip_edges=False):
    return _frame_generic(ref(), "", x, L, stp, snip_edges)


def _frame_flt_generic(lib, prefix, x, L, W, c0, stp):
    x = np.ascontiguousarray(x)
    s = {"float32": "s", "float64": "d", "complex64": "c", "complex128": "z"}[x.dtype.name]
    ct = C.c_float if x.dtype in (np.float32, np.complex64) else C.c_double
    Y = np.zeros((W, L), dtype=x.dtype)
    f = getattr(lib, f"{prefix}frame_univar_flt_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(x), _sz(x.size), _sz(L), _sz(W), ct(c0), ct(stp))
    return rc, Y


def frame_univar_flt(x, L, W, c0, stp):
    return _frame_flt_generic(orc(), "orc_", x, L, W, c0, stp)


def ref_frame_univar_flt(x, L, W, c0, stp):
    return _frame_flt_generic(ref(), "", x, L, W, c0, stp)


def _dims4(shape):
    shape = tuple(shape) + (1,) * (4 - len(shape))
    return shape


def _apply_win_generic(lib, prefix, x, win, dim, iscolmajor):
    """x is given as a numpy array whose *logical* shape is x.shape (<= 4-D); memory order is C for
    row-major and Fortran for col-major."""
    order = "F" if iscolmajor else "C"
    x = np.asarray(x)
    xm = np.asarray(x, order=order)
    s = {"float32": "s", "float64": "d", "complex64": "c", "complex128": "z"}[x.dtype.name]
    win = np.ascontiguousarray(win, dtype=x.dtype)
    R, Cc, S, H = _dims4(x.shape)
    Y = np.zeros(x.shape, dtype=x.dtype, order=order)
    f = getattr(lib, f"{prefix}apply_win_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(xm), _ptr(win), _sz(R), _sz(Cc), _sz(S), _sz(H), _sz(win.size), _sz(dim), _int(iscolmajor))
    return rc, Y


def apply_win(x, win, dim, iscolmajor=False):
    return _apply_win_generic(orc(), "orc_", x, win, dim, iscolmajor)


def ref_apply_win(x, win, dim, iscolmajor=False):
    return _apply_win_generic(ref(), "", x, win, dim, iscolmajor)


def fft_r2hc(x, dim, nfft, iscolmajor=False, sc=False, exact=True):
    """Oracle for fft_fftw_r2hc_s/_d.  Returns (rc, complex Y) with Ly = nfft//2+1 along dim."""
    order = "F" if iscolmajor else "C"
    x = np.asarray(x)
    xm = np.asarray(x, order=order)
    s, _ = _sfx(x.dtype)
    R, Cc, S, H = _dims4(x.shape)
    oshape = list(_dims4(x.shape))
    oshape[dim] = nfft // 2 + 1
    oshape = tuple(oshape[: max(x.ndim, dim + 1)])
    cdt = np.complex64 if x.dtype == np.float32 else np.complex128
    Y = np.zeros(oshape, dtype=cdt, order=order)
    f = getattr(orc(), f"orc_fft_r2hc_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(xm), _sz(R), _sz(Cc), _sz(S), _sz(H), _int(iscolmajor), _sz(dim), _sz(nfft), _int(sc),
           _int(0 if exact else 1))
    return rc, Y


def fft(x, dim, nfft, iscolmajor=False, sc=False):
    """Oracle for fft_fftw_s/_d/_c/_z (reference c/fft.fftw.c:29-111,194-272): the published forward DFT
    X[k] = sum_n x[n] exp(-2 pi i k n / nfft) of every vector along dim, zero-padded to nfft, evaluated
    in double precision (the arithmetic itself is FFTW's, which is not in the reference tree); real
    input keeps bins 0..nfft/2, complex input all nfft; sc multiplies by 1/sqrt(2 nfft) (c/fft.fftw.c:104-108).
    The memory order of the tensor is numpy's business here: the result has the order asked for.
    Returns (rc, complex Y)."""
    x = np.asarray(x)
    if dim > 3:
        return 1, None
    shape4 = _dims4(x.shape)
    if nfft < shape4[dim]:
        return 1, None
    real = x.dtype in (np.float32, np.float64)
    cdt = np.complex64 if x.dtype in (np.float32, np.complex64) else np.complex128
    x4 = x.reshape(shape4).astype(np.complex128)
    Y = np.fft.fft(x4, n=nfft, axis=dim) if nfft > 0 else np.zeros(shape4, np.complex128)
    if real:
        Y = np.take(Y, np.arange(nfft // 2 + 1), axis=dim)
    if sc and nfft > 0:
        Y = Y * (1.0 / np.sqrt(2.0 * nfft))
    oshape = Y.shape[: max(x.ndim, dim + 1)]
    return 0, np.asarray(Y.reshape(oshape).astype(cdt), order="F" if iscolmajor else "C")


def ref_fft_rad2_c(xc):
    """The reference's own radix-2 complex FFT (c/fft.rad2.c, fft_rad2_c/_z) of each row of xc[V, nfft]
    (nfft a power of two, rows already of full length: the entry point's own zero-padding loop does not
    advance its pointer, c/fft.rad2.c:576)."""
    xc = np.ascontiguousarray(xc)
    V, n = xc.shape
    s = "c" if xc.dtype == np.complex64 else "z"
    Y = np.zeros((V, n), dtype=xc.dtype)
    f = getattr(ref(), f"fft_rad2_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(xc), _sz(V), _sz(n), _sz(1), _sz(1), _int(0), _sz(1), _sz(n), _int(0))
    return rc, Y


def ref_fft_rad2(frames, nfft):
    """The reference's own radix-2 FFT (c/fft.rad2.c) applied to each row of frames[W, L], zero-padded
    to nfft, returning the first nfft//2+1 bins (the layout of fft.fftw.r2hc).

    Goes through the complex entry point fft_rad2_c/_z with rows already padded to nfft: the
    real-input entry points fft_rad2_s/_d skip the bit-reversal swaps of index pairs that both lie
    above Nyquist (c/fft.rad2.c:159-163) and are wrong for nfft >= 16, and the complex entry
    point's own zero-padding loop does not advance its pointer (c/fft.rad2.c:576)."""
    frames = np.ascontiguousarray(frames)
    W, L = frames.shape
    cdt, s = (np.complex64, "c") if frames.dtype == np.float32 else (np.complex128, "z")
    xc = np.zeros((W, nfft), dtype=cdt)
    xc[:, :L] = frames
    Y = np.zeros((W, nfft), dtype=cdt)
    f = getattr(ref(), f"fft_rad2_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(xc), _sz(W), _sz(nfft), _sz(1), _sz(1), _int(0), _sz(1), _sz(nfft), _int(0))
    return rc, np.ascontiguousarray(Y[:, : nfft // 2 + 1])


def fft_power(xc):
    xc = np.ascontiguousarray(xc)
    s, rdt = _csfx(xc.dtype)
    Y = np.zeros(xc.shape, dtype=rdt)
    f = getattr(orc(), f"orc_fft_power_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(xc), _sz(xc.size))
    return rc, Y


def ref_fft_power(xc):
    xc = np.ascontiguousarray(xc)
    s, rdt = _csfx(xc.dtype)
    Y = np.zeros(xc.shape, dtype=rdt)
    f = getattr(ref(), f"fft_power_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(xc), _sz(xc.size))
    return rc, Y


def _wu_types(x):
    """suffix, real dtype and C scalar type of a real or complex signal"""
    return {"float32": ("s", np.float32, C.c_float), "float64": ("d", np.float64, C.c_double),
            "complex64": ("c", np.float32, C.c_float), "complex128": ("z", np.float64, C.c_double)}[x.dtype.name]


def ref_window_univar(x, win, stp, snip_edges=False):
    """reference c/window_univar.c: frame + window, no FFT (x real or complex, win real)."""
    x = np.ascontiguousarray(x)
    s, rdt, _ = _wu_types(x)
    win = np.ascontiguousarray(win, dtype=rdt)
    W = num_frames(x.size, win.size, stp, snip_edges)
    Y = np.zeros((W, win.size), dtype=x.dtype)
    f = getattr(ref(), f"window_univar_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(x), _ptr(win), _sz(x.size), _sz(win.size), _sz(stp), _int(snip_edges))
    return rc, Y


def ref_window_univar_flt(x, win, W, c0, stp):
    """reference c/window_univar_flt.c: frame + window with the _flt conventions (x real or complex)."""
    x = np.ascontiguousarray(x)
    s, rdt, ct = _wu_types(x)
    win = np.ascontiguousarray(win, dtype=rdt)
    Y = np.zeros((W, win.size), dtype=x.dtype)
    f = getattr(ref(), f"window_univar_flt_{s}")
    f.restype = _int
    rc = f(_ptr(Y), _ptr(x), _ptr(win), _sz(x.size), _sz(win.size), _sz(W), ct(c0), ct(stp))
    return rc, Y


def pow_compress(x, p, preg, use_ref=False):
    """Oracle for pow_compress_s/_d (reference c/pow_compress.c); use_ref runs the reference's own code."""
    x = np.ascontiguousarray(x)
    s, _ = _sfx(x.dtype)
    Y = np.zeros_like(x)
    f = getattr(ref(), f"pow_compress_{s}") if use_ref else getattr(orc(), f"orc_pow_compress_{s}")
    ft = C.c_float if s == "s" else C.c_double
    f.restype = _int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, ft, ft]
    rc = f(_ptr(Y), _ptr(x), x.size, float(p), float(preg))
    return rc, Y


def get_stft_freqs(nfft, F, fs, dtype, use_ref=False):
    """Oracle for get_stft_freqs_s/_d (reference c/get_stft_freqs.c)."""
    s, _ = _sfx(dtype)
    Y = np.zeros(F, dtype=dtype)
    f = getattr(ref(), f"get_stft_freqs_{s}") if use_ref else getattr(orc(), f"orc_get_stft_freqs_{s}")
    ft = C.c_float if s == "s" else C.c_double
    f.restype = _int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, ft]
    rc = f(_ptr(Y), nfft, F, float(fs))
    return rc, Y


def ref_window_cplx(name, L, dtype, norm=0, par=None, exact=0):
    """The complex forms of the reference's window generators (the _c/_z functions of c/hann.c ... c/planck.c): 2 L
    interleaved values, returned as a complex array; guard elements on each side must come back untouched."""
    rdt = np.float32 if np.dtype(dtype) == np.complex64 else np.float64
    s = "c" if rdt == np.float32 else "z"
    buf = np.full(2 * L + 4, 77.0, dtype=rdt)
    Y = buf[2:2 * L + 2]
    f = getattr(ref(), f"{name}_{s}")
    f.restype = _int
    if name in ("gauss", "tukey", "planck"):
        rc = f(_ptr(Y), _sz(L), (C.c_float if s == "c" else C.c_double)(par), _sz(norm))
    elif name == "blackman":
        rc = f(_ptr(Y), _sz(L), _int(exact), _sz(norm))
    else:
        rc = f(_ptr(Y), _sz(L), _sz(norm))
    assert rc == 0
    assert np.all(buf[:2] == 77.0) and np.all(buf[-2:] == 77.0), f"{name} L={L}: the reference wrote outside its 2 L values"
    return Y.copy().view(np.complex64 if rdt == np.float32 else np.complex128)


def ref_window(name, L, dtype, norm=0, par=None, exact=0):
    """reference window generators c/hann.c, c/hamming.c, c/povey.c, c/blackmanharris.c, c/blackman.c
    (exact), c/bartlett.c, c/triangular.c, c/rectangular.c, c/flattop.c, c/gauss.c, c/tukey.c, c/planck.c
    (par).  The output buffer carries a guard element on each side, which must come back untouched."""
    s, _ = _sfx(dtype)
    buf = np.full(L + 2, 77.0, dtype=dtype)
    Y = buf[1:L + 1]
    f = getattr(ref(), f"{name}_{s}")
    f.restype = _int
    if name in ("gauss", "tukey", "planck"):
        rc = f(_ptr(Y), _sz(L), (C.c_float if s == "s" else C.c_double)(par), _sz(norm))
    elif name == "blackman":
        rc = f(_ptr(Y), _sz(L), _int(exact), _sz(norm))
    else:
        rc = f(_ptr(Y), _sz(L), _sz(norm))
    assert rc == 0
    assert buf[0] == 77.0 and buf[-1] == 77.0, f"{name} L={L}: the reference wrote outside its L values"
    return Y.copy()


def ref_convert_freqs(frqs, in_scale, out_scale):
    """reference c/convert_freqs.c, on a copy."""
    y = np.array(frqs, copy=True)
    s, _ = _sfx(y.dtype)
    f = getattr(ref(), f"convert_freqs_{s}")
    f.restype = _int
    rc = f(_ptr(y), _sz(y.size), in_scale.encode(), out_scale.encode())
    return rc, y


def normalised_err(Y, Yref):
    """max over frames of max_f|Y - Yref| / max_f|Yref| (per-frame-normalised max error)."""
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, Y.shape[-1])
    Yref = np.asarray(Yref, dtype=np.float64).reshape(-1, Yref.shape[-1])
    if Y.size == 0:
        return 0.0
    den = np.max(np.abs(Yref), axis=1)
    num = np.max(np.abs(Y - Yref), axis=1)
    den = np.where(den > 0, den, 1.0)
    return float(np.max(num / den))
