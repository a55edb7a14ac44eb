"""Host-side mirror of the reference's operator interface for the STFT path.

Every function here is a thin ctypes call into libdsp_stft_hip.so (include/dsp_stft.h); names,
argument meaning and error behaviour follow the reference's C entry points (c/stft.c:29-30,
c/stft_flt.c:29-30, c/frame_univar.c:35-38, c/frame_univar_flt.c:23-26, c/apply_win.c:15-23,
c/fft.fftw.r2hc.c:20-21, c/fft_power.c:13-14).  A non-zero return code raises DspError; the
library has already written the reference-style "error in <fn>: ..." line to stderr.

The numpy functions use the host-pointer entry points (copy in, compute on the GPU, copy out).
The *_dev functions take CUDA/HIP torch tensors and enqueue on torch's current stream.
"""
import ctypes as C

import numpy as np

from ._lib import lib


class DspError(RuntimeError):
    """A libdsp_stft_hip entry point returned 1."""


def _p(a):
    return C.c_void_p(a.ctypes.data)


def nextpow2(L):
    """nfft as the reference's command-line tools derive it (srci/stft.cpp:101-102)."""
    n = 1
    while n < L:
        n *= 2
    return n


def num_frames(N, L, stp, snip_edges=False):
    """W of stft_s/stft_d (reference c/stft.c:40)."""
    return int(lib().dsp_stft_num_frames(N, L, stp, int(bool(snip_edges))))


def _real_sfx(dtype):
    if dtype == np.float32:
        return "s"
    if dtype == np.float64:
        return "d"
    raise TypeError(f"unsupported dtype {dtype}: the reference's data types are float32 and float64")


def _any_sfx(dtype):
    return {"float32": "s", "float64": "d", "complex64": "c", "complex128": "z"}[np.dtype(dtype).name]


def _check(rc, name):
    if rc != 0:
        raise DspError(f"{name} failed (see stderr)")


def _check_out(out, shape, like, what):
    """A caller-supplied device output must be exactly what the kernel writes: a wrong shape, dtype, device or
    layout would make it write past the tensor."""
    if tuple(out.shape) != tuple(shape) or not out.is_contiguous() or out.dtype != like.dtype or out.device != like.device:
        raise ValueError(f"out must be a contiguous {what} tensor of x's dtype on x's device")


def stft(x, win, stp=160, snip_edges=False, mn0=False, amp=False, lg=False, nfft=None):
    """stft_s / stft_d.  x: 1-D signal, win: 1-D window; returns Y[W, nfft//2+1]."""
    x = np.ascontiguousarray(x)
    s = _real_sfx(x.dtype)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    N, L = x.size, win.size
    nfft = nextpow2(L) if nfft is None else int(nfft)
    W = num_frames(N, L, stp, snip_edges)
    Y = np.empty((W, nfft // 2 + 1), dtype=x.dtype)
    rc = getattr(lib(), f"stft_{s}")(_p(Y), _p(x), _p(win), N, L, nfft, stp, int(bool(snip_edges)), int(bool(mn0)),
                                     int(bool(amp)), int(bool(lg)))
    _check(rc, f"stft_{s}")
    return Y


def stft_batch(x, win, stp=160, snip_edges=False, mn0=False, amp=False, lg=False, nfft=None):
    """dsp_stft_batch_s / _d.  x: [B, N] (rows may be strided); returns Y[B, W, F]."""
    x = np.asarray(x)
    if x.ndim != 2 or x.strides[1] != x.itemsize or (x.shape[0] > 1 and x.strides[0] % x.itemsize):
        x = np.ascontiguousarray(x)
    s = _real_sfx(x.dtype)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    B, N = x.shape
    ldx = x.strides[0] // x.itemsize if B > 1 else N
    L = win.size
    nfft = nextpow2(L) if nfft is None else int(nfft)
    W = num_frames(N, L, stp, snip_edges)
    Y = np.empty((B, W, nfft // 2 + 1), dtype=x.dtype)
    rc = getattr(lib(), f"dsp_stft_batch_{s}")(_p(Y), _p(x), _p(win), B, N, ldx, L, nfft, stp, int(bool(snip_edges)),
                                               int(bool(mn0)), int(bool(amp)), int(bool(lg)))
    _check(rc, f"dsp_stft_batch_{s}")
    return Y


def stft_flt_batch(x, win, W, c0=0.0, stp=160.0, mn0=False, amp=False, lg=False, nfft=None):
    """dsp_stft_flt_batch_s / _d.  x: [B, N] (rows may be strided); returns Y[B, W, F]."""
    x = np.asarray(x)
    if x.ndim != 2 or x.strides[1] != x.itemsize or (x.shape[0] > 1 and x.strides[0] % x.itemsize):
        x = np.ascontiguousarray(x)
    s = _real_sfx(x.dtype)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    B, N = x.shape
    ldx = x.strides[0] // x.itemsize if B > 1 else N
    L = win.size
    nfft = nextpow2(L) if nfft is None else int(nfft)
    Y = np.empty((B, W, nfft // 2 + 1), dtype=x.dtype)
    rc = getattr(lib(), f"dsp_stft_flt_batch_{s}")(_p(Y), _p(x), _p(win), B, N, ldx, L, W, nfft, float(c0), float(stp),
                                                   int(bool(mn0)), int(bool(amp)), int(bool(lg)))
    _check(rc, f"dsp_stft_flt_batch_{s}")
    return Y


def stft_flt_multi(x, win, W, c0=0.0, stp=160.0, mn0=False, amp=False, lg=False, nfft=None, devices=(0,)):
    """dsp_stft_flt_multi_s / _d: the host batch x[B, N] split over the GPUs in `devices`, one thread per GPU."""
    x = np.ascontiguousarray(x)
    s = _real_sfx(x.dtype)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    B, N = x.shape
    L = win.size
    nfft = nextpow2(L) if nfft is None else int(nfft)
    Y = np.empty((B, W, nfft // 2 + 1), dtype=x.dtype)
    devs = (C.c_int * len(devices))(*devices)
    rc = getattr(lib(), f"dsp_stft_flt_multi_{s}")(_p(Y), _p(x), _p(win), B, N, N, L, W, nfft, float(c0), float(stp), int(bool(mn0)),
                                                   int(bool(amp)), int(bool(lg)), C.cast(devs, C.c_void_p), len(devices))
    _check(rc, f"dsp_stft_flt_multi_{s}")
    return Y


def shard_range(B, rank, world):
    """dsp_shard_range: the block [lo, hi) of B signals that GPU `rank` of `world` takes."""
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    lib().dsp_shard_range(B, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def stft_multi(x, win, stp=160, snip_edges=False, mn0=False, amp=False, lg=False, nfft=None, devices=(0,)):
    """dsp_stft_multi_s / _d: the host batch x[B, N] split over the GPUs in `devices`, one thread per GPU."""
    x = np.ascontiguousarray(x)
    s = _real_sfx(x.dtype)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    B, N = x.shape
    L = win.size
    nfft = nextpow2(L) if nfft is None else int(nfft)
    W = num_frames(N, L, stp, snip_edges)
    Y = np.empty((B, W, nfft // 2 + 1), dtype=x.dtype)
    devs = (C.c_int * len(devices))(*devices)
    rc = getattr(lib(), f"dsp_stft_multi_{s}")(_p(Y), _p(x), _p(win), B, N, N, L, nfft, stp, int(bool(snip_edges)), int(bool(mn0)),
                                               int(bool(amp)), int(bool(lg)), C.cast(devs, C.c_void_p), len(devices))
    _check(rc, f"dsp_stft_multi_{s}")
    return Y


def stft_multi_dev(x_blocks, wins, stp=160, snip_edges=False, mn0=False, amp=False, lg=False, nfft=None, devices=None, gather=True):
    """dsp_stft_multi_dev_s / _d.  x_blocks[g]: GPU g's block of the batch (a dense device tensor [hi-lo, N] on
    devices[g], blocks as shard_range gives them), wins[g] the window on that GPU.  Returns (list of output blocks,
    Y_all on devices[0] or None)."""
    import torch
    ngpu = len(x_blocks)
    devices = list(range(ngpu)) if devices is None else list(devices)
    B = sum(int(b.shape[0]) for b in x_blocks)
    N = int(x_blocks[0].shape[1])
    s = _torch_sfx(x_blocks[0])
    L = wins[0].numel()
    nfft = nextpow2(L) if nfft is None else int(nfft)
    W = num_frames(N, L, stp, snip_edges)
    F = nfft // 2 + 1
    for g, xb in enumerate(x_blocks):
        lo, hi = shard_range(B, g, ngpu)
        if tuple(xb.shape) != (hi - lo, N) or not xb.is_contiguous():
            raise ValueError(f"block {g} must be a dense [{hi - lo}, {N}] tensor")
    ys = [torch.empty((xb.shape[0], W, F), dtype=xb.dtype, device=xb.device) for xb in x_blocks]
    y_all = torch.empty((B, W, F), dtype=x_blocks[0].dtype, device=x_blocks[0].device) if gather else None
    vp = C.c_void_p
    yp = (vp * ngpu)(*[y.data_ptr() for y in ys])
    xp = (vp * ngpu)(*[xb.data_ptr() for xb in x_blocks])
    wp = (vp * ngpu)(*[w.data_ptr() for w in wins])
    devs = (C.c_int * ngpu)(*devices)
    # the entry point launches on streams of its own on every GPU of the list, which are not ordered with torch's:
    # whatever still produces the blocks and the windows there (copies, earlier kernels) has to be finished first
    for dv in sorted(set(devices)):
        torch.cuda.synchronize(dv)
    rc = getattr(lib(), f"dsp_stft_multi_dev_{s}")(C.cast(yp, vp), C.cast(xp, vp), C.cast(wp, vp), vp(y_all.data_ptr() if gather else 0), B, N, N, L,
                                                   nfft, stp, int(bool(snip_edges)), int(bool(mn0)), int(bool(amp)), int(bool(lg)),
                                                   C.cast(devs, vp), ngpu)
    _check(rc, f"dsp_stft_multi_dev_{s}")
    return ys, y_all


def stft_flt(x, win, W, c0=0.0, stp=160.0, mn0=False, amp=False, lg=False, nfft=None):
    """stft_flt_s / stft_flt_d.  Returns Y[W, nfft//2+1]."""
    x = np.ascontiguousarray(x)
    s = _real_sfx(x.dtype)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    N, L = x.size, win.size
    nfft = nextpow2(L) if nfft is None else int(nfft)
    Y = np.empty((W, nfft // 2 + 1), dtype=x.dtype)
    rc = getattr(lib(), f"stft_flt_{s}")(_p(Y), _p(x), _p(win), N, L, W, nfft, c0, stp, int(bool(mn0)), int(bool(amp)),
                                         int(bool(lg)))
    _check(rc, f"stft_flt_{s}")
    return Y


def set_flt_mode(intended):
    """False (default): stft_flt frames as the reference's code computes them; True: as documented."""
    lib().dsp_stft_flt_set_mode(int(bool(intended)))


def force_generic(on):
    lib().dsp_stft_force_generic(int(bool(on)))


def kernel_family(nfft, dtype):
    return int(lib().dsp_stft_kernel_family(nfft, np.dtype(dtype).itemsize))


def frame_univar(x, L, stp, snip_edges=False):
    """frame_univar_{s,d,c,z}.  Returns Y[W, L]."""
    x = np.ascontiguousarray(x)
    s = _any_sfx(x.dtype)
    W = num_frames(x.size, L, stp, snip_edges)
    Y = np.empty((W, L), dtype=x.dtype)
    rc = getattr(lib(), f"frame_univar_{s}")(_p(Y), _p(x), x.size, L, stp, int(bool(snip_edges)))
    _check(rc, f"frame_univar_{s}")
    return Y


def frame_univar_flt(x, L, W, c0, stp):
    """frame_univar_flt_{s,d,c,z}.  Returns Y[W, L]."""
    x = np.ascontiguousarray(x)
    s = _any_sfx(x.dtype)
    Y = np.empty((W, L), dtype=x.dtype)
    rc = getattr(lib(), f"frame_univar_flt_{s}")(_p(Y), _p(x), x.size, L, W, c0, stp)
    _check(rc, f"frame_univar_flt_{s}")
    return Y


def _dims4(shape):
    if len(shape) > 4:
        raise ValueError("tensors have at most 4 dimensions (R, C, S, H)")
    return tuple(shape) + (1,) * (4 - len(shape))


def apply_win(x, win, dim, iscolmajor=False, inplace=False):
    """apply_win_{s,d,c,z} / apply_win_inplace_*: x .* win with win broadcast along dim.  The logical
    shape is x.shape; memory order C for row-major, Fortran for column-major."""
    order = "F" if iscolmajor else "C"
    x = np.asarray(x)
    s = _any_sfx(x.dtype)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    R, Cc, S, H = _dims4(x.shape)
    if inplace:
        if not (x.flags.f_contiguous if iscolmajor else x.flags.c_contiguous):
            raise ValueError("in-place apply_win needs a contiguous array in the stated major order")
        rc = getattr(lib(), f"apply_win_inplace_{s}")(_p(x), _p(win), R, Cc, S, H, win.size, dim, int(bool(iscolmajor)))
        _check(rc, f"apply_win_inplace_{s}")
        return x
    xm = np.asarray(x, order=order)
    Y = np.empty(x.shape, dtype=x.dtype, order=order)
    rc = getattr(lib(), f"apply_win_{s}")(_p(Y), _p(xm), _p(win), R, Cc, S, H, win.size, dim, int(bool(iscolmajor)))
    _check(rc, f"apply_win_{s}")
    return Y


def fft_r2hc(x, dim, nfft, iscolmajor=False, sc=False):
    """fft_fftw_r2hc_s / _d: real FFT along dim, nfft//2+1 complex outputs along dim."""
    order = "F" if iscolmajor else "C"
    x = np.asarray(x)
    s = _real_sfx(x.dtype)
    xm = np.asarray(x, order=order)
    R, Cc, S, H = _dims4(x.shape)
    oshape = list(_dims4(x.shape))
    oshape[dim] = nfft // 2 + 1
    oshape = tuple(oshape[: max(x.ndim, dim + 1)])
    Y = np.empty(oshape, dtype=np.complex64 if x.dtype == np.float32 else np.complex128, order=order)
    rc = getattr(lib(), f"fft_fftw_r2hc_{s}")(_p(Y), _p(xm), R, Cc, S, H, int(bool(iscolmajor)), dim, nfft, int(bool(sc)))
    _check(rc, f"fft_fftw_r2hc_{s}")
    return Y


def fft(x, dim, nfft, iscolmajor=False, sc=False):
    """fft_fftw_s / _d / _c / _z: FFT along dim; nfft//2+1 complex outputs for real x, nfft for complex x."""
    order = "F" if iscolmajor else "C"
    x = np.asarray(x)
    s = {"float32": "s", "float64": "d", "complex64": "c", "complex128": "z"}[x.dtype.name]
    xm = np.asarray(x, order=order)
    R, Cc, S, H = _dims4(x.shape)
    oshape = list(_dims4(x.shape))
    oshape[dim] = nfft // 2 + 1 if s in "sd" else nfft
    oshape = tuple(oshape[: max(x.ndim, dim + 1)])
    Y = np.empty(oshape, dtype=np.complex64 if s in "sc" else np.complex128, order=order)
    rc = getattr(lib(), f"fft_fftw_{s}")(_p(Y), _p(xm), R, Cc, S, H, int(bool(iscolmajor)), dim, nfft, int(bool(sc)))
    _check(rc, f"fft_fftw_{s}")
    return Y


def fft_power(xc):
    """fft_power_c / _z: |x|^2 element-wise over a complex array."""
    xc = np.ascontiguousarray(xc)
    s = {"complex64": "c", "complex128": "z"}[xc.dtype.name]
    Y = np.empty(xc.shape, dtype=np.float32 if s == "c" else np.float64)
    rc = getattr(lib(), f"fft_power_{s}")(_p(Y), _p(xc), xc.size)
    _check(rc, f"fft_power_{s}")
    return Y


_ANY_SFX = {"float32": "s", "float64": "d", "complex64": "c", "complex128": "z"}
_REAL_OF = {"s": np.float32, "d": np.float64, "c": np.float32, "z": np.float64}


def window_univar(x, win, stp, snip_edges=False):
    """window_univar_s / _d / _c / _z: frames of x (real or complex), each multiplied by the real window
    win; returns Y[W, L]."""
    x = np.ascontiguousarray(x)
    s = _ANY_SFX[x.dtype.name]
    win = np.ascontiguousarray(win, dtype=_REAL_OF[s])
    W = num_frames(x.size, win.size, stp, snip_edges)
    Y = np.empty((W, win.size), dtype=x.dtype)
    rc = getattr(lib(), f"window_univar_{s}")(_p(Y), _p(x), _p(win), x.size, win.size, stp, int(bool(snip_edges)))
    _check(rc, f"window_univar_{s}")
    return Y


def window_univar_flt(x, win, W, c0, stp):
    """window_univar_flt_s / _d / _c / _z: W zero-edged frames centred at round(c0 + w*stp), each
    multiplied by the real window win."""
    x = np.ascontiguousarray(x)
    s = _ANY_SFX[x.dtype.name]
    win = np.ascontiguousarray(win, dtype=_REAL_OF[s])
    Y = np.empty((W, win.size), dtype=x.dtype)
    rc = getattr(lib(), f"window_univar_flt_{s}")(_p(Y), _p(x), _p(win), x.size, win.size, W, float(c0), float(stp))
    _check(rc, f"window_univar_flt_{s}")
    return Y


WINDOW_IDS = {"hann": 0, "hamming": 1, "povey": 2, "blackmanharris": 3, "blackman": 4, "bartlett": 5, "triangular": 6,
              "rectangular": 7, "flattop": 8, "gauss": 9, "tukey": 10, "planck": 11}
WINDOW_PAR = {"gauss": 2.5, "tukey": 0.5, "planck": 0.1}        # generators with a parameter, and its default


def window(name, L, norm=0, dtype=np.float32, par=None, exact=False):
    """hann_s/_d/_c/_z ... planck_s/_d/_c/_z: the window computed by the library.  par: gauss a, tukey r, planck
    epsilon; exact: blackman's exact coefficients."""
    if name not in WINDOW_IDS:
        raise ValueError(f"unknown window {name}")
    s = _any_sfx(np.dtype(dtype))               # complex dtypes: the _c/_z forms (zero imaginary parts)
    Y = np.empty(L, dtype=dtype)
    fn = getattr(lib(), f"{name}_{s}")
    if name in WINDOW_PAR:
        rc = fn(_p(Y), L, float(WINDOW_PAR[name] if par is None else par), norm)
    elif name == "blackman":
        rc = fn(_p(Y), L, int(bool(exact)), norm)
    else:
        rc = fn(_p(Y), L, norm)
    _check(rc, f"{name}_{s}")
    return Y


def window_dev(name, L, norm=0, dtype=None, device="cuda", par=None, exact=False):
    """dsp_window_par_dev_s/_d: the window as a device tensor, enqueued on torch's current stream."""
    import torch
    dtype = torch.float32 if dtype is None else dtype
    out = torch.empty(L, dtype=dtype, device=device)
    s = _torch_sfx(out)
    par = WINDOW_PAR.get(name, 0.0) if par is None else par
    with torch.cuda.device(out.device):
        rc = getattr(lib(), f"dsp_window_par_dev_{s}")(C.c_void_p(out.data_ptr()), WINDOW_IDS[name], L, float(par), int(bool(exact)), norm, _stream_ptr(out))
    _check(rc, f"dsp_window_par_dev_{s}")
    return out


def convert_freqs(frqs, in_scale, out_scale):
    """convert_freqs_s / _d: a copy of the frequency axis on another scale ("hz", "sqrt", "cbrt", "octave")."""
    y = np.array(frqs, copy=True)
    s = _real_sfx(y.dtype)
    rc = getattr(lib(), f"convert_freqs_{s}")(_p(y), y.size, in_scale.encode(), out_scale.encode())
    _check(rc, f"convert_freqs_{s}")
    return y


def pow_compress(x, p, preg=0.0, inplace=False):
    """pow_compress_s / _d (and _inplace_): f(x + preg) element-wise, f chosen by the exponent p in [0, 1]."""
    x = np.ascontiguousarray(x)
    s = _real_sfx(x.dtype)
    if inplace:
        rc = getattr(lib(), f"pow_compress_inplace_{s}")(_p(x), x.size, float(p), float(preg))
        _check(rc, f"pow_compress_inplace_{s}")
        return x
    Y = np.empty_like(x)
    rc = getattr(lib(), f"pow_compress_{s}")(_p(Y), _p(x), x.size, float(p), float(preg))
    _check(rc, f"pow_compress_{s}")
    return Y


def get_stft_freqs(nfft, F, fs, dtype=np.float32):
    """get_stft_freqs_s / _d: the first F bin frequencies of an nfft-point STFT at sample rate fs."""
    s = _real_sfx(np.dtype(dtype))
    Y = np.empty(F, dtype=dtype)
    rc = getattr(lib(), f"get_stft_freqs_{s}")(_p(Y), nfft, F, float(fs))
    _check(rc, f"get_stft_freqs_{s}")
    return Y


def stft_pc_batch(x, win, p, preg=0.0, stp=160, snip_edges=False, mn0=False, amp=False, lg=False, nfft=None):
    """dsp_stft_pc_batch_s / _d: stft_batch with pow_compress(., p, preg) fused into the store.  x: [B, N]."""
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[None, :]
    if x.ndim != 2 or x.strides[1] != x.itemsize or (x.shape[0] > 1 and x.strides[0] % x.itemsize):
        x = np.ascontiguousarray(x)
    s = _real_sfx(x.dtype)
    win = np.ascontiguousarray(win, dtype=x.dtype)
    B, N = x.shape
    ldx = x.strides[0] // x.itemsize if B > 1 else N
    L = win.size
    nfft = nextpow2(L) if nfft is None else int(nfft)
    W = num_frames(N, L, stp, snip_edges)
    Y = np.empty((B, W, nfft // 2 + 1), dtype=x.dtype)
    rc = getattr(lib(), f"dsp_stft_pc_batch_{s}")(_p(Y), _p(x), _p(win), B, N, ldx, L, nfft, stp, int(bool(snip_edges)),
                                                  int(bool(mn0)), int(bool(amp)), int(bool(lg)), float(p), float(preg))
    _check(rc, f"dsp_stft_pc_batch_{s}")
    return Y


# ------------------------------------------------------------------------------------------------
# device-resident forms (torch tensors are only the carrier of device memory and of the stream)
# ------------------------------------------------------------------------------------------------
def _torch_sfx(t):
    import torch
    if t.dtype == torch.float32:
        return "s"
    if t.dtype == torch.float64:
        return "d"
    raise TypeError(f"unsupported dtype {t.dtype}")


def _stream_ptr(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def stft_dev(x, win, stp=160, snip_edges=False, mn0=False, amp=False, lg=False, nfft=None, out=None):
    """dsp_stft_batch_dev_s / _d.  x: device tensor [B, N], win: device tensor [L]; returns the device tensor
    Y[B, W, F].  Enqueued on torch's current stream, not synchronised.  A view with a sample stride (such as
    t.T of a row-major [N, C] tensor of interleaved channels) goes through dsp_stft_batch_dev_strided_s / _d."""
    import torch
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if not x.is_cuda or x.dim() != 2 or x.stride(1) < 1 or x.stride(0) < 0:
        raise ValueError("x must be a device tensor [B, N] with positive strides")
    s = _torch_sfx(x)
    win = win.to(device=x.device, dtype=x.dtype).contiguous()
    B, N = x.shape
    ldx = x.stride(0) if B > 1 else N
    if x.stride(1) != 1:
        L = win.numel()
        nfft = nextpow2(L) if nfft is None else int(nfft)
        W = num_frames(N, L, stp, snip_edges)
        if out is None:
            out = torch.empty((B, W, nfft // 2 + 1), dtype=x.dtype, device=x.device)
        else:
            _check_out(out, (B, W, nfft // 2 + 1), x, "[B, W, F]")
        with torch.cuda.device(x.device):
            rc = getattr(lib(), f"dsp_stft_batch_dev_strided_{s}")(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()),
                                                                   C.c_void_p(win.data_ptr()), B, N, x.stride(0) if B > 1 else 0,
                                                                   x.stride(1), L, nfft, stp, int(bool(snip_edges)), int(bool(mn0)),
                                                                   int(bool(amp)), int(bool(lg)), _stream_ptr(x))
        _check(rc, f"dsp_stft_batch_dev_strided_{s}")
        return out
    L = win.numel()
    nfft = nextpow2(L) if nfft is None else int(nfft)
    W = num_frames(N, L, stp, snip_edges)
    if out is None:
        out = torch.empty((B, W, nfft // 2 + 1), dtype=x.dtype, device=x.device)
    else:
        _check_out(out, (B, W, nfft // 2 + 1), x, "[B, W, F]")
    with torch.cuda.device(x.device):
        rc = getattr(lib(), f"dsp_stft_batch_dev_{s}")(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()),
                                                       C.c_void_p(win.data_ptr()), B, N, ldx, L, nfft, stp,
                                                       int(bool(snip_edges)), int(bool(mn0)), int(bool(amp)),
                                                       int(bool(lg)), _stream_ptr(x))
    _check(rc, f"dsp_stft_batch_dev_{s}")
    return out


def stft_cplx_dev(x, win, stp=160, snip_edges=False, sc=False, nfft=None, out=None):
    """dsp_stft_cplx_batch_dev_s / _d: the complex spectrogram Y[B, W, F] (torch complex64 / complex128) of the device
    tensor x[B, N] (or x[N]), enqueued on the current stream."""
    import torch
    single = x.dim() == 1
    if single:
        x = x[None, :]
    if not x.is_cuda or x.dim() != 2:
        raise ValueError("x must be a device tensor [B, N] or [N]")
    if x.stride(1) != 1:
        x = x.contiguous()
    s = _torch_sfx(x)
    win = win.to(device=x.device, dtype=x.dtype).contiguous()
    B, N = x.shape
    L = win.numel()
    nfft = nextpow2(L) if nfft is None else int(nfft)
    W = num_frames(N, L, stp, snip_edges)
    cdt = torch.complex64 if x.dtype == torch.float32 else torch.complex128
    if out is None:
        out = torch.empty((B, W, nfft // 2 + 1), dtype=cdt, device=x.device)
    else:
        _check_out(out, (B, W, nfft // 2 + 1), out, "stft_cplx_dev")
        if out.dtype != cdt or out.device != x.device:
            raise ValueError("stft_cplx_dev: out must be a complex tensor of x's precision on x's device")
    ldx = x.stride(0) if B > 1 else N
    with torch.cuda.device(x.device):                     # the library takes its tables from, and launches on, the current device
        rc = getattr(lib(), f"dsp_stft_cplx_batch_dev_{s}")(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(win.data_ptr()),
                                                            B, N, ldx, L, nfft, stp, int(bool(snip_edges)), int(bool(sc)), _stream_ptr(x))
    _check(rc, f"dsp_stft_cplx_batch_dev_{s}")
    return out[0] if single else out


def stft_pc_dev(x, win, p, preg=0.0, stp=160, snip_edges=False, mn0=False, amp=False, lg=False, nfft=None, out=None):
    """dsp_stft_pc_batch_dev_s / _d: stft_dev with pow_compress(., p, preg) fused into the store."""
    import torch
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be a device tensor [B, N] with unit stride along N")
    s = _torch_sfx(x)
    win = win.to(device=x.device, dtype=x.dtype).contiguous()
    B, N = x.shape
    ldx = x.stride(0) if B > 1 else N
    L = win.numel()
    nfft = nextpow2(L) if nfft is None else int(nfft)
    W = num_frames(N, L, stp, snip_edges)
    if out is None:
        out = torch.empty((B, W, nfft // 2 + 1), dtype=x.dtype, device=x.device)
    else:
        _check_out(out, (B, W, nfft // 2 + 1), x, "[B, W, F]")
    with torch.cuda.device(x.device):
        rc = getattr(lib(), f"dsp_stft_pc_batch_dev_{s}")(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()),
                                                          C.c_void_p(win.data_ptr()), B, N, ldx, L, nfft, stp,
                                                          int(bool(snip_edges)), int(bool(mn0)), int(bool(amp)),
                                                          int(bool(lg)), float(p), float(preg), _stream_ptr(x))
    _check(rc, f"dsp_stft_pc_batch_dev_{s}")
    return out


def pow_compress_dev(x, p, preg=0.0, out=None):
    """dsp_pow_compress_dev_s / _d over a contiguous device tensor (out may be x)."""
    import torch
    if not x.is_cuda or not x.is_contiguous():
        raise ValueError("x must be a contiguous device tensor")
    s = _torch_sfx(x)
    if out is None:
        out = torch.empty_like(x)
    else:
        _check_out(out, x.shape, x, "x-shaped")
    with torch.cuda.device(x.device):
        rc = getattr(lib(), f"dsp_pow_compress_dev_{s}")(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), x.numel(),
                                                         float(p), float(preg), _stream_ptr(x))
    _check(rc, f"dsp_pow_compress_dev_{s}")
    return out


def stft_flt_dev(x, win, W, c0=0.0, stp=160.0, mn0=False, amp=False, lg=False, nfft=None, out=None):
    """dsp_stft_flt_batch_dev_s / _d.  x: device tensor [B, N]; returns Y[B, W, F].  A view with a sample stride goes
    through dsp_stft_flt_batch_dev_strided_s / _d."""
    import torch
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if not x.is_cuda or x.dim() != 2 or x.stride(1) < 1 or x.stride(0) < 0:
        raise ValueError("x must be a device tensor [B, N] with positive strides")
    s = _torch_sfx(x)
    win = win.to(device=x.device, dtype=x.dtype).contiguous()
    B, N = x.shape
    ldx = x.stride(0) if B > 1 else N
    L = win.numel()
    nfft = nextpow2(L) if nfft is None else int(nfft)
    if out is None:
        out = torch.empty((B, W, nfft // 2 + 1), dtype=x.dtype, device=x.device)
    else:
        _check_out(out, (B, W, nfft // 2 + 1), x, "[B, W, F]")
    if x.stride(1) != 1:
        with torch.cuda.device(x.device):
            rc = getattr(lib(), f"dsp_stft_flt_batch_dev_strided_{s}")(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()),
                                                                       C.c_void_p(win.data_ptr()), B, N, x.stride(0) if B > 1 else 0,
                                                                       x.stride(1), L, W, nfft, c0, stp, int(bool(mn0)), int(bool(amp)),
                                                                       int(bool(lg)), _stream_ptr(x))
        _check(rc, f"dsp_stft_flt_batch_dev_strided_{s}")
        return out
    with torch.cuda.device(x.device):
        rc = getattr(lib(), f"dsp_stft_flt_batch_dev_{s}")(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()),
                                                           C.c_void_p(win.data_ptr()), B, N, ldx, L, W, nfft, c0, stp,
                                                           int(bool(mn0)), int(bool(amp)), int(bool(lg)), _stream_ptr(x))
    _check(rc, f"dsp_stft_flt_batch_dev_{s}")
    return out


def fft_power_dev(xc, out=None):
    """dsp_fft_power_dev_c / _z on a complex device tensor."""
    import torch
    xc = xc.contiguous()
    s = {torch.complex64: "c", torch.complex128: "z"}[xc.dtype]
    rdt = torch.float32 if s == "c" else torch.float64
    if out is None:
        out = torch.empty(xc.shape, dtype=rdt, device=xc.device)
    elif tuple(out.shape) != tuple(xc.shape) or not out.is_contiguous() or out.dtype != rdt or out.device != xc.device:
        raise ValueError("out must be a contiguous real tensor of xc's shape and precision on xc's device")
    with torch.cuda.device(xc.device):
        rc = getattr(lib(), f"dsp_fft_power_dev_{s}")(C.c_void_p(out.data_ptr()), C.c_void_p(xc.data_ptr()), xc.numel(),
                                                      _stream_ptr(xc))
    _check(rc, f"dsp_fft_power_dev_{s}")
    return out
