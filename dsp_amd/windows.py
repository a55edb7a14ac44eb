"""Window generators used to synthesise inputs: host-side restatements of the reference formulas
(symmetric, period L-1, first half computed then mirrored, centre forced to 1 for odd L;
reference c/hann.c:25-29, c/hamming.c, c/povey.c:25-29, c/blackmanharris.c:26-30).  norm = 0 only."""
import numpy as np


def _mirror(first_half, L, dtype):
    y = np.empty(L, dtype=dtype)
    h = L // 2
    y[:h] = first_half
    if L % 2:
        y[h] = 1.0
    y[L - h:] = first_half[::-1]
    return y


def _phase(L, dtype):
    h = L // 2
    if dtype == np.float32:
        p = np.float32(2.0 * np.pi / np.float64(max(L - 1, 1))) if L > 1 else np.float32(np.inf)
        return (np.arange(h, dtype=np.float32) * p).astype(np.float32), p
    p = 2.0 * np.pi / np.float64(L - 1) if L > 1 else np.inf
    return np.arange(h, dtype=np.float64) * p, p


def hann(L, dtype=np.float32):
    dtype = np.dtype(dtype).type
    a, _ = _phase(L, dtype)
    return _mirror((dtype(0.5) - dtype(0.5) * np.cos(a)).astype(dtype), L, dtype)


def hamming(L, dtype=np.float32):
    dtype = np.dtype(dtype).type
    a, _ = _phase(L, dtype)
    return _mirror((dtype(0.54) - dtype(0.46) * np.cos(a)).astype(dtype), L, dtype)


def povey(L, dtype=np.float32):
    dtype = np.dtype(dtype).type
    a, _ = _phase(L, dtype)
    return _mirror(np.power((dtype(0.5) - dtype(0.5) * np.cos(a)).astype(dtype), dtype(0.85)).astype(dtype), L, dtype)


def blackmanharris(L, dtype=np.float32):
    dtype = np.dtype(dtype).type
    h = L // 2
    _, p = _phase(L, dtype)
    l = np.arange(h)
    one = l.astype(dtype) * p
    two = (2 * l).astype(dtype) * p
    three = (3 * l).astype(dtype) * p
    y = (dtype(0.35875) - dtype(0.48829) * np.cos(one) + dtype(0.14128) * np.cos(two)
         - dtype(0.01168) * np.cos(three)).astype(dtype)
    return _mirror(y, L, dtype)


WINDOWS = {"hann": hann, "hamming": hamming, "povey": povey, "blackmanharris": blackmanharris}
