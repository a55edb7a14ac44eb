"""Wall time of the reference's own host-pointer calls on one hour-long signal (one channel of configurations 3 and 4)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, dsp_amd
rng = np.random.default_rng(1)
def t(f, n=3):
    f()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best
x = (rng.standard_normal(57_600_000) * 0.1)
win = dsp_amd.povey(400, np.float64)
W = (x.size - 1 - 200) // 160
s = t(lambda: dsp_amd.stft_flt(x, win, W, 200.0, 160.0))
byts = x.nbytes + W * 257 * 8
print('stft_flt_d, 1 h @ 16 kHz float64: %.1f ms  %.1f GB/s of host traffic' % (s * 1e3, byts / s / 1e9))
x = (rng.standard_normal(172_800_000) * 0.1).astype(np.float32)
win = dsp_amd.blackmanharris(4096, np.float32)
s = t(lambda: dsp_amd.stft(x, win, 1024, nfft=4096))
Wn = dsp_amd.num_frames(x.size, 4096, 1024)
byts = x.nbytes + Wn * 2049 * 4
print('stft_s nfft 4096, 1 h @ 48 kHz float32: %.1f ms  %.1f GB/s of host traffic' % (s * 1e3, byts / s / 1e9))
x = x[:57_600_000]
win = dsp_amd.hann(400, np.float32)
s = t(lambda: dsp_amd.stft(x, win, 160))
Wn = dsp_amd.num_frames(x.size, 400, 160)
print('stft_s nfft 512, 1 h @ 16 kHz float32: %.1f ms  %.1f GB/s of host traffic' % (s * 1e3, (x.nbytes + Wn * 257 * 4) / s / 1e9))
# the same stft_flt_d call into an output array that is reused (no first-touch page faults in the copy out)
import ctypes as C
from dsp_amd._lib import lib
x = (rng.standard_normal(57_600_000) * 0.1)
win = dsp_amd.povey(400, np.float64)
W = (x.size - 1 - 200) // 160
Y = np.empty((W, 257), np.float64)
Y[:] = 0
p = lambda a: a.ctypes.data_as(C.c_void_p)
call = lambda: lib().stft_flt_d(p(Y), p(x), p(win), x.size, 400, W, 512, 200.0, 160.0, 0, 0, 0)
s = t(call)
print('stft_flt_d into a reused output array: %.1f ms  %.1f GB/s of host traffic' % (s * 1e3, (x.nbytes + Y.nbytes) / s / 1e9))
