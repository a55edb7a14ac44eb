import os, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
from dsp_amd._lib import lib
B, N = 1024, 160000
x = torch.randn((B, N), device='cuda') * 0.1
win = torch.from_numpy(dsp_amd.hann(400)).cuda()
y = dsp_amd.stft_dev(x, win, 160)
for _ in range(3): dsp_amd.stft_dev(x, win, 160, out=y)
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
h = C.CDLL(os.environ['DSP_AMD_LIB'])
rc = h.dsp_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
a = buf.reshape(8192, 8)[:4096].astype(np.float64)
tot = a.sum(1)
names = ['p1 read+fft', 'sync+exwrite+sync', 'p2 read+fft', 'sync+zwrite+sync', 'output+stores', 'sync+commit(wait loads)+sync', 'advance+issue loads', 'ring form: wait for the span']
print('rc', rc, 'mean total cycles per wave', tot.mean(), 'steps/wave', 256000 / 4096)
for i, n in enumerate(names):
    print(f'{n:32s} {a[:, i].mean() / (256000 / 4096):9.0f} cyc/step  {100 * a[:, i].mean() / tot.mean():5.1f}%')
nz = tot[tot > 0]
print('waves with data', nz.size, 'min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f mean %.0f' % (nz.min(), np.percentile(nz, 10), np.percentile(nz, 50), np.percentile(nz, 90), nz.max(), nz.mean()))
wg = nz.reshape(-1, 4).mean(1)
print('per-WG mean: min %.0f p50 %.0f max %.0f ; first 16:' % (wg.min(), np.percentile(wg, 50), wg.max()), [int(v) for v in wg[:16]])
print('hist of WG means:', np.histogram(wg, bins=8))
wgx = nz.reshape(-1, 4).mean(1)
print('per xcd mean/max:', [(int(wgx[x::8].mean()), int(wgx[x::8].max())) for x in range(8)])
print('by WG position in xcd range (first half / second half of WG index):', int(wgx[:256].mean()), int(wgx[256:].mean()))
print('std within WG %.0f' % nz.reshape(-1, 4).std(1).mean())
