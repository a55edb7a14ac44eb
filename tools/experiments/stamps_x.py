"""Per-phase cycle sums of the nfft 1024 / 2048 kernel (a build with -DR512X_STAMPS=1, DSP_AMD_LIB pointing at it)."""
import os, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
nfft, hop = int(sys.argv[1]), int(sys.argv[2])
B, N = 8, 1 << 25
x = torch.randn((B, N), device='cuda') * 0.1
win = torch.from_numpy(dsp_amd.hann(nfft)).cuda()
y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
for _ in range(40): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 40
buf = np.zeros(2048 * 8, dtype=np.uint64)
h = C.CDLL(os.environ['DSP_AMD_LIB'])
rc = h.dsp_debug_read_stamps_x(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
a = buf.reshape(2048, 8).astype(np.float64)
tot = a.sum(1)
fps = 4 // (nfft // 512)
steps = y.shape[0] * ((y.shape[1] + fps - 1) // fps) / 2048
names = ['wait for the span', 'input stage', 'dft16 + draw + span fetch issue', 'twiddle + exchange writes', 'pass-2 reads + dft16', 'split', 'staging', 'stores issued']
print('ms per launch %.4f' % ms, 'implied shader clock %.3f GHz' % (tot.mean() / (ms * 1e6)))
print('rc', rc, 'nfft', nfft, 'hop', hop, 'mean total cycles per wave', tot.mean(), 'steps/wave', steps)
for i, n in enumerate(names):
    print(f'{n:34s} {a[:, i].mean() / steps:8.0f} cyc/step  {100 * a[:, i].mean() / tot.mean():5.1f}%')
print('total per wave: min %.0f  p50 %.0f  max %.0f' % (tot.min(), np.percentile(tot, 50), tot.max()))
