import os, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
B, N = 8, 172800000 // 4
x = torch.randn((B, N), device='cuda') * 0.1
win = torch.from_numpy(dsp_amd.blackmanharris(4096)).cuda()
y = dsp_amd.stft_dev(x, win, 1024, nfft=4096)
for _ in range(60): dsp_amd.stft_dev(x, win, 1024, nfft=4096, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): dsp_amd.stft_dev(x, win, 1024, nfft=4096, out=y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 40
buf = np.zeros(2048 * 10, dtype=np.uint64)
h = C.CDLL(os.environ['DSP_AMD_LIB'])
rc = h.dsp_debug_read_stamps4096(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
full = buf.reshape(2048, 10).astype(np.float64)
a = full[:, :8]
frames = y.shape[0] * y.shape[1] / max(1, int((a.sum(1) > 0).sum()))
names = ['dft16x2+combine', 'tw+swap+write+sync+locate+early', 'p2 read+dft32', 'sync+zwrite+sync', 'late loads issue', 'split+stores', 'window (wait for loads)', 'shift+issue loads']
nw = int((a.sum(1) > 0).sum())
life = full[:nw, 8:]
a = a[:nw]
tot = a.sum(1)
print('rc', rc, 'waves', nw, 'ms per launch %.4f' % ms, 'implied shader clock %.3f GHz' % (tot.mean() / (ms * 1e6)), 'mean total cycles per wave', tot.mean(), 'frames/wave', frames)
for i, n in enumerate(names):
    print(f'{n:32s} {a[:, i].mean() / frames:9.0f} cyc/frame  {100 * a[:, i].mean() / tot.mean():5.1f}%')
print('total per wave: min %.0f  p50 %.0f  p90 %.0f  max %.0f' % (tot.min(), np.percentile(tot, 50), np.percentile(tot, 90), tot.max()))
w = tot.reshape(256, -1)
print('per xcd mean/max:', [(int(w[x::8].mean()), int(w[x::8].max())) for x in range(8)])
print('mean by wave slot:', [int(v) for v in w.mean(0)])
print('std across WGs of WG-mean: %.0f ; mean within-WG std: %.0f' % (w.mean(1).std(), w.std(1).mean()))

# the shader clock as a ratio of two counters read by the same wave at its start and at its end: s_memtime (shader cycles)
# over s_memrealtime (constant 100 MHz); no launch time enters
clk = life[:, 0] / np.maximum(life[:, 1], 1) * 0.1
busy = life[:, 1] / 100e3 / ms      # the wave's life as a share of the launch
print('shader clock by counter ratio, GHz: mean %.3f  min %.3f  p10 %.3f  p50 %.3f  p90 %.3f  max %.3f' % (clk.mean(), clk.min(), np.percentile(clk, 10), np.percentile(clk, 50), np.percentile(clk, 90), clk.max()))
print('wave life / launch time: mean %.3f  min %.3f  max %.3f' % (busy.mean(), busy.min(), busy.max()))
