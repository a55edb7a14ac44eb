"""Launch times of the nfft-512 instantiations whose pass-1 twiddles come from LDS (double; float with 2048-sample spans)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd

def t(run, n=30):
    for _ in range(n): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

out = []
# config 3 chunk: stft_flt_d, 8 channels x 1 h @ 16 kHz
x = torch.randn((8, 57600000), device='cuda', dtype=torch.float64) * 0.1
win = torch.from_numpy(dsp_amd.povey(400, np.float64)).cuda()
W = (57600000 - 1 - 200) // 160
y = dsp_amd.stft_flt_dev(x, win, W, 200.0, 160.0)
ms = t(lambda: dsp_amd.stft_flt_dev(x, win, W, 200.0, 160.0, out=y), 15)
out.append('flt_d hop 160.0: %.3f ms %.1f %%' % (ms, 8 * W * 3336 / ms / 8e9 * 100))
del x, y
for (L, hop, dt, B) in [(512, 256, torch.float64, 512), (400, 200, torch.float64, 512), (400, 320, torch.float64, 512), (400, 161, torch.float64, 512),
                        (400, 320, torch.float32, 1024), (400, 400, torch.float32, 1024), (512, 512, torch.float32, 1024)]:
    x = torch.randn((B, 160000), device='cuda', dtype=dt) * 0.1
    win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=512)
    ms = t(lambda: dsp_amd.stft_dev(x, win, hop, nfft=512, out=y))
    out.append('L %d hop %d %s: %.3f ms %.1f %%' % (L, hop, str(dt)[6:], ms, B * y.shape[1] * (hop + 257) * x.element_size() / ms / 8e9 * 100))
x = torch.randn((64, 9600000), device='cuda') * 0.1
win = torch.from_numpy(dsp_amd.povey(400)).cuda()
W = int((9600000 - 1 - 200) // 300.25)
y = dsp_amd.stft_flt_dev(x, win, W, 200.0, 300.25)
ms = t(lambda: dsp_amd.stft_flt_dev(x, win, W, 200.0, 300.25, out=y))
out.append('flt_s hop 300.25: %.3f ms %.1f %%' % (ms, 64 * W * (300.25 + 257) * 4 / ms / 8e9 * 100))
print(' | '.join(out))
