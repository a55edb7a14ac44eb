"""float64 launch times of the power-of-two lengths without a kernel of their own (L = nfft, hop nfft / 4, 8 x 2^24 samples)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
out = []
for nfft in (64, 256, 1024, 2048, 4096):
    x = torch.randn((8, 1 << 24), device='cuda', dtype=torch.float64) * 0.1
    win = torch.from_numpy(dsp_amd.hann(nfft, np.float64)).cuda()
    hop = nfft // 4
    y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
    for _ in range(10): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    out.append('%d: %.3f ms %.1f %%' % (nfft, ms, 8 * y.shape[1] * (hop + nfft // 2 + 1) * 8 / ms / 8e9 * 100))
print(' | '.join(out))
