import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for nfft in (16, 32, 64, 128, 256):
    for dt in (torch.float32, torch.float64):
        L, hop = nfft, nfft // 4
        x = torch.randn((8, 1 << (24 if dt == torch.float32 else 23)), device='cuda', dtype=dt) * 0.1
        win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
        y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
        for _ in range(5): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f'nfft {nfft} {str(dt)[6:]}: {ms:.3f} ms  {y.shape[0] * y.shape[1] / ms / 1e3:.0f} Mfr/s')
