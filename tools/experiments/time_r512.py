import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for (L, hop, dt) in [(400, 160, torch.float32), (400, 170, torch.float32), (400, 172, torch.float32), (400, 200, torch.float32), (512, 256, torch.float32), (512, 128, torch.float32), (400, 161, torch.float32), (512, 256, torch.float64), (400, 200, torch.float64)]:
    B, N = 1024, 160000
    if dt == torch.float64: B = 512
    x = torch.randn((B, N), device='cuda', dtype=dt) * 0.1
    win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=512)
    for _ in range(50): dsp_amd.stft_dev(x, win, hop, nfft=512, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): dsp_amd.stft_dev(x, win, hop, nfft=512, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    W = y.shape[1]; es = x.element_size()
    byts = B * W * (hop + 257) * es
    print(f'L {L} hop {hop} {str(dt)[6:]}: {ms:.3f} ms  {byts / ms / 8e9 * 100:.1f} % of 8 TB/s  {B * W / ms / 1e3:.0f} Mfr/s')
