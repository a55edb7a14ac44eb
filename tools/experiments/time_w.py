import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for (N, dt) in [(160000, torch.float32), (160160, torch.float32), (160320, torch.float32), (160480, torch.float32), (160160, torch.float64)]:
    B = 1024 if dt == torch.float32 else 512
    x = torch.randn((B, N), device='cuda', dtype=dt) * 0.1
    win = torch.from_numpy(dsp_amd.hann(400, np.float32 if dt == torch.float32 else np.float64)).cuda()
    y = dsp_amd.stft_dev(x, win, 160)
    for _ in range(100): dsp_amd.stft_dev(x, win, 160, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): dsp_amd.stft_dev(x, win, 160, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 100
    W = y.shape[1]; es = x.element_size()
    print(f'N {N} W {W} {str(dt)[6:]}: {ms:.3f} ms  {B * W * (160 + 257) * es / ms / 8e9 * 100:.1f} % of 8 TB/s')
