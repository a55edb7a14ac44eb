"""nfft 512 headline shape with frame counts that are / are not multiples of 4 (row blocks on / off 16-byte boundaries)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for N in (160000, 160160, 160320, 160480):
    B = 1024
    x = torch.randn((B, N), device='cuda') * 0.1
    win = torch.from_numpy(dsp_amd.hann(400, np.float32)).cuda()
    y = dsp_amd.stft_dev(x, win, 160, nfft=512)
    for _ in range(50): dsp_amd.stft_dev(x, win, 160, nfft=512, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): dsp_amd.stft_dev(x, win, 160, nfft=512, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    W = y.shape[1]
    print(f'N {N} W {W}: {ms:.4f} ms  {B * W * (160 + 257) * 4 / ms / 8e9 * 100:.1f} % of 8 TB/s', flush=True)
