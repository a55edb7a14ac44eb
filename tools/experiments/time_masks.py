"""Shapes whose instantiations carry per-row lane masks (windows shorter than nfft, and the general-window instantiation of the
512-point kernel, which also serves L = 512): kernel time and % of the HBM peak."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for (nfft, L, hop) in [(512, 512, 128), (512, 512, 256), (512, 480, 128), (512, 480, 160), (512, 320, 160), (512, 256, 128), (512, 300, 100), (1024, 1024, 220), (1024, 551, 220), (1024, 800, 200), (1024, 800, 320), (1024, 882, 441), (1024, 960, 480), (1024, 700, 220), (2048, 2048, 480), (2048, 1200, 480), (2048, 1102, 441), (2048, 1500, 512), (4096, 4096, 1024), (4096, 4096, 960), (4096, 2400, 960), (4096, 3000, 1024), (256, 256, 64), (256, 240, 64), (256, 240, 80), (256, 160, 80), (256, 200, 80), (128, 128, 32)]:
    B, N = 8, 1 << 24
    x = torch.randn((B, N), device='cuda') * 0.1
    win = torch.from_numpy(dsp_amd.hann(L, np.float32)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
    import time
    t0 = time.time()
    while time.time() - t0 < 0.3:                         # the clocks ramp up under load: 0.3 s of launches first
        for _ in range(8): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 30
    print(f'nfft {nfft} L {L} hop {hop}: {ms:.4f} ms  {B * y.shape[1] * (hop + nfft // 2 + 1) * 4 / ms / 8e9 * 100:.1f} %', flush=True)
    del x, y
