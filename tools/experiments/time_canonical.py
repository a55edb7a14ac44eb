"""L = nfft, hop = nfft / 4 (the usual 75 % overlap) and hop = nfft / 2 for every kernel family, float32 and float64."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
def run(nfft, hop, dt):
    tdt = torch.float32 if dt == np.float32 else torch.float64
    B, N = 8, (1 << 24) if dt == np.float32 else (1 << 23)
    x = torch.randn((B, N), device='cuda', dtype=tdt) * 0.1
    win = torch.from_numpy(dsp_amd.hann(nfft, dt)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
    import time
    t0 = time.time()
    while time.time() - t0 < 0.3:                         # the clocks ramp up under load: 0.3 s of launches first
        for _ in range(8): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    byts = B * y.shape[1] * (hop + nfft // 2 + 1) * x.element_size()
    return f'{ms:.3f} ms {byts / ms / 8e9 * 100:.0f} %'
for nfft in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    print(f'nfft {nfft}: f32 hop/4 {run(nfft, nfft // 4, np.float32)} | hop/2 {run(nfft, nfft // 2, np.float32)} | f64 (family {dsp_amd.kernel_family(nfft, np.float64)}) hop/4 {run(nfft, nfft // 4, np.float64)} | hop/2 {run(nfft, nfft // 2, np.float64)}')
