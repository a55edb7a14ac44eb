import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for (L, hop) in [(4096, 1024), (4096, 2048), (4096, 512), (4096, 1000), (2048, 512), (3000, 750), (4096, 1023)]:
    B, N = 8, 1 << 25
    x = torch.randn((B, N), device='cuda') * 0.1
    win = torch.from_numpy(dsp_amd.blackmanharris(L)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=4096)
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, nfft=4096, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, nfft=4096, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    W = y.shape[1]
    print(f'L {L} hop {hop}: {ms:.3f} ms  {B * W * (hop + 2049) * 4 / ms / 8e9 * 100:.1f} % of 8 TB/s  {B * W / ms / 1e3:.0f} Mfr/s')
