import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for (stp, hop512) in [(160.0, None), (160.5, None), (300.25, None)]:
    B, N, L = 64, 16000 * 600, 400
    x = torch.randn((B, N), device='cuda') * 0.1
    win = torch.from_numpy(dsp_amd.povey(L)).cuda()
    W = int((N - 1 - L // 2) // stp)
    y = dsp_amd.stft_flt_dev(x, win, W, float(L // 2), stp)
    for _ in range(20): dsp_amd.stft_flt_dev(x, win, W, float(L // 2), stp, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dsp_amd.stft_flt_dev(x, win, W, float(L // 2), stp, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f'stft_flt float32 hop {stp}: {ms:.3f} ms  {B * W * (stp + 257) * 4 / ms / 8e9 * 100:.1f} % of 8 TB/s  {B * W / ms / 1e3:.0f} Mfr/s')
# nfft 512 with a hop above 256 (span mode of the general kernel)
for hop in (320, 400):
    x = torch.randn((1024, 160000), device='cuda') * 0.1
    win = torch.from_numpy(dsp_amd.hann(400)).cuda()
    y = dsp_amd.stft_dev(x, win, hop)
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f'stft float32 L 400 hop {hop}: {ms:.3f} ms  {1024 * y.shape[1] * (hop + 257) * 4 / ms / 8e9 * 100:.1f} % of 8 TB/s')
# the same in float64 (512 signals)
for hop in (320, 400):
    x = torch.randn((512, 160000), device='cuda', dtype=torch.float64) * 0.1
    win = torch.from_numpy(dsp_amd.hann(400, np.float64)).cuda()
    y = dsp_amd.stft_dev(x, win, hop)
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    dsp_amd.force_generic(True)
    for _ in range(5): dsp_amd.stft_dev(x, win, hop, out=y)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10): dsp_amd.stft_dev(x, win, hop, out=y)
    b.record(); torch.cuda.synchronize()
    dsp_amd.force_generic(False)
    print(f'stft float64 L 400 hop {hop}: {ms:.3f} ms  {512 * y.shape[1] * (hop + 257) * 8 / ms / 8e9 * 100:.1f} % of 8 TB/s (generic family: {a.elapsed_time(b) / 10:.3f} ms)')
