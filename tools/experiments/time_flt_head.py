"""stft_flt in the reference's framing mode with first frames that hang over the start of the signal (c0 < L/2):
one launch on the general kernels before, head + tail launches now."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
def run(dt, B, N, L, nfft, c0, stp, W):
    tdt = torch.float32 if dt == np.float32 else torch.float64
    x = torch.randn((B, N), device='cuda', dtype=tdt) * 0.1
    win = torch.from_numpy(dsp_amd.povey(L, dt)).cuda()
    y = dsp_amd.stft_flt_dev(x, win, W, c0, stp, nfft=nfft)
    for _ in range(5): dsp_amd.stft_flt_dev(x, win, W, c0, stp, nfft=nfft, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dsp_amd.stft_flt_dev(x, win, W, c0, stp, nfft=nfft, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    byts = B * W * (stp + nfft // 2 + 1) * x.element_size()
    print(f'{np.dtype(dt).name} B {B} N {N} L {L} nfft {nfft} c0 {c0} stp {stp} W {W}: {ms:.3f} ms  {byts / ms / 8e9 * 100:.1f} % of 8 TB/s')
for c0 in (200.0, 0.0, 37.0):
    run(np.float64, 8, 57_600_000 // 8, 400, 512, c0, 160.0, 57_600_000 // 8 // 160 - 2)
    run(np.float64, 8, 57_600_000 // 8, 400, 512, c0, 160.5, 57_600_000 // 8 // 161 - 2)
    run(np.float32, 1024, 160000, 400, 512, c0, 160.0, 998)
    run(np.float32, 8, 1 << 24, 1024, 1024, c0, 256.0, (1 << 24) // 256 - 4)
