"""nfft 64 / 128 / 256, float32: the register-resident kernel against the LDS-pass kernel (force_generic -> family 2 is the
generic one; the LDS-pass kernel, family 5, ran these sizes at about 22 % of the HBM peak)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
def run(nfft, L, hop, B, N):
    x = torch.randn((B, N), device='cuda') * 0.1
    win = torch.from_numpy(dsp_amd.hann(L)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
    for _ in range(10): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 30
    W = y.shape[1]
    byts = B * W * (hop + nfft // 2 + 1) * 4
    return ms, byts / ms / 8e9 * 100, B * W / ms / 1e6
for (nfft, L, hop, B, N) in [(64, 64, 16, 8, 1 << 24), (64, 64, 32, 8, 1 << 24), (64, 50, 20, 1024, 40000),
                             (128, 128, 32, 8, 1 << 24), (128, 128, 64, 8, 1 << 24), (128, 100, 40, 1024, 80000), (128, 128, 128, 8, 1 << 24),
                             (256, 256, 64, 8, 1 << 25), (256, 200, 80, 1024, 80000), (256, 256, 128, 8, 1 << 25)]:
    ms, pct, gf = run(nfft, L, hop, B, N)
    print(f'nfft {nfft} L {L} hop {hop} B {B} N {N}: family {dsp_amd.kernel_family(nfft, np.float32)} {ms:.3f} ms  {pct:.1f} % of 8 TB/s  {gf:.2f} G frames/s')
