import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
def run(nfft, L, hop, B, N, dt):
    x = torch.randn((B, N), device='cuda', dtype=dt) * 0.1
    win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
    for _ in range(10): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    W = y.shape[1]; es = x.element_size()
    byts = B * W * (hop + nfft // 2 + 1) * es
    return ms, byts / ms / 8e9 * 100, B * W / ms / 1e3
for nfft in (16, 32, 64, 128, 256, 1024, 2048, 4096, 8192):
    L, hop = nfft, nfft // 4
    N = 1 << 26
    B = 8
    for dt in (torch.float32, torch.float64):
        out = []
        for gen in (False, True):
            dsp_amd.force_generic(gen)
            fam = dsp_amd.kernel_family(nfft, np.float32 if dt == torch.float32 else np.float64)
            ms, pct, mf = run(nfft, L, hop, B, N // 4 if dt == torch.float64 else N // 2, dt)
            out.append(f'family {fam}: {ms:.3f} ms {pct:.1f} % {mf:.0f} Mfr/s')
        dsp_amd.force_generic(False)
        print(f'nfft {nfft} hop {hop} {str(dt)[6:]}:  ' + '   |   '.join(out))
