"""nfft 512 with hops above 512 (frames further apart than a transform length): % of the HBM peak, algorithmic bytes
= (min(hop, 512)... every sample a frame reads + 257 outputs) per frame: the samples between frames are never touched."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for dt in (torch.float32, torch.float64):
    for (L, hop) in [(512, 600), (512, 1024), (400, 800), (512, 513)]:
        B, N = 64, 1 << 22
        x = torch.randn((B, N), device='cuda', dtype=dt) * 0.1
        win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
        y = dsp_amd.stft_dev(x, win, hop, nfft=512)
        for _ in range(20): dsp_amd.stft_dev(x, win, hop, nfft=512, out=y)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): dsp_amd.stft_dev(x, win, hop, nfft=512, out=y)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 50
        W = y.shape[1]; es = x.element_size()
        print(f'{dt} L {L} hop {hop}: {ms:.3f} ms  {B * W * (min(hop, L) + 257) * es / ms / 8e9 * 100:.1f} % of 8 TB/s (bytes the frames read and write)  {B * W / ms / 1e6:.2f} Gfr/s')
