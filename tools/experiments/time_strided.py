"""Interleaved channels ([N][C] row-major) through the strided entry point against the same channels dense."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for dt in (torch.float32, torch.float64):
    N, C = (1 << 22, 64) if dt == torch.float32 else (1 << 21, 64)
    t = torch.randn((N, C), device='cuda', dtype=dt) * 0.1
    x = t.T
    dense = x.contiguous()
    win = torch.from_numpy(dsp_amd.hann(400, np.float32 if dt == torch.float32 else np.float64)).cuda()
    y = dsp_amd.stft_dev(dense, win, 160)
    res = {}
    for name, src in (('dense', dense), ('interleaved', x)):
        t0 = time.time()
        while time.time() - t0 < 0.3:
            for _ in range(5): dsp_amd.stft_dev(src, win, 160, out=y)
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): dsp_amd.stft_dev(src, win, 160, out=y)
        b.record(); torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / 20
    gb = 2 * N * C * t.element_size() / 1e9
    print(f'{str(dt)[6:]} {C} channels x {N}: dense {res["dense"]:.3f} ms, interleaved {res["interleaved"]:.3f} ms; the gather alone {res["interleaved"] - res["dense"]:.3f} ms = {gb / (res["interleaved"] - res["dense"]):.2f} TB/s')
