"""nfft 2048, full window, hop 512 and 1024: the ring-of-quarters form against the step form (DSP_R512X_MIN_UNITS=0), both precisions."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
for dt, tdt, n in ((np.float32, torch.float32, 1 << 24), (np.float64, torch.float64, 1 << 23)):
    x = torch.randn((8, n), device='cuda', dtype=tdt) * 0.1
    win = torch.from_numpy(dsp_amd.hann(2048, dt)).cuda()
    for hop in (512, 1024):
        y = dsp_amd.stft_dev(x, win, hop, nfft=2048)
        for name, mu in (('steps', '0'), ('ring', None), ('steps', '0'), ('ring', None)):
            if mu is None: os.environ.pop('DSP_R512X_MIN_UNITS', None)
            else: os.environ['DSP_R512X_MIN_UNITS'] = mu
            for _ in range(30): dsp_amd.stft_dev(x, win, hop, nfft=2048, out=y)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): dsp_amd.stft_dev(x, win, hop, nfft=2048, out=y)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 30
            print(f'{np.dtype(dt).name} hop {hop} {name}: {ms:.4f} ms  {8 * y.shape[1] * (hop + 1025) * x.element_size() / ms / 8e9 * 100:.1f} %', flush=True)
