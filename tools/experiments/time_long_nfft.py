"""Kernel time of a long transform (default nfft 4096) at hop nfft/4 in float64 and float32, register kernel and generic
kernel, on 8 x 2^24 and 8 x 43.2 M samples: python tools/experiments/time_4096_f64.py [reps] [nfft]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dsp_amd as D

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nfft = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    for B, N in ((8, 1 << 24), (8, 172_800_000 // 4)):
        x = torch.randn((B, N), device="cuda", dtype=tdt) * 0.1
        w = torch.from_numpy(D.blackmanharris(nfft, dt)).cuda()
        y = D.stft_dev(x, w, nfft // 4, nfft=nfft)
        for forced in (False, True):
            D.force_generic(forced)
            for _ in range(2):
                D.stft_dev(x, w, nfft // 4, nfft=nfft, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                D.stft_dev(x, w, nfft // 4, nfft=nfft, out=y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            nbytes = (x.numel() + y.numel()) * x.element_size()
            print(f"{np.dtype(dt).name} B={B} N={N} {'generic' if forced else 'register'}: {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s  {nbytes / ms / 8e9 * 100:.1f} % of 8 TB/s", flush=True)
        D.force_generic(False)
        del x, y
