"""Device-resident timings of the steps around the fused kernel (not the headline benchmark: bench.py is).
Streaming steps against the HBM roofline, and the fused stft | pow_compress against the unfused pair.
Usage: python tools/bench_steps.py [--out profiles/r01_steps.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsp_amd  # noqa: E402


def timed(fn, reps=50, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rows = []
    peak = 8000.0

    def add(name, ms, byts):
        gbs = byts / ms / 1e6
        rows.append({"step": name, "ms": ms, "bytes": byts, "GB/s": gbs, "frac_of_8TBps": gbs / peak})
        print(f"{name:58s} {ms:8.4f} ms  {gbs:7.0f} GB/s  {100 * gbs / peak:5.1f} %")

    n = 257 * 1024 * 1000                        # the spectrogram of the headline workload
    for dt, cdt in ((torch.float32, torch.complex64), (torch.float64, torch.complex128)):
        es = torch.empty(0, dtype=dt).element_size()
        z = torch.randn(n // 2, dtype=cdt, device="cuda")
        y = torch.empty(n // 2, dtype=dt, device="cuda")
        add(f"fft_power {str(dt)[6:]} ({n // 2} bins)", timed(lambda: dsp_amd.fft_power_dev(z, out=y)), 3 * es * (n // 2))
        del z, y
        x = torch.rand(n // (1 if es == 4 else 2), dtype=dt, device="cuda") + 0.1
        o = torch.empty_like(x)
        for p in (0.0, 0.5, 1.0 / 3.0, 0.3):
            add(f"pow_compress {str(dt)[6:]} p={p:.3f} ({x.numel()} values)", timed(lambda: dsp_amd.pow_compress_dev(x, p, 1e-3, out=o)), 2 * es * x.numel())
        del x, o
    # fused stft | pow_compress against stft then pow_compress in place (headline workload)
    x = torch.randn((1024, 160000), device="cuda") * 0.1
    win = torch.from_numpy(dsp_amd.hann(400)).cuda()
    y = dsp_amd.stft_dev(x, win, 160)
    algo = 1024 * 1000 * (160 + 257) * 4
    add("stft float32 (headline workload)", timed(lambda: dsp_amd.stft_dev(x, win, 160, out=y)), algo)
    add("stft | pow_compress p=1/3, fused", timed(lambda: dsp_amd.stft_pc_dev(x, win, 1.0 / 3.0, 1e-3, 160, out=y)), algo)

    def unfused():
        dsp_amd.stft_dev(x, win, 160, out=y)
        dsp_amd.pow_compress_dev(y, 1.0 / 3.0, 1e-3, out=y)
    add("stft then pow_compress in place, two kernels", timed(unfused), algo + 2 * 4 * y.numel())
    for name in ("hann", "planck"):
        add(f"window {name} L=4096 on the device (latency)", timed(lambda: dsp_amd.window_dev(name, 4096), reps=200), 4096 * 4)
    if args.out:
        json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
