"""The float64 forms of the 64 / 128 / 256 and 1024 / 2048 kernels and the float32 64 / 128 forms: many launches on the same
input, every result compared with the first by an exact checksum per signal (tools/experiments/soak.py does the same for
the benchmark kernels)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd

def chk(y):
    return y.view(torch.int32 if y.dtype == torch.float32 else torch.int64).to(torch.int64).sum(dim=(1, 2))

def soak(name, run, n):
    h0 = chk(run().clone())
    bad = 0
    t0 = time.time()
    for i in range(n):
        bad += int(not torch.equal(chk(run()), h0))
    torch.cuda.synchronize()
    print(f'{name}: {n} launches, differing {bad}, {time.time() - t0:.1f} s', flush=True)
    return bad == 0

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ok = True
g = torch.Generator(device='cuda').manual_seed(6)
for dt, ndt in ((torch.float64, np.float64), (torch.float32, np.float32)):
    for nfft in ((64, 128, 256, 1024, 2048) if dt == torch.float64 else (64, 128)):
        B, N = (512, 80000) if nfft <= 256 else (32, 2646000)
        x = torch.randn((B, N), device='cuda', dtype=dt, generator=g) * 0.1
        win = torch.from_numpy(dsp_amd.hann(nfft, ndt)).cuda()
        y = dsp_amd.stft_dev(x, win, nfft // 4, nfft=nfft)
        ok &= soak(f'{str(dt)[6:]} nfft {nfft} hop {nfft // 4}', lambda: dsp_amd.stft_dev(x, win, nfft // 4, nfft=nfft, out=y), n)
        ok &= soak(f'{str(dt)[6:]} nfft {nfft} hop {nfft // 4 + 1} -m -l', lambda: dsp_amd.stft_dev(x, win, nfft // 4 + 1, mn0=True, lg=True, nfft=nfft), max(50, n // 10))
        del x, y
print('OK' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
