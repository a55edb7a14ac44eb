"""Many launches of the benchmark kernels on the same input, every result compared with the first by an exact (integer,
order-independent) checksum per signal: the kernels that deal their work statically or by drawn steps must repeat bit
for bit (the nfft-4096 run form deals runs at run time and rotates frames accordingly: compared to 1e-6 instead)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd

def chk(y):
    return y.view(torch.int32 if y.dtype == torch.float32 else torch.int64).to(torch.int64).sum(dim=(1, 2))

def soak(name, run, n, exact=True):
    y0 = run().clone()
    h0 = chk(y0)
    bad = 0
    worst = 0.0
    t0 = time.time()
    for i in range(n):
        y = run()
        if exact:
            bad += int(not torch.equal(chk(y), h0))
        else:
            worst = max(worst, float(((y - y0).abs().amax() / y0.amax())))
    torch.cuda.synchronize()
    print(f'{name}: {n} launches, differing {bad}, worst relative difference {worst:.2e}, {time.time() - t0:.1f} s', flush=True)
    return bad == 0 and worst <= 1e-6

if __name__ == '__main__':
    ok = True
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn((1024, 160000), device='cuda', generator=g) * 0.1
    win = torch.from_numpy(dsp_amd.hann(400)).cuda()
    y = torch.empty((1024, 1000, 257), device='cuda')
    ok &= soak('headline float32', lambda: dsp_amd.stft_dev(x, win, 160, out=y), int(sys.argv[1]) if len(sys.argv) > 1 else 3000)
    ok &= soak('headline -m -l', lambda: dsp_amd.stft_dev(x, win, 160, mn0=True, lg=True, out=y), 500)
    x = torch.randn((1024, 80000), device='cuda', generator=g) * 0.1
    win = torch.from_numpy(dsp_amd.hann(200)).cuda()
    ok &= soak('nfft 256', lambda: dsp_amd.stft_dev(x, win, 80, nfft=256), 500)
    x = torch.randn((64, 2646000), device='cuda', generator=g) * 0.1
    for nfft in (1024, 2048):
        win = torch.from_numpy(dsp_amd.hann(nfft)).cuda()
        ok &= soak(f'nfft {nfft}', lambda: dsp_amd.stft_dev(x, win, nfft // 4, nfft=nfft), 300)
    x = torch.randn((8, 14400000), device='cuda', dtype=torch.float64, generator=g) * 0.1
    win = torch.from_numpy(dsp_amd.povey(400, np.float64)).cuda()
    W = (14400000 - 1 - 200) // 160
    ok &= soak('stft_flt float64', lambda: dsp_amd.stft_flt_dev(x, win, W, 200.0, 160.0), 500)
    Wt = int((14400000 - 1 - 200) // 160.5)
    ok &= soak('stft_flt float64 hop 160.5', lambda: dsp_amd.stft_flt_dev(x, win, Wt, 200.0, 160.5), 300)
    x = torch.randn((8, 43200000), device='cuda', generator=g) * 0.1
    win = torch.from_numpy(dsp_amd.blackmanharris(4096)).cuda()
    ok &= soak('nfft 4096 run form', lambda: dsp_amd.stft_dev(x, win, 1024, nfft=4096), 300, exact=False)
    print('SOAK', 'OK' if ok else 'FAILED')
