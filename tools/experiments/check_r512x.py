"""nfft 256 / 1024 / 2048 float32 kernels: results against the generic kernel family on the same device (which the parity
tests pin to the oracle), launch-to-launch equality, and launch time against the LDS-pass kernel it replaces
(DSP_R512X_OFF=1 in the environment selects that one)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd

def one(nfft, L, hop, B, N, snip, mn0, amp, lg, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = (torch.randn((B, N), generator=g) * 0.3 + 0.05).cuda()
    win = torch.from_numpy(dsp_amd.hann(L, np.float32)).cuda()
    kw = dict(nfft=nfft, snip_edges=snip, mn0=mn0, amp=amp, lg=lg)
    y = dsp_amd.stft_dev(x, win, hop, **kw)
    y2 = dsp_amd.stft_dev(x, win, hop, **kw)
    dsp_amd.force_generic(True)
    try:
        ref = dsp_amd.stft_dev(x, win, hop, **kw)
    finally:
        dsp_amd.force_generic(False)
    torch.cuda.synchronize()
    same = bool(torch.equal(y, y2))
    if lg:                                   # back to the linear domain: bins near zero have no relative accuracy
        y, ref = torch.exp(y.double()), torch.exp(ref.double())
    scale = ref.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)
    err = float(((y - ref).abs() / scale).max())
    return err, same, tuple(y.shape)

bad = 0
cases = []
for nfft in (256, 1024, 2048):
    for L in (nfft, nfft - 24, nfft - 56, nfft // 2 + 3, 7):
        for hop in (nfft // 4, nfft // 2, 80, 160, 161, 441, nfft, 1):
            for snip in (1, 0):
                cases.append((nfft, L, hop, snip))
rng = np.random.default_rng(5)
for i, (nfft, L, hop, snip) in enumerate(cases):
    B = int(rng.integers(1, 4))
    N = int(rng.integers(max(L, nfft) + 1, 9 * nfft)) if hop > 1 else int(rng.integers(nfft + 1, nfft + 600))
    mn0, amp, lg = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
    err, same, shp = one(nfft, L, hop, B, N, snip, mn0, amp, lg, i)
    tol = 1e-5
    ok = err <= tol and same
    bad += not ok
    if not ok or i % 16 == 0:
        print(f'nfft {nfft} L {L} hop {hop} snip {snip} B {B} N {N} mn0 {mn0} amp {amp} lg {lg} -> {shp}: err {err:.2e} same {same} {"ok" if ok else "FAIL"}', flush=True)
print('cases', len(cases), 'failures', bad, flush=True)

def timeit(nfft, L, hop, B, N):
    x = torch.randn((B, N), device='cuda') * 0.1
    win = torch.from_numpy(dsp_amd.hann(L, np.float32)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
    for _ in range(30): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    W = y.shape[1]
    byts = B * W * (min(hop, nfft) + nfft // 2 + 1) * 4
    return ms, byts / ms / 8e9 * 100

if bad == 0 or '--time' in sys.argv:
    for nfft, L, hop in ((256, 256, 64), (256, 200, 80), (256, 256, 128), (256, 256, 33), (1024, 1024, 256), (1024, 1024, 512), (1024, 800, 200), (1024, 1024, 128), (2048, 2048, 512), (2048, 2048, 1024), (2048, 1200, 300)):
        ms, pct = timeit(nfft, L, hop, 8, 1 << 25)
        print(f'time nfft {nfft} L {L} hop {hop}: {ms:.3f} ms  {pct:.1f} % of 8 TB/s', flush=True)
sys.exit(1 if bad else 0)
