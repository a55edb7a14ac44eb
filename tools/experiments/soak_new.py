"""Many launches of the forms added last (hops above 512, interleaved channels read in place, the float64 4096-point kernel, the
8192-point kernel, the 2048-point kernel with its partner fetch by ds_bpermute, the forms for windows shorter than the transform), every result compared with the first by an
exact checksum per signal (the 4096-point run form deals runs at run time: compared to 1e-6 / 1e-14 instead)."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tools/experiments')
import numpy as np, torch, dsp_amd
from soak import soak

ok = True
g = torch.Generator(device='cuda').manual_seed(6)
x = torch.randn((64, 1 << 22), device='cuda', generator=g) * 0.1
win = torch.from_numpy(dsp_amd.hann(512)).cuda()
ok &= soak('nfft 512 hop 600', lambda: dsp_amd.stft_dev(x, win, 600, nfft=512), 1000)
ok &= soak('nfft 512 hop 1024', lambda: dsp_amd.stft_dev(x, win, 1024, nfft=512), 1000)
win4 = torch.from_numpy(dsp_amd.hann(400)).cuda()
for C in (16, 32, 8):
    t = torch.randn((1 << 21, C), device='cuda', generator=g) * 0.1
    xi = t.T
    ok &= soak(f'{C} interleaved channels in place', lambda: dsp_amd.stft_dev(xi, win4, 160), 1500)
xd = torch.randn((8, 1 << 23), device='cuda', dtype=torch.float64, generator=g) * 0.1
wd = torch.from_numpy(dsp_amd.blackmanharris(4096, np.float64)).cuda()
ok &= soak('nfft 4096 float64 run form', lambda: dsp_amd.stft_dev(xd, wd, 1024, nfft=4096), 300, exact=False)
ok &= soak('nfft 4096 float64 frame form', lambda: dsp_amd.stft_dev(xd, wd, 1000, nfft=4096), 300)
x8 = torch.randn((8, 1 << 24), device='cuda', generator=g) * 0.1
w8 = torch.from_numpy(dsp_amd.hann(8192)).cuda()
ok &= soak('nfft 8192', lambda: dsp_amd.stft_dev(x8, w8, 2048, nfft=8192), 500)
w2 = torch.from_numpy(dsp_amd.hann(2048)).cuda()
ok &= soak('nfft 2048', lambda: dsp_amd.stft_dev(x8, w2, 512, nfft=2048), 500)
xd2 = xd[:, :1 << 22]
w2d = torch.from_numpy(dsp_amd.hann(2048, np.float64)).cuda()
ok &= soak('nfft 2048 float64', lambda: dsp_amd.stft_dev(xd2, w2d, 512, nfft=2048), 500)
# windows shorter than the transform: the unmasked slot of the 2048-point kernel (zeros written behind the window once a span has landed),
# the row-count instantiations of the 1024-, 512- and 256-point kernels
for (nfft, L, hop) in ((2048, 1102, 441), (2048, 1201, 480), (1024, 551, 220), (512, 480, 160), (512, 512, 128), (256, 240, 80)):
    wl = torch.from_numpy(dsp_amd.povey(L, np.float32)).cuda()
    ok &= soak(f'nfft {nfft} L {L} hop {hop}', lambda: dsp_amd.stft_dev(x8, wl, hop, nfft=nfft), 500)
wl = torch.from_numpy(dsp_amd.povey(1103, np.float64)).cuda()
ok &= soak('nfft 2048 L 1103 hop 441 float64', lambda: dsp_amd.stft_dev(xd2, wl, 441, nfft=2048), 500)
print('SOAK', 'OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
