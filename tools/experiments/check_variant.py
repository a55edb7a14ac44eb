"""Parity of a variant build (DSP_AMD_LIB) on the benchmark shape.  Every signal: an exact (integer, order-independent)
checksum of the output bits, compared between repeated launches and with the checksums another build left in
gpurun_out/hash_ref.npy (argument `ref` writes that file); a few signals against the CPU oracle."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
from oracle import oracle as O
B, N = 1024, 160000
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn((B, N), device='cuda', generator=g) * 0.1
win = torch.from_numpy(dsp_amd.hann(400)).cuda()
def run():
    y = dsp_amd.stft_dev(x, win, 160)
    torch.cuda.synchronize()
    return y, y.view(torch.int32).to(torch.int64).sum(dim=(1, 2)).cpu().numpy()
y, h = run()
bad = 0
for i in range(60):
    _, h2 = run()
    bad += int(not np.array_equal(h, h2))
print('launches that differ from the first: %d of 60' % bad)
ref = 'gpurun_out/hash_ref.npy'
if len(sys.argv) > 1 and sys.argv[1] == 'ref':
    np.save(ref, h)
elif os.path.exists(ref):
    d = np.nonzero(np.load(ref) != h)[0]
    print('signals whose bits differ from the reference build: %d' % d.size, d[:16])
xh, wh = x.cpu().numpy(), win.cpu().numpy()
worst = 0.0
for b in (0, 1, 2, 3, 255, 256, 777, 1023):
    rc, r = O.stft(xh[b], wh, 160, 512)
    worst = max(worst, O.normalised_err(y[b].cpu().numpy(), r))
print('worst normalised error over 8 signals: %.3e' % worst, 'OK' if worst <= 1e-5 and bad == 0 else 'FAIL')
