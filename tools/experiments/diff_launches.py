"""Where two launches of one build differ on the benchmark shape (a race hunt): signal, frame, bin, both values."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
B, N = 1024, 160000
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn((B, N), device='cuda', generator=g) * 0.1
win = torch.from_numpy(dsp_amd.hann(400)).cuda()
ref = dsp_amd.stft_dev(x, win, 160)
torch.cuda.synchronize()
seen = 0
for it in range(40):
    y = dsp_amd.stft_dev(x, win, 160)
    torch.cuda.synchronize()
    d = torch.nonzero(y != ref)
    if d.numel():
        d = d.cpu().numpy()
        fr = np.unique(d[:, :2], axis=0)
        print('launch', it, ':', d.shape[0], 'values differ in', fr.shape[0], 'frames; first frames (b, w):', fr[:6].tolist(),
              'bins of the first frame:', d[(d[:, 0] == fr[0, 0]) & (d[:, 1] == fr[0, 1])][:, 2][:12].tolist())
        b, w, f = d[0]
        print('   y', float(y[b, w, f]), 'ref', float(ref[b, w, f]), ' step', (b * 250 + w // 4), 'frame in step', w % 4)
        seen += 1
        if seen >= 8: break
print('done, launches with differences:', seen)
