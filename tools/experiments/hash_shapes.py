"""Bits of the output of a build (DSP_AMD_LIB) on a fixed list of shapes of the span-slot kernels (nfft 64 ... 2048, both
precisions, full and short windows, mean removal, compression, both framings, equally spaced and table frames, signals
that end inside a step), as exact checksums: `python tools/experiments/hash_shapes.py ref` writes them to
gpurun_out/hash_shapes.npy, without the argument they are compared with that file.  For changes that must not alter a
single bit (address arithmetic, instruction selection)."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd

cases = []
for dt in (torch.float32, torch.float64):
    for nfft in (64, 128, 256, 512, 1024, 2048):
        for L in (nfft, nfft - nfft // 8 - 3):
            for hop in (nfft // 4, nfft // 2, nfft // 4 + 5, nfft // 8):
                for B, N in ((6, 70000 + 13 * nfft), (3, 2 * nfft + 7)):
                    for snip, mn0, amp in ((0, 0, 0), (1, 1, 0), (0, 0, 1)):
                        cases.append((dt, nfft, L, hop, B, N, snip, mn0, amp))
hs = []
for dt, nfft, L, hop, B, N, snip, mn0, amp in cases:
    g = torch.Generator(device='cpu').manual_seed(nfft + L + hop + B)
    x = (torch.randn((B, N), generator=g, dtype=dt) * 0.3 + 0.05).cuda()
    win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
    y = dsp_amd.stft_dev(x, win, hop, nfft=nfft, snip_edges=snip, mn0=mn0, amp=amp)
    v = y.view(torch.int32 if dt == torch.float32 else torch.int64).to(torch.int64)
    hs.append(int((v * (torch.arange(v.numel(), device='cuda').reshape(v.shape) % 8191 + 1)).sum().item()))
    # fractional hops: the table form
    if hop == nfft // 4 and B == 6:
        Wf = int((N - L) / (hop + 0.5))
        y = dsp_amd.stft_flt_dev(x, win, Wf, 3.0, hop + 0.5, nfft=nfft, mn0=mn0, amp=amp)
        v = y.view(torch.int32 if dt == torch.float32 else torch.int64).to(torch.int64)
        hs.append(int((v * (torch.arange(v.numel(), device='cuda').reshape(v.shape) % 8191 + 1)).sum().item()))
torch.cuda.synchronize()
h = np.array(hs, dtype=np.int64)
ref = 'gpurun_out/hash_shapes.npy'
if len(sys.argv) > 1 and sys.argv[1] == 'ref':
    os.makedirs('gpurun_out', exist_ok=True)
    np.save(ref, h)
    print('wrote %d checksums' % h.size)
else:
    r = np.load(ref)
    d = np.nonzero(r != h)[0]
    print('%d checksums, %d differ from the reference build' % (h.size, d.size), d[:20], 'OK' if d.size == 0 else 'FAIL')
    sys.exit(0 if d.size == 0 else 1)
