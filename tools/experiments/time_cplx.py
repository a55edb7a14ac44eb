"""Launch time of the complex spectrogram (dsp_stft_cplx_batch_dev) on the benchmark shapes, beside torch.stft on the same input."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
def t(f, n=50):
    for _ in range(n // 2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (dt, B, N, L, hop) in ((torch.float32, 1024, 160000, 400, 160), (torch.float64, 512, 160000, 400, 160), (torch.float32, 1024, 160000, 512, 256)):
    x = torch.randn((B, N), device='cuda', dtype=dt) * 0.1
    win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
    y = dsp_amd.stft_cplx_dev(x, win, hop)
    ms = t(lambda: dsp_amd.stft_cplx_dev(x, win, hop, out=y))
    es = x.element_size()
    byts = B * y.shape[1] * (hop + 2 * 257) * es
    wpad = torch.zeros(512, device='cuda', dtype=dt); wpad[(512 - L) // 2:(512 - L) // 2 + L] = win
    f = lambda: torch.stft(x, 512, hop_length=hop, window=wpad, center=True, pad_mode='reflect', return_complex=True)
    try:
        mt = t(f, 10)
    except Exception as e:
        mt = float('nan')
    print('%s L %d hop %d: %.3f ms  %.1f %% of 8 TB/s (%d B per frame)   torch.stft: %.3f ms' % (str(dt)[6:], L, hop, ms, byts / ms / 8e9 * 100, (hop + 514) * es, mt), flush=True)
for (nfft, hop) in ((1024, 256), (2048, 512)):
    x = torch.randn((64, 2646000), device='cuda') * 0.1
    win = torch.from_numpy(dsp_amd.hann(nfft)).cuda()
    y = dsp_amd.stft_cplx_dev(x, win, hop, nfft=nfft)
    ms = t(lambda: dsp_amd.stft_cplx_dev(x, win, hop, nfft=nfft, out=y), 20)
    byts = 64 * y.shape[1] * (hop + nfft + 2) * 4
    mt = t(lambda: torch.stft(x, nfft, hop_length=hop, window=win, center=True, pad_mode='reflect', return_complex=True), 6)
    print('float32 nfft %d hop %d: %.3f ms  %.1f %% of 8 TB/s   torch.stft: %.3f ms' % (nfft, hop, ms, byts / ms / 8e9 * 100, mt), flush=True)
x = torch.randn((1024, 80000), device='cuda') * 0.1
win = torch.from_numpy(dsp_amd.hann(200)).cuda()
y = dsp_amd.stft_cplx_dev(x, win, 80, nfft=256)
ms = t(lambda: dsp_amd.stft_cplx_dev(x, win, 80, nfft=256, out=y))
print('float32 nfft 256 L 200 hop 80: %.3f ms  %.1f %% of 8 TB/s' % (ms, 1024 * y.shape[1] * (80 + 258) * 4 / ms / 8e9 * 100), flush=True)
x = torch.randn((8, 43200000), device='cuda') * 0.1
win = torch.from_numpy(dsp_amd.blackmanharris(4096)).cuda()
y = dsp_amd.stft_cplx_dev(x, win, 1024, nfft=4096)
ms = t(lambda: dsp_amd.stft_cplx_dev(x, win, 1024, nfft=4096, out=y), 10)
mt = t(lambda: torch.stft(x, 4096, hop_length=1024, window=win, center=True, pad_mode='reflect', return_complex=True), 4)
print('float32 nfft 4096 hop 1024: %.3f ms  %.1f %% of 8 TB/s   torch.stft: %.3f ms' % (ms, 8 * y.shape[1] * (1024 + 4098) * 4 / ms / 8e9 * 100, mt), flush=True)
