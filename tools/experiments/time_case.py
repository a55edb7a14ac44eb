import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
L = int(sys.argv[1]); dt = torch.float32 if sys.argv[2] == 'f32' else torch.float64
flags = [bool(int(c)) for c in (sys.argv[3] if len(sys.argv) > 3 else '000')]
B, N = (1024, 160000) if dt == torch.float32 else (512, 160000)
x = torch.randn((B, N), device='cuda', dtype=dt) * 0.1
win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
y = dsp_amd.stft_dev(x, win, 160, False, *flags)
for _ in range(800): dsp_amd.stft_dev(x, win, 160, False, *flags, out=y)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
for a, b in ev:
    a.record(); dsp_amd.stft_dev(x, win, 160, False, *flags, out=y); b.record()
torch.cuda.synchronize()
print('L', L, sys.argv[2], flags, 'kernel_ms %.4f' % np.mean([a.elapsed_time(b) for a, b in ev]))
