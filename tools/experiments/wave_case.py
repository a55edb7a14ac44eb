import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
nfft = int(sys.argv[1]); dt = torch.float32 if sys.argv[2] == 'f32' else torch.float64
L, hop = nfft, nfft // 4
x = torch.randn((8, 1 << 23), device='cuda', dtype=dt) * 0.1
win = torch.from_numpy(dsp_amd.hann(L, np.float32 if dt == torch.float32 else np.float64)).cuda()
y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
for _ in range(3): dsp_amd.stft_dev(x, win, hop, nfft=nfft, out=y)
torch.cuda.synchronize()
print('frames', y.shape[0] * y.shape[1])
