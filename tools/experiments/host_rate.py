import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
rng = np.random.default_rng(0)
win = dsp_amd.hann(400)
# C1: one utterance through the reference's own entry point (host pointers, copies included)
x = (rng.standard_normal(160000) * 0.1).astype(np.float32)
for _ in range(5): dsp_amd.stft(x, win, 160)
t = []
for _ in range(50):
    t0 = time.perf_counter(); dsp_amd.stft(x, win, 160); t.append(time.perf_counter() - t0)
print('C1 stft_s host-pointer: median %.1f us, min %.1f us per call (1000 frames) -> %.2f M frames/s' % (np.median(t) * 1e6, np.min(t) * 1e6, 1000 / np.median(t) / 1e6))
xd = torch.from_numpy(x).cuda(); wd = torch.from_numpy(win).cuda()
y = dsp_amd.stft_dev(xd, wd, 160)
torch.cuda.synchronize()
t = []
for _ in range(50):
    t0 = time.perf_counter(); dsp_amd.stft_dev(xd, wd, 160, out=y); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
print('C1 device-resident call+sync: median %.1f us' % (np.median(t) * 1e6))
# C2-sized batch through the host-pointer batched entry point (pageable numpy memory)
xb = (rng.standard_normal((1024, 160000)) * 0.1).astype(np.float32)
dsp_amd.stft_batch(xb[:8], win, 160)
t = []
for _ in range(3):
    t0 = time.perf_counter(); Y = dsp_amd.stft_batch(xb, win, 160); t.append(time.perf_counter() - t0)
print('C2 dsp_stft_batch_s host-pointer (pageable): min %.1f ms -> %.1f M frames/s, %.2f GB/s of host traffic' % (np.min(t) * 1e3, 1.024e6 / np.min(t) / 1e6, (xb.nbytes + Y.nbytes) / np.min(t) / 1e9))
