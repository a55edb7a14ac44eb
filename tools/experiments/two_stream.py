import sys
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd
B, N = 1024, 160000
x = torch.randn((B, N), device='cuda') * 0.1
win = torch.from_numpy(dsp_amd.hann(400)).cuda()
y = dsp_amd.stft_dev(x, win, 160)
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()
def single():
    dsp_amd.stft_dev(x, win, 160, out=y)
def split(nb):
    e = torch.cuda.Event(); e.record(main)
    dsp_amd.stft_dev(x[:B - nb], win, 160, out=y[:B - nb])
    with torch.cuda.stream(s2):
        s2.wait_event(e)
        dsp_amd.stft_dev(x[B - nb:], win, 160, out=y[B - nb:])
        e2 = torch.cuda.Event(); e2.record(s2)
    main.wait_event(e2)
def timed(fn, reps=300):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print('single %.4f' % timed(single))
for nb in (32, 64, 96, 128):
    print('split tail', nb, '%.4f' % timed(lambda: split(nb)))
print('single %.4f' % timed(single))
