"""A/B: interleaved channels ([N][C] row-major, float32, C channels x 2^22 samples, Hann 400 / hop 160 / nfft 512) through the
strided entry point:
 (a) gathered into a dense batch, then the dense kernel (DSP_ILV_MAX_CHANNELS=0);
 (b) read in place by the interleaved form of the 512-point kernel (a step = one frame of 4 adjacent channels);
beside the dense kernel on a dense copy:  python tools/experiments/ilv_ab.py [C ...]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, dsp_amd

N = 1 << 22
for C in ([int(v) for v in sys.argv[1:]] or [64, 32, 16, 8, 4]):
    t = torch.randn((N, C), device='cuda') * 0.1            # [N][C]
    x = t.T                                                  # a view: channels strided
    dense = x.contiguous()
    win = torch.from_numpy(dsp_amd.hann(400, np.float32)).cuda()
    y = dsp_amd.stft_dev(dense, win, 160)
    y2 = dsp_amd.stft_dev(x, win, 160)
    err = float((y2 - y).abs().max() / y.abs().max())
    assert err < 1e-5
    res = {}
    for name, src, env in (('dense', dense, None), ('gather + dense', x, '0'), ('in place', x, None)):
        if env is None:
            os.environ.pop('DSP_ILV_MAX_CHANNELS', None)
        else:
            os.environ['DSP_ILV_MAX_CHANNELS'] = env
        t0 = time.time()
        while time.time() - t0 < 0.3:
            for _ in range(5): dsp_amd.stft_dev(src, win, 160, out=y)
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): dsp_amd.stft_dev(src, win, 160, out=y)
        b.record(); torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / 20
    os.environ.pop('DSP_ILV_MAX_CHANNELS', None)
    nbytes = (t.numel() + y.numel()) * 4
    print(f'{C} channels x {N} float32 ({nbytes / 1e9:.2f} GB in + out; in place against dense: max |diff| / max = {err:.1e}): '
          + ', '.join(f'{k} {v:.3f} ms ({nbytes / v / 8e9 * 100:.1f} %)' for k, v in res.items()), flush=True)
    del t, x, dense, y, y2
