"""The allocation-free entry point dsp_stft_batch_dev_noalloc_s (include/dsp_stft.h), called through the C
ABI, against dsp_stft_batch_dev_s on the same device buffers: two calls in a row, the second with a smaller
input, for every kernel family it reaches (nfft 512, 4096 in both loop forms, a wavefront-per-frame size, a
generic size).  Bar: the path's float32 tolerance (1e-5 per-frame-normalised); the twiddle tables of the two
entry points are rounded from different precisions (long double on the host, double on the device), so
bit-equality is reported, not required."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import TOL

pytestmark = pytest.mark.gpu


def _call(name, y, x, win, stp, nfft, snip, mn0, amp, lg):
    import torch
    from dsp_amd._lib import lib
    B, N = x.shape
    rc = getattr(lib(), name)(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(win.data_ptr()), B, N, x.stride(0),
                              win.numel(), nfft, stp, int(snip), int(mn0), int(amp), int(lg),
                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    return rc


def _both(O, B, N, L, stp, nfft, snip=False, mn0=False, amp=False, lg=False, seed=0):
    import torch
    import dsp_amd
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((B, N), device="cuda", generator=g) * 0.1
    win = torch.from_numpy(dsp_amd.hann(L)).cuda()
    W = dsp_amd.num_frames(N, L, stp, snip)
    F = nfft // 2 + 1
    y_ref = torch.full((B, W, F), float("nan"), device="cuda")
    y_na = torch.full((B, W, F), float("nan"), device="cuda")
    assert _call("dsp_stft_batch_dev_s", y_ref, x, win, stp, nfft, snip, mn0, amp, lg) == 0
    assert _call("dsp_stft_batch_dev_noalloc_s", y_na, x, win, stp, nfft, snip, mn0, amp, lg) == 0
    torch.cuda.synchronize()
    a, b = y_na.cpu().numpy(), y_ref.cpu().numpy()
    assert np.all(np.isfinite(a) == np.isfinite(b)) and not np.any(np.isnan(a))
    if lg:
        a, b = np.exp(a.astype(np.float64)), np.exp(b.astype(np.float64))
    err = max(O.normalised_err(a[i], b[i]) for i in range(B))
    assert err <= TOL[np.dtype(np.float32)], err
    return bool(np.array_equal(a, b))


@pytest.mark.parametrize("nfft,L,stp,big,small", [
    (512, 400, 160, (64, 48000), (3, 16000)),
    (4096, 4096, 1024, (4, 300000), (2, 20000)),
    (256, 200, 80, (16, 30000), (2, 5000)),
    (128, 100, 40, (16, 30000), (2, 5000)),
    (64, 64, 16, (16, 20000), (3, 700)),
    (1024, 1024, 256, (8, 60000), (2, 5000)),
    (8, 8, 3, (4, 1000), (1, 100)),
])
def test_noalloc_matches_batch_dev_twice(O, nfft, L, stp, big, small):
    same1 = _both(O, big[0], big[1], L, stp, nfft, seed=1)
    same2 = _both(O, small[0], small[1], L, stp, nfft, mn0=True, lg=True, seed=2)
    print(f"nfft {nfft}: bit-equal {same1} {same2}")


def test_noalloc_4096_run_form_twice(O):
    """The 4096-point loop that deals runs from counters (normally chosen for >= 4096 runs): forced at a
    small size, so that the counters in the static array are used and cleared on consecutive calls."""
    old = os.environ.get("DSP_R4096_MIN_RUNS")
    os.environ["DSP_R4096_MIN_RUNS"] = "1"
    try:
        _both(O, 6, 1024 * 200, 4096, 1024, 4096, seed=3)
        _both(O, 2, 1024 * 70, 4096, 1024, 4096, amp=True, seed=4)
        _both(O, 5, 1024 * 133 + 17, 4096, 1024, 4096, seed=5)
    finally:
        if old is None:
            del os.environ["DSP_R4096_MIN_RUNS"]
        else:
            os.environ["DSP_R4096_MIN_RUNS"] = old


def test_noalloc_rejects_sizes_beyond_its_limit(capfd):
    import torch
    x = torch.zeros((1, 20000), device="cuda")
    win = torch.ones(8192, device="cuda")
    y = torch.zeros((1, 10, 4097), device="cuda")
    assert _call("dsp_stft_batch_dev_noalloc_s", y, x, win, 1024, 8192, False, False, False, False) == 1
    assert "error in dsp_stft_batch_dev_noalloc_s: nfft must be a power of two <= 4096" in capfd.readouterr().err
    win = torch.ones(300, device="cuda")
    assert _call("dsp_stft_batch_dev_noalloc_s", y, x, win, 100, 300, False, False, False, False) == 1
    assert "power of two" in capfd.readouterr().err


@pytest.mark.parametrize("nfft,L,stp,B,N,run_form", [(512, 400, 160, 4, 160000, False), (4096, 4096, 1024, 3, 1024 * 150, True)])
def test_noalloc_under_graph_capture(O, nfft, L, stp, B, N, run_form):
    """What the entry point is for: it is called while its stream is being captured into a HIP graph (a call that
    allocated would fail the capture), and the graph is replayed on new input.  The replays equal direct calls on
    the same input bit for bit (the 4096 run form, whose work is dealt at run time: to the tolerance); its
    counter clear (a memset node) replays with the kernel."""
    import torch
    import dsp_amd
    old = os.environ.get("DSP_R4096_MIN_RUNS")
    if run_form:
        os.environ["DSP_R4096_MIN_RUNS"] = "1"
    try:
        g = torch.Generator(device="cuda").manual_seed(11)
        x = torch.randn((B, N), device="cuda", generator=g) * 0.1
        win = torch.from_numpy(dsp_amd.hann(L)).cuda()
        W, F = dsp_amd.num_frames(N, L, stp, False), nfft // 2 + 1
        y_direct = torch.empty((B, W, F), device="cuda")
        y_graph = torch.full((B, W, F), float("nan"), device="cuda")
        # first call outside the capture: the tables of the static array are filled once per device
        assert _call("dsp_stft_batch_dev_noalloc_s", y_direct, x, win, stp, nfft, False, True, False, False) == 0
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            rc = _call("dsp_stft_batch_dev_noalloc_s", y_graph, x, win, stp, nfft, False, True, False, False)
        assert rc == 0
        for seed in (12, 13):
            x.copy_(torch.randn((B, N), device="cuda", generator=g.manual_seed(seed)) * 0.1)
            graph.replay()
            assert _call("dsp_stft_batch_dev_noalloc_s", y_direct, x, win, stp, nfft, False, True, False, False) == 0
            torch.cuda.synchronize()
            a, b = y_graph.cpu().numpy(), y_direct.cpu().numpy()
            assert not np.any(np.isnan(a))
            if run_form:
                assert max(O.normalised_err(a[i], b[i]) for i in range(B)) <= 1e-6
            else:
                assert np.array_equal(a, b)
        rc, ref = O.stft(x[1].cpu().numpy(), win.cpu().numpy(), stp, nfft, 0, 1, exact=False)
        assert O.normalised_err(y_graph[1].cpu().numpy(), ref) <= TOL[np.dtype(np.float32)]
    finally:
        if run_form:
            if old is None:
                del os.environ["DSP_R4096_MIN_RUNS"]
            else:
                os.environ["DSP_R4096_MIN_RUNS"] = old
