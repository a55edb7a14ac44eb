"""Seeded random shapes through every kernel family against the oracle: transform lengths from 16 to 8192,
any window length up to nfft, any hop (odd, above the window, above the span limits), signals shorter than a
window, every flag, snipped edges, and stft_flt with integral and fractional hops in both framing modes."""
import os

import numpy as np
import pytest

from test_gpu_parity import TOL, check_spec, signal

pytestmark = pytest.mark.gpu

# DSP_FUZZ_SEED shifts the seeds, DSP_FUZZ_SCALE multiplies the number of shapes (longer runs by hand; the defaults are
# what the suite runs)
SEED = int(os.environ.get("DSP_FUZZ_SEED", "0"))
SCALE = int(os.environ.get("DSP_FUZZ_SCALE", "1"))


@pytest.fixture(scope="module")
def D():
    import dsp_amd
    return dsp_amd


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_random_shapes_stft(O, D, dt):
    rng = np.random.default_rng(20240 + np.dtype(dt).itemsize + 1000 * SEED)
    for it in range(600 * SCALE):
        nfft = int(2 ** rng.integers(4, 14))
        L = int(rng.integers(1, nfft + 1)) if rng.random() < 0.6 else nfft
        stp = int(rng.integers(1, 2 * L + 2)) if rng.random() < 0.7 else max(1, L // int(rng.integers(2, 6)))
        N = int(rng.integers(1, 12 * L + 40))
        if N * nfft // max(stp, 1) > 3_000_000:           # keep the oracle's double DFT in seconds
            stp = max(stp, N * nfft // 3_000_000 + 1)
        if (N // stp + 1) * nfft * nfft > 2_000_000_000:
            stp = max(stp, N * nfft * nfft // 2_000_000_000 + 1)
        snip = bool(rng.integers(0, 2)) and L <= N
        flags = tuple(int(v) for v in rng.integers(0, 2, 3))
        x = signal(1000 + it, N, dt) + dt(0.05)
        win = D.povey(L, dt) if L > 1 else np.ones(1, dt)
        if nfft >= 4096:          # float64 FFT oracle: the long-double DFT is O(nfft^2) per frame
            rc, Yref = O.stft(x.astype(np.float64), win.astype(np.float64), stp, nfft, snip, *flags, exact=False)
            Yref = Yref.astype(dt)
        else:
            rc, Yref = O.stft(x, win, stp, nfft, snip, *flags)
        assert rc == 0, (nfft, L, stp, N, snip, flags)
        for generic in (False, True):
            D.force_generic(generic)
            try:
                Y = D.stft(x, win, stp, snip, *[bool(f) for f in flags], nfft=nfft)
            finally:
                D.force_generic(False)
            try:
                check_spec(O, Y, Yref, flags[1], flags[2])
            except AssertionError as e:
                raise AssertionError(f"nfft {nfft} L {L} stp {stp} N {N} snip {snip} flags {flags} generic {generic}: {e}")


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_random_shapes_stft_flt(O, D, dt):
    rng = np.random.default_rng(777 + np.dtype(dt).itemsize + 1000 * SEED)
    for it in range(400 * SCALE):
        nfft = int(2 ** rng.integers(4, 14))
        L = int(rng.integers(1, nfft + 1)) if rng.random() < 0.6 else nfft
        N = int(rng.integers(L + 1, 10 * L + 60))
        stp = float(rng.integers(1, L + 2)) if rng.random() < 0.5 else float(rng.uniform(0.5, 1.5 * L))
        W = int(min(N, rng.integers(1, 40), max(1, 2_000_000_000 // (nfft * nfft))))
        c0 = float(rng.uniform(-1.5 * L, N - 1))
        flags = tuple(int(v) for v in rng.integers(0, 2, 3))
        intended = bool(rng.integers(0, 2))
        x = signal(5000 + it, N, dt) + dt(0.05)
        win = D.povey(L, dt) if L > 1 else np.ones(1, dt)
        D.set_flt_mode(intended)
        try:
            if nfft >= 4096:
                # float64 FFT oracle: it also accumulates the frame centres in double, the float32 entry point in
                # float, so only integral c0 and hops (exact in both) give the same frames
                c0, stp = float(round(c0)), float(max(1, round(stp)))
                rc, Yref = O.stft_flt(x.astype(np.float64), win.astype(np.float64), W, c0, stp, nfft, *flags, intended=intended, exact=False)
                Yref = Yref.astype(dt)
            else:
                rc, Yref = O.stft_flt(x, win, W, c0, stp, nfft, *flags, intended=intended)
            assert rc == 0, (nfft, L, stp, N, W, c0)
            for generic in (False, True):
                D.force_generic(generic)
                try:
                    Y = D.stft_flt(x, win, W, c0, stp, *[bool(f) for f in flags], nfft=nfft)
                finally:
                    D.force_generic(False)
                try:
                    check_spec(O, Y, Yref, flags[1], flags[2])
                except AssertionError as e:
                    raise AssertionError(f"nfft {nfft} L {L} stp {stp} N {N} W {W} c0 {c0} flags {flags} intended {intended} generic {generic}: {e}")
        finally:
            D.set_flt_mode(False)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_random_large_shapes_fast_kernels_against_generic(O, D, dt):
    """The kernels with work distribution over the whole chip (drawn steps, ranges per XCD, span slots refilled step after
    step) on inputs of many steps per wave: random batches of long signals, transform lengths 64 to 8192, window lengths,
    hops inside and outside the span limits, flags, snipped edges; against the generic kernels (one workgroup per frame,
    checked against the oracle by the tests above) on the same input, to the oracle's bound."""
    import torch
    rng = np.random.default_rng(4242 + np.dtype(dt).itemsize + 1000 * SEED)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    g = torch.Generator(device="cuda").manual_seed(99 + SEED)
    for it in range(48 * SCALE):
        nfft = int(2 ** rng.integers(6, 14))
        L = int(rng.integers(nfft // 2, nfft + 1)) if rng.random() < 0.5 else nfft
        stp = int(rng.choice([nfft // 4, nfft // 2, nfft // 8 + 1, int(rng.integers(1, nfft + 2)), L]))
        stp = max(stp, nfft // 64)
        B = int(rng.integers(1, 6))
        W = int(rng.integers(2000, 12000))
        N = W * stp + int(rng.integers(0, L))
        if B * N > 30_000_000:
            N = 30_000_000 // B
        snip = bool(rng.integers(0, 2)) and L <= N
        flags = [bool(v) for v in rng.integers(0, 2, 3)]
        x = torch.randn((B, N), device="cuda", dtype=tdt, generator=g) * 0.1 + 0.01
        win = torch.from_numpy(D.povey(L, dt)).cuda()
        Y = D.stft_dev(x, win, stp, snip, *flags, nfft=nfft)
        D.force_generic(True)
        try:
            Yg = D.stft_dev(x, win, stp, snip, *flags, nfft=nfft)
        finally:
            D.force_generic(False)
        torch.cuda.synchronize()
        try:
            for b in range(B):
                check_spec(O, Y[b].cpu().numpy(), Yg[b].cpu().numpy(), int(flags[1]), int(flags[2]), tol=2 * TOL[np.dtype(dt)])   # (each side is within the bound of the oracle)
        except AssertionError as e:
            raise AssertionError(f"nfft {nfft} L {L} stp {stp} B {B} N {N} snip {snip} flags {flags}: {e}")
        del x, Y, Yg
