"""The CPU oracle against fixtures produced by EXECUTED reference code (tests/golden/ref_*.npz, made by
oracle/make_golden.py from the reference's FFTW-free C sources).  Framing and windowing must match
bit for bit; the spectrum/power is limited by the accuracy of the reference's own radix-2 FFT, whose
twiddles come from a float recurrence (c/fft.rad2.c:96-113): 3e-6 (float32) / 2e-14 (float64)."""
import numpy as np
import pytest

from conftest import load_golden

CHAIN = ["base", "short_multifold", "oddL", "hop_gt_L", "N_eq_L", "nfft4096", "nfft_gt_pow2", "tiny"]
PTOL = {"s": 3e-6, "d": 2e-14}


@pytest.mark.parametrize("tag", ["s", "d"])
@pytest.mark.parametrize("name", CHAIN)
def test_chain_power(O, tag, name):
    g = load_golden(f"ref_chain_{tag}.npz")
    x, win = g[f"{name}.x"], g[f"{name}.win"]
    N, L, stp, nfft = (int(v) for v in g[f"{name}.params"])
    for snip in (0, 1):
        key = f"{name}.snip{snip}.power"
        if key not in g:
            continue
        rc, Y = O.stft(x, win, stp, nfft, snip)
        assert rc == 0
        P = g[key]
        assert Y.shape == P.shape
        assert O.normalised_err(Y, P) <= PTOL[tag]


@pytest.mark.parametrize("tag", ["s", "d"])
@pytest.mark.parametrize("name", ["base", "short_multifold", "tiny"])
def test_chain_intermediates(O, tag, name):
    g = load_golden(f"ref_chain_{tag}.npz")
    x, win = g[f"{name}.x"], g[f"{name}.win"]
    N, L, stp, nfft = (int(v) for v in g[f"{name}.params"])
    for snip in (0, 1):
        if f"{name}.snip{snip}.frames" not in g:
            continue
        rc, fr = O.frame_univar(x, L, stp, snip)
        assert rc == 0 and np.array_equal(fr, g[f"{name}.snip{snip}.frames"])
        rc, wn = O.apply_win(fr, win, 1, False)
        assert rc == 0 and np.array_equal(wn, g[f"{name}.snip{snip}.windowed"])
        rc, z = O.fft_r2hc(wn, 1, nfft)
        zr = g[f"{name}.snip{snip}.spectrum"]
        assert rc == 0 and np.max(np.abs(z - zr)) <= PTOL[tag] * np.max(np.abs(zr))
        rc, p = O.fft_power(zr)
        assert rc == 0 and np.array_equal(p, g[f"{name}.snip{snip}.power"])


@pytest.mark.parametrize("tag", ["s", "d"])
def test_flt_intended_matches_reference_framing(O, tag):
    g = load_golden(f"ref_flt_{tag}.npz")
    x, win = g["x"], g["win"]
    for i, (c0, stp, W) in enumerate(g["cases"]):
        W = int(W)
        rc, Y = O.stft_flt(x, win, W, c0, stp, intended=True)
        assert rc == 0
        assert O.normalised_err(Y, g[f"case{i}.power"]) <= PTOL[tag]
        if f"case{i}.frames" in g:
            rc, fr = O.frame_univar_flt(x, win.size, W, c0, stp)
            assert rc == 0 and np.array_equal(fr, g[f"case{i}.frames"])
            rc, wn = O.apply_win(fr, win, 1, False)
            assert np.array_equal(wn, g[f"case{i}.windowed"])


@pytest.mark.parametrize("tag", ["s", "d"])
def test_flt_compat_equals_intended_only_when_first_start_is_zero(O, tag):
    """c/stft_flt.c never advances X1 to the first non-negative start: with c0 = L/2 that start is 0 and
    both framings agree; otherwise every frame from there on is read `shift` samples early."""
    g = load_golden(f"ref_flt_{tag}.npz")
    x, win = g["x"], g["win"]
    L = win.size
    for i, (c0, stp, W) in enumerate(g["cases"]):
        W = int(W)
        rc, Yc = O.stft_flt(x, win, W, c0, stp, intended=False)
        rc2, Yi = O.stft_flt(x, win, W, c0, stp, intended=True)
        assert rc == 0 and rc2 == 0
        ss = O.flt_starts(L, W, c0, stp, x.dtype)
        nonneg = ss[(ss >= 0) & (ss <= x.size - 1)]
        shift = int(nonneg[0]) if nonneg.size else 0
        if shift == 0:
            assert np.array_equal(Yc, Yi)
        else:
            # frames that start left of 0 are unaffected; the others equal the intended STFT of x shifted
            rc, Ys = O.stft_flt(np.concatenate([np.zeros(shift, x.dtype), x])[: x.size], win, W, c0, stp, intended=True)
            for w in range(W):
                if ss[w] < 0:
                    assert np.array_equal(Yc[w], Yi[w])
                elif ss[w] + L <= x.size:
                    assert np.array_equal(Yc[w], Ys[w])


@pytest.mark.parametrize("tag,dt", [("s", np.float32), ("d", np.float64)])
def test_window_generators(tag, dt):
    """dsp_amd.windows restates c/hann.c etc.; libm's cos/pow differ from numpy's by an ulp at most."""
    from dsp_amd import windows
    g = load_golden(f"ref_windows_{tag}.npz")
    for key in g.files:
        parts = key.split(".")
        name, L = parts[0], int(parts[1])
        if L < 2 or len(parts) > 2:          # normalised windows are the library's own generators' business
            continue
        w = windows.WINDOWS[name](L, dt)
        assert w.dtype == dt
        np.testing.assert_allclose(w, g[key], rtol=0, atol=4 * np.finfo(dt).eps)


@pytest.mark.parametrize("tag", ["s", "d"])
def test_complex_framing_and_apply_win(O, tag):
    g = load_golden(f"ref_misc_{tag}.npz")
    xc = g["x"]
    for snip in (0, 1):
        rc, fr = O.frame_univar(xc, 48, 20, snip)
        assert rc == 0 and np.array_equal(fr, g[f"frames.snip{snip}"])
    rc, fr = O.frame_univar_flt(xc, 48, 30, -10.0, 17.3)
    assert rc == 0 and np.array_equal(fr, g["frames_flt"])
    a = g["aw.x"]
    for col in (0, 1):
        rc, y0 = O.apply_win(a, g["aw.w0"], 0, bool(col))
        rc, y1 = O.apply_win(a, g["aw.w1"], 1, bool(col))
        assert np.array_equal(np.ascontiguousarray(y0), g[f"aw.dim0.col{col}"])
        assert np.array_equal(np.ascontiguousarray(y1), g[f"aw.dim1.col{col}"])


@pytest.mark.parametrize("tag", ["s", "d"])
def test_fft_oracle_against_reference_radix2(O, tag):
    """The DFT restatement behind fft_fftw_* vs what the reference's own fft_rad2_c/_z returned (fixtures);
    the bound is that code's float twiddle recurrence, as for the chain fixtures.  Real input: the bins
    0..nfft/2 of the complex transform, which is also what the r2hc oracle returns."""
    g = load_golden(f"ref_fft_c_{tag}.npz")
    for n in (2, 8, 64, 512):
        x, yref = g[f"x.{n}"], g[f"y.{n}"]
        rc, y = O.fft(x, 1, n)
        assert rc == 0 and y.shape == yref.shape and y.dtype == yref.dtype
        assert np.max(np.abs(y - yref)) <= PTOL[tag] * np.max(np.abs(yref))
        xr = np.ascontiguousarray(x.real)
        rc, yr = O.fft(xr, 1, n, sc=True)
        rc2, yh = O.fft_r2hc(xr, 1, n, False, True)
        assert rc == 0 and rc2 == 0
        assert np.max(np.abs(yr - yh)) <= 4 * np.finfo(yr.real.dtype).eps * np.max(np.abs(yh))
    assert O.fft(np.zeros((3, 9), np.float32), 1, 8)[0] == 1
