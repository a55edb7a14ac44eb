"""Known-answer and flag tests of the CPU oracle (no GPU): what the reference's fused loop computes
(c/stft.c:60-96) stated as analytic facts, plus its argument checks."""
import numpy as np
import pytest


def test_dc_rectangular_window(O):
    # constant signal, all-ones window: power L^2 in bin 0 of the unpadded transform
    x = np.full(1000, 2.0)
    rc, Y = O.stft(x, np.ones(64), 16, 64, True)
    assert rc == 0 and Y.shape == (1 + (1000 - 64) // 16, 33)
    np.testing.assert_allclose(Y[:, 0], (2.0 * 64) ** 2, rtol=1e-14)
    assert np.max(np.abs(Y[:, 1:])) < 1e-20
    rc, Yz = O.stft(x, np.ones(64), 16, 64, True, mn0=True)
    assert np.max(np.abs(Yz)) < 1e-20


def test_impulse_gives_flat_spectrum(O):
    x = np.zeros(512)
    x[200] = 1.0
    win = np.linspace(0.1, 1.0, 64)
    rc, Y = O.stft(x, win, 64, 64, True)      # frame 3 covers samples 192..255, impulse at l = 8
    np.testing.assert_allclose(Y[3], win[8] ** 2, rtol=1e-13)
    assert np.max(np.abs(np.delete(Y, 3, axis=0))) == 0.0


def test_cosine_bin(O):
    n = np.arange(4096)
    x = np.cos(2 * np.pi * 8 * n / 64)
    rc, Y = O.stft(x, np.ones(64), 64, 64, True)
    np.testing.assert_allclose(Y[:, 8], 32.0 ** 2, rtol=1e-12)
    assert np.max(np.delete(Y, 8, axis=1)) < 1e-20


def test_flags_amp_lg(O):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(3000)
    win = np.hanning(400)
    rc, P = O.stft(x, win, 160, 512)
    rc, A = O.stft(x, win, 160, 512, amp=True)
    rc, Lg = O.stft(x, win, 160, 512, lg=True)
    rc, AL = O.stft(x, win, 160, 512, amp=True, lg=True)
    np.testing.assert_array_equal(A, np.sqrt(P))
    np.testing.assert_allclose(Lg, np.log(P), rtol=0, atol=4e-15)          # libm vs numpy: an ulp
    np.testing.assert_allclose(AL, np.log(np.sqrt(P)), rtol=0, atol=4e-15)


def test_log_of_zero_is_minus_inf(O):
    rc, Y = O.stft(np.zeros(600, np.float32), np.ones(64, np.float32), 64, 64, True, lg=True)
    assert rc == 0 and np.all(np.isneginf(Y))


def test_mn0_subtracts_mean_over_L_only(O):
    rng = np.random.default_rng(4)
    x = rng.standard_normal(2000) + 5.0
    win = np.hanning(400)
    rc, Y = O.stft(x, win, 160, 512, mn0=True)
    # independent restatement
    W = (2000 + 80) // 160
    for w in (0, 3, W - 1):
        idx = np.arange(80 - 200 + w * 160, 80 - 200 + w * 160 + 400)
        idx = np.where(idx < 0, -idx - 1, idx)
        idx = np.where(idx >= 2000, 2 * 2000 - 1 - idx, idx)
        v = x[idx] * win
        v = v - v.mean()
        X = np.fft.rfft(v, 512)
        np.testing.assert_allclose(Y[w], np.abs(X) ** 2, rtol=0, atol=1e-12 * np.max(np.abs(X) ** 2))


def test_num_frames_formula(O):
    assert O.num_frames(160000, 400, 160, False) == 1000
    assert O.num_frames(160000, 400, 160, True) == 998
    assert O.num_frames(50, 400, 160, False) == 0
    assert O.num_frames(400, 400, 160, True) == 1


@pytest.mark.parametrize("args,msg", [
    (dict(L=0), "error in stft_s: L must be positive"),
    (dict(stp=0), "error in stft_s: stp must be positive"),
    (dict(N=100, snip=1), "error in stft_s: L must be < N if snip_edges"),
    (dict(nfft=256), "error in stft_s: nfft must be >= L"),
])
def test_stft_error_paths(O, capfd, args, msg):
    N, L, stp, nfft, snip = args.get("N", 2000), args.get("L", 400), args.get("stp", 160), args.get("nfft", 512), args.get("snip", 0)
    rc, _ = O.stft(np.zeros(N, np.float32), np.zeros(L, np.float32), stp, nfft, snip)
    assert rc == 1
    assert msg in capfd.readouterr().err


@pytest.mark.parametrize("kw,msg", [
    (dict(W=3000), "W must be <= N"),
    (dict(L=2500), "L must be <= N"),
    (dict(c0=2000.0), "c0 (center samp of 1st frame) must be < N"),
    (dict(stp=0.0), "stp (step size) must be positive"),
])
def test_stft_flt_error_paths(O, capfd, kw, msg):
    N = 2000
    L = kw.get("L", 400)
    nfft = 4096 if L > 512 else 512
    rc, _ = O.stft_flt(np.zeros(N), np.zeros(L), kw.get("W", 10), kw.get("c0", 0.0), kw.get("stp", 160.0), nfft=nfft)
    assert rc == 1
    err = capfd.readouterr().err
    assert "error in stft_flt_d: " in err and msg in err


def test_w_zero_is_success_and_touches_nothing(O):
    rc, Y = O.stft(np.zeros(50), np.ones(400), 160, 512, False)
    assert rc == 0 and Y.shape == (0, 257)


def test_odd_nfft_keeps_reference_pairing(O):
    """For odd nfft the reference's second power loop adds i_{f+1}^2 to bin f (c/stft.c:81-82 walks the
    half-complex array upward while Y walks down from F-2)."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal(200)
    win = np.ones(9)
    rc, Y = O.stft(x, win, 9, 9, True)
    X = np.fft.fft(x[:9], 9)
    F = 5
    want = X.real[:F] ** 2
    for f in range(1, F - 1):
        want[f] += X.imag[f + 1] ** 2
    np.testing.assert_allclose(Y[0], want, rtol=1e-12)


def test_fast_mode_close_to_exact(O):
    rng = np.random.default_rng(11)
    for dt, tol in ((np.float32, 2e-6), (np.float64, 1e-14)):
        x = rng.standard_normal(16000).astype(dt)
        win = np.hanning(400).astype(dt)
        rc, Ye = O.stft(x, win, 160, 512, exact=True)
        rc, Yf = O.stft(x, win, 160, 512, exact=False)
        assert O.normalised_err(Yf, Ye) <= tol


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_pow_compress_and_freqs_oracle_equals_reference(O, dt):
    """The restatements of c/pow_compress.c and c/get_stft_freqs.c return the reference's own bits."""
    x = np.abs(np.random.default_rng(3).standard_normal(5000)).astype(dt)
    x[:3] = 0
    for p, preg in [(1.0, 0.0), (1.0, 0.5), (0.0, 1e-6), (0.5, 0.0), (1.0 / 3.0, 1e-3), (0.3, 1e-3), (0.9, 0.0)]:
        rc, a = O.pow_compress(x, p, preg)
        rc2, b = O.pow_compress(x, p, preg, use_ref=True)
        assert rc == 0 and rc2 == 0 and np.array_equal(a, b, equal_nan=True)
    for p, preg in [(-0.1, 0.0), (1.1, 0.0), (0.5, -1.0)]:
        assert O.pow_compress(x, p, preg)[0] == 1 and O.pow_compress(x, p, preg, use_ref=True)[0] == 1
    for nfft, F, fs in [(512, 257, 16000.0), (512, 512, 16000.0), (512, 300, 44100.0), (4096, 2049, 48000.0), (2, 2, 1.0), (8, 1, 8.0)]:
        rc, a = O.get_stft_freqs(nfft, F, fs, dt)
        rc2, b = O.get_stft_freqs(nfft, F, fs, dt, use_ref=True)
        assert rc == 0 and rc2 == 0 and np.array_equal(a, b)
    assert O.get_stft_freqs(8, 9, 1.0, dt)[0] == 1 and O.get_stft_freqs(8, 9, 1.0, dt, use_ref=True)[0] == 1


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_complex_window_forms_of_the_reference_are_the_real_ones_with_zero_imaginary_parts(O, dt):
    """What the library's _c/_z generators return is stated from this: the reference's own hann_c ... planck_z
    (c/hann.c:67-113 and the like, compiled as they lie) give the values of the _s/_d forms, each followed by 0 -- all
    but gauss_c/_z, whose second half is a defect of the reference."""
    if O.ref() is None:
        pytest.skip("oracle/_ref not built")
    cdt = np.complex64 if dt == np.float32 else np.complex128
    pars = {"gauss": 2.5, "tukey": 0.5, "planck": 0.1}
    for name in ("hann", "hamming", "povey", "blackmanharris", "blackman", "bartlett", "triangular", "rectangular", "flattop",
                 "gauss", "tukey", "planck"):
        for L in (16, 31, 400):
            for norm in (0, 1, 2, 3):
                re = O.ref_window(name, L, dt, norm, par=pars.get(name))
                cx = O.ref_window_cplx(name, L, cdt, norm, par=pars.get(name))
                if name == "gauss":
                    # c/gauss.c:78,107: the second half is mirrored with the REAL form's offset (2 l + 1 + L % 2 elements
                    # back instead of 4 l + 2 + 2 (L % 2)) and so copies the wrong elements; the first half is right.
                    # The library returns the symmetric window there (include/dsp_stft.h).
                    if norm == 0:
                        assert np.all(cx.imag[: L // 2] == 0)
                        assert np.allclose(cx.real[: L // 2], re[: L // 2], rtol=4 * np.finfo(dt).eps, atol=0), (name, L)
                        assert not np.allclose(cx.real[L - L // 2:], re[L - L // 2:], rtol=1e-3, atol=0)
                    continue
                assert np.all(cx.imag == 0), (name, L, norm)
                assert np.allclose(cx.real, re, rtol=4 * np.finfo(dt).eps, atol=0), (name, L, norm)
