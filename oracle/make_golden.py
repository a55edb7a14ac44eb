"""Generates tests/golden/*.npz from EXECUTED reference code (oracle/_ref/libdspref.so, the reference's
FFTW-free C sources compiled where they lie).  Run in the build container only:

    make -C oracle && python oracle/make_golden.py

Each fixture holds seeded inputs and what the reference's own functions returned for them:
frame_univar / frame_univar_flt (framing), apply_win / window_univar / window_univar_flt (windowing),
fft_rad2_c/_z (the reference's own radix-2 FFT, used because its FFTW-backed fft.fftw.r2hc cannot be
built here) and fft_power.  The chain frame_univar -> apply_win -> FFT -> fft_power is the unfused
form of stft_s/stft_d with mn0 = amp = lg = 0.  Test infrastructure; data only, no reference source.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def nextpow2(L):
    n = 1
    while n < L:
        n *= 2
    return n


def chain_cases():
    # (name, N, L, stp, nfft, window)
    return [
        ("base", 2000, 400, 160, 512, "hann"),
        ("short_multifold", 300, 400, 160, 512, "hann"),      # N < L: repeated reflection (no snip only)
        ("oddL", 2000, 401, 160, 512, "povey"),
        ("hop_gt_L", 2000, 64, 100, 64, "hamming"),
        ("N_eq_L", 400, 400, 160, 512, "hann"),
        ("nfft4096", 9000, 4096, 1024, 4096, "blackmanharris"),
        ("nfft_gt_pow2", 2000, 400, 160, 1024, "hann"),
        ("tiny", 37, 8, 3, 8, "hann"),
    ]


FLAG_CASES = ("base", "oddL", "hop_gt_L", "N_eq_L", "tiny")


def flags_fixture():
    """ref_chain_flags_{s,d}.npz: the amp / lg outputs of stft.  c/stft.c:86-95 applies sqrtf / logf / logf(sqrtf())
    to the finished power row; the reference's own compiled pow_compress_s/_d (c/pow_compress.c:28-47) does exactly
    that for (p, preg) = (0.5, 0) / (0, 0) / one after the other, so its outputs on the executed chain's power rows
    are the reference's amp / lg rows.  Inputs are the cases of ref_chain_*.npz (same seeds)."""
    assert O.ref() is not None, "oracle/_ref/libdspref.so missing (make -C oracle)"
    for dt, tag in ((np.float32, "s"), (np.float64, "d")):
        d = {}
        for ci, (name, N, L, stp, nfft, wname) in enumerate(chain_cases()):
            if name not in FLAG_CASES:
                continue
            rng = np.random.default_rng(1000 + ci)
            x = rng.standard_normal(N).astype(dt)
            h = O.ref_window(wname, L, dt)
            d[f"{name}.x"] = x
            d[f"{name}.win"] = h
            d[f"{name}.params"] = np.array([N, L, stp, nfft], dtype=np.int64)
            for snip in (0, 1):
                if snip and L > N:
                    continue
                rc, wu = O.ref_window_univar(x, h, stp, snip)
                assert rc == 0
                rc, z = O.ref_fft_rad2(wu, nfft)
                assert rc == 0
                rc, p = O.ref_fft_power(z)
                assert rc == 0
                rc, a = O.pow_compress(p, 0.5, 0.0, use_ref=True)
                assert rc == 0
                rc, lg = O.pow_compress(p, 0.0, 0.0, use_ref=True)
                assert rc == 0
                rc, al = O.pow_compress(a, 0.0, 0.0, use_ref=True)
                assert rc == 0
                d[f"{name}.snip{snip}.power"] = p
                d[f"{name}.snip{snip}.amp"] = a
                d[f"{name}.snip{snip}.lg"] = lg
                d[f"{name}.snip{snip}.amplg"] = al
        np.savez_compressed(os.path.join(OUT, f"ref_chain_flags_{tag}.npz"), **d)


def main():
    assert O.ref() is not None, "oracle/_ref/libdspref.so missing (make -C oracle)"
    os.makedirs(OUT, exist_ok=True)
    for dt, tag in ((np.float32, "s"), (np.float64, "d")):
        d = {}
        for ci, (name, N, L, stp, nfft, wname) in enumerate(chain_cases()):
            rng = np.random.default_rng(1000 + ci)
            x = rng.standard_normal(N).astype(dt)
            h = O.ref_window(wname, L, dt)
            d[f"{name}.x"] = x
            d[f"{name}.win"] = h
            d[f"{name}.params"] = np.array([N, L, stp, nfft], dtype=np.int64)
            for snip in (0, 1):
                if snip and L > N:
                    continue
                rc, fr = O.ref_frame_univar(x, L, stp, snip)
                assert rc == 0
                rc, wn = O.ref_apply_win(fr, h, 1, False)
                assert rc == 0
                rc, wu = O.ref_window_univar(x, h, stp, snip)
                assert rc == 0 and np.array_equal(wu, wn)
                rc, z = O.ref_fft_rad2(wn, nfft)
                assert rc == 0
                rc, p = O.ref_fft_power(z)
                assert rc == 0
                if name in ("base", "tiny", "short_multifold"):      # intermediates only where they are small
                    d[f"{name}.snip{snip}.frames"] = fr
                    d[f"{name}.snip{snip}.windowed"] = wn
                    d[f"{name}.snip{snip}.spectrum"] = z
                d[f"{name}.snip{snip}.power"] = p
        np.savez_compressed(os.path.join(OUT, f"ref_chain_{tag}.npz"), **d)

        d = {}
        rng = np.random.default_rng(77)
        N, L, nfft = 2000, 400, 512
        x = rng.standard_normal(N).astype(dt)
        h = O.ref_window("povey", L, dt)
        d["x"] = x
        d["win"] = h
        cases = [(0.0, 160.0, 14), (300.0, 160.0, 12), (-50.0, 160.0, 14), (200.0, 160.0, 13), (-700.0, 133.37, 20),
                 (200.0, 160.0, 20), (1999.0, 0.75, 9)]
        d["cases"] = np.array(cases, dtype=np.float64)
        for i, (c0, stp, W) in enumerate(cases):
            W = int(W)
            rc, fr = O.ref_frame_univar_flt(x, L, W, c0, stp)
            assert rc == 0
            rc, wn = O.ref_window_univar_flt(x, h, W, c0, stp)
            assert rc == 0
            rc, z = O.ref_fft_rad2(wn, nfft)
            rc, p = O.ref_fft_power(z)
            if i in (0, 4):
                d[f"case{i}.frames"] = fr
                d[f"case{i}.windowed"] = wn
            d[f"case{i}.power"] = p
        np.savez_compressed(os.path.join(OUT, f"ref_flt_{tag}.npz"), **d)

        d = {}
        for wname in ("hann", "hamming", "povey", "blackmanharris"):
            for L in (1, 2, 7, 400, 401):
                d[f"{wname}.{L}"] = O.ref_window(wname, L, dt)
            for L in (7, 400, 401):
                for norm in (1, 2, 3):
                    d[f"{wname}.{L}.n{norm}"] = O.ref_window(wname, L, dt, norm)
        d["blackmanharris.4096"] = O.ref_window("blackmanharris", 4096, dt)
        np.savez_compressed(os.path.join(OUT, f"ref_windows_{tag}.npz"), **d)

        # the other generators: keys name.L.nK, parameter variants name@par.L.nK, blackman exact: blackman@1
        d = {}
        variants = [("blackman", None, 0), ("blackman", None, 1), ("bartlett", None, 0), ("triangular", None, 0),
                    ("rectangular", None, 0), ("flattop", None, 0), ("gauss", 2.5, 0), ("gauss", 0.7, 0),
                    ("tukey", 0.5, 0), ("tukey", 0.0, 0), ("tukey", 1.0, 0), ("tukey", 0.37, 0),
                    ("planck", 0.1, 0), ("planck", 0.25, 0), ("planck", 0.45, 0)]
        for wname, par, exact in variants:
            tagname = wname + (f"@{par}" if par is not None else ("@1" if exact else ""))
            for L in (1, 2, 3, 7, 8, 400, 401):
                if wname == "planck" and L % 2 and int(np.ceil(par * L)) > L // 2:
                    continue                       # the reference leaves the centre unwritten there
                for norm in ((0, 1, 2, 3) if L in (7, 400, 401) else (0,)):
                    d[f"{tagname}.{L}.n{norm}"] = O.ref_window(wname, L, dt, norm, par, exact)
        np.savez_compressed(os.path.join(OUT, f"ref_windows2_{tag}.npz"), **d)

        d = {}
        hz = O.get_stft_freqs(512, 257, 16000.0, dt, use_ref=True)[1][1:]
        scales = ("hz", "sqrt", "cbrt", "octave")
        d["hz"] = hz
        for a in scales:
            rc, xa = O.ref_convert_freqs(hz, "hz", a)
            assert rc == 0
            for b in scales:
                rc, y = O.ref_convert_freqs(xa, a, b)
                assert rc == 0
                d[f"{a}.{b}"] = y
            d[f"in.{a}"] = xa
        np.savez_compressed(os.path.join(OUT, f"ref_convert_freqs_{tag}.npz"), **d)

        # complex FFT rows through the reference's own radix-2 code (fft_rad2_c/_z)
        d = {}
        cdt = np.complex64 if tag == "s" else np.complex128
        for n in (2, 8, 64, 512):
            xc = (rng.standard_normal((5, n)) + 1j * rng.standard_normal((5, n))).astype(cdt)
            rc, yc = O.ref_fft_rad2_c(xc)
            assert rc == 0
            d[f"x.{n}"] = xc
            d[f"y.{n}"] = yc
        np.savez_compressed(os.path.join(OUT, f"ref_fft_c_{tag}.npz"), **d)

        # window_univar_c/z and window_univar_flt_c/z on a complex signal with a real window.  After a
        # frame that lies inside the signal window_univar_c/z steps back L-stp floats, not complex values
        # (c/window_univar.c:156-159,184-186), and reads past the signal from then on: only hop = L and
        # signals shorter than the window (every frame reflected) give defined output
        d = {}
        rngc = np.random.default_rng(31)
        xc = (rngc.standard_normal(700) + 1j * rngc.standard_normal(700)).astype(cdt)
        h = O.ref_window("hamming", 48, dt)
        d["x"] = xc; d["win"] = h
        for stp in (48,):
            rc, y = O.ref_window_univar(xc, h, stp, False)
            assert rc == 0
            d[f"wu.stp{stp}"] = y
        rc, y = O.ref_window_univar(xc[:30], h, 20, False)      # N < L: repeated reflection
        assert rc == 0
        d["wu.short"] = y
        for i, (W, c0, stp) in enumerate([(30, -10.0, 17.3), (40, 24.0, 20.0), (25, 600.0, 9.5)]):
            rc, y = O.ref_window_univar_flt(xc, h, W, c0, stp)
            assert rc == 0
            d[f"wuf.{i}"] = y
            d[f"wuf.{i}.par"] = np.array([W, c0, stp])
        np.savez_compressed(os.path.join(OUT, f"ref_wu_c_{tag}.npz"), **d)

        # complex framing / windowing (frame_univar_c/z, frame_univar_flt_c/z, apply_win_c/z in place form)
        d = {}
        rng = np.random.default_rng(5)
        cdt = np.complex64 if dt == np.float32 else np.complex128
        xc = (rng.standard_normal(500) + 1j * rng.standard_normal(500)).astype(cdt)
        d["x"] = xc
        for snip in (0, 1):
            rc, fr = O.ref_frame_univar(xc, 48, 20, snip)
            assert rc == 0
            d[f"frames.snip{snip}"] = fr
        rc, fr = O.ref_frame_univar_flt(xc, 48, 30, -10.0, 17.3)
        assert rc == 0
        d["frames_flt"] = fr
        # apply_win on 2-D real, both majors, both dims (the cases where the reference's strided branch is sound)
        a = rng.standard_normal((6, 5)).astype(dt)
        d["aw.x"] = a
        w0 = rng.standard_normal(6).astype(dt)
        w1 = rng.standard_normal(5).astype(dt)
        d["aw.w0"] = w0
        d["aw.w1"] = w1
        for col in (0, 1):
            rc, y0 = O.ref_apply_win(a, w0, 0, bool(col))
            rc, y1 = O.ref_apply_win(a, w1, 1, bool(col))
            d[f"aw.dim0.col{col}"] = np.ascontiguousarray(y0)
            d[f"aw.dim1.col{col}"] = np.ascontiguousarray(y1)
        np.savez_compressed(os.path.join(OUT, f"ref_misc_{tag}.npz"), **d)
    flags_fixture()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "flags":      # only the amp / lg fixtures (the others stay as committed)
        flags_fixture()
    else:
        main()
