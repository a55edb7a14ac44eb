"""GPU parity tests: every entry point of libdsp_stft_hip.so, called through the C ABI, against the CPU
oracle on the same seeded inputs, and against the fixtures made from executed reference code.

Tolerance of the path (stated once, used everywhere): per-frame-normalised max error
    max_f |Y - Yref| / max_f |Yref|  <=  1e-5 (float32), 1e-12 (float64)
for power and amplitude outputs.  Log outputs are compared twice: after exp() with the same bound (an
element-wise bound in the log domain is meaningless on bins that are 1e-12 of the frame maximum, and
a bin that is pure rounding residue, such as DC after mean removal, may be exactly 0 -> -inf in one
implementation and 1e-12 of the maximum in the other), and in the log domain itself on the bins that
reach LOG_FLOOR = 1e-3 of their frame's maximum, where an error of tol * max in the value is an
absolute error of at most tol / LOG_FLOOR in its logarithm: 1e-2 (float32) / 1e-9 (float64); log of
an all-zero frame must be -inf (test_log_of_silence_is_minus_inf).  Copies (framing) are bit-exact.
"""
import os

import numpy as np
import pytest

from conftest import TOL, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    import dsp_amd
    return dsp_amd


LOG_FLOOR = 1e-3


def check_log_domain(Y, Yref, tol):
    """Absolute error of log outputs on the bins within LOG_FLOOR of their frame's maximum (log of the value or of its
    square root alike: the bound on the logarithm of a power is twice that of its root, the larger one is used)."""
    Y = Y.astype(np.float64).reshape(-1, Y.shape[-1])
    Yref = Yref.astype(np.float64).reshape(-1, Yref.shape[-1])
    with np.errstate(invalid="ignore"):
        top = np.max(np.where(np.isfinite(Yref), Yref, -np.inf), axis=-1, keepdims=True)
        loud = np.isfinite(Yref) & (Yref >= top + np.log(LOG_FLOOR))
    if not loud.any():
        return 0.0
    err = float(np.max(np.abs(Y[loud] - Yref[loud])))
    assert err <= tol / LOG_FLOOR, f"log-domain error {err:.3e} > {tol / LOG_FLOOR:.1e} on bins >= {LOG_FLOOR} of the frame maximum"
    return err


def check_spec(O, Y, Yref, amp=False, lg=False, tol=None):
    assert Y.shape == Yref.shape and Y.dtype == Yref.dtype
    tol = TOL[Y.dtype] if tol is None else tol
    if lg:
        assert not np.any(np.isposinf(Y)) and not np.any(np.isposinf(Yref))
        check_log_domain(Y, Yref, tol)
        with np.errstate(over="ignore"):
            Y = np.exp(Y.astype(np.float64))
            Yref = np.exp(Yref.astype(np.float64))
    assert np.all(np.isfinite(Y) == np.isfinite(Yref))
    Y = np.where(np.isfinite(Y), Y, 0)
    Yref = np.where(np.isfinite(Yref), Yref, 0)
    err = O.normalised_err(Y, Yref)
    assert err <= tol, f"normalised error {err:.3e} > {tol:.1e}"
    return err


@pytest.mark.parametrize("tag", ["s", "d"])
def test_stft_amp_lg_against_reference_fixtures(O, D, tag):
    """stft with -a / -l / both against rows made by EXECUTED reference code: the reference's chain for the power
    (window_univar -> fft_rad2 -> fft_power) followed by its compiled pow_compress with (p, preg) = (0.5, 0), (0, 0)
    and one after the other, which is what c/stft.c:86-95 does to a power row (sqrtf, logf, logf(sqrtf())).  The
    power rows of the fixture carry the error of the reference's own radix-2 FFT (3e-6 / 2e-14 of the frame maximum),
    so: amplitude rows on the bins >= 1e-3 of the frame's largest power within err_p / (2 sqrt(1e-3)) of the largest
    amplitude, log rows there within err_p / 1e-3 absolute."""
    g = load_golden(f"ref_chain_flags_{tag}.npz")
    perr = {"s": 3e-6, "d": 2e-14}[tag] + TOL[g["base.x"].dtype]
    for name in ("base", "oddL", "hop_gt_L", "N_eq_L", "tiny"):
        x, win = g[f"{name}.x"], g[f"{name}.win"]
        N, L, stp, nfft = (int(v) for v in g[f"{name}.params"])
        for snip in (0, 1):
            if f"{name}.snip{snip}.power" not in g:
                continue
            P = g[f"{name}.snip{snip}.power"].astype(np.float64)
            loud = P >= LOG_FLOOR * P.max(axis=-1, keepdims=True)
            for key, amp, lg in (("amp", 1, 0), ("lg", 0, 1), ("amplg", 1, 1)):
                R = g[f"{name}.snip{snip}.{key}"].astype(np.float64)
                Y = D.stft(x, win, stp, nfft=nfft, snip_edges=bool(snip), amp=bool(amp), lg=bool(lg)).astype(np.float64)
                assert Y.shape == R.shape
                if lg:
                    err = np.max(np.abs(Y[loud] - R[loud]))
                    assert err <= perr / LOG_FLOOR, (name, snip, key, err)
                else:
                    err = np.max(np.abs(Y - R)[loud] / np.broadcast_to(R.max(axis=-1, keepdims=True), R.shape)[loud])
                    assert err <= perr / (2 * np.sqrt(LOG_FLOOR)), (name, snip, key, err)


def signal(seed, N, dt):
    return np.random.default_rng(seed).standard_normal(N).astype(dt)


CASES = [
    # N, L, stp, nfft
    (2000, 400, 160, 512),
    (16000, 400, 160, 512),
    (300, 400, 160, 512),        # N < L: repeated reflection
    (2000, 401, 160, 512),       # odd L
    (2001, 400, 161, 512),       # odd hop: unaligned frame starts
    (9000, 512, 700, 512),       # hop > nfft
    (3000, 300, 100, 512),       # nfft > nextpow2(L) is allowed by the C API
    (2000, 1, 7, 512),
    (2000, 64, 100, 64),
    (400, 400, 160, 512),
    (5000, 400, 160, 1024),
    (9000, 4096, 1024, 4096),
    (1000, 30, 10, 32),
    (100, 2, 1, 2),
    (500, 3, 2, 4),
    (2000, 300, 160, 300),       # even, not a power of two: direct family
    (600, 45, 20, 45),           # odd nfft: the reference's own pairing of the half-complex array
    (40000, 16384, 5000, 16384),
]


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d_L%d_s%d_n%d" % c)
def test_stft_matches_oracle(O, D, dt, case):
    N, L, stp, nfft = case
    x = signal(hash(case) % 1000, N, dt)
    win = (np.hanning(L) + 0.1).astype(dt) if L > 2 else np.ones(L, dt)
    for snip in (0, 1):
        if snip and L > N:
            continue
        rc, Yref = O.stft(x, win, stp, nfft, snip)
        assert rc == 0
        Y = D.stft(x, win, stp, bool(snip), nfft=nfft)
        check_spec(O, Y, Yref)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("mn0,amp,lg", [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1), (0, 1, 1)])
def test_stft_flags(O, D, dt, generic, mn0, amp, lg):
    x = signal(5, 16000, dt) + dt(0.3)
    win = D.hann(400, dt)
    D.force_generic(generic)
    try:
        for snip in (0, 1):
            rc, Yref = O.stft(x, win, 160, 512, snip, mn0, amp, lg)
            Y = D.stft(x, win, 160, bool(snip), bool(mn0), bool(amp), bool(lg))
            check_spec(O, Y, Yref, amp, lg)
    finally:
        D.force_generic(False)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_both_512_families_agree_with_oracle(O, D, dt):
    """nfft 512 has two kernel families; each must meet the bound on its own (W not a multiple of 4,
    hops at, below and above the span-staging limit of 512, odd hops, frame starts that are not 16-byte aligned)."""
    for (N, L, stp) in [(7013, 400, 160), (7013, 512, 33), (30000, 320, 1500), (1900, 400, 160), (700, 400, 160),
                        (9001, 400, 170), (9001, 402, 160), (9001, 399, 166),    # spans that start off 16-byte boundaries
                        (9001, 512, 256), (9001, 400, 200), (5000, 512, 172), (9001, 300, 254),   # spans of 1028 to 1280 samples
                        (9001, 400, 161), (9001, 512, 255), (2000, 100, 1),                       # odd hops
                        (9001, 401, 257), (20001, 400, 320), (20001, 512, 512), (20001, 480, 511), (12000, 400, 400),   # spans of up to 2048 samples
                        (20001, 512, 513), (30000, 100, 777)]:                                    # above the span-staging limit
        x = signal(N, N, dt)
        win = D.povey(L, dt)
        rc, Yref = O.stft(x, win, stp, 512, 0)
        for generic in (False, True):
            D.force_generic(generic)
            try:
                Y = D.stft(x, win, stp, nfft=512)
            finally:
                D.force_generic(False)
            check_spec(O, Y, Yref)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_both_power_of_two_families_agree_with_oracle(O, D, dt):
    """Power-of-two nfft: the Stockham kernels (family 5: nfft 16 to 2048; one wave, part of a wave or a workgroup per
    frame; 64 to 4096 and float32 8192 have kernels of their own in front of it) and the generic kernel
    (family 2) must each meet the bound:
    every size, short and odd windows, odd hops, a signal shorter than the window, every flag."""
    top = 8192 if dt == np.float32 else 4096
    assert D.kernel_family(16, dt) == 5 and D.kernel_family(2 * top, dt) == 2
    assert D.kernel_family(8192, dt) == (8 if dt == np.float32 else 2)
    cases = [(nfft, nfft, max(1, nfft // 4), (0, 0, 0)) for nfft in (16, 32, 64, 128, 256, 1024, 2048, 4096, 8192) if nfft <= top]
    cases += [(top, top - 37, top // 3 + 1, (1, 1, 0)), (2048, 1500, 511, (1, 0, 1))]
    cases += [(16, 9, 3, (1, 0, 0)), (64, 33, 7, (0, 1, 0)), (256, 200, 80, (1, 1, 1)), (1024, 801, 321, (1, 0, 1)),
              (1024, 1, 500, (0, 0, 0)), (128, 128, 1000, (0, 1, 1))]
    for (nfft, L, stp, flags) in cases:
        for N in (20 * L + 13, max(2, L // 2)):
            x = signal(nfft + N, N, dt) + dt(0.1)
            win = D.povey(L, dt)
            for snip in (0, 1):
                if snip and L > N:
                    continue
                if nfft >= 2048:      # the long-double DFT oracle is O(nfft^2) per frame: float64 FFT oracle instead
                    rc, Yref = O.stft(x.astype(np.float64), win.astype(np.float64), stp, nfft, snip, *flags, exact=False)
                    Yref = Yref.astype(dt)
                else:
                    rc, Yref = O.stft(x, win, stp, nfft, snip, *flags)
                assert rc == 0
                for generic in (False, True):
                    D.force_generic(generic)
                    try:
                        Y = D.stft(x, win, stp, bool(snip), *[bool(f) for f in flags], nfft=nfft)
                    finally:
                        D.force_generic(False)
                    check_spec(O, Y, Yref, flags[1], flags[2])


def test_256_1024_2048_kernels_agree_with_oracle(O, D):
    """nfft 64, 128 and 256 (family 7), 1024 and 2048 (family 6), float32, have kernels of their own: full, short and
    odd windows, hops of every alignment up to the span limit and past it (where the family-5 kernel takes over),
    odd frame counts (a last step with one frame), a signal shorter than the window, every flag, both edge
    rules, and outputs that start 4, 8 and 12 bytes off a 16-byte boundary."""
    import torch
    dt = np.float32
    assert D.kernel_family(1024, dt) == 6 and D.kernel_family(2048, dt) == 6
    assert D.kernel_family(256, dt) == 7 and D.kernel_family(128, dt) == 7 and D.kernel_family(64, dt) == 7 and D.kernel_family(32, dt) == 5
    shapes = [(128, 128, 32, (0, 0, 0)), (128, 100, 40, (1, 0, 0)), (128, 127, 17, (0, 1, 0)), (128, 65, 128, (1, 1, 1)), (128, 104, 73, (0, 0, 1)),
              (128, 97, 129, (0, 0, 0)), (128, 2, 1, (1, 0, 0)), (128, 128, 64, (0, 0, 0)),
              (64, 64, 16, (0, 0, 0)), (64, 50, 20, (1, 0, 0)), (64, 63, 9, (0, 1, 0)), (64, 33, 47, (1, 1, 1)), (64, 52, 48, (0, 0, 1)),
              (64, 49, 64, (0, 0, 0)), (64, 2, 1, (1, 0, 0)), (64, 64, 32, (0, 1, 1)),
              (256, 256, 64, (0, 0, 0)), (256, 200, 80, (1, 0, 0)), (256, 255, 33, (0, 1, 0)), (256, 129, 256, (1, 1, 1)), (256, 208, 146, (0, 0, 1)),
              (256, 193, 257, (0, 0, 0)), (256, 2, 1, (1, 0, 0)),
              (1024, 1024, 256, (0, 0, 0)), (1024, 1024, 255, (1, 0, 0)), (1024, 800, 200, (0, 1, 0)), (1024, 513, 1024, (1, 1, 1)),
              (1024, 1000, 1025, (0, 0, 0)), (1024, 3, 77, (1, 0, 1)), (1024, 1024, 1, (0, 0, 0)),
              (2048, 2048, 512, (0, 0, 0)), (2048, 2047, 441, (1, 0, 0)), (2048, 1200, 300, (0, 1, 1)), (2048, 1025, 5000, (1, 1, 0)),
              (2048, 2048, 1, (0, 0, 1))]
    for (nfft, L, stp, flags) in shapes:
        for N in (max(12 * nfft + 77, 3001) if stp > 1 else nfft + 300, L // 2 + 1):
            x = signal(nfft + N + stp, N, dt) + dt(0.15)
            win = D.povey(L, dt)
            for snip in (0, 1):
                if snip and L > N:
                    continue
                rc, Yref = O.stft(x.astype(np.float64), win.astype(np.float64), stp, nfft, snip, *flags, exact=False)
                assert rc == 0
                Y = D.stft(x, win, stp, bool(snip), *[bool(f) for f in flags], nfft=nfft)
                check_spec(O, Y, Yref.astype(dt), flags[1], flags[2])
    # rows that start off 16-byte boundaries, several signals, odd and even frame counts
    for (nfft, stp, N) in [(64, 16, 3000), (64, 20, 2777), (128, 40, 5000), (128, 32, 4157),
                           (256, 80, 5000), (256, 64, 4157), (1024, 256, 9000), (1024, 300, 9301), (2048, 512, 20000)]:
        xs = np.stack([signal(10 * k + nfft, N, dt) for k in range(3)])
        win = D.hann(nfft, dt)
        want = np.stack([O.stft(r.astype(np.float64), win.astype(np.float64), stp, nfft, 0, exact=False)[1] for r in xs]).astype(dt)
        xd, wd = torch.from_numpy(xs).cuda(), torch.from_numpy(win).cuda()
        for off in (0, 1, 2, 3):
            buf = torch.full((want.size + 8,), 7.0, dtype=torch.float32, device="cuda")
            out = buf[off:off + want.size].view(want.shape)
            D.stft_dev(xd, wd, stp, nfft=nfft, out=out)
            torch.cuda.synchronize()
            check_spec(O, out.cpu().numpy(), want)
            assert float(buf[:off].sum()) == 7.0 * off and bool((buf[off + want.size:] == 7.0).all())   # nothing outside the rows
    # a full grid of workgroups: every launch gives the same bits (the span slots are refilled behind the reads of the
    # step before), and interior frames do not depend on where a signal lies in the batch
    for (nfft, stp) in [(64, 16), (128, 40), (256, 80), (1024, 256), (2048, 512)]:
        xd = torch.randn((64, 1 << 20), device="cuda", generator=torch.Generator(device="cuda").manual_seed(nfft))
        wd = torch.from_numpy(D.hann(nfft, dt)).cuda()
        y0 = D.stft_dev(xd, wd, stp, nfft=nfft)
        ref_sum = int(y0.view(torch.int32).to(torch.int64).sum())
        y1 = torch.empty_like(y0)
        for _ in range(10):
            D.stft_dev(xd, wd, stp, nfft=nfft, out=y1)
            assert int(y1.view(torch.int32).to(torch.int64).sum()) == ref_sum
        assert torch.equal(y0, y1)
        alone = D.stft_dev(xd[37:38].clone(), wd, stp, nfft=nfft)
        assert torch.equal(alone[0], y0[37])
        del xd, y0, y1
    # zero edges and a first frame off the signal (stft_flt with an integral hop); a non-integral hop goes to family 5
    x = signal(78, 30000, dt)
    # (the hops that scale with nfft stay inside the span limits of the small kernels: integral ones by hop, the others by
    # a table of frame starts)
    for nfft in (64, 128, 256, 1024, 2048):
        win = D.blackmanharris(nfft, dt)
        for intended in (False, True):
            D.set_flt_mode(intended)
            try:
                for (c0, stp, W) in [(0.0, 256.0, 31), (700.0, 300.0, 40), (-1500.0, 512.0, 20), (100.0, 250.5, 20),
                                     (0.0, nfft / 4.0, 75), (-3.0 * nfft, nfft / 2.0, 41), (100.0, nfft / 4.0 + 0.5, 67), (29000.0, nfft / 4.0 + 0.25, 70)]:
                    rc, Yref = O.stft_flt(x.astype(np.float64), win.astype(np.float64), W, c0, stp, nfft, 0, 0, 0, intended=intended, exact=False)
                    assert rc == 0
                    Y = D.stft_flt(x, win, W, c0, stp, nfft=nfft)
                    check_spec(O, Y, Yref.astype(dt))
            finally:
                D.set_flt_mode(False)


def test_64_128_256_kernels_float64(O, D):
    """The float64 form of the nfft 64 / 128 / 256 kernel (family 7; 4 waves per workgroup, rows stored element by
    element): the shapes of the float32 test (full, short and odd windows, hops of every alignment up to the span limit and
    past it, every flag, both edge rules, a signal shorter than the window), outputs at both 8-byte alignments with the
    bytes around them untouched, bit-equal repeats on a full grid, stft_flt with whole and fractional hops in both framings."""
    import torch
    dt = np.float64
    assert D.kernel_family(256, dt) == 7 and D.kernel_family(128, dt) == 7 and D.kernel_family(64, dt) == 7 and D.kernel_family(32, dt) == 5
    shapes = [(256, 256, 64, (0, 0, 0)), (256, 200, 80, (1, 0, 0)), (256, 255, 33, (0, 1, 0)), (256, 129, 256, (1, 1, 1)), (256, 208, 146, (0, 0, 1)),
              (256, 193, 257, (0, 0, 0)), (256, 2, 1, (1, 0, 0)),
              (128, 128, 32, (0, 0, 0)), (128, 100, 40, (1, 0, 0)), (128, 127, 17, (0, 1, 0)), (128, 65, 128, (1, 1, 1)), (128, 104, 73, (0, 0, 1)),
              (128, 97, 129, (0, 0, 0)), (128, 2, 1, (1, 0, 0)), (128, 128, 64, (0, 0, 0)),
              (64, 64, 16, (0, 0, 0)), (64, 50, 20, (1, 0, 0)), (64, 63, 9, (0, 1, 0)), (64, 33, 47, (1, 1, 1)), (64, 52, 48, (0, 0, 1)),
              (64, 49, 64, (0, 0, 0)), (64, 2, 1, (1, 0, 0)), (64, 64, 32, (0, 1, 1))]
    for (nfft, L, stp, flags) in shapes:
        for N in (max(12 * nfft + 77, 3001) if stp > 1 else nfft + 300, L // 2 + 1):
            x = signal(nfft + N + stp, N, dt) + dt(0.15)
            win = D.povey(L, dt)
            for snip in (0, 1):
                if snip and L > N:
                    continue
                rc, Yref = O.stft(x, win, stp, nfft, snip, *flags)
                assert rc == 0
                Y = D.stft(x, win, stp, bool(snip), *[bool(f) for f in flags], nfft=nfft)
                check_spec(O, Y, Yref, flags[1], flags[2])
    for (nfft, stp, N) in [(64, 16, 3000), (64, 20, 2777), (128, 40, 5000), (128, 32, 4157), (256, 80, 5000), (256, 64, 4157)]:
        xs = np.stack([signal(10 * k + nfft, N, dt) for k in range(3)])
        win = D.hann(nfft, dt)
        want = np.stack([O.stft(r, win, stp, nfft, 0)[1] for r in xs])
        xd, wd = torch.from_numpy(xs).cuda(), torch.from_numpy(win).cuda()
        for off in (0, 1):
            buf = torch.full((want.size + 4,), 7.0, dtype=torch.float64, device="cuda")
            out = buf[off:off + want.size].view(want.shape)
            D.stft_dev(xd, wd, stp, nfft=nfft, out=out)
            torch.cuda.synchronize()
            check_spec(O, out.cpu().numpy(), want)
            assert float(buf[:off].sum()) == 7.0 * off and bool((buf[off + want.size:] == 7.0).all())
    for (nfft, stp) in [(64, 16), (128, 40), (256, 80)]:
        xd = torch.randn((64, 1 << 19), device="cuda", dtype=torch.float64, generator=torch.Generator(device="cuda").manual_seed(nfft))
        wd = torch.from_numpy(D.hann(nfft, dt)).cuda()
        y0 = D.stft_dev(xd, wd, stp, nfft=nfft)
        y1 = torch.empty_like(y0)
        for _ in range(5):
            D.stft_dev(xd, wd, stp, nfft=nfft, out=y1)
            assert torch.equal(y0, y1)
        alone = D.stft_dev(xd[37:38].clone(), wd, stp, nfft=nfft)
        assert torch.equal(alone[0], y0[37])
        del xd, y0, y1
    x = signal(78, 30000, dt)
    for nfft in (64, 128, 256):
        win = D.blackmanharris(nfft, dt)
        for intended in (False, True):
            D.set_flt_mode(intended)
            try:
                for (c0, stp, W) in [(0.0, nfft / 4.0, 75), (-3.0 * nfft, nfft / 2.0, 41), (100.0, nfft / 4.0 + 0.5, 67), (29000.0, nfft / 4.0 + 0.25, 70),
                                     (700.0, 300.0, 40)]:
                    rc, Yref = O.stft_flt(x, win, W, c0, stp, nfft, 0, 0, 0, intended=intended)
                    assert rc == 0
                    Y = D.stft_flt(x, win, W, c0, stp, nfft=nfft)
                    check_spec(O, Y, Yref)
            finally:
                D.set_flt_mode(False)


def test_1024_2048_kernels_float64(O, D):
    """The float64 form of the nfft 1024 / 2048 kernel (family 6; 4 waves per workgroup, lane moves in two halves, rows
    stored element by element): the shapes of the float32 test (full, short and odd windows, hops of every alignment up to
    the span limit and past it, every flag, both edge rules, a signal shorter than the window), outputs at both 8-byte
    alignments with the bytes around them untouched, bit-equal repeats on a full grid, stft_flt in both framings."""
    import torch
    dt = np.float64
    assert D.kernel_family(1024, dt) == 6 and D.kernel_family(2048, dt) == 6
    shapes = [(1024, 1024, 256, (0, 0, 0)), (1024, 1024, 255, (1, 0, 0)), (1024, 800, 200, (0, 1, 0)), (1024, 513, 1024, (1, 1, 1)),
              (1024, 1000, 1025, (0, 0, 0)), (1024, 3, 77, (1, 0, 1)), (1024, 1024, 1, (0, 0, 0)), (1024, 1024, 128, (0, 0, 1)),
              (2048, 2048, 512, (0, 0, 0)), (2048, 2047, 441, (1, 0, 0)), (2048, 1200, 300, (0, 1, 1)), (2048, 1025, 5000, (1, 1, 0)),
              (2048, 2048, 1, (0, 0, 1))]
    for (nfft, L, stp, flags) in shapes:
        for N in ((12 * nfft + 77) if stp > 1 else nfft + 300, L // 2 + 1):
            x = signal(nfft + N + stp, N, dt) + dt(0.15)
            win = D.povey(L, dt)
            for snip in (0, 1):
                if snip and L > N:
                    continue
                rc, Yref = O.stft(x, win, stp, nfft, snip, *flags, exact=False)
                assert rc == 0
                Y = D.stft(x, win, stp, bool(snip), *[bool(f) for f in flags], nfft=nfft)
                check_spec(O, Y, Yref, flags[1], flags[2])
    for (nfft, stp, N) in [(1024, 256, 9000), (1024, 300, 9301), (2048, 512, 20000)]:
        xs = np.stack([signal(10 * k + nfft, N, dt) for k in range(3)])
        win = D.hann(nfft, dt)
        want = np.stack([O.stft(r, win, stp, nfft, 0, exact=False)[1] for r in xs])
        xd, wd = torch.from_numpy(xs).cuda(), torch.from_numpy(win).cuda()
        for off in (0, 1):
            buf = torch.full((want.size + 4,), 7.0, dtype=torch.float64, device="cuda")
            out = buf[off:off + want.size].view(want.shape)
            D.stft_dev(xd, wd, stp, nfft=nfft, out=out)
            torch.cuda.synchronize()
            check_spec(O, out.cpu().numpy(), want)
            assert float(buf[:off].sum()) == 7.0 * off and bool((buf[off + want.size:] == 7.0).all())
    for (nfft, stp) in [(1024, 256), (2048, 512)]:
        xd = torch.randn((64, 1 << 19), device="cuda", dtype=torch.float64, generator=torch.Generator(device="cuda").manual_seed(nfft))
        wd = torch.from_numpy(D.hann(nfft, dt)).cuda()
        y0 = D.stft_dev(xd, wd, stp, nfft=nfft)
        y1 = torch.empty_like(y0)
        for _ in range(5):
            D.stft_dev(xd, wd, stp, nfft=nfft, out=y1)
            assert torch.equal(y0, y1)
        alone = D.stft_dev(xd[37:38].clone(), wd, stp, nfft=nfft)
        assert torch.equal(alone[0], y0[37])
        del xd, y0, y1
    x = signal(78, 30000, dt)
    for nfft in (1024, 2048):
        win = D.blackmanharris(nfft, dt)
        for intended in (False, True):
            D.set_flt_mode(intended)
            try:
                for (c0, stp, W) in [(0.0, 256.0, 31), (700.0, 300.0, 40), (-1500.0, 512.0, 20), (100.0, 250.5, 20), (100.0, nfft / 4.0 + 0.5, 67)]:
                    rc, Yref = O.stft_flt(x, win, W, c0, stp, nfft, 0, 0, 0, intended=intended, exact=False)
                    assert rc == 0
                    Y = D.stft_flt(x, win, W, c0, stp, nfft=nfft)
                    check_spec(O, Y, Yref)
            finally:
                D.set_flt_mode(False)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_both_4096_families_agree_with_oracle(O, D, dt):
    """nfft 4096 has a register-resident kernel beside the generic one (float64: 4 waves per workgroup, the window
    read from memory); each must meet the bound on its own: full and short windows, odd hop (unaligned frame
    starts), a signal shorter than the window (repeated reflection), every flag, and the zero-edged stft_flt framing."""
    assert D.kernel_family(4096, dt) == 4
    for (N, L, stp, flags) in [(40000, 4096, 1024, (0, 0, 0)), (40001, 4096, 1023, (1, 0, 0)), (30000, 3000, 700, (0, 1, 0)),
                               (30000, 2049, 512, (1, 1, 1)), (3000, 4096, 1024, (0, 0, 1)), (20000, 100, 5000, (1, 0, 0)),
                               (9000, 4095, 1, (0, 0, 0))]:
        x = signal(N + L, N, dt) + dt(0.2)
        win = D.blackmanharris(L, dt) if L > 3000 else D.povey(L, dt)
        if stp == 1:
            x = x[:4200]
        for snip in (0, 1):
            if snip and L > len(x):
                continue
            # float64 FFT oracle on the same (exactly representable) inputs
            rc, Yref = O.stft(x.astype(np.float64), win.astype(np.float64), stp, 4096, snip, *flags, exact=False)
            assert rc == 0
            for generic in (False, True):
                D.force_generic(generic)
                try:
                    Y = D.stft(x, win, stp, bool(snip), *[bool(f) for f in flags], nfft=4096)
                finally:
                    D.force_generic(False)
                check_spec(O, Y, Yref.astype(dt), flags[1], flags[2])
    x = signal(77, 30000, dt)
    win = D.blackmanharris(4096, dt)
    for intended in (False, True):
        D.set_flt_mode(intended)
        try:
            for (c0, stp, W) in [(0.0, 1024.0, 30), (3000.0, 1000.5, 20), (-6000.0, 2048.0, 20)]:
                rc, Yref = O.stft_flt(x.astype(np.float64), win.astype(np.float64), W, c0, stp, 4096, 0, 0, 0, intended=intended, exact=False)
                assert rc == 0
                Y = D.stft_flt(x, win, W, c0, stp, nfft=4096)
                check_spec(O, Y, Yref.astype(dt))
        finally:
            D.set_flt_mode(False)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_full_window_instantiations(O, D, dt):
    """L = nfft for nfft 64 ... 512: instantiations of the span-slot kernels without the lane masks of the rows that may reach past
    the window.  Hops nfft/4 and nfft/2 (the aligned span forms), an odd hop (which keeps the general instantiation), every flag,
    both framings, signals that end inside a step and shorter than the window; against the oracle, and bit for bit against the
    general instantiation (selected by the same window without its last tap, which is zero in both)."""
    import torch
    for nfft in (64, 128, 256, 512):
        win = D.hann(nfft, dt)
        wd = torch.from_numpy(win).cuda()
        for (B, N, snip, flags) in [(3, 40 * nfft + 13, 0, (0, 0, 0)), (2, 9 * nfft, 1, (1, 0, 0)), (1, nfft - 5, 0, (0, 1, 1)), (4, 17 * nfft + 1, 1, (1, 1, 1))]:
            x = np.stack([signal(5 * b + N + nfft, N, dt) + dt(0.1) for b in range(B)])
            xd = torch.from_numpy(x).cuda()
            for hop in (nfft // 4, nfft // 2, nfft // 4 + 1):
                if snip and N < nfft:
                    continue
                got = D.stft_dev(xd, wd, hop, bool(snip), *[bool(f) for f in flags], nfft=nfft).cpu().numpy()
                for b in (0, B - 1):
                    rc, Yref = O.stft(x[b].astype(np.float64), win.astype(np.float64), hop, nfft, snip, *flags, exact=False)
                    assert rc == 0
                    check_spec(O, got[b], Yref.astype(dt), flags[1], flags[2])
        # the general instantiation on the same taps: a window whose last tap is zero, with and without that tap
        wz = win.copy()
        wz[-1] = 0
        x = np.stack([signal(77 + b, 33 * nfft + 3, dt) for b in range(2)])
        xd = torch.from_numpy(x).cuda()
        # (frames from sample 0, so that the window length does not move them; N leaves both lengths the same number of frames)
        full = D.stft_dev(xd, torch.from_numpy(wz).cuda(), nfft // 4, True, nfft=nfft)
        gen = D.stft_dev(xd, torch.from_numpy(wz[:-1].copy()).cuda(), nfft // 4, True, nfft=nfft)
        assert full.shape == gen.shape and torch.equal(full, gen), nfft


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_row_count_instantiations_of_the_512_point_kernel(O, D, dt):
    """nfft 512 with windows of 8 to 15 rows of 32 samples (instantiations that skip the empty rows and mask only the last one that
    holds data), at the ends of the ranges and just outside them; narrow, wide and odd hops; every flag; NaN behind the
    window stays out."""
    import torch
    for L in (224, 225, 256, 257, 288, 289, 320, 321, 352, 353, 384, 385, 416, 417, 448, 449, 480, 481):
        win = D.povey(L, dt)
        wd = torch.from_numpy(win).cuda()
        for (hop, N, snip, flags) in [(160, 40 * 160 + L, 0, (0, 0, 0)), (256, 9000, 1, (1, 0, 0)), (161, 7001, 0, (1, 1, 1)), (100, 5000, 1, (0, 1, 0))]:
            x = np.stack([signal(L + b, N, dt) + dt(0.1) for b in range(3)])
            got = D.stft_dev(torch.from_numpy(x).cuda(), wd, hop, bool(snip), *[bool(f) for f in flags], nfft=512).cpu().numpy()
            for b in (0, 2):
                rc, Yref = O.stft(x[b].astype(np.float64), win.astype(np.float64), hop, 512, snip, *flags, exact=False)
                assert rc == 0
                check_spec(O, got[b], Yref.astype(dt), flags[1], flags[2])
        # a NaN right behind the window of frame 3 (snip_edges framing: frames start at w * hop)
        x = signal(L, 30 * 160 + 512, dt)
        x[3 * 160 + L] = np.nan
        Y = D.stft_dev(torch.from_numpy(x[None]).cuda(), wd, 160, True, nfft=512)[0].cpu().numpy()
        bad = np.isnan(Y).any(axis=1)
        assert not bad[3] and bad[4] and bad.sum() == len([w for w in range(Y.shape[0]) if w * 160 <= 3 * 160 + L < w * 160 + L]), L


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_row_count_instantiations_of_the_256_point_kernel(O, D, dt):
    """nfft 256, windows of 10, 13 and 15 rows of 16 samples, at both ends of each range and just outside it; NaN behind the
    window stays out."""
    import torch
    for L in (144, 145, 160, 161, 193, 208, 224, 225, 240, 241):
        win = D.povey(L, dt)
        wd = torch.from_numpy(win).cuda()
        for (hop, N, snip, flags) in [(80, 60 * 80 + L, 0, (0, 0, 0)), (64, 5000, 1, (1, 1, 0)), (81, 4001, 0, (1, 0, 1))]:
            x = np.stack([signal(L + b, N, dt) + dt(0.1) for b in range(3)])
            got = D.stft_dev(torch.from_numpy(x).cuda(), wd, hop, bool(snip), *[bool(f) for f in flags], nfft=256).cpu().numpy()
            for b in (0, 2):
                rc, Yref = O.stft(x[b].astype(np.float64), win.astype(np.float64), hop, 256, snip, *flags, exact=False)
                assert rc == 0
                check_spec(O, got[b], Yref.astype(dt), flags[1], flags[2])
        x = signal(L, 60 * 80 + 256, dt)
        x[5 * 80 + L] = np.nan
        Y = D.stft_dev(torch.from_numpy(x[None]).cuda(), wd, 80, True, nfft=256)[0].cpu().numpy()
        bad = np.isnan(Y).any(axis=1)
        assert not bad[5] and bad[6] and bad.sum() == len([w for w in range(Y.shape[0]) if w * 80 <= 5 * 80 + L < w * 80 + L]), L


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_row_count_instantiations_of_the_1024_point_kernel(O, D, dt):
    """nfft 1024, windows of 8 to 15 rows of 64 samples (one masked row, the rows above it not read), at the ends of the
    ranges and just outside them; the 1280- and 2048-sample span forms, an odd hop (general instantiation), compression flags;
    NaN and Inf behind the window of either frame of a step stay out."""
    import torch
    for L in (448, 449, 512, 513, 551, 576, 577, 640, 700, 768, 769, 800, 832, 833, 882, 896, 897, 960, 961):
        win = D.povey(L, dt)
        wd = torch.from_numpy(win).cuda()
        for (hop, N, snip, flags) in [(220, 30 * 220 + L, 0, (0, 0, 0)), (256, 9000, 1, (0, 1, 0)), (400, 12001, 0, (0, 1, 1)), (221, 8000, 1, (0, 0, 0))]:
            x = np.stack([signal(L + b, N, dt) + dt(0.1) for b in range(3)])
            got = D.stft_dev(torch.from_numpy(x).cuda(), wd, hop, bool(snip), *[bool(f) for f in flags], nfft=1024).cpu().numpy()
            for b in (0, 2):
                rc, Yref = O.stft(x[b].astype(np.float64), win.astype(np.float64), hop, 1024, snip, *flags, exact=False)
                assert rc == 0
                check_spec(O, got[b], Yref.astype(dt), flags[1], flags[2])
        x = signal(L, 40 * 220 + 1024, dt)
        x[4 * 220 + L] = np.nan            # right behind the window of frame 4 (the first frame of its step)
        x[11 * 220 + L + 1] = np.inf       # ... and one sample further behind frame 11 (the second frame of its step)
        Y = D.stft_dev(torch.from_numpy(x[None]).cuda(), wd, 220, True, nfft=1024)[0].cpu().numpy()
        bad = ~np.isfinite(Y).all(axis=1)
        want = np.array([any(w * 220 <= i < w * 220 + L for i in (4 * 220 + L, 11 * 220 + L + 1)) for w in range(Y.shape[0])])
        assert np.array_equal(bad, want), L


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_2048_ring_of_quarters(O, D, dt, monkeypatch):
    """nfft 2048 with a full window and hop 512 or 1024: large inputs are transformed in units of consecutive frames per wave, the
    span slot kept as a ring of four quarters of which only the new one is fetched per frame.  Forced here at small sizes
    (DSP_R512X_MIN_UNITS=1): units that cross either edge of the signal, a last unit shorter than the others, signals
    shorter than one unit and shorter than the window, several signals, every flag, both edge rules, the complex form;
    against the oracle and, bit for bit, against the step form of the same kernel."""
    import torch
    win = D.hann(2048, dt)
    wd = torch.from_numpy(win).cuda()
    tdt = torch.float32 if dt == np.float32 else torch.float64
    for (B, N, snip, flags) in [(3, 512 * 100 + 7, 0, (0, 0, 0)), (2, 512 * 37, 1, (1, 0, 0)), (1, 3000, 0, (0, 1, 1)), (5, 512 * 18 + 2048, 1, (0, 0, 0)),
                                (2, 512 * 7 + 100, 0, (1, 1, 0)), (1, 1500, 0, (0, 0, 1)), (4, 512 * 6 * 9, 0, (1, 1, 1))]:
        x = np.stack([signal(7 * b + N, N, dt) + dt(0.1) for b in range(B)])
        xd = torch.from_numpy(x).cuda()
        for hop in (512, 1024):
            got = {}
            for mu in ("1", "0"):
                monkeypatch.setenv("DSP_R512X_MIN_UNITS", mu)
                got[mu] = D.stft_dev(xd, wd, hop, bool(snip), *[bool(f) for f in flags], nfft=2048).cpu().numpy()
            assert np.array_equal(got["1"], got["0"]), (B, N, snip, flags, hop)
            for b in (0, B - 1):
                rc, Yref = O.stft(x[b].astype(np.float64), win.astype(np.float64), hop, 2048, snip, *flags, exact=False)
                assert rc == 0
                check_spec(O, got["1"][b], Yref.astype(dt), flags[1], flags[2])
    # the complex form, and an output that is a view with other rows around it
    x = np.stack([signal(31 + b, 512 * 50, dt) for b in range(2)])
    xd = torch.from_numpy(x).cuda()
    got = {}
    for mu in ("1", "0"):
        monkeypatch.setenv("DSP_R512X_MIN_UNITS", mu)
        got[mu] = D.stft_cplx_dev(xd, wd, 512, nfft=2048, sc=True)
    assert torch.equal(got["1"], got["0"])
    monkeypatch.setenv("DSP_R512X_MIN_UNITS", "1")
    want = D.stft_dev(xd, wd, 512, nfft=2048)
    buf = torch.full((want.numel() + 6,), 7.0, dtype=tdt, device="cuda")
    out = buf[3:3 + want.numel()].view(want.shape)
    D.stft_dev(xd, wd, 512, nfft=2048, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, want) and bool((buf[:3] == 7.0).all()) and bool((buf[3 + want.numel():] == 7.0).all())
    # stft_flt with an integral hop of 512 takes the same form (zero edges, read shift)
    xs = signal(5, 40000, dt)
    for intended in (False, True):
        D.set_flt_mode(intended)
        try:
            for (c0, W) in [(0.0, 60), (1024.0, 70), (-3000.0, 50)]:
                rc, Yref = O.stft_flt(xs.astype(np.float64), win.astype(np.float64), W, c0, 512.0, 2048, 0, 0, 0, intended=intended, exact=False)
                assert rc == 0
                Y = D.stft_flt(xs, win, W, c0, 512.0, nfft=2048)
                check_spec(O, Y, Yref.astype(dt))
        finally:
            D.set_flt_mode(False)


def test_8192_kernel_agrees_with_oracle(O, D):
    """nfft 8192 (float32) has a register-resident kernel (a 64 x 64 transform per wave, 4 waves per workgroup): full
    and short windows, odd hops (unaligned frame starts), a signal shorter than the window, every flag, a batch with
    outputs at every 4-byte alignment and untouched neighbours, bit-equal repeats on a full grid, and the zero-edged
    stft_flt framings (tables included)."""
    import torch
    dt = np.float32
    assert D.kernel_family(8192, dt) == 8
    for (N, L, stp, flags) in [(70000, 8192, 2048, (0, 0, 0)), (70001, 8192, 2047, (1, 0, 0)), (60000, 6000, 1400, (0, 1, 0)),
                               (60000, 4097, 1024, (1, 1, 1)), (6000, 8192, 2048, (0, 0, 1)), (40000, 100, 9000, (1, 0, 0)),
                               (9000, 8191, 1, (0, 0, 0)), (50000, 8192, 4096, (1, 1, 0))]:
        x = signal(N + L, N, dt) + dt(0.2)
        win = D.blackmanharris(L, dt) if L > 5000 else D.povey(L, dt)
        if stp == 1:
            x = x[:8300]
        for snip in (0, 1):
            if snip and L > len(x):
                continue
            rc, Yref = O.stft(x.astype(np.float64), win.astype(np.float64), stp, 8192, snip, *flags, exact=False)
            assert rc == 0
            Y = D.stft(x, win, stp, bool(snip), *[bool(f) for f in flags], nfft=8192)
            check_spec(O, Y, Yref.astype(dt), flags[1], flags[2])
    xs = np.stack([signal(10 * k + 8192, 50000, dt) for k in range(3)])
    win = D.hann(8192, dt)
    want = np.stack([O.stft(r.astype(np.float64), win.astype(np.float64), 2048, 8192, 0, exact=False)[1] for r in xs]).astype(dt)
    xd, wd = torch.from_numpy(xs).cuda(), torch.from_numpy(win).cuda()
    for off in (0, 1, 2, 3):
        buf = torch.full((want.size + 8,), 7.0, dtype=torch.float32, device="cuda")
        out = buf[off:off + want.size].view(want.shape)
        D.stft_dev(xd, wd, 2048, nfft=8192, out=out)
        torch.cuda.synchronize()
        check_spec(O, out.cpu().numpy(), want)
        assert float(buf[:off].sum()) == 7.0 * off and bool((buf[off + want.size:] == 7.0).all())
    xd = torch.randn((16, 1 << 21), device="cuda", generator=torch.Generator(device="cuda").manual_seed(8192))
    y0 = D.stft_dev(xd, wd, 2048, nfft=8192)
    y1 = torch.empty_like(y0)
    for _ in range(5):
        D.stft_dev(xd, wd, 2048, nfft=8192, out=y1)
        assert torch.equal(y0, y1)
    alone = D.stft_dev(xd[11:12].clone(), wd, 2048, nfft=8192)
    assert torch.equal(alone[0], y0[11])
    del xd, y0, y1
    x = signal(79, 60000, dt)
    win = D.blackmanharris(8192, dt)
    for intended in (False, True):
        D.set_flt_mode(intended)
        try:
            for (c0, stp, W) in [(0.0, 2048.0, 30), (3000.0, 2000.5, 20), (-12000.0, 4096.0, 20)]:
                rc, Yref = O.stft_flt(x.astype(np.float64), win.astype(np.float64), W, c0, stp, 8192, 0, 0, 0, intended=intended, exact=False)
                assert rc == 0
                Y = D.stft_flt(x, win, W, c0, stp, nfft=8192)
                check_spec(O, Y, Yref.astype(dt))
        finally:
            D.set_flt_mode(False)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_4096_run_form_agrees_with_oracle(O, D, monkeypatch, dt):
    """Full window with hop 1024: large inputs are transformed in runs of 32 consecutive frames per wave
    (each sample loaded once, the frame held rotated in registers).  Forced here at small sizes: runs
    that cross either edge of the signal, a last run shorter than 32, a signal shorter than one run,
    several signals, every flag; then the frame form on the same input."""
    import torch
    win = D.blackmanharris(4096, dt)
    for (N, flags) in [(40000, (0, 0, 0)), (32 * 1024 * 2 + 700, (1, 0, 0)), (100000, (0, 1, 1)), (5000, (1, 1, 0)), (4096 + 1024 * 63, (0, 0, 1))]:
        x = signal(N, N, dt) + dt(0.2)
        for snip in (0, 1):
            rc, Yref = O.stft(x.astype(np.float64), win.astype(np.float64), 1024, 4096, snip, *flags, exact=False)
            assert rc == 0
            for min_runs in ("1", "0"):
                monkeypatch.setenv("DSP_R4096_MIN_RUNS", min_runs)
                Y = D.stft(x, win, 1024, bool(snip), *[bool(f) for f in flags], nfft=4096)
                check_spec(O, Y, Yref.astype(dt), flags[1], flags[2])
    monkeypatch.setenv("DSP_R4096_MIN_RUNS", "1")
    xb = np.stack([signal(s, 70000, dt) for s in (1, 2, 3)])
    Yb = D.stft_dev(torch.from_numpy(xb).cuda(), torch.from_numpy(win).cuda(), 1024).cpu().numpy()
    for b in range(3):
        rc, Yref = O.stft(xb[b].astype(np.float64), win.astype(np.float64), 1024, 4096, exact=False)
        check_spec(O, Yb[b], Yref.astype(dt))


@pytest.mark.parametrize("tag", ["s", "d"])
def test_stft_against_reference_fixtures(O, D, tag):
    """Fixtures hold what the reference's own frame_univar -> apply_win -> FFT -> fft_power returned."""
    g = load_golden(f"ref_chain_{tag}.npz")
    for name in ["base", "short_multifold", "oddL", "hop_gt_L", "N_eq_L", "nfft4096", "nfft_gt_pow2", "tiny"]:
        x, win = g[f"{name}.x"], g[f"{name}.win"]
        N, L, stp, nfft = (int(v) for v in g[f"{name}.params"])
        for snip in (0, 1):
            key = f"{name}.snip{snip}.power"
            if key not in g:
                continue
            Y = D.stft(x, win, stp, bool(snip), nfft=nfft)
            check_spec(O, Y, g[key])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("intended", [False, True])
def test_stft_flt_matches_oracle(O, D, dt, intended):
    x = signal(21, 3000, dt)
    win = D.povey(400, dt)
    D.set_flt_mode(intended)
    try:
        for (c0, stp, W, flags) in [(0.0, 160.0, 18, (0, 0, 0)), (300.0, 160.0, 12, (1, 0, 0)), (-50.0, 160.0, 14, (0, 1, 0)),
                                    (200.0, 160.0, 23, (0, 0, 1)), (-700.0, 133.37, 30, (1, 1, 1)), (2999.0, 0.75, 9, (0, 0, 0)),
                                    (-5000.0, 1000.0, 9, (0, 0, 0)), (200.0, 160.0, 1, (0, 0, 0)), (10.5, 2.5, 700, (0, 0, 0))]:
            rc, Yref = O.stft_flt(x, win, W, c0, stp, 512, *flags, intended=intended)
            assert rc == 0
            Y = D.stft_flt(x, win, W, c0, stp, *[bool(f) for f in flags])
            check_spec(O, Y, Yref, flags[1], flags[2])
            D.force_generic(True)
            try:
                Y = D.stft_flt(x, win, W, c0, stp, *[bool(f) for f in flags])
            finally:
                D.force_generic(False)
            check_spec(O, Y, Yref, flags[1], flags[2])
    finally:
        D.set_flt_mode(False)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_stft_flt_other_sizes(O, D, dt):
    """stft_flt on the transform lengths of the wavefront-per-frame kernels and of the generic one: integral
    and non-integral hops (start tables), frames off both edges, both framing modes, both families."""
    x = signal(23, 9000, dt)
    for intended in (False, True):
        D.set_flt_mode(intended)
        try:
            for (L, nfft, c0, stp, W, flags) in [(200, 256, 100.0, 80.0, 60, (0, 0, 0)), (200, 256, -300.0, 80.5, 120, (1, 0, 0)),
                                                 (1000, 1024, 500.0, 250.0, 30, (0, 1, 0)), (1000, 1024, 17.0, 333.3, 40, (1, 1, 1)),
                                                 (64, 64, 8000.0, 33.0, 40, (0, 0, 0)), (2048, 2048, 1024.0, 512.0, 14, (0, 0, 1))]:
                win = D.povey(L, dt)
                rc, Yref = O.stft_flt(x, win, W, c0, stp, nfft, *flags, intended=intended)
                assert rc == 0
                for generic in (False, True):
                    D.force_generic(generic)
                    try:
                        Y = D.stft_flt(x, win, W, c0, stp, *[bool(f) for f in flags], nfft=nfft)
                    finally:
                        D.force_generic(False)
                    check_spec(O, Y, Yref, flags[1], flags[2])
        finally:
            D.set_flt_mode(False)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_stft_flt_start_tables_512_batched(O, D, dt):
    """nfft 512 with non-integral hops on a batch: frame starts that are not equally spaced go through the
    span-staging kernel while frames w and w + 3 start at most 1536 samples apart (hops up to about 511.6), and
    through the general one beyond; frame counts that are no multiple of 4, frames off both ends, every flag."""
    import torch
    B, N = 5, 40000
    xh = np.stack([signal(40 + b, N, dt) for b in range(B)])
    x = torch.from_numpy(xh).cuda()
    for (L, c0, stp, W, flags) in [(400, 200.0, 160.5, 246, (0, 0, 0)), (400, -333.0, 160.5, 251, (1, 0, 0)), (512, 256.0, 255.6, 153, (0, 1, 1)),
                                   (512, 100.0, 256.4, 155, (0, 0, 0)), (300, 39000.0, 77.77, 41, (1, 1, 0)),
                                   (400, 200.0, 300.25, 131, (0, 0, 0)), (512, 0.0, 511.6, 77, (1, 0, 1)), (512, 300.0, 512.7, 77, (0, 0, 0)), (400, 7.0, 1.3, 1021, (0, 0, 1))]:
        win = D.povey(L, dt)
        for intended in (False, True):
            D.set_flt_mode(intended)
            try:
                Y = D.stft_flt_dev(x, torch.from_numpy(win).cuda(), W, c0, stp, *[bool(f) for f in flags]).cpu().numpy()
            finally:
                D.set_flt_mode(False)
            for b in range(B):
                rc, Yref = O.stft_flt(xh[b], win, W, c0, stp, 512, *flags, intended=intended)
                assert rc == 0
                check_spec(O, Y[b], Yref, flags[1], flags[2])


@pytest.mark.parametrize("tag", ["s", "d"])
def test_stft_flt_intended_against_reference_fixtures(O, D, tag):
    g = load_golden(f"ref_flt_{tag}.npz")
    x, win = g["x"], g["win"]
    D.set_flt_mode(True)
    try:
        for i, (c0, stp, W) in enumerate(g["cases"]):
            Y = D.stft_flt(x, win, int(W), float(c0), float(stp))
            check_spec(O, Y, g[f"case{i}.power"])
    finally:
        D.set_flt_mode(False)


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_framing_is_bit_exact(O, D, dt):
    rng = np.random.default_rng(8)
    x = rng.standard_normal(1500)
    if np.dtype(dt).kind == "c":
        x = x + 1j * rng.standard_normal(1500)
    x = x.astype(dt)
    for (L, stp, snip) in [(400, 160, 0), (400, 160, 1), (64, 100, 0), (1, 1, 1), (1499, 1, 0), (7, 3000, 0)]:
        rc, ref = O.frame_univar(x, L, stp, snip)
        assert rc == 0
        assert np.array_equal(D.frame_univar(x, L, stp, bool(snip)), ref)
    for (L, W, c0, stp) in [(400, 14, 0.0, 160.0), (48, 30, -10.0, 17.3), (400, 20, -700.0, 133.37), (9, 1500, 3.3, 0.999)]:
        ct = np.float32 if np.dtype(dt).name in ("float32", "complex64") else np.float64
        rc, ref = O.frame_univar_flt(x, L, W, ct(c0), ct(stp))
        assert rc == 0
        assert np.array_equal(D.frame_univar_flt(x, L, W, c0, stp), ref)


@pytest.mark.parametrize("tag", ["s", "d"])
def test_framing_against_reference_fixtures(D, tag):
    g = load_golden(f"ref_misc_{tag}.npz")
    xc = g["x"]
    for snip in (0, 1):
        assert np.array_equal(D.frame_univar(xc, 48, 20, bool(snip)), g[f"frames.snip{snip}"])
    assert np.array_equal(D.frame_univar_flt(xc, 48, 30, -10.0, 17.3), g["frames_flt"])
    a = g["aw.x"]
    for col in (0, 1):
        assert np.array_equal(np.ascontiguousarray(D.apply_win(a, g["aw.w0"], 0, bool(col))), g[f"aw.dim0.col{col}"])
        assert np.array_equal(np.ascontiguousarray(D.apply_win(a, g["aw.w1"], 1, bool(col))), g[f"aw.dim1.col{col}"])
    gc = load_golden(f"ref_chain_{tag}.npz")
    for snip in (0, 1):
        fr = D.frame_univar(gc["base.x"], 400, 160, bool(snip))
        assert np.array_equal(fr, gc[f"base.snip{snip}.frames"])
        assert np.array_equal(D.apply_win(fr, gc["base.win"], 1), gc[f"base.snip{snip}.windowed"])


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_apply_win(O, D, dt):
    rng = np.random.default_rng(2)
    cplx = np.dtype(dt).kind == "c"
    for shape in [(6, 5), (3, 4, 5), (2, 3, 4, 5), (7,), (1, 9)]:
        x = rng.standard_normal(shape)
        if cplx:
            x = x + 1j * rng.standard_normal(shape)
        x = x.astype(dt)
        for dim in range(len(shape)):
            w = rng.standard_normal(shape[dim])
            if cplx:
                w = w + 1j * rng.standard_normal(shape[dim])
            w = w.astype(dt)
            for col in (False, True):
                rc, ref = O.apply_win(x, w, dim, col)
                assert rc == 0
                y = D.apply_win(x, w, dim, col)
                if cplx:        # products and sums may be fused on the device
                    np.testing.assert_allclose(y, ref, rtol=0, atol=4 * np.finfo(dt).eps * np.max(np.abs(ref)))
                else:
                    assert np.array_equal(np.asarray(y), np.asarray(ref))
                xi = np.array(x, order="F" if col else "C")
                D.apply_win(xi, w, dim, col, inplace=True)
                np.testing.assert_array_equal(np.asarray(xi), np.asarray(y))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_fft_r2hc(O, D, dt):
    rng = np.random.default_rng(6)
    tol = 10 * TOL[np.dtype(dt)]
    for shape, dim, nfft in [((13, 400), 1, 512), ((400, 13), 0, 512), ((3, 50, 4), 1, 64), ((2, 3, 4, 20), 3, 32),
                             ((5, 30), 1, 45), ((5, 30), 1, 30), ((4, 7), 1, 8), ((9,), 0, 16), ((4, 1), 1, 1),
                             ((6, 2000), 1, 4096)]:
        x = rng.standard_normal(shape).astype(dt)
        for col in (False, True):
            for sc in (False, True):
                rc, ref = O.fft_r2hc(x, dim, nfft, col, sc)
                assert rc == 0
                y = D.fft_r2hc(x, dim, nfft, col, sc)
                assert y.shape == ref.shape
                assert np.max(np.abs(np.asarray(y) - np.asarray(ref))) <= tol * np.max(np.abs(ref))
                # imag of DC (and of Nyquist for even nfft) is exactly zero, as the reference writes it
                ym = np.moveaxis(np.asarray(y), dim, -1)
                assert np.all(ym[..., 0].imag == 0)
                if nfft % 2 == 0 and nfft > 1:
                    assert np.all(ym[..., nfft // 2].imag == 0)


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_fft_fftw(O, D, dt):
    """fft_fftw_s/_d/_c/_z along any dim of tensors of up to 4 dimensions, both memory orders, with and
    without sc, power-of-two and other nfft, zero-padded; complex rows also against what the reference's
    own radix-2 code returned (fixtures)."""
    rng = np.random.default_rng(16)
    rdt = np.float32 if dt in (np.float32, np.complex64) else np.float64
    tol = 10 * TOL[np.dtype(rdt)]
    cplx = dt in (np.complex64, np.complex128)
    for shape, dim, nfft in [((13, 400), 1, 512), ((400, 13), 0, 512), ((3, 50, 4), 1, 64), ((2, 3, 4, 20), 3, 32),
                             ((5, 30), 1, 45), ((5, 30), 1, 30), ((4, 7), 1, 8), ((9,), 0, 16), ((4, 1), 1, 1),
                             ((3, 2, 5), 2, 5), ((6, 2000), 1, 4096)]:
        x = rng.standard_normal(shape).astype(rdt)
        if cplx:
            x = (x + 1j * rng.standard_normal(shape)).astype(dt)
        for col in (False, True):
            for sc in (False, True):
                rc, ref = O.fft(x, dim, nfft, col, sc)
                assert rc == 0
                y = D.fft(x, dim, nfft, col, sc)
                assert y.shape == ref.shape and y.dtype == ref.dtype
                assert y.shape[dim] == (nfft if cplx else nfft // 2 + 1)
                assert np.max(np.abs(np.asarray(y) - np.asarray(ref))) <= tol * np.max(np.abs(ref))
    if cplx:
        g = load_golden(f"ref_fft_c_{'s' if dt == np.complex64 else 'd'}.npz")
        for n in (2, 8, 64, 512):
            y = D.fft(g[f"x.{n}"], 1, n)
            assert np.max(np.abs(y - g[f"y.{n}"])) <= (3e-6 if dt == np.complex64 else 2e-14) * np.max(np.abs(g[f"y.{n}"]))
    else:
        # the real transform is the r2hc one
        x = rng.standard_normal((7, 300)).astype(dt)
        assert np.array_equal(D.fft(x, 1, 512, False, True), D.fft_r2hc(x, 1, 512, False, True))
    with pytest.raises(D.DspError):
        D.fft(np.zeros((3, 9), dt), 1, 8)


@pytest.mark.parametrize("dt", [np.complex64, np.complex128])
def test_fft_power(O, D, dt):
    rng = np.random.default_rng(7)
    for n in (1, 7, 257000, 1 << 20):
        z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dt)
        rc, ref = O.fft_power(z)
        y = D.fft_power(z)
        np.testing.assert_allclose(y, ref, rtol=2 * np.finfo(dt).eps, atol=0)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_unfused_chain_equals_fused(O, D, dt):
    """frame_univar | apply_win | fft.fftw.r2hc | fft_power is the unfused form of stft."""
    x = signal(31, 16000, dt)
    win = D.hann(400, dt)
    for snip in (False, True):
        fr = D.frame_univar(x, 400, 160, snip)
        wn = D.apply_win(fr, win, 1, inplace=False)
        z = D.fft_r2hc(wn, 1, 512)
        p = D.fft_power(z)
        check_spec(O, p, D.stft(x, win, 160, snip))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_batch_host_and_device(O, D, dt):
    import torch
    rng = np.random.default_rng(12)
    B, N = 5, 4000
    big = rng.standard_normal((B, N + 37)).astype(dt)
    x = big[:, :N]                      # row stride N+37
    win = D.hamming(400, dt)
    Yb = D.stft_batch(x, win, 160)
    for b in range(B):
        assert np.array_equal(Yb[b], D.stft(np.ascontiguousarray(x[b]), win, 160))
    xt = torch.from_numpy(big).cuda()[:, :N]
    wt = torch.from_numpy(win).cuda()
    Yd = D.stft_dev(xt, wt, 160)
    torch.cuda.synchronize()
    assert np.array_equal(Yd.cpu().numpy(), Yb)
    D.set_flt_mode(True)
    try:
        Yf = D.stft_flt_dev(xt, wt, 20, 200.0, 160.0).cpu().numpy()
        for b in range(B):
            assert np.array_equal(Yf[b], D.stft_flt(np.ascontiguousarray(x[b]), win, 20, 200.0, 160.0))
    finally:
        D.set_flt_mode(False)
    zt = torch.view_as_complex(torch.from_numpy(rng.standard_normal((1000, 2)).astype(dt)).cuda())
    np.testing.assert_array_equal(D.fft_power_dev(zt).cpu().numpy(), D.fft_power(zt.cpu().numpy()))


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_batch_host_streamed_equals_device(O, D, dt):
    """Batches of 64 MB and more go through the host-pointer entry points in chunks (pinned stages, copies
    overlapped with the kernel): same bits as the device-resident call on the whole batch, for dense and for
    strided rows, and for a batch that does not divide into equal chunks."""
    import torch
    rng = np.random.default_rng(5)
    B, N = (67, 160000) if dt == np.float32 else (35, 150000)
    buf = (rng.standard_normal((B, N + 24)) * 0.1).astype(dt)
    win = D.hann(400, dt)
    for xs in (np.ascontiguousarray(buf[:, :N]), buf[:, 8:N + 8]):
        Yh = D.stft_batch(xs, win, 160)
        Yd = D.stft_dev(torch.from_numpy(np.ascontiguousarray(xs)).cuda(), torch.from_numpy(win).cuda(), 160)
        torch.cuda.synchronize()
        assert np.array_equal(Yh, Yd.cpu().numpy())
    rc, Yref = O.stft(np.ascontiguousarray(buf[B - 1, :N]), win, 160, 512, exact=False)
    check_spec(O, Yh_last := D.stft_batch(np.ascontiguousarray(buf[:, :N]), win, 160)[B - 1], Yref)
    # twice in a row with different sizes (the stages are reused and regrown)
    Y2 = D.stft_batch(np.ascontiguousarray(buf[:40, :N]), win, 160, mn0=True, lg=True)
    Y2d = D.stft_dev(torch.from_numpy(np.ascontiguousarray(buf[:40, :N])).cuda(), torch.from_numpy(win).cuda(), 160, mn0=True, lg=True)
    torch.cuda.synchronize()
    assert np.array_equal(Y2, Y2d.cpu().numpy())


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_interleaved_channels_equal_channels_alone(O, D, dt, monkeypatch):
    """A multivariate tensor with interleaved channels, row-major [N][C] (sample stride C, channel stride 1: the
    reference's tensors have stride K along the transformed axis, c/fft.fftw.r2hc.c:63-65), through the strided
    entry points with the channels gathered into a dense batch first: every channel has the bits of the same channel
    transformed from a dense copy, and matches the oracle; channel counts that are no multiple of the gather tile, a
    sub-block of the channels, both framings.  (The in-place read of the 512-point kernel is switched off here; it has
    a test of its own below.)"""
    import torch
    monkeypatch.setenv("DSP_ILV_MAX_CHANNELS", "0")
    tdt = torch.float32 if dt == np.float32 else torch.float64
    g = torch.Generator(device="cuda").manual_seed(17)
    win = torch.from_numpy(D.povey(400, dt)).cuda()
    # (64 / 8 / 68 / 132 channels with a sample count that is a multiple of 4: the gather with 16-byte accesses, whole
    # and partial tiles; the others: the element-wise gather)
    for (N, Cc) in [(48000, 64), (20011, 7), (9000, 130), (16384, 8), (5000, 68), (3332, 132)]:
        t = torch.randn((N, Cc), generator=g, device="cuda", dtype=tdt)           # [N][C]
        x = t.T                                                                    # [C][N] view, stride (1, C)
        dense = x.contiguous()
        Y = D.stft_dev(x, win, 160)
        Yd = D.stft_dev(dense, win, 160)
        W = int((N - 1 - 200) // 160.5)
        Z = D.stft_flt_dev(x, win, W, 200.0, 160.5, mn0=True)
        Zd = D.stft_flt_dev(dense, win, W, 200.0, 160.5, mn0=True)
        sub = D.stft_dev(x[1:5], win, 160)                                         # a block of the channels
        torch.cuda.synchronize()
        assert torch.equal(Y, Yd) and torch.equal(Z, Zd) and torch.equal(sub, Yd[1:5])
        if Cc >= 8:
            assert torch.equal(D.stft_dev(x[4:8], win, 160), Yd[4:8])              # a block that starts on a 16-byte boundary
        c = Cc - 1
        rc, Yref = O.stft(dense[c].cpu().numpy(), win.cpu().numpy(), 160, 512, exact=False)
        check_spec(O, Y[c].cpu().numpy(), Yref)


def test_interleaved_channels_read_in_place(O, D, monkeypatch):
    """float32, nfft 512, up to 32 adjacent channels of a [N][pitch] array whose count and pitch are multiples of 4: the
    512-point kernel reads them in place (a step = one frame of 4 channels).  Against the oracle channel by channel and
    against the gather + dense path (same values to rounding: another instantiation of the kernel): full, short and
    odd windows, hops up to and past 512, both edge rules, every flag, a signal shorter than the window, a block of
    the channels of a wider array, the bytes around the output untouched; and what is not eligible (a channel count
    that is no multiple of 4, an unaligned block) still takes the gather and gives the dense bits."""
    import torch
    dt = np.float32
    g = torch.Generator(device="cuda").manual_seed(23)
    cases = [(48000, 64, 400, 160, 0, (0, 0, 0)), (20000, 8, 512, 128, 1, (1, 0, 0)), (9001, 12, 333, 77, 0, (0, 1, 1)),
             (30000, 4, 400, 600, 1, (1, 1, 0)), (300, 16, 400, 160, 0, (0, 0, 1)), (16000, 32, 1, 512, 0, (0, 0, 0)),
             (7000, 20, 512, 1, 1, (0, 1, 0))]
    for (N, Cc, L, hop, snip, flags) in cases:
        if hop == 1:
            N = 700
        t = torch.randn((N, Cc), generator=g, device="cuda") * 0.3 + 0.1
        x = t.T
        win = torch.from_numpy(D.povey(L, dt)).cuda()
        kw = dict(snip_edges=bool(snip), mn0=bool(flags[0]), amp=bool(flags[1]), lg=bool(flags[2]), nfft=512)
        monkeypatch.setenv("DSP_ILV_MAX_CHANNELS", "0")
        want = D.stft_dev(x, win, hop, **kw)
        monkeypatch.setenv("DSP_ILV_MAX_CHANNELS", "64")       # (the default limit is 32 channels)
        buf = torch.full((want.numel() + 8,), 7.0, device="cuda")
        out = buf[4:4 + want.numel()].view(want.shape)
        D.stft_dev(x, win, hop, out=out, **kw)
        torch.cuda.synchronize()
        assert bool((buf[:4] == 7.0).all()) and bool((buf[4 + want.numel():] == 7.0).all())
        if 384 < L <= 416 and hop <= 170:
            # the dense kernel has an instantiation of its own for these windows and hops (13 of 16 rows of pass 1 hold data), the in-place
            # form does not: equal to rounding, not to the bit -- which also shows that the in-place form is what ran
            assert not torch.equal(out, want)
        check_spec(O, out.cpu().numpy().reshape(-1, 257), want.cpu().numpy().reshape(-1, 257), flags[1], flags[2])
        for c in (0, Cc - 1):
            rc, Yref = O.stft(t[:, c].cpu().numpy().copy(), win.cpu().numpy(), hop, 512, snip, *flags, exact=False)
            assert rc == 0
            check_spec(O, out[c].cpu().numpy(), Yref, flags[1], flags[2])
    # a block of 8 channels of a 24-channel array (pitch 24), and what falls back to the gather
    monkeypatch.delenv("DSP_ILV_MAX_CHANNELS")
    t = torch.randn((12000, 24), generator=g, device="cuda")
    x = t.T
    win = torch.from_numpy(D.hann(400, dt)).cuda()
    dense = D.stft_dev(x.contiguous(), win, 160)
    blk = D.stft_dev(x[8:16], win, 160)
    check_spec(O, blk.cpu().numpy().reshape(-1, 257), dense[8:16].cpu().numpy().reshape(-1, 257))
    assert not torch.equal(blk, dense[8:16])
    assert torch.equal(D.stft_dev(x[1:9], win, 160), dense[1:9])          # not on a 16-byte boundary: gathered
    assert torch.equal(D.stft_dev(x[4:10], win, 160), dense[4:10])        # 6 channels: gathered
    monkeypatch.setenv("DSP_ILV_MAX_CHANNELS", "4")
    assert torch.equal(D.stft_dev(x[8:16], win, 160), dense[8:16])        # above the limit: gathered


def test_multi_gpu_entry_points_on_one_device(O, D):
    """The single-process multi-GPU entry points with every 'GPU' mapped to device 0 (all a one-GPU box allows): the
    host form (one thread per entry of `devices`, each through the host path for its block) and the device form with
    the blocks copied into one tensor; both equal the single-GPU result, also for a batch that leaves short and
    empty blocks."""
    import torch
    dt = np.float32
    win = D.hann(400, dt)
    wd = torch.from_numpy(win).cuda()
    for (B, devs) in [(7, [0, 0, 0]), (2, [0, 0, 0, 0]), (5, [0])]:
        x = np.stack([signal(900 + b, 24000, dt) for b in range(B)])
        ref = D.stft_batch(x, win, 160, mn0=True)
        assert np.array_equal(D.stft_multi(x, win, 160, mn0=True, devices=devs), ref)
        blocks = []
        for g in range(len(devs)):
            lo, hi = D.shard_range(B, g, len(devs))
            blocks.append(torch.from_numpy(x[lo:hi]).cuda().contiguous())
        ys, y_all = D.stft_multi_dev(blocks, [wd] * len(devs), 160, mn0=True, devices=devs)
        assert np.array_equal(y_all.cpu().numpy(), ref)
        assert np.array_equal(torch.cat(ys).cpu().numpy(), ref)
    rc, Yref = O.stft(x[4], win, 160, 512, 0, 1)
    check_spec(O, ref[4], Yref)
    # the stft_flt framing over the same device lists (integral hop, and a table of frame starts)
    x = np.stack([signal(950 + b, 24000, dt) for b in range(7)])
    for stp in (160.0, 133.37):
        W = int((24000 - 1 - 200) // stp)
        ref = D.stft_flt_batch(x, win, W, 200.0, stp, lg=True)
        for devs in ([0, 0, 0], [0] * 9, [0]):
            assert np.array_equal(D.stft_flt_multi(x, win, W, 200.0, stp, lg=True, devices=devs), ref)


def test_nan_stays_in_its_own_frames(O, D):
    """A NaN sample must only reach the frames that contain it (reflection included)."""
    x = signal(1, 8000, np.float32)
    x[4000] = np.nan
    win = D.hann(400, np.float32)
    rc, Yref = O.stft(x, win, 160, 512)
    for generic in (False, True):
        D.force_generic(generic)
        try:
            Y = D.stft(x, win, 160)
        finally:
            D.force_generic(False)
        assert np.array_equal(np.isnan(Y).any(axis=1), np.isnan(Yref).any(axis=1))
        ok = ~np.isnan(Yref).any(axis=1)
        check_spec(O, Y[ok], Yref[ok])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_nan_behind_a_short_window_stays_out(O, D, dt):
    """Windows shorter than the transform on the 1024- and 2048-point kernels (the latter reads its span slot unmasked and makes what
    lies behind the window in the slot harmless instead): NaN and Inf samples reach exactly the frames whose window contains them,
    for even and odd window lengths (the samples behind the window inside the last 16 bytes of a span) and frames at both edges."""
    import torch
    for nfft, L, hop in ((2048, 1102, 441), (2048, 1103, 441), (2048, 1201, 480), (2048, 9, 700), (1024, 551, 220), (1024, 800, 200)):
        N = 40 * hop + L
        x = signal(3 + L, N, dt)
        bad = [0, L, L + 1, 5 * hop + L, 5 * hop + L + 1, 5 * hop + L + 2, 17 * hop - 1, N - 1]
        for k, i in enumerate(bad):
            x[i] = np.nan if k % 2 == 0 else np.inf
        win = D.povey(L, dt)
        for snip in (True, False):
            rc, Yref = O.stft(x.astype(np.float64), win.astype(np.float64), hop, nfft, int(snip), 0, 0, 0, exact=False)
            assert rc == 0
            Y = D.stft_dev(torch.from_numpy(x[None]).cuda(), torch.from_numpy(win).cuda(), hop, snip, nfft=nfft)[0].cpu().numpy()
            assert Y.shape == Yref.shape
            finite_ref = np.isfinite(Yref).all(axis=1)
            assert np.array_equal(np.isfinite(Y).all(axis=1), finite_ref), (nfft, L, hop, snip)
            assert finite_ref.sum() > 10
            check_spec(O, Y[finite_ref], Yref[finite_ref].astype(dt))


def test_log_of_silence_is_minus_inf(D):
    Y = D.stft(np.zeros(4000, np.float32), D.hann(400), 160, lg=True)
    assert np.all(np.isneginf(Y))
    Y = D.stft(np.zeros(4000, np.float64), D.hann(400, np.float64), 160, amp=True, lg=True)
    assert np.all(np.isneginf(Y))


@pytest.mark.parametrize("nfft,L,stp", [(512, 400, 160), (4096, 4096, 1024), (256, 200, 80)])
def test_very_quiet_signal_float32(O, D, nfft, L, stp):
    """A float32 signal so quiet that its power values are denormal (below 2^-126): -a and -l must give what
    sqrtf / logf give for them (the reference calls those, c/stft.c:86-95), not 0 / -inf, and so must the power
    compression of pow_compress (c/pow_compress.c:30-46) on such values."""
    dt = np.float32
    x = (signal(31, 60000, dt) * dt(3e-21 * (400.0 / L) ** 0.5)).astype(dt)
    win = D.hann(L, dt)
    for (amp, lg) in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        rc, Yref = O.stft(x, win, stp, nfft, 0, 0, amp, lg)
        assert rc == 0
        P = Yref if not lg else np.exp(Yref.astype(np.float64))
        assert 0 < np.median(P) < (1e-19 if amp else 3e-39)              # the case is what it claims to be
        Y = D.stft(x, win, stp, False, False, bool(amp), bool(lg), nfft=nfft)
        assert np.all(np.isfinite(Y) == np.isfinite(Yref)) and not np.any(np.isnan(Y))
        # compared as power values (a denormal power carries only a few bits, so its square root or logarithm
        # differs between two correct implementations by far more than the path's bound): the bound of the path
        # plus four steps of the denormal grid, 2^-149
        def as_power(A):
            A = A.astype(np.float64)
            A = np.exp(A) if lg else A
            return A * A if amp else A
        Pg, Pr = as_power(Y), as_power(Yref)
        fmax = Pr.max(axis=1, keepdims=True)
        # (a logarithm near -90 is itself only given to 8e-6: four steps of its float32 grid, as a factor on the power)
        slack = Pr * (4 * np.spacing(np.abs(Yref)).astype(np.float64) * (2 if amp else 1)) if lg else 0.0
        assert np.all(np.abs(Pg - Pr) <= 1e-5 * fmax + 5.7e-45 + slack), (amp, lg)
        assert np.count_nonzero(Pg) > 0.99 * Pg.size                     # not flushed to zero
    rc, Ypow = O.stft(x, win, stp, nfft)
    for p in (0.0, 0.5, 1.0 / 3.0, 0.3, 0.97):
        got = D.pow_compress(Ypow, p, 0.0)
        with np.errstate(divide="ignore"):
            want = (np.log(Ypow.astype(np.float64)) if p == 0.0 else np.power(Ypow.astype(np.float64), p))
        ok = np.isfinite(want) & (Ypow > 1e-42)                       # (a few denormal bits: nothing to compare below)
        if p == 0.0:
            assert np.max(np.abs(got[ok] - want[ok])) < 2e-5, p
        else:
            # (relative bound of the path plus two steps of the denormal grid, 2^-149, on which such results lie)
            assert np.all(np.abs(got[ok] - want[ok]) <= 2e-5 * np.abs(want[ok]) + 2.9e-45), p
        assert np.all(got[Ypow == 0] == (-np.inf if p == 0.0 else 0.0))


# ---------------------------------------------------------------------------------------------------
# full benchmark sizes: size-independent properties + sampled oracle checks
# ---------------------------------------------------------------------------------------------------
def test_full_size_batch_1024_utterances(O, D):
    """1024 x 10 s @ 16 kHz float32, Hann 400 / hop 160 / nfft 512 (1 024 000 frames)."""
    import torch
    B, N = 1024, 160000
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn((B, N), generator=g, device="cuda", dtype=torch.float32) * 0.1
    win = torch.from_numpy(D.hann(400)).cuda()
    Y = D.stft_dev(x, win, 160)
    torch.cuda.synchronize()
    assert tuple(Y.shape) == (B, 1000, 257)
    # (1) sampled signals against the oracle (first, last, a few in between; edge frames included)
    for b in (0, 1, 511, 1023):
        rc, Yref = O.stft(x[b].cpu().numpy(), win.cpu().numpy(), 160, 512, exact=False)
        check_spec(O, Y[b].cpu().numpy(), Yref)
    # (2) shift by one hop moves interior frames by one row, bit for bit (same arithmetic, another wave, another
    # place in its step): every signal of the batch, so every interior frame is checked against a second launch
    xs = torch.roll(x, shifts=-160, dims=1)
    Ys = D.stft_dev(xs, win, 160)
    torch.cuda.synchronize()
    assert torch.equal(Ys[:, 2:990], Y[:, 3:991])
    del xs, Ys
    # (3) Parseval per frame: sum_k c_k P_k = nfft * sum_l v_l^2 (c = 1 for DC and Nyquist, else 2)
    fr = x[:4, 40:].unfold(1, 400, 160)[:, :995] * win     # frame w of the output starts at 160 w - 120
    e_time = (fr.double() ** 2).sum(-1) * 512
    Pd = Y[:4, 1:996].double()
    e_freq = 2 * Pd.sum(-1) - Pd[..., 0] - Pd[..., 256]
    assert torch.allclose(e_freq, e_time, rtol=2e-5)
    # (4) scaling by 2 scales power by exactly 4
    Y2 = D.stft_dev(x[:8] * 2, win, 160)
    torch.cuda.synchronize()
    assert torch.equal(Y2, Y[:8] * 4)
    # (5) repeated launches return the same bits (which wave takes which step differs from launch to launch; a
    # span read before it has landed, or overwritten before it was read, shows as one frame in a million): an
    # exact, order-independent checksum of every signal, 30 launches
    def checksum(t):
        return t.view(torch.int32).to(torch.int64).sum(dim=(1, 2))
    ref = checksum(Y)
    for _ in range(30):
        D.stft_dev(x, win, 160, out=Y)
        torch.cuda.synchronize()
        assert torch.equal(checksum(Y), ref)


def test_full_size_share_of_sharded_batch_8192_utterances(O, D):
    """One GPU's share of the sharded configuration (65 536 utterances over 8 GPUs): 8192 x 10 s @ 16 kHz float32,
    Hann 400 / hop 160 / nfft 512: 5.2 GB in, 8.4 GB out, one launch.  Sampled signals against the oracle; the batch
    cut into two launches, and into the eight blocks dsp_shard_range would give eight ranks, returns the same bits;
    repeated launches return the same bits (exact checksum per signal)."""
    import torch
    B, N = 8192, 160000
    g = torch.Generator(device="cuda").manual_seed(4321)
    x = torch.randn((B, N), generator=g, device="cuda", dtype=torch.float32) * 0.1
    win = torch.from_numpy(D.hann(400)).cuda()
    Y = D.stft_dev(x, win, 160)
    torch.cuda.synchronize()
    assert tuple(Y.shape) == (B, 1000, 257)
    for b in (0, 4095, 4096, 8191):
        rc, Yref = O.stft(x[b].cpu().numpy(), win.cpu().numpy(), 160, 512, exact=False)
        check_spec(O, Y[b].cpu().numpy(), Yref)

    def checksum(t):
        return t.view(torch.int32).to(torch.int64).sum(dim=(1, 2))
    ref = checksum(Y)
    part = torch.empty((B // 2, 1000, 257), device="cuda", dtype=torch.float32)
    for h in range(2):
        D.stft_dev(x[h * (B // 2):(h + 1) * (B // 2)], win, 160, out=part)
        torch.cuda.synchronize()
        assert torch.equal(part, Y[h * (B // 2):(h + 1) * (B // 2)])
    del part
    for r in range(8):
        lo, hi = D.shard_range(B, r, 8)
        blk = D.stft_dev(x[lo:hi], win, 160)
        torch.cuda.synchronize()
        assert torch.equal(checksum(blk), ref[lo:hi])
        del blk
    for _ in range(5):
        D.stft_dev(x, win, 160, out=Y)
        torch.cuda.synchronize()
        assert torch.equal(checksum(Y), ref)


def test_bench_two_rank_rehearsal_in_a_fresh_process():
    """bench.py's N > 1 branch, started the way the driver starts it (python -m torch.distributed.run, one process per
    rank), as a fresh child: two ranks on this one GPU over gloo (DSP_BENCH_REHEARSAL=1), the 1024-utterance workload so
    that both ranks' batches fit beside each other.  The JSON line must carry n_gpus 2, the whole-job value and a gather
    object without an error."""
    import json
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSP_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    with socket.socket() as sk:                      # a port nobody listens on (a fixed one may be taken on a shared host)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--workload", "utt1024x10s_f32_hann400_hop160_nfft512", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=150, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 1024 * 1000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert isinstance(d.get("gather"), dict) and "error" not in d["gather"], d.get("gather")


def test_counted_waits_against_full_waits():
    """The span-staging kernels wait for their LDS-DMA loads with counted waits that leave the row stores in flight.
    libdsp_stft_hip_safe.so is the same code with every such wait built as a wait for everything outstanding
    (-DDSP_SAFE_WAITS=1): both builds must return the same bits -- on the benchmark's batch (every signal's exact checksum,
    three launches each) and on the shapes of the 256-, 1024- and 2048-point kernels.  The safe build runs in a child
    process (one library per process)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    safe = os.path.join(root, "dsp_amd", "libdsp_stft_hip_safe.so")
    assert os.path.exists(safe), "make -C dsp_amd/csrc builds it"
    prog = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np, torch, dsp_amd
out = {}
def checksum(t):
    return t.view(torch.int32).to(torch.int64).sum(dim=(1, 2)).cpu().tolist()
for name, B, N, L, hop, nfft in (("utt", 1024, 160000, 400, 160, 512), ("tel", 256, 80000, 200, 80, 256), ("m1024", 16, 1323000, 1024, 256, 1024),
                                 ("m2048", 16, 1323000, 2048, 512, 2048), ("odd", 64, 160003, 400, 161, 512)):
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn((B, N), generator=g, device="cuda", dtype=torch.float32) * 0.1
    win = torch.from_numpy(dsp_amd.hann(L)).cuda()
    sums = []
    for rep in range(3):
        y = dsp_amd.stft_dev(x, win, hop, nfft=nfft)
        torch.cuda.synchronize()
        sums.append(checksum(y))
    assert sums[0] == sums[1] == sums[2]
    out[name] = sums[0]
print("RESULT " + json.dumps(out))
""" % root
    def run(lib):
        env = dict(os.environ)
        if lib:
            env["DSP_AMD_LIB"] = lib
        else:
            env.pop("DSP_AMD_LIB", None)
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=280, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
        return json.loads(line[7:])
    counted, full = run(None), run(safe)
    assert counted.keys() == full.keys()
    for k in counted:
        assert counted[k] == full[k], k


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hops_above_512_on_the_span_kernel(O, D, dt):
    """nfft 512 with frames further apart than a transform length (hop > 512): the span-staging kernel fetches the four
    frames of a step as four pieces and skips what lies between them.  Signals long enough for interior steps, both
    edge rules (stft reflects, stft_flt pads with zeros), odd hops and short windows; every signal against the oracle."""
    import torch
    rng = np.random.default_rng(512)
    for L, hop, N in ((512, 600, 40000), (512, 513, 30011), (400, 801, 50000), (512, 1024, 66000), (33, 2000, 70001)):
        x = (rng.standard_normal((3, N)) * 0.1).astype(dt)
        win = D.hamming(L, dt)
        for snip in (False, True):
            Y = D.stft_dev(torch.from_numpy(x).cuda(), torch.from_numpy(win).cuda(), hop, nfft=512, snip_edges=snip).cpu().numpy()
            for b in range(3):
                rc, Yref = O.stft(x[b], win, hop, 512, int(snip))
                assert rc == 0
                check_spec(O, Y[b], Yref)
        W = (N - 1) // hop
        Z = D.stft_flt_dev(torch.from_numpy(x).cuda(), torch.from_numpy(win).cuda(), W, float(L // 2), float(hop), nfft=512, mn0=True).cpu().numpy()
        for b in range(3):
            rc, Zref = O.stft_flt(x[b], win, W, float(L // 2), float(hop), 512, 1)
            assert rc == 0
            check_spec(O, Z[b], Zref)


def test_full_size_chunks_equal_channels_alone(O, D, monkeypatch):
    """The multi-channel launches of the long-signal configurations: 64 and 8 channels of 1 h @ 16 kHz float64
    through stft_flt (Povey 400, hop 160.0, c0 200), 8 channels of 1 h @ 48 kHz float32 with nfft 4096 / hop 1024.
    Every channel of the batched output equals, bit for bit, the same channel transformed alone; head and tail rows
    of two channels against the oracle."""
    import torch
    # float64 stft_flt
    N, L = 57_600_000, 400
    W = int((N - 1 - 200) // 160)
    win = torch.from_numpy(D.povey(L, np.float64)).cuda()
    g = torch.Generator(device="cuda").manual_seed(2024)
    x = torch.randn((64, N), generator=g, device="cuda", dtype=torch.float64)
    Y = D.stft_flt_dev(x, win, W, 200.0, 160.0)
    torch.cuda.synchronize()
    assert tuple(Y.shape) == (64, W, 257)
    y1 = torch.empty((1, W, 257), device="cuda", dtype=torch.float64)
    for b in range(64):
        D.stft_flt_dev(x[b:b + 1], win, W, 200.0, 160.0, out=y1)
        torch.cuda.synchronize()
        assert torch.equal(y1[0], Y[b]), b
    Y8 = D.stft_flt_dev(x[8:16], win, W, 200.0, 160.0)
    torch.cuda.synchronize()
    assert torch.equal(Y8, Y[8:16])
    wh = win.cpu().numpy()
    for b in (0, 63):
        xh = x[b].cpu().numpy()
        rc, Yref = O.stft_flt(xh[:20000], wh, 100, 200.0, 160.0, intended=True)
        check_spec(O, Y[b, :100].cpu().numpy(), Yref)
        w0 = W - 50
        rc, Yref = O.stft_flt(xh[w0 * 160:], wh, 50, 200.0, 160.0, intended=True)
        check_spec(O, Y[b, w0:].cpu().numpy(), Yref)
    del x, Y, Y8, y1
    torch.cuda.empty_cache()
    # float32, nfft 4096
    N = 172_800_000
    W = (N + 512) // 1024
    win = torch.from_numpy(D.blackmanharris(4096)).cuda()
    x = torch.randn((8, N), generator=g, device="cuda", dtype=torch.float32)
    Y = D.stft_dev(x, win, 1024)
    torch.cuda.synchronize()
    assert tuple(Y.shape) == (8, W, 2049)
    y1 = torch.empty((1, W, 2049), device="cuda", dtype=torch.float32)
    for b in range(8):
        D.stft_dev(x[b:b + 1], win, 1024, out=y1)
        torch.cuda.synchronize()
        assert torch.equal(y1[0], Y[b]), b
    wh = win.cpu().numpy()
    for b in (0, 7):
        xh = x[b].cpu().numpy()
        rc, Yref = O.stft(xh[:1024 * 40], wh, 1024, 4096, exact=False)
        check_spec(O, Y[b, :30].cpu().numpy(), Yref[:30])
    # the last 12 frames (reflection at the right edge): oracle on the tail, whose last frames are the signal's
    xh = x[7].cpu().numpy()
    tail = xh[-1024 * 30:]
    rc, Yref = O.stft(tail, wh, 1024, 4096, exact=False)
    check_spec(O, Y[7, -12:].cpu().numpy(), Yref[-12:])


def test_4096_run_forms_on_two_streams(O, D, monkeypatch):
    """Two 4096-point launches whose loops deal runs from counters, with different flags (different kernel
    instantiations), in flight together on two streams, 80 pairs in a row: neither disturbs the other's counters."""
    import torch
    monkeypatch.setenv("DSP_R4096_MIN_RUNS", "1")
    win = torch.from_numpy(D.blackmanharris(4096)).cuda()
    g = torch.Generator(device="cuda").manual_seed(11)
    xa = torch.randn((6, 1024 * 300), generator=g, device="cuda") + 0.3
    xb = torch.randn((5, 1024 * 260 + 77), generator=g, device="cuda") + 0.3
    ya_ref = D.stft_dev(xa, win, 1024)
    yb_ref = D.stft_dev(xb, win, 1024, mn0=True)
    torch.cuda.synchronize()
    wh = win.cpu().numpy()
    rc, Yref = O.stft(xa[2].cpu().numpy(), wh, 1024, 4096, exact=False)
    check_spec(O, ya_ref[2].cpu().numpy(), Yref)
    rc, Yref = O.stft(xb[4].cpu().numpy(), wh, 1024, 4096, 0, 1, exact=False)
    check_spec(O, yb_ref[4].cpu().numpy(), Yref)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ya = [torch.empty_like(ya_ref) for _ in range(4)]
    yb = [torch.empty_like(yb_ref) for _ in range(4)]
    for it in range(80):
        with torch.cuda.stream(sa):
            D.stft_dev(xa, win, 1024, out=ya[it % 4])
        with torch.cuda.stream(sb):
            D.stft_dev(xb, win, 1024, mn0=True, out=yb[it % 4])
        if it % 4 == 3:
            torch.cuda.synchronize()
            for k in range(4):
                assert torch.equal(ya[k], ya_ref), (it, k)
                assert torch.equal(yb[k], yb_ref), (it, k)


def test_stft_flt_dev_repeated_calls_only_enqueue(O, D):
    """200 device-resident stft_flt calls with a non-integral hop (frame starts from a table) alternating over two
    streams, nothing synchronised in between: after the first call of a framing the entry point neither walks the
    centres on the host nor allocates or copies (the host time of a repeated call is measured and bounded), and
    every output equals the oracle's."""
    import ctypes as C
    import time
    import torch
    from dsp_amd._lib import lib
    dt = np.float32
    N, L, W, c0, stp = 400_000, 400, 2990, 133.0, 133.37
    x = torch.from_numpy(np.stack([signal(70 + b, N, dt) for b in range(3)])).cuda()
    win = torch.from_numpy(D.povey(L, dt)).cuda()
    D.set_flt_mode(True)
    try:
        ref = D.stft_flt_dev(x, win, W, c0, stp)            # first use of this framing: builds and uploads the table
        torch.cuda.synchronize()
        for b in range(3):
            rc, Yref = O.stft_flt(x[b].cpu().numpy(), win.cpu().numpy(), W, c0, stp, intended=True)
            check_spec(O, ref[b].cpu().numpy(), Yref)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.empty_like(ref) for _ in range(8)]
        fn = lib().dsp_stft_flt_batch_dev_s
        t_host = 0.0
        for it in range(200):
            s = streams[it % 2]
            y = outs[it % 8]
            t0 = time.perf_counter()
            rc = fn(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(win.data_ptr()), 3, N, N, L, W, 512,
                    c0, stp, 0, 0, 0, C.c_void_p(s.cuda_stream))
            t_host += time.perf_counter() - t0
            assert rc == 0
            if it % 8 == 7:
                torch.cuda.synchronize()
                for k in range(8):
                    assert torch.equal(outs[k], ref), (it, k)
                    outs[k].zero_()
        print(f"host time per repeated dsp_stft_flt_batch_dev_s call, alternating streams: {t_host / 200 * 1e6:.1f} us")
        # the entry point's own host time: one stream (the runtime's change of stream costs more than the call)
        s0 = C.c_void_p(streams[0].cuda_stream)
        y = outs[0]
        t0 = time.perf_counter()
        for it in range(200):
            fn(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(win.data_ptr()), 3, N, N, L, W, 512, c0, stp, 0, 0, 0, s0)
        per_call_us = (time.perf_counter() - t0) / 200 * 1e6
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
        print(f"host time per repeated dsp_stft_flt_batch_dev_s call, one stream: {per_call_us:.1f} us")
        assert per_call_us < 20, per_call_us        # (W = 2990 centres on the host alone would be ~10 us, an upload ~50)
    finally:
        D.set_flt_mode(False)


def test_full_size_long_flt_double(O, D):
    """One channel of an hour-long float64 signal through stft_flt (Povey 400, hop 160.0, c0 = 200)."""
    import torch
    N = 57_600_000
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn((1, N), generator=g, device="cuda", dtype=torch.float64)
    win = torch.from_numpy(D.povey(400, np.float64)).cuda()
    W = int((N - 1 - 200) // 160)
    Y = D.stft_flt_dev(x, win, W, 200.0, 160.0)
    torch.cuda.synchronize()
    assert tuple(Y.shape) == (1, W, 257)
    xh = x[0].cpu().numpy()
    # head and tail against the oracle (the tail includes the zero-padded right edge)
    rc, Yref = O.stft_flt(xh[:20000], win.cpu().numpy(), 100, 200.0, 160.0, intended=True)
    check_spec(O, Y[0, :100].cpu().numpy(), Yref)
    w0 = W - 50
    s0 = w0 * 160
    rc, Yref = O.stft_flt(xh[s0:], win.cpu().numpy(), 50, 200.0, 160.0, intended=True)
    check_spec(O, Y[0, w0:].cpu().numpy(), Yref)
    # middle: equals the integer-hop stft with snipped edges on the same samples
    mid = 180000
    Ym = D.stft_dev(x[:, mid * 160: mid * 160 + 160 * 99 + 400], win, 160, snip_edges=True)
    torch.cuda.synchronize()
    assert torch.equal(Ym[0], Y[0, mid: mid + 100])


def test_full_size_long_window_4096(O, D):
    """One channel of 1 h @ 48 kHz float32, Blackman-Harris L = nfft = 4096, hop 1024."""
    import torch
    N = 172_800_000
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((1, N), generator=g, device="cuda", dtype=torch.float32)
    win = torch.from_numpy(D.blackmanharris(4096)).cuda()
    Y = D.stft_dev(x, win, 1024)
    torch.cuda.synchronize()
    W = (N + 512) // 1024
    assert tuple(Y.shape) == (1, W, 2049)
    xh = x[0, :1024 * 40].cpu().numpy()
    rc, Yref = O.stft(xh, win.cpu().numpy(), 1024, 4096, exact=False)
    check_spec(O, Y[0, :30].cpu().numpy(), Yref[:30])
    Y2 = D.stft_dev(x * 2, win, 1024)
    torch.cuda.synchronize()
    assert torch.equal(Y2, Y * 4)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_pow_compress_and_freqs(O, D, dt):
    """pow_compress_* is a pure element-wise map: element-wise relative bound (a few ulp of the
    library functions: 4e-6 float32, 2e-14 float64; exact for the identity branch)."""
    x = np.abs(signal(31, 20000, dt)) * dt(3)
    x[:4] = 0
    rtol = 4e-6 if dt == np.float32 else 2e-14
    for p, preg in [(1.0, 0.25), (0.0, 1e-6), (0.5, 0.0), (1.0 / 3.0, 1e-3), (0.3, 1e-3), (0.85, 0.0)]:
        rc, ref = O.pow_compress(x, p, preg)
        y = D.pow_compress(x, p, preg)
        if p == 1.0:
            assert np.array_equal(y, ref)
        else:
            np.testing.assert_allclose(y, ref, rtol=rtol, atol=rtol)
        z = x.copy()
        assert D.pow_compress(z, p, preg, inplace=True) is z and np.array_equal(z, y)
    for p, preg in [(-0.1, 0.0), (1.5, 0.0), (0.5, -1.0)]:
        with pytest.raises(D.DspError):
            D.pow_compress(x, p, preg)
    for nfft, F, fs in [(512, 257, 16000.0), (512, 400, 44100.0), (4096, 2049, 48000.0)]:
        assert np.array_equal(D.get_stft_freqs(nfft, F, fs, dt), O.get_stft_freqs(nfft, F, fs, dt)[1])
    with pytest.raises(D.DspError):
        D.get_stft_freqs(8, 9, 1.0, dt)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_fused_power_compression_matches_two_step_oracle(O, D, dt):
    """dsp_stft_pc_batch_*: STFT with the compression fused into the store == oracle stft followed by oracle
    pow_compress, in every kernel family (512 fast and generic, 4096, a direct-DFT size, a generic power of two)."""
    import torch

    def undo(y, p, preg):
        y = y.astype(np.float64)
        with np.errstate(over="ignore"):
            v = np.exp(y) if p == 0.0 else y if p == 1.0 else np.power(y, 1.0 / p)
        return (v - preg).astype(dt)

    for (N, L, stp, nfft) in [(16000, 400, 160, 512), (7013, 512, 33, 512), (30000, 4096, 1024, 4096), (3000, 300, 160, 300), (5000, 400, 160, 1024)]:
        x = np.stack([signal(N + b, N, dt) for b in range(3)])
        win = D.povey(L, dt) if L != 4096 else D.blackmanharris(L, dt)
        for p, preg, flags in [(0.0, 1e-6, (0, 0, 0)), (0.5, 0.0, (1, 0, 0)), (1.0 / 3.0, 1e-3, (0, 0, 0)), (0.3, 1e-2, (0, 1, 0)), (1.0, 0.5, (0, 0, 0))]:
            ref = []
            for b in range(3):
                rc, P = O.stft(x[b], win, stp, nfft, 0, *flags, exact=(nfft <= 1024))
                rc2, Pc = O.pow_compress(P, p, preg)
                assert rc == 0 and rc2 == 0
                ref.append(Pc)
            ref = np.stack(ref)
            for generic in (False, True):
                D.force_generic(generic)
                try:
                    Y = D.stft_pc_batch(x, win, p, preg, stp, False, *[bool(f) for f in flags], nfft=nfft)
                    Yd = D.stft_pc_dev(torch.from_numpy(x).cuda(), torch.from_numpy(win).cuda(), p, preg, stp, False,
                                       *[bool(f) for f in flags], nfft=nfft).cpu().numpy()
                finally:
                    D.force_generic(False)
                assert np.array_equal(Y, Yd)
                # compared in the domain the bound is stated for: the compression is undone in float64
                # first (a cube root turns a 1e-7-of-the-maximum rounding residue into 5e-3 of it)
                check_spec(O, undo(Y, p, preg), undo(ref, p, preg))
    with pytest.raises(D.DspError):
        D.stft_pc_batch(np.zeros((1, 2000), dt), np.ones(400, dt), 1.5)
    # device-resident pow_compress, in place
    t = torch.rand(100000, dtype=torch.float32 if dt == np.float32 else torch.float64, device="cuda")
    ref = O.pow_compress(t.cpu().numpy(), 0.3, 1e-3)[1]
    out = D.pow_compress_dev(t, 0.3, 1e-3, out=t)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=4e-6 if dt == np.float32 else 2e-14)


@pytest.mark.parametrize("tag", ["s", "d"])
def test_more_window_generators_against_reference_fixtures(D, tag):
    """blackman (both coefficient sets), bartlett, triangular, rectangular, flattop, gauss, tukey, planck
    on the device vs the reference's own generators (fixtures; keys name[@par].L.nK), lengths 1, 2, 3, 7, 8,
    400, 401, all four norms.  Same bounds as for the first four windows; the exponentials of gauss and
    planck are a few ulp from libm's: 1e-6 float32."""
    import torch
    g = load_golden(f"ref_windows2_{tag}.npz")
    dt = np.float32 if tag == "s" else np.float64
    seen = 0
    for key in g.files:
        head, L, norm = key.split(".n")[0].rsplit(".", 1)[0], int(key.split(".n")[0].rsplit(".", 1)[1]), int(key.rsplit(".n", 1)[1])
        name, _, par = head.partition("@")
        kw = {}
        if name == "blackman":
            kw["exact"] = par == "1"
        elif par:
            kw["par"] = float(par)
        ref = g[key]
        y = D.window(name, L, norm, dt, **kw)
        assert y.dtype == ref.dtype and y.shape == ref.shape
        tol = (1e-6 if name in ("gauss", "planck") else 3e-7) if tag == "s" else 1e-14
        bound = tol * (10 if norm in (1, 2) else 1) * np.max(np.abs(ref))
        assert np.max(np.abs(y.astype(np.float64) - ref)) <= bound, (key, np.max(np.abs(y - ref)))
        yd = D.window_dev(name, L, norm, torch.float32 if tag == "s" else torch.float64, **kw).cpu().numpy()
        assert np.array_equal(y, yd)
        seen += 1
    assert seen >= 15 * (4 + 3 * 4) - 20
    for name, bad in [("gauss", 0.0), ("gauss", -1.0), ("tukey", 1.5), ("tukey", -0.1), ("planck", 0.6), ("planck", -0.1)]:
        with pytest.raises(D.DspError):
            D.window(name, 16, 0, dt, par=bad)
    with pytest.raises(D.DspError):
        D.window("flattop", 16, 4, dt)
    # where the reference writes past its L values or leaves the centre unwritten
    assert np.array_equal(D.window("planck", 9, 0, dt, par=0.0), np.ones(9, dt))
    y = D.window("planck", 9, 0, dt, par=0.5)
    assert y[0] == 0 and np.array_equal(y, y[::-1]) and 0.5 < y[4] <= 1


@pytest.mark.parametrize("tag", ["s", "d"])
def test_window_generators_against_reference_fixtures(D, tag):
    """hann/hamming/povey/blackmanharris computed on the device vs what the reference's own generators
    returned (fixtures), all four norms.  Bound: absolute, relative to the window's largest element:
    3e-7 float32 (the reference evaluates cosf/powf in float, a few ulp from the device's), 1e-14 float64;
    for norms 1 and 2, whose divisor the reference accumulates serially in T over L terms, 3e-6 and 1e-13."""
    import torch
    g = load_golden(f"ref_windows_{tag}.npz")
    dt = np.float32 if tag == "s" else np.float64
    tol = 3e-7 if tag == "s" else 1e-14
    seen = 0
    for key in g.files:
        parts = key.split(".")
        name, L = parts[0], int(parts[1])
        norm = int(parts[2][1:]) if len(parts) > 2 else 0
        ref = g[key]
        y = D.window(name, L, norm, dt)
        assert y.dtype == ref.dtype and y.shape == ref.shape
        scale = np.max(np.abs(ref)) if L else 1.0
        bound = tol * (10 if norm in (1, 2) else 1)
        assert np.max(np.abs(y.astype(np.float64) - ref)) <= bound * scale, (key, np.max(np.abs(y - ref)) / scale)
        yd = D.window_dev(name, L, norm, torch.float32 if tag == "s" else torch.float64).cpu().numpy()
        assert np.array_equal(y, yd)
        seen += 1
    assert seen >= 57
    with pytest.raises(D.DspError):
        D.window("hann", 16, 4, dt)
    assert D.window("hann", 0, 0, dt).size == 0
    # a generated window drives the transform like a host-made one
    x = signal(3, 8000, dt)
    win = D.window_dev("povey", 400, 0, torch.float32 if tag == "s" else torch.float64)
    Y = D.stft_dev(torch.from_numpy(x).cuda(), win, 160).cpu().numpy()[0]
    assert np.array_equal(Y, D.stft(x, win.cpu().numpy(), 160))


@pytest.mark.parametrize("tag", ["s", "d"])
def test_window_univar_against_reference_fixtures(O, D, tag):
    """window_univar_* / window_univar_flt_* (framing and windowing in one step) are a copy and one
    multiplication: bit-exact against what the reference's own code returned."""
    g = load_golden(f"ref_chain_{tag}.npz")
    seen = 0
    for name in ["base", "short_multifold", "tiny"]:
        x, win = g[f"{name}.x"], g[f"{name}.win"]
        N, L, stp, nfft = (int(v) for v in g[f"{name}.params"])
        for snip in (0, 1):
            key = f"{name}.snip{snip}.windowed"
            if key not in g:
                continue
            assert np.array_equal(D.window_univar(x, win, stp, bool(snip)), g[key])
            seen += 1
    assert seen >= 4
    g = load_golden(f"ref_flt_{tag}.npz")
    x, win = g["x"], g["win"]
    for i, (c0, stp, W) in enumerate(g["cases"]):
        if f"case{i}.windowed" in g:
            assert np.array_equal(D.window_univar_flt(x, win, int(W), c0, stp), g[f"case{i}.windowed"])
    rc, fr = O.frame_univar(x, win.size, 77, 0)
    assert np.array_equal(D.window_univar(x, win, 77), fr * win[None, :])
    with pytest.raises(D.DspError):
        D.window_univar(x[:100], win, 10, True)
    # complex signal, real window (window_univar_c/_z, window_univar_flt_c/_z)
    g = load_golden(f"ref_wu_c_{tag}.npz")
    xc, h = g["x"], g["win"]
    assert np.array_equal(D.window_univar(xc, h, 48), g["wu.stp48"])
    assert np.array_equal(D.window_univar(xc[:30], h, 20), g["wu.short"])
    # other hops: the reference's complex form steps back floats instead of complex values after a frame
    # inside the signal (c/window_univar.c:184-186) and reads past it; the complex framing itself is pinned
    for stp in (20, 61):
        rc, fr = O.frame_univar(xc, h.size, stp, 0)
        assert np.array_equal(D.window_univar(xc, h, stp), fr * h[None, :])
    for i in range(3):
        W, c0, stp = g[f"wuf.{i}.par"]
        assert np.array_equal(D.window_univar_flt(xc, h, int(W), c0, stp), g[f"wuf.{i}"])
    # with snipped edges the reference's complex form mis-steps; frame w starts at w * stp here
    rc, fr = O.frame_univar(xc, h.size, 20, 1)
    assert np.array_equal(D.window_univar(xc, h, 20, True), fr * h[None, :])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_complex_window_forms(D, dt):
    """hann_c/_z ... planck_c/_z: the values of the real forms (bit for bit), each followed by a zero imaginary part
    (what the reference's complex forms are: tests/test_oracle_semantics.py runs them)."""
    cdt = np.complex64 if dt == np.float32 else np.complex128
    for name in D.api.WINDOW_IDS:
        for L in (1, 2, 31, 400, 4096):
            for norm in (0, 1, 2, 3):
                for exact in ((False, True) if name == "blackman" else (False,)):
                    re = D.window(name, L, norm, dt, exact=exact)
                    cx = D.window(name, L, norm, cdt, exact=exact)
                    assert cx.dtype == cdt and cx.shape == (L,)
                    assert np.array_equal(cx.real, re, equal_nan=True) and np.all(cx.imag == 0)
    assert D.window("hann", 0, 0, cdt).size == 0
    with pytest.raises(D.DspError):
        D.window("hann", 8, 4, cdt)


def test_host_entry_points_from_several_threads(O, D):
    """The C entry points are re-entrant (the reference is not: FFTW's planner, c/stft.c:53): six host threads call
    them at once for a while -- single host-pointer transforms of both precisions, a streamed host batch (pinned
    stages, copy threads), stft_flt with more distinct frame-start tables than the per-device cache holds (entries
    are evicted while other threads launch with theirs), the nfft-4096 run form (its counter sets) and a thread
    with the documented stft_flt framing selected (a per-thread setting that must not leak into the others).
    Every result equals what the same call gave alone (the run form: to the tolerance, it deals its work at run time)."""
    import threading
    import torch
    rng = np.random.default_rng(77)
    x32 = (rng.standard_normal(48000) * 0.1).astype(np.float32)
    x64 = (rng.standard_normal(30000) * 0.1).astype(np.float64)
    xb = (rng.standard_normal((110, 160000)) * 0.1).astype(np.float32)            # 70 MB: streamed
    xl = torch.randn((8, 1 << 24), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 0.1   # 4096 runs of 32 frames
    w32, w64, wp = D.hann(400, np.float32), D.hann(400, np.float64), D.povey(400, np.float32)
    wl = torch.from_numpy(D.blackmanharris(4096)).cuda()
    hops = [150.25 + 0.5 * i for i in range(12)]                                   # 12 tables: more than the cache keeps
    Wf = int((48000 - 1 - 200) // hops[-1])

    def flt_all():
        return [D.stft_flt(x32, wp, Wf, 200.0, h) for h in hops]

    def flt_intended():
        D.set_flt_mode(True)
        try:
            return D.stft_flt(x32, wp, Wf, -45.0, 100.0)
        finally:
            D.set_flt_mode(False)

    def long_window():
        y = D.stft_dev(xl, wl, 1024, nfft=4096)
        torch.cuda.synchronize()
        return y

    jobs = {
        "stft_s": lambda: D.stft(x32, w32, 160, mn0=True),
        "stft_d": lambda: D.stft(x64, w64, 160, lg=True),
        "batch": lambda: D.stft_batch(xb, w32, 160),
        "flt_tables": flt_all,
        "flt_intended": flt_intended,
        "flt_compat": lambda: D.stft_flt(x32, wp, Wf, -45.0, 100.0),
        "r4096": long_window,
    }
    alone = {k: f() for k, f in jobs.items()}
    assert not np.array_equal(alone["flt_intended"], alone["flt_compat"])          # the two framings differ here
    errors = []

    def worker(name, reps):
        try:
            for _ in range(reps):
                got = jobs[name]()
                want = alone[name]
                if name == "r4096":
                    assert float((got - want).abs().max() / want.max()) <= 1e-6
                elif isinstance(want, list):
                    assert all(np.array_equal(a, b) for a, b in zip(got, want))
                else:
                    assert np.array_equal(got, want), name
        except Exception as e:                                                     # noqa: BLE001 - reported below
            errors.append((name, repr(e)))

    reps = {"stft_s": 40, "stft_d": 40, "batch": 3, "flt_tables": 4, "flt_intended": 30, "flt_compat": 30, "r4096": 6}
    threads = [threading.Thread(target=worker, args=(k, reps[k])) for k in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    rc, Yref = O.stft(x32, w32, 160, 512, 0, 1)
    check_spec(O, alone["stft_s"], Yref)
    rc, Yref = O.stft(xl[3, :40000].cpu().numpy(), wl.cpu().numpy(), 1024, 4096, exact=False)
    check_spec(O, alone["r4096"][3, :Yref.shape[0] - 3].cpu().numpy(), Yref[:-3])


def test_stft_flt_reference_framing_head_and_tail(O, D):
    """stft_flt in the reference's framing (c/stft_flt.c:66-88) with first frames that hang over the start of the signal:
    the frames that start inside the signal are read `shift` samples early, the ones before them where they start.  The
    library transforms such a call in two launches (head frames by the table; the others as frames of the signal cut to
    N - shift samples, on the fast kernels): batches of several signals (the tail's rows of signal b start W rows after
    those of signal b - 1, not W - head), every kernel family, whole and fractional hops, a head longer than one step, a
    frame count that ends inside and beyond the signal, the flags; against the oracle's restatement of the reference's
    walk, signal by signal."""
    import torch
    rng = np.random.default_rng(123)
    cases = [(np.float32, 512, 400, 0.0, 160.0, 0), (np.float32, 512, 400, 37.0, 160.5, 1), (np.float64, 512, 400, 0.0, 160.0, 2),
             (np.float64, 512, 400, 150.0, 100.25, 0), (np.float32, 512, 512, -300.0, 64.0, 3), (np.float32, 256, 200, 0.0, 80.0, 0),
             (np.float32, 128, 100, 10.0, 40.5, 1), (np.float32, 64, 64, 0.0, 16.0, 0), (np.float32, 1024, 1024, 100.0, 256.0, 2),
             (np.float32, 2048, 2048, 0.0, 512.5, 0), (np.float32, 4096, 4096, 1000.0, 1024.0, 0), (np.float64, 1024, 800, 0.0, 200.0, 1),
             (np.float32, 100, 100, 0.0, 25.0, 3), (np.float32, 512, 400, 0.0, 3.0, 0)]
    flag_sets = [(0, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 0)]
    for (dt, nfft, L, c0, stp, fl) in cases:
        B = 3
        for N in (60000, 9000):
            for W in (int((N - 1 - c0) // stp) - 1, min(N, int((N + 3 * L - c0) // stp))):
                x = (rng.standard_normal((B, N)) * 0.1 + 0.05).astype(dt)
                win = D.povey(L, dt)
                mn0, amp, lg = flag_sets[fl]
                Y = D.stft_flt_dev(torch.from_numpy(x).cuda(), torch.from_numpy(win).cuda(), W, c0, stp, bool(mn0), bool(amp), bool(lg), nfft=nfft)
                Y = Y.cpu().numpy()
                for b in range(B):
                    rc, Yref = O.stft_flt(x[b].astype(np.float64), win.astype(np.float64), W, c0, stp, nfft, mn0, amp, lg, intended=False, exact=False)
                    assert rc == 0
                    check_spec(O, Y[b], Yref.astype(dt), amp, lg)


def test_host_streamed_short_last_range(O, D):
    """A long signal whose last range of frames holds only two frames (float32: ranges of 30 080 frames): the kernels that
    stage a step's whole span then reflect samples from well before the range's own first sample; the staged range must
    reach down to them.  Same bits as one launch over the signal on the device; the signal's last frames against the oracle."""
    import torch
    fw = 30080
    W = 6 * fw + 2
    N = W * 160 - 80
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(N) * 0.1).astype(np.float32)
    win = D.hann(400, np.float32)
    assert D.num_frames(N, 400, 160) == W
    Yh = D.stft(x, win, 160)
    Yd = D.stft_dev(torch.from_numpy(x).cuda()[None, :], torch.from_numpy(win).cuda(), 160)[0].cpu().numpy()
    assert np.array_equal(Yh, Yd)
    tail = 160 * 60 + (N % 160)
    rc, Yref = O.stft(x[-tail:], win, 160, 512)
    check_spec(O, Yh[-40:], Yref[-40:])


def test_host_long_signal_streamed_by_frames(O, D):
    """One long signal through the reference's own host-pointer entry points (stft_s/_d, stft_flt_s/_d: configurations
    3 and 4 of the benchmark list are 64 / 128 such calls): from 64 MB of input + output on the call goes through in
    ranges of frames, each with the samples it reads (the reflected ones at the ends of the signal included), copies
    overlapped with the kernel.  Same bits as one launch over the whole signal on the device (nfft 4096: to the
    tolerance, its run form transforms a frame rotated by a multiple of a quarter of its length, which depends on
    where a range starts), and equal to the oracle at both ends and across the first boundary between two ranges."""
    import torch
    rng = np.random.default_rng(99)
    for dt, N in ((np.float32, 30_000_000), (np.float64, 12_000_017)):
        x = (rng.standard_normal(N) * 0.1).astype(dt)
        xd = torch.from_numpy(x).cuda()
        win = D.hann(400, dt)
        wd = torch.from_numpy(win).cuda()
        Yh = D.stft(x, win, 160, mn0=True)
        Yd = D.stft_dev(xd[None, :], wd, 160, mn0=True)[0].cpu().numpy()
        assert np.array_equal(Yh, Yd)
        # both ends against the oracle (the last frames of the signal's tail are those of the whole signal)
        rc, Yref = O.stft(x[:20000], win, 160, 512, 0, 1)
        check_spec(O, Yh[:100], Yref[:100])
        tail = 160 * 120 + (N % 160)
        rc, Yref = O.stft(x[-tail:], win, 160, 512, 0, 1)
        check_spec(O, Yh[-100:], Yref[-100:])
        # stft_flt with a start of the first frame above 0 (reference framing: every read 100 samples early) and with
        # the documented framing
        W = (N - 1 - 300) // 160
        for intended in (False, True):
            D.set_flt_mode(intended)
            try:
                Zh = D.stft_flt(x, win, W, 300.0, 160.0)
                Zd = D.stft_flt_dev(xd[None, :], wd, W, 300.0, 160.0)[0].cpu().numpy()
            finally:
                D.set_flt_mode(False)
            assert np.array_equal(Zh, Zd)
        # a hop that is no whole number: the frames come from a table of starts, cut into the same ranges
        W = int((N - 1 - 200) // 160.5)
        Zh = D.stft_flt(x, win, W, 200.0, 160.5, lg=True)
        Zd = D.stft_flt_dev(xd[None, :], wd, W, 200.0, 160.5, lg=True)[0].cpu().numpy()
        assert np.array_equal(Zh, Zd)
        # reference framing with first frames that hang over the start of the signal (two launches per range of frames:
        # what is left of the head by the table, the rest as frames of the cut signal), whole and fractional hop
        for (c0, stp) in ((0.0, 160.0), (37.0, 160.5)):
            W = int((N - 1 - c0) // stp) - 2
            Zh = D.stft_flt(x, win, W, c0, stp)
            Zd = D.stft_flt_dev(xd[None, :], wd, W, c0, stp)[0].cpu().numpy()
            assert np.array_equal(Zh, Zd)
            rc, Zref = O.stft_flt(x[:40000].astype(np.float64), win.astype(np.float64), 200, c0, stp, 512, 0, 0, 0, intended=False, exact=False)
            assert rc == 0
            check_spec(O, Zh[:200], Zref.astype(dt))
        del Yh, Yd, Zh, Zd
    # other kernel families: nfft 1024 (hop 256), nfft 4096 (hop 1024: run form inside the ranges), odd hop
    x = (rng.standard_normal(40_000_000) * 0.1).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    for (L, hop, nfft, exact) in ((1024, 256, 1024, True), (4096, 1024, 4096, False), (400, 161, 512, True), (200, 80, 256, True)):
        win = D.blackmanharris(L, np.float32)
        Yh = D.stft(x, win, hop, nfft=nfft)
        Yd = D.stft_dev(xd[None, :], torch.from_numpy(win).cuda(), hop, nfft=nfft)[0].cpu().numpy()
        if exact:
            assert np.array_equal(Yh, Yd), (L, hop, nfft)
        else:
            assert O.normalised_err(Yh, Yd) <= 1e-6
        del Yh, Yd


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_stft_flt_host_batch(O, D, dt):
    """dsp_stft_flt_batch_s/_d, the host-pointer form of stft_flt for B signals (a multivariate recording in one call):
    every signal has the bits of the reference's own stft_flt call on it; rows with a stride; a batch large enough to be
    streamed (groups of signals for short ones, ranges of frames for long ones), with an integral hop and with a table
    of frame starts."""
    rng = np.random.default_rng(31)
    win = D.povey(400, dt)
    buf = (rng.standard_normal((5, 20000 + 16)) * 0.1).astype(dt)
    for xs in (np.ascontiguousarray(buf[:, :20000]), buf[:, 8:20008]):
        for (c0, stp) in ((200.0, 160.0), (-45.0, 100.0), (300.0, 160.5)):
            W = int((20000 - 1 - max(c0, 0)) // stp)
            Y = D.stft_flt_batch(xs, win, W, c0, stp, mn0=True)
            for b in range(5):
                assert np.array_equal(Y[b], D.stft_flt(np.ascontiguousarray(xs[b]), win, W, c0, stp, mn0=True))
            rc, ref = O.stft_flt(np.ascontiguousarray(xs[4]), win, W, c0, stp, 512, True, False, False)
            check_spec(O, Y[4], ref)
    # streamed: 40 signals of 10 s (groups of signals), 3 signals of 12 M samples (ranges of frames)
    for (B, N) in ((300 if dt == np.float32 else 150, 160000), (3, 12_000_000 if dt == np.float32 else 6_000_000)):
        x = (rng.standard_normal((B, N)) * 0.1).astype(dt)
        for stp in (160.0, 160.5):
            W = int((N - 1 - 200) // stp)
            Y = D.stft_flt_batch(x, win, W, 200.0, stp)
            for b in (0, B - 1):
                assert np.array_equal(Y[b], D.stft_flt(x[b], win, W, 200.0, stp))


def test_release_buffers_between_host_calls(O, D):
    """dsp_stft_release_buffers gives back the workspace, the pinned stages and the cached stft_flt tables of the device;
    the calls after it allocate again and return the same bits (an unstreamed call, a streamed batch, a table framing)."""
    import torch
    from dsp_amd._lib import lib
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((70, 160000)) * 0.1).astype(np.float32)
    win = D.hann(400, np.float32)
    W = int((160000 - 1 - 200) // 160.5)
    before = (D.stft(x[0], win, 160), D.stft_batch(x, win, 160), D.stft_flt(x[1], win, W, 200.0, 160.5))
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    assert lib().dsp_stft_release_buffers() == 0
    assert torch.cuda.mem_get_info()[0] > free0                  # device memory came back
    after = (D.stft(x[0], win, 160), D.stft_batch(x, win, 160), D.stft_flt(x[1], win, W, 200.0, 160.5))
    for a, b in zip(before, after):
        assert np.array_equal(a, b)
    assert lib().dsp_stft_release_buffers() == 0 and lib().dsp_stft_release_buffers() == 0   # idempotent


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_complex_spectrogram(O, D, dt):
    """dsp_stft_cplx_batch_dev_s/_d: frames times window, then the real FFT of each frame (what window_univar followed
    by fft_fftw_r2hc gives, c/window_univar.c:41-89, c/fft.fftw.r2hc.c:69-78), against the oracle's framing and r2hc
    restatements applied one after the other; its squared magnitude equals the power spectrogram of stft on the same
    input.  Shapes of the fused 512-point kernel (all span forms, odd hop, 13- and 16-row windows, blocks of rows off
    16-byte boundaries, a batch), of the 64 / 128 / 256- and 1024 / 2048-point kernels and of the unfused fallback (other nfft, hop
    above 512, a transform length that is no power of two); the 1/sqrt(2 nfft) scaling."""
    import torch
    tol = TOL[np.dtype(dt)]
    rng = np.random.default_rng(41)
    cases = [(3, 16000, 400, 160, 512, False), (2, 16001, 400, 161, 512, False), (2, 9000, 512, 256, 512, False), (1, 20000, 300, 400, 512, True),
             (2, 7000, 400, 200, 512, False), (1, 5000, 512, 128, 512, False), (5, 1603, 400, 160, 512, False),
             (2, 6000, 256, 64, 256, False), (1, 9000, 1024, 256, 1024, False), (1, 9000, 400, 600, 512, False), (2, 3000, 100, 50, 120, False),
             # the 1024 / 2048-point kernel's complex form
             (2, 20000, 800, 200, 1024, False), (3, 9000, 1024, 512, 1024, True), (2, 30000, 2048, 512, 2048, False), (2, 12001, 1200, 301, 2048, False),
             (1, 9001, 1024, 255, 1024, False), (5, 4099, 2048, 1024, 2048, False),
             # the 256-point kernel's
             (3, 8000, 200, 80, 256, False), (2, 8001, 256, 81, 256, True), (4, 3001, 256, 128, 256, False), (1, 9000, 130, 200, 256, False),
             # the 128- and 64-point forms of that kernel
             (3, 8000, 100, 40, 128, False), (2, 8001, 128, 41, 128, True), (4, 3001, 128, 64, 128, False), (1, 9000, 70, 128, 128, False),
             (3, 4000, 50, 20, 64, False), (2, 4001, 64, 21, 64, True), (4, 3001, 64, 32, 64, False), (1, 5000, 33, 47, 64, False), (2, 4000, 64, 16, 64, False)]
    for (B, N, L, hop, nfft, snip) in cases:
        x = (rng.standard_normal((B, N)) * 0.1).astype(dt)
        win = D.povey(L, dt) if L > 1 else np.ones(1, dt)
        xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(win).cuda()
        for sc in (False, True):
            Y = D.stft_cplx_dev(xd, wd, hop, snip_edges=snip, sc=sc, nfft=nfft)
            P = D.stft_dev(xd, wd, hop, snip_edges=snip, nfft=nfft)
            torch.cuda.synchronize()
            Yh, Ph = Y.cpu().numpy(), P.cpu().numpy()
            k = (1.0 / np.sqrt(2.0 * nfft)) if sc else 1.0
            for b in range(B):
                rc, fr = O.frame_univar(x[b], L, hop, snip)
                assert rc == 0
                rc, ref = O.fft_r2hc(np.ascontiguousarray(fr * win[None, :]), 1, nfft, False, sc)
                assert rc == 0 and Yh[b].shape == ref.shape, (B, N, L, hop, nfft)
                # per frame, relative to the frame's largest bin: amplitudes carry half the digits of the power bound
                scale = np.abs(ref).max(axis=1, keepdims=True)
                err = float((np.abs(Yh[b] - ref) / np.maximum(scale, 1e-300)).max())
                assert err <= (3e-6 if dt == np.float32 else 1e-14), ((B, N, L, hop, nfft, snip, sc), err)
                assert np.all(Yh[b][:, 0].imag == 0) and np.all(Yh[b][:, -1].imag == 0)
                mag2 = (Yh[b].real.astype(np.float64) ** 2 + Yh[b].imag.astype(np.float64) ** 2) / (k * k)
                assert O.normalised_err(mag2.astype(dt), Ph[b]) <= tol


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_complex_spectrogram_4096(O, D, dt):
    """The 4096-point kernel's complex form: its frame loop (full and short windows) and its run loop (forced at a small
    size), where the frame in the registers is rotated by a multiple of a quarter of its length and the complex form turns
    every bin back by the matching power of i."""
    import torch
    rng = np.random.default_rng(43)
    old = os.environ.get("DSP_R4096_MIN_RUNS")
    try:
        for (B, N, L, hop, run) in ((1, 40000, 4096, 1024, False), (2, 30000, 3000, 750, False), (2, 1024 * 150 + 5, 4096, 1024, True),
                                    (3, 4096 * 9, 4096, 2048, False)):
            if run:
                os.environ["DSP_R4096_MIN_RUNS"] = "1"
            elif "DSP_R4096_MIN_RUNS" in os.environ:
                del os.environ["DSP_R4096_MIN_RUNS"]
            x = (rng.standard_normal((B, N)) * 0.1).astype(dt)
            win = D.blackmanharris(L, dt)
            xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(win).cuda()
            Y = D.stft_cplx_dev(xd, wd, hop, nfft=4096, sc=True)
            P = D.stft_dev(xd, wd, hop, nfft=4096)
            torch.cuda.synchronize()
            Yh, Ph = Y.cpu().numpy(), P.cpu().numpy()
            for b in range(B):
                rc, fr = O.frame_univar(x[b], L, hop, False)
                rc2, ref = O.fft_r2hc(np.ascontiguousarray(fr * win[None, :]).astype(np.float64), 1, 4096, False, True, exact=False)
                assert rc == 0 and rc2 == 0 and Yh[b].shape == ref.shape
                scale = np.abs(ref).max(axis=1, keepdims=True)
                err = float((np.abs(Yh[b] - ref) / np.maximum(scale, 1e-300)).max())
                assert err <= (3e-6 if dt == np.float32 else 1e-14), ((B, N, L, hop, run), err)
                assert np.all(Yh[b][:, 0].imag == 0) and np.all(Yh[b][:, -1].imag == 0)
                mag2 = (Yh[b].real.astype(np.float64) ** 2 + Yh[b].imag.astype(np.float64) ** 2) * (2.0 * 4096)
                assert O.normalised_err(mag2.astype(dt), Ph[b]) <= TOL[np.dtype(dt)]
    finally:
        if old is None:
            os.environ.pop("DSP_R4096_MIN_RUNS", None)
        else:
            os.environ["DSP_R4096_MIN_RUNS"] = old
