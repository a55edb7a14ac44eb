"""The stft / stft_flt command-line tools and their tensor-file layer (NumPy .npy, Armadillo binary).

CPU part: the file layer through bin/dsp_tensor against NumPy's own reader/writer, the option grammar and
every check the reference tool makes before it computes (reference src/stft.cpp:118-175,
srci/stft_flt.cpp:80-104).  GPU part: whole runs of the tools against the oracle.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import TOL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, os.environ.get("DSP_CLI_BIN", "bin"))      # bin_san: the sanitizer build (make -C cli SAN=1)


def run(args, stdin=None, **kw):
    return subprocess.run(args, input=stdin, capture_output=True, timeout=120, **kw)


@pytest.fixture(scope="module")
def tools():
    for t in ("stft", "stft_flt", "dsp_tensor"):
        assert os.path.exists(os.path.join(BIN, t)), "run __graft_entry__.build() (make -C cli) first"
    return BIN


def arma_bytes(a):
    a = np.asarray(a)
    code = {np.dtype(np.float32): "FN004", np.dtype(np.float64): "FN008", np.dtype(np.complex64): "FC008", np.dtype(np.complex128): "FC016"}[a.dtype]
    if a.ndim == 3:
        head = f"ARMA_CUB_BIN_{code}\n{a.shape[0]} {a.shape[1]} {a.shape[2]}\n"
    else:
        a2 = a.reshape(a.shape[0], -1)
        head = f"ARMA_MAT_BIN_{code}\n{a2.shape[0]} {a2.shape[1]}\n"
    return head.encode() + np.asfortranarray(a).tobytes(order="F")


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
def test_npy_headers_round_trip_through_numpy(tools, tmp_path, dt):
    rng = np.random.default_rng(1)
    for shape, order in [((7,), "C"), ((3, 5), "C"), ((3, 5), "F"), ((2, 3, 4), "C"), ((2, 3, 4, 5), "F"), ((1, 9), "C"), ((0,), "C")]:
        a = rng.standard_normal(shape).astype(dt)
        a = np.asfortranarray(a) if order == "F" else np.ascontiguousarray(a)
        src, dst = tmp_path / "a.npy", tmp_path / "b.npy"
        np.save(src, a)
        r = run([os.path.join(tools, "dsp_tensor"), "info", str(src)])
        assert r.returncode == 0, r.stderr
        f = r.stdout.decode().split()
        dims = list(shape) + [1] * (4 - len(shape))
        assert f[0] == "npy" and int(f[1]) == {4: 1, 8: 2}[a.real.dtype.itemsize] + (100 if np.iscomplexobj(a) else 0)
        assert [int(v) for v in f[2:6]] == dims
        col = a.ndim > 1 and a.flags["F_CONTIGUOUS"] and not a.flags["C_CONTIGUOUS"]
        assert f[6] == ("colmajor" if col else "rowmajor")
        r = run([os.path.join(tools, "dsp_tensor"), "copy", str(src), "-o", str(dst)])
        assert r.returncode == 0, r.stderr
        b = np.load(dst)
        assert b.dtype == a.dtype and b.size == a.size
        assert np.array_equal(b.reshape(-1, order="A"), a.reshape(-1, order="A")) or a.size == 0
        if a.ndim >= 2:
            assert b.shape == a.shape and np.array_equal(a, b)


def test_npy_version_2_header_and_stdin(tools, tmp_path):
    a = np.arange(10, dtype=np.float32)
    import numpy.lib.format as fmt
    p = tmp_path / "v2.npy"
    with open(p, "wb") as fh:
        fmt.write_array(fh, a, version=(2, 0))
    r = run([os.path.join(tools, "dsp_tensor"), "info"], stdin=p.read_bytes())
    assert r.returncode == 0 and r.stdout.decode().split()[:3] == ["npy", "1", "10"]


def test_armadillo_headers(tools, tmp_path):
    rng = np.random.default_rng(2)
    for a in [rng.standard_normal((4, 3)).astype(np.float32), rng.standard_normal((5, 1)), rng.standard_normal((2, 3, 2)).astype(np.float32),
              (rng.standard_normal((3, 2)) + 1j * rng.standard_normal((3, 2))).astype(np.complex64)]:
        src = tmp_path / "a.bin"
        src.write_bytes(arma_bytes(a))
        r = run([os.path.join(tools, "dsp_tensor"), "info", str(src)])
        f = r.stdout.decode().split()
        assert r.returncode == 0 and f[0] == "arma" and f[6] == "colmajor"
        assert [int(v) for v in f[2:5]] == (list(a.shape) + [1])[:3]
        # Armadillo -> NumPy keeps the values and the column-major order; and back gives the same bytes
        dst = tmp_path / "a.npy"
        assert run([os.path.join(tools, "dsp_tensor"), "copy", "-f", "npy", str(src), "-o", str(dst)]).returncode == 0
        b = np.load(dst)
        assert np.array_equal(b, a.reshape(b.shape)) and b.flags["F_CONTIGUOUS"]
        back = run([os.path.join(tools, "dsp_tensor"), "copy", "-f", "arma", str(dst)])
        assert back.returncode == 0 and back.stdout == arma_bytes(a)
    # a row-major tensor cannot become an Armadillo file
    p = tmp_path / "c.npy"
    np.save(p, np.zeros((2, 3), np.float32))
    r = run([os.path.join(tools, "dsp_tensor"), "copy", "-f", "arma", str(p), "-o", str(tmp_path / "x")])
    assert r.returncode == 1 and b"column-major" in r.stderr


def test_bad_files_are_rejected(tools, tmp_path):
    for payload, msg in [(b"", b"empty input"), (b"hello world", b"unknown file format"), (b"\x93NUMPY\x01\x00\x10\x00short", b"truncated"),
                         (b"ARMA_MAT_BIN_IS004\n2 2\n", b"element type")]:
        r = run([os.path.join(tools, "dsp_tensor"), "info"], stdin=payload)
        assert r.returncode == 1 and msg in r.stderr, r.stderr
    p = tmp_path / "i.npy"
    np.save(p, np.arange(4))
    r = run([os.path.join(tools, "dsp_tensor"), "info", str(p)])
    assert r.returncode == 1 and b"dtype" in r.stderr


def test_option_grammar_and_checks_before_compute(tools, tmp_path):
    x32, w32, w64, mat, cx = (tmp_path / n for n in ("x.npy", "w.npy", "w64.npy", "m.npy", "c.npy"))
    np.save(x32, np.zeros(2000, np.float32)); np.save(w32, np.ones(400, np.float32)); np.save(w64, np.ones(400))
    np.save(mat, np.zeros((4, 5), np.float32)); np.save(cx, np.zeros(8, np.complex64))
    stft, flt = os.path.join(tools, "stft"), os.path.join(tools, "stft_flt")
    out = str(tmp_path / "y.npy")
    cube, w5, matF = tmp_path / "cube.npy", tmp_path / "w5.npy", tmp_path / "mF.npy"
    np.save(cube, np.zeros((2, 3, 4), np.float32)); np.save(w5, np.ones(3, np.float32)); np.save(matF, np.asfortranarray(np.zeros((4, 5), np.float32)))
    cases = [
        ([stft, "-h"], 1, b"", b"Usage: stft"),
        ([stft, "--help"], 1, b"", b"--snip-edges"),
        ([stft, "-s0", str(x32), str(w32), "-o", out], 1, b"stp must be positive", b""),
        ([stft, "--step=-3", str(x32), str(w32), "-o", out], 1, b"stp must be positive", b""),
        ([stft, "-s", "abc", str(x32), str(w32)], 1, b"invalid argument", b""),
        ([stft, "-q", str(x32), str(w32)], 1, b"invalid option", b""),
        ([stft, "-e", "-e", str(x32), str(w32)], 1, b"excess option", b""),
        ([stft, str(x32), str(w32), str(w32)], 1, b"unexpected argument", b""),
        ([stft, str(x32), str(w64), "-o", out], 1, b"inputs must have the same data type", b""),
        ([stft, str(cx), str(w32), "-o", out], 1, b"inputs must be real-valued", b""),
        ([stft, str(mat), str(w32), "-o", out], 1, b"input 1 (X1) must be a vector", b""),
        ([stft, str(x32), str(mat), "-o", out], 1, b"input 2 (X2) must be a vector", b""),
        ([stft, str(tmp_path / "missing.npy"), str(w32), "-o", out], 1, b"problem opening input file 1", b""),
        ([flt, "-h"], 1, b"", b"Usage: stft_flt"),
        ([flt, "-c", "5000", str(x32), str(w32), "-o", out], 1, b"c0 (center of first frame) must be <= N-1", b""),
        ([flt, "-s0", str(x32), str(w32), "-o", out], 1, b"stp (step size) must be positive", b""),
        ([flt, "-w0", str(x32), str(w32), "-o", out], 1, b"W (nframes) must be positive", b""),
        ([flt, str(x32), str(mat), "-o", out], 1, b"input 2 (X2) must be a vector", b""),
        # the opt-in additions: a matrix of signals, devices
        ([stft, "-d2", str(mat), str(w32), "-o", out], 1, b"dim must be in {0,1}", b""),
        ([stft, "-d0", str(cube), str(w32), "-o", out], 1, b"input 1 (X1) must be a matrix or a vector with -d", b""),
        ([stft, "--device", "0", "--ngpu", "2", str(x32), str(w32), "-o", out], 1, b"--device and --ngpu cannot be combined", b""),
        ([stft, "--ngpu", "0", str(x32), str(w32), "-o", out], 1, b"ngpu must be in [1 64]", b""),
        ([stft, "--device", "99", str(x32), str(w32), "-o", out], 1, b"device must be in [0 63]", b""),
        ([stft, "-d0", "--ngpu", "2", str(mat), str(w5), "-o", out], 1, b"with --ngpu the signals must be contiguous", b""),
        ([flt, "-d", "1", "--ngpu=2", str(matF), str(w5), "-o", out], 1, b"with --ngpu the signals must be contiguous", b""),
        ([flt, "-d0", "-c", "4", str(mat), str(w5), "-o", out], 1, b"c0 (center of first frame) must be <= N-1", b""),
    ]
    for args, rc, err, outtxt in cases:
        r = run(args)
        assert r.returncode == rc and err in r.stderr and outtxt in r.stdout, (args, r)
    # two stdin inputs
    r = run([stft, "-", "-"], stdin=b"")
    assert r.returncode == 1 and b"can only use stdin for one input" in r.stderr
    r = run([stft], stdin=b"")
    assert r.returncode == 1 and b"can only use stdin for one input" in r.stderr


# ------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def check(O, Y, Yref, lg=False):
    assert Y.shape == Yref.shape and Y.dtype == Yref.dtype
    tol = TOL[np.dtype(Yref.dtype)]
    if lg:
        Y, Yref = np.exp(Y.astype(np.float64)), np.exp(Yref.astype(np.float64))
    err = O.normalised_err(Y, Yref)
    assert err <= tol, f"normalised error {err:.3e} > {tol:.1e}"


@gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_stft_tool_matches_oracle(tools, O, tmp_path, dt):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(16000).astype(dt)
    import dsp_amd
    win = dsp_amd.hamming(401, dt)
    xp, wp, yp = tmp_path / "x.npy", tmp_path / "w.npy", tmp_path / "y.npy"
    np.save(xp, x); np.save(wp, win)
    stft = os.path.join(tools, "stft")
    for flags, kw in [([], {}), (["-e"], dict(snip_edges=1)), (["-s65", "-z"], dict(mn0=1)), (["-ea", "--step", "100"], dict(snip_edges=1, amp=1)),
                      (["-l", "-z", "-a"], dict(mn0=1, amp=1, lg=1))]:
        stp = 65 if "-s65" in flags else 100 if "100" in flags else 160
        r = run([stft] + flags + [str(xp), str(wp), "-o", str(yp)])
        assert r.returncode == 0, r.stderr
        Y = np.load(yp)
        rc, Yref = O.stft(x, win, stp, 512, kw.get("snip_edges", 0), kw.get("mn0", 0), kw.get("amp", 0), kw.get("lg", 0))
        assert rc == 0 and Y.flags["C_CONTIGUOUS"]
        check(O, Y, Yref, kw.get("lg", 0))
    # the signal through stdin, the spectrogram through stdout
    r = run([stft, "-e", "-", str(wp)], stdin=xp.read_bytes())
    assert r.returncode == 0, r.stderr
    (tmp_path / "so.npy").write_bytes(r.stdout)
    rc, Yref = O.stft(x, win, 160, 512, 1)
    check(O, np.load(tmp_path / "so.npy"), Yref)
    # the window through stdin
    r = run([stft, "-e", str(xp)], stdin=wp.read_bytes())
    assert r.returncode == 0, r.stderr
    (tmp_path / "so2.npy").write_bytes(r.stdout)
    check(O, np.load(tmp_path / "so2.npy"), Yref)


@gpu
def test_stft_tool_on_the_reference_configuration(tools, O, tmp_path):
    """The reference's own case through the tool, exactly: 10 s mono @ 16 kHz float32, 400-sample Hann window, hop 160
    (the tool's default), nfft 512 = nextpow2(400), power spectrogram: 1000 x 257."""
    rng = np.random.default_rng(16000)
    x = (rng.standard_normal(160000) * 0.1).astype(np.float32)
    import dsp_amd
    win = dsp_amd.hann(400, np.float32)
    xp, wp, yp = tmp_path / "x.npy", tmp_path / "w.npy", tmp_path / "y.npy"
    np.save(xp, x); np.save(wp, win)
    r = run([os.path.join(tools, "stft"), str(xp), str(wp), "-o", str(yp)])
    assert r.returncode == 0, r.stderr
    Y = np.load(yp)
    assert Y.shape == (1000, 257) and Y.dtype == np.float32
    rc, Yref = O.stft(x, win, 160, 512)
    assert rc == 0
    check(O, Y, Yref)
    assert np.array_equal(Y, dsp_amd.stft(x, win, 160))          # the tool is the library call, nothing else


@gpu
def test_stft_tool_major_order_and_armadillo(tools, O, tmp_path):
    rng = np.random.default_rng(6)
    x = rng.standard_normal(5000).astype(np.float32)
    import dsp_amd
    win = dsp_amd.povey(400, np.float32)
    rc, Yref = O.stft(x, win, 160, 512, 0)
    W, F = Yref.shape
    stft = os.path.join(tools, "stft")
    # column-major .npy input (a 1 x N Fortran-ordered row vector): output is F x W column-major, same bytes
    xp, wp, yp = tmp_path / "x.npy", tmp_path / "w.npy", tmp_path / "y.npy"
    np.save(xp, np.asfortranarray(x.reshape(1, -1))); np.save(wp, win)
    import numpy.lib.format as fmt
    with open(xp, "wb") as fh:
        fmt.write_array_header_1_0(fh, {"descr": "<f4", "fortran_order": True, "shape": (1, 5000)})
        fh.write(x.tobytes())
    assert run([stft, str(xp), str(wp), "-o", str(yp)]).returncode == 0
    Y = np.load(yp)
    assert Y.shape == (F, W) and Y.flags["F_CONTIGUOUS"]
    check(O, np.ascontiguousarray(Y.T), Yref)
    # Armadillo column vector in, Armadillo F x W matrix out
    xa, wa, ya = tmp_path / "x.bin", tmp_path / "w.bin", tmp_path / "y.bin"
    xa.write_bytes(arma_bytes(x.reshape(-1, 1))); wa.write_bytes(arma_bytes(win.reshape(-1, 1)))
    r = run([stft, str(xa), str(wa), "-o", str(ya)])
    assert r.returncode == 0, r.stderr
    raw = ya.read_bytes()
    head = f"ARMA_MAT_BIN_FN004\n{F} {W}\n".encode()
    assert raw.startswith(head)
    Ya = np.frombuffer(raw[len(head):], np.float32).reshape(W, F)
    check(O, Ya, Yref)
    # snip-edges with a window longer than the signal: the library's message, exit code 1
    np.save(xp, x[:300])
    r = run([stft, "-e", str(xp), str(wp), "-o", str(yp)])
    assert r.returncode == 1 and b"error in stft_s: L must be < N if snip_edges" in r.stderr and b"problem during function call" in r.stderr


@gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_stft_flt_tool_matches_oracle(tools, O, tmp_path, dt):
    rng = np.random.default_rng(7)
    x = rng.standard_normal(3000).astype(dt)
    import dsp_amd
    win = dsp_amd.povey(400, dt)
    xp, wp, yp = tmp_path / "x.npy", tmp_path / "w.npy", tmp_path / "y.npy"
    np.save(xp, x); np.save(wp, win)
    flt = os.path.join(tools, "stft_flt")
    for args, c0, stp, W, intended in [([], 0.0, 160.0, None, False), (["-c200"], 200.0, 160.0, None, False), (["-c", "-50", "-s133.37", "-w", "20"], -50.0, 133.37, 20, False),
                                       (["--c0=300", "--stp=160", "-i"], 300.0, 160.0, None, True), (["-c300"], 300.0, 160.0, None, False)]:
        if W is None:
            W = int((float(len(x) - 1) - c0) / stp)
        r = run([flt] + args + [str(xp), str(wp), "-o", str(yp)])
        assert r.returncode == 0, r.stderr
        rc, Yref = O.stft_flt(x, win, W, c0, stp, 512, 0, 0, 0, intended=intended)
        assert rc == 0
        check(O, np.load(yp), Yref)


@gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_tools_on_a_matrix_of_signals(tools, O, tmp_path, dt):
    """-d: every vector along one axis of a matrix is a signal.  All four layouts (axis x major order): interleaved channels
    go through the strided entry points, contiguous ones through the batched ones; the output is one W x F block per
    signal, B x W x F for a row-major file and F x W x B for a column-major one (the same bytes).  Each block against the
    oracle's result for that channel alone and, bit for bit, against the tool run on the channel as a vector."""
    rng = np.random.default_rng(11)
    C_, N = 5, 6000
    sig = (rng.standard_normal((C_, N)) * 0.1).astype(dt)            # [channel][sample]
    import dsp_amd
    win = dsp_amd.hamming(400, dt)
    wp = tmp_path / "w.npy"
    np.save(wp, win)
    stft, flt = os.path.join(tools, "stft"), os.path.join(tools, "stft_flt")
    layouts = {
        "rows_are_signals_rowmajor": (np.ascontiguousarray(sig), "1"),                # [C][N], contiguous signals
        "cols_are_signals_rowmajor": (np.ascontiguousarray(sig.T), "0"),              # [N][C], interleaved
        "cols_are_signals_colmajor": (np.asfortranarray(sig.T), "0"),                 # N x C column-major: contiguous
        "rows_are_signals_colmajor": (np.asfortranarray(sig), "1"),                   # C x N column-major: interleaved
    }
    ref = [O.stft(sig[c], win, 160, 512, 0, 1)[1] for c in range(C_)]
    W = int((N - 1 - 200) // 160.5)
    ref_flt = [O.stft_flt(sig[c], win, W, 200.0, 160.5)[1] for c in range(C_)]
    alone = []
    for c in range(C_):
        xp, yp = tmp_path / "x1.npy", tmp_path / "y1.npy"
        np.save(xp, sig[c])
        r = run([stft, "-z", str(xp), str(wp), "-o", str(yp)])
        assert r.returncode == 0, r.stderr
        alone.append(np.load(yp))
    for name, (arr, d) in layouts.items():
        xp, yp = tmp_path / (name + ".npy"), tmp_path / (name + "_y.npy")
        np.save(xp, arr)
        r = run([stft, "-z", "-d", d, "--device", "0", str(xp), str(wp), "-o", str(yp)])
        assert r.returncode == 0, (name, r.stderr)
        Y = np.load(yp)
        col = arr.flags["F_CONTIGUOUS"] and not arr.flags["C_CONTIGUOUS"]
        assert Y.shape == ((257, ref[0].shape[0], C_) if col else (C_, ref[0].shape[0], 257)), (name, Y.shape)
        Yb = Y.transpose(2, 1, 0) if col else Y                    # [b][w][f]
        for c in range(C_):
            check(O, np.ascontiguousarray(Yb[c]), ref[c])
            assert np.array_equal(Yb[c], alone[c]), (name, c)
        r = run([flt, "-d" + d, "-c200", "-s160.5", "-w", str(W), "--ngpu", "1", str(xp), str(wp), "-o", str(yp)])
        assert r.returncode == 0, (name, r.stderr)
        Y = np.load(yp)
        Yb = Y.transpose(2, 1, 0) if col else Y
        for c in range(C_):
            check(O, np.ascontiguousarray(Yb[c]), ref_flt[c])
    # a device the box does not have: the library's own error line, exit code 1
    r = run([stft, "--device", "63", str(tmp_path / "x1.npy"), str(wp), "-o", str(tmp_path / "z.npy")])
    assert r.returncode == 1 and b"error" in r.stderr
