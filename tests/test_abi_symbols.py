"""The C-ABI library loads without a GPU and exports every symbol include/dsp_stft.h declares; argument
checks that the reference makes before any work return 1 with its message (no device needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "dsp_stft.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b([a-z_0-9]+)\s*\(", src)) - {"defined", "if"})


def test_header_and_loader_agree():
    from dsp_amd import _lib
    names = header_functions()
    assert len(names) >= 38
    assert sorted(_lib.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    from dsp_amd import _lib
    handle = C.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert getattr(handle, name) is not None
    assert b"gfx950" in _lib.lib().dsp_stft_version()


def test_shard_range_in_the_library_is_the_python_partition():
    """dsp_shard_range (what the single-process multi-GPU entry points split by) against dsp_amd.dist.shard_range
    (what the one-process-per-GPU path splits by): the same contiguous blocks, covering the batch exactly."""
    import dsp_amd
    from dsp_amd.dist import shard_range
    for B in (0, 1, 5, 8, 1023, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [dsp_amd.shard_range(B, r, world) for r in range(world)]
            assert spans == [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
    assert dsp_amd.shard_range(65536, 3, 8) == (3 * 8192, 4 * 8192)


def test_num_frames_host_logic():
    import dsp_amd
    assert dsp_amd.num_frames(160000, 400, 160, False) == 1000
    assert dsp_amd.num_frames(160000, 400, 160, True) == 998
    assert dsp_amd.num_frames(172800000, 4096, 1024, False) == 168750
    assert dsp_amd.num_frames(100, 400, 160, True) == 0
    assert dsp_amd.kernel_family(512, np.float32) == 3
    assert dsp_amd.kernel_family(4096, np.float64) == 4 and dsp_amd.kernel_family(8192, np.float64) == 2 and dsp_amd.kernel_family(8192, np.float32) == 8
    assert dsp_amd.kernel_family(256, np.float32) == 7 and dsp_amd.kernel_family(256, np.float64) == 7 and dsp_amd.kernel_family(1024, np.float64) == 6
    assert dsp_amd.kernel_family(1024, np.float32) == 6 and dsp_amd.kernel_family(2048, np.float64) == 6 and dsp_amd.kernel_family(32, np.float64) == 5
    assert dsp_amd.kernel_family(16384, np.float32) == 2
    assert dsp_amd.kernel_family(300, np.float32) == 1
    assert dsp_amd.kernel_family(1, np.float32) == 0


@pytest.mark.parametrize("call,msg", [
    (lambda d: d.stft(np.zeros(100, np.float32), np.zeros(0, np.float32)), "error in stft_s: L must be positive"),
    (lambda d: d.stft(np.zeros(100), np.zeros(10), stp=0), "error in stft_d: stp must be positive"),
    (lambda d: d.stft(np.zeros(100), np.zeros(400), snip_edges=True), "error in stft_d: L must be < N if snip_edges"),
    (lambda d: d.stft(np.zeros(1000), np.zeros(400), nfft=256), "error in stft_d: nfft must be >= L"),
    (lambda d: d.stft_flt(np.zeros(100, np.float32), np.zeros(10, np.float32), 200), "error in stft_flt_s: W must be <= N (length X)"),
    (lambda d: d.stft_flt(np.zeros(100), np.zeros(200), 10), "error in stft_flt_d: L must be <= N (length X)"),
    (lambda d: d.stft_flt(np.zeros(100), np.zeros(10), 10, c0=100.0), "error in stft_flt_d: c0 (center samp of 1st frame) must be < N (length X)"),
    (lambda d: d.stft_flt(np.zeros(100), np.zeros(10), 10, stp=0.0), "error in stft_flt_d: stp (step size) must be positive"),
    (lambda d: d.frame_univar(np.zeros(100), 0, 10), "error in frame_univar_d: L must be positive"),
    (lambda d: d.frame_univar_flt(np.zeros(100, np.complex64), 10, 200, 0.0, 1.0), "error in frame_univar_flt_c: W must be <= N (length X)"),
    (lambda d: d.apply_win(np.zeros((4, 5)), np.zeros(4), 1), "error in apply_win_d: L (winlength) must equal C (ncols X) for dim=1"),
    (lambda d: d.apply_win(np.zeros((4, 5), np.float32), np.zeros(4, np.float32), 4), "error in apply_win_s: dim must be in [0 3]"),
    (lambda d: d.fft_r2hc(np.zeros((4, 50)), 1, 32), "error in fft_fftw_r2hc_d: nfft must be >= L (vec length)"),
])
def test_argument_checks_precede_device_work(capfd, call, msg):
    import dsp_amd
    with pytest.raises(dsp_amd.DspError):
        call(dsp_amd)
    assert msg in capfd.readouterr().err


def test_zero_frames_is_success_without_device():
    import dsp_amd
    Y = dsp_amd.stft(np.zeros(50, np.float32), np.ones(400, np.float32))
    assert Y.shape == (0, 257)
    assert dsp_amd.stft_flt(np.zeros(50), np.ones(10), 0).shape == (0, 9)
