"""The reference-side binding: dropin/c holds files with the reference's file names that declare the library's entry
points inside namespace codee (the reference's tools include "stft.c" etc. by name, src/stft.cpp:16), and
dropin/check_main.cpp is a caller written the way those tools are.  Without a GPU: it compiles and links against the
library alone (no FFTW), and every reference entry point of the path is reachable through the forwarding files.  On
the GPU: its results against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import TOL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "dropin_check")


def lcg(n, seed):
    """The numbers dropin/check_main.cpp fills its signal with."""
    out = np.empty(n, np.float64)
    s = seed
    for i in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[i] = ((s >> 11) / 9007199254740992.0 - 0.5) * 0.2
    return out


def test_forwarding_files_compile_and_link_like_the_reference_tools(tmp_path):
    assert os.path.exists(TOOL), "run __graft_entry__.build() (make -C dropin) first"
    # a translation unit that includes every forwarding file, as C++ (namespace codee) and as C, and takes the address
    # of one function per file
    names = sorted(f for f in os.listdir(os.path.join(ROOT, "dropin", "c")) if f.endswith(".c"))
    assert {"stft.c", "stft_flt.c", "fft_power.c", "frame_univar.c", "apply_win.c", "fft.fftw.r2hc.c"} <= set(names)
    inc = "".join(f'#include "{n}"\n' for n in names)
    cpp = tmp_path / "all.cpp"
    cpp.write_text(inc + "int main() { return (codee::stft_s && codee::stft_flt_d && codee::fft_power_z && codee::frame_univar_flt_c && "
                         "codee::apply_win_inplace_s && codee::fft_fftw_r2hc_d && codee::window_univar_s && codee::pow_compress_s && codee::planck_d) ? 0 : 1; }\n")
    c = tmp_path / "all.c"
    c.write_text(inc + "int main(void) { return (stft_s && stft_flt_d && fft_power_z) ? 0 : 1; }\n")
    flags = ["-I" + os.path.join(ROOT, "dropin", "c"), "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "dsp_amd"),
             "-ldsp_stft_hip", "-Wl,-rpath," + os.path.join(ROOT, "dsp_amd"), "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-Wno-address", "-o", str(tmp_path / "a_cpp"), str(cpp)] + flags, check=True)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-Wno-address", "-o", str(tmp_path / "a_c"), str(c)] + flags, check=True)
    # the C++ build's calls bind to the unmangled names, as the reference's do (c/stft.c:24-27)
    syms = subprocess.run(["nm", "-u", TOOL], check=True, capture_output=True, text=True).stdout.split()
    assert {"stft_s", "stft_flt_d", "fft_power_c", "hann_s", "povey_d"} <= set(syms)


@pytest.mark.gpu
def test_reference_style_caller_against_oracle(O, tmp_path):
    out = tmp_path / "out.bin"
    r = subprocess.run([TOOL, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "error in stft_s: L must be positive" in r.stderr        # the deliberate bad call at the end
    N, L, nfft, stp, F = 16000, 400, 512, 160, 257
    xd = lcg(N, 12345)
    xs = xd.astype(np.float32)
    raw = np.fromfile(out, np.uint8)
    W, Wf = (N + stp // 2) // stp, 90
    p = 0
    ws = raw[p:p + 4 * L].view(np.float32); p += 4 * L
    Ys = raw[p:p + 4 * W * F].view(np.float32).reshape(W, F); p += 4 * W * F
    wd = raw[p:p + 8 * L].view(np.float64); p += 8 * L
    Yd = raw[p:p + 8 * Wf * F].view(np.float64).reshape(Wf, F); p += 8 * Wf * F
    P = raw[p:p + 4 * 500].view(np.float32); p += 4 * 500
    assert p == raw.size
    rc, ref = O.stft(xs, ws.copy(), stp, nfft, 0, 1)
    assert rc == 0 and O.normalised_err(Ys, ref) <= TOL[np.dtype(np.float32)]
    rc, ref = O.stft_flt(xd, wd.copy(), Wf, 200.0, 160.5, nfft, False, True, False)
    assert rc == 0 and O.normalised_err(Yd, ref) <= TOL[np.dtype(np.float64)]
    z = xs[:1000].astype(np.float64)
    assert np.allclose(P, z[0::2] ** 2 + z[1::2] ** 2, rtol=1e-6, atol=0)
