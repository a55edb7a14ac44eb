"""convert_freqs_s/_d (host code of the library, no device involved) against what the reference's own
c/convert_freqs.c returned for every pair of scales (fixtures made by oracle/make_golden.py), and its
error paths (c/convert_freqs.c:26-27,53-55,125-127)."""
import numpy as np
import pytest

from conftest import load_golden

SCALES = ("hz", "sqrt", "cbrt", "octave")


@pytest.mark.parametrize("tag", ["s", "d"])
def test_convert_freqs_matches_reference(tag):
    import dsp_amd as D
    g = load_golden(f"ref_convert_freqs_{tag}.npz")
    # libm of the build host vs the fixture's: identical functions, a last-place difference at most
    rtol = 3e-7 if tag == "s" else 5e-16
    for a in SCALES:
        xa = g[f"in.{a}"]
        for b in SCALES:
            y = D.convert_freqs(xa, a, b)
            assert y.dtype == xa.dtype
            if a == b:
                assert np.array_equal(y, xa)
            np.testing.assert_allclose(y, g[f"{a}.{b}"], rtol=rtol, atol=0)
    # scales are told apart by their leading characters, as in the reference
    x = g["hz"]
    assert np.array_equal(D.convert_freqs(x, "hzzz", "sqrt-root"), D.convert_freqs(x, "hz", "sqrt"))
    assert np.array_equal(D.convert_freqs(x, "sq", "hz"), D.convert_freqs(x, "sqrt", "hz"))
    assert D.convert_freqs(x[:0], "hz", "octave").size == 0


def test_convert_freqs_errors(capfd):
    import dsp_amd as D
    x = np.arange(1, 5, dtype=np.float32)
    for a, b in [("h", "hz"), ("hz", "s"), ("mel", "hz"), ("hz", "mel"), ("oct", "hz"), ("hz", "sq")]:
        with pytest.raises(D.DspError):
            D.convert_freqs(x, a, b)
    assert "error in convert_freqs_s" in capfd.readouterr().err
