import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.orc()
    return oracle


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


# per-frame-normalised tolerances of the path (float32 / float64)
TOL = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-12}
