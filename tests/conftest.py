import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def load_golden(name):
    """Load a fixture written by tests/golden/make_golden.py, expanding the
    sparse / bit-packed arrays of the 400-state case."""
    z = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
    if "C_idx" in z:
        C = np.zeros(tuple(z["C_shape"]))
        C[tuple(z["C_idx"])] = z["C_val"]
        z["C"] = C
    if "mask_shape" in z:
        shape = tuple(z["mask_shape"])
        z["mask"] = np.unpackbits(z["mask"])[: shape[0] * shape[1]].reshape(shape).astype(np.float64)
    return z


@pytest.fixture(scope="session")
def golden():
    return load_golden


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
