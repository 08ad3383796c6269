import os
import sys


def _cpu_budget():
    """CPUs the container's CFS quota really grants (bench.py, cpu_budget(): the GPU boxes report 256 CPUs under a quota of
    16; 256 numpy / torch threads get the whole cgroup throttled)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max" and int(p) > 0:
            n = min(n, max(1, int(q) // int(p)))
    except (OSError, ValueError):
        pass
    return max(1, n)


for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):   # before numpy / torch load their thread pools
    os.environ.setdefault(_k, str(max(1, _cpu_budget() - max(1, _cpu_budget() // 4))))
# the library honours its test hooks (CB_BANK_UNFUSED, CB_BANK_KG, CB_EIGH_HOST, CB_NO_HYBRID, CB_NO_SYM, CB_FAULT_INJECT,
# CB_EIGH_SHORT_PLAN, CB_BANK_TEST_NO_CLAIM) only together with this gate (csrc/cb_internal.hip.h, cb_test_hook)
os.environ.setdefault("CB_TEST_HOOKS", "1")
os.environ.setdefault("KMP_BLOCKTIME", "0")
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np   # noqa: E402
import pytest   # noqa: E402

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
