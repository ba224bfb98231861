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
    bt = os.path.join(ROOT, "tools", "native", "libsegv_bt.so")
    if os.environ.get("TRACE_SEGV_BT") and os.path.exists(bt):      # diagnostic: native frames of a host crash
        import ctypes
        ctypes.CDLL(bt).segv_bt_install()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def synth_boxes(rng, n, k=2.0, size=224.0):
    """SURVEY.md 8(d) synthetic boxes."""
    x1 = rng.uniform(0, 160, n)
    y1 = rng.uniform(0, 160, n)
    w = rng.uniform(16, 64 * k, n)
    h = rng.uniform(16, 64 * k, n)
    return np.stack([x1, y1, np.minimum(x1 + w, size - 1), np.minimum(y1 + h, size - 1)], 1).astype(np.float32)


def distinct_scores(rng, n):
    return (rng.permutation(n).astype(np.float32) / max(n, 1) * 0.95 + 0.05).astype(np.float32)
