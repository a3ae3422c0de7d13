import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cyclical-visual-captioning_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


class Golden:
    """Read-only view of one tests/golden/*.npz fixture (made by tools/make_golden.py from the
    reference itself).  `sub(prefix)` strips a key prefix; missing-grad markers become None."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))
        self.keys = list(self.z.keys())

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z

    def sub(self, prefix):
        out = {}
        for k in self.keys:
            if k.startswith(prefix):
                kk = k[len(prefix):]
                if kk.endswith(".is_none"):
                    out[kk[:-len(".is_none")]] = None
                else:
                    out[kk] = self.z[k]
        return out


@pytest.fixture(scope="session")
def g1():
    return Golden("g1_tiny.npz")


@pytest.fixture(scope="session")
def g2():
    return Golden("g2_cfg1.npz")


@pytest.fixture(scope="session")
def g3():
    return Golden("g3_shards.npz")
