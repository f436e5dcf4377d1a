import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


class Golden:
    """Lazy reader for tests/golden/<group>.npz (keys use '|' in place of '/')."""

    def __init__(self):
        self._z = {}

    def group(self, name):
        if name not in self._z:
            self._z[name] = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
        return self._z[name]

    def get(self, group, key):
        return self.group(group)[key.replace("/", "|")]

    def keys(self, group):
        return [k.replace("|", "/") for k in self.group(group).files]


@pytest.fixture(scope="session")
def golden():
    return Golden()


def rel_err(a, b):
    """max |a-b| / max|b|  (the 'rel-err' of BASELINE.json: relative to the tensor's scale)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) / den
