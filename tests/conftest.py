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
    config.addinivalue_line("markers", "lab: needs the LAB build of the library (tuning knobs, fault injection, event-bracketed "
                                       "launches: include/dpm_lab.h); skipped on the product library and run by "
                                       "test_lab_suite_* in a subprocess with DPM_SOLVER_AMD_LIB=tools/_variants/lab/libdpm_lab.so")


def pytest_collection_modifyitems(config, items):
    from dpm_solver_amd import _lib as L
    if L.IS_LAB:
        return
    skip = pytest.mark.skip(reason="needs the lab build (run by test_lab_suite_* on tools/_variants/lab/libdpm_lab.so)")
    for item in items:
        if "lab" in item.keywords:
            item.add_marker(skip)


def run_lab_suite(marker_expr, timeout=1500):
    """the lab-marked tests in a subprocess on the lab library; returns the number that passed"""
    import re
    import subprocess
    from dpm_solver_amd import _lib as L
    if not os.path.exists(L.LAB_LIB_PATH):
        pytest.skip("no lab build next to the library (__graft_entry__.build() makes it)")
    assert os.path.getmtime(L.LAB_LIB_PATH) >= os.path.getmtime(os.path.join(ROOT, "dpm_solver_amd", "libdpm_hip.so")) - 3600, \
        "the lab build is older than the library"
    env = dict(os.environ, DPM_SOLVER_AMD_LIB=L.LAB_LIB_PATH)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-q", "-x", "-m", marker_expr, "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout.splitlines()[-1])
    assert m and "failed" not in r.stdout.splitlines()[-1], tail
    return int(m.group(1))


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
