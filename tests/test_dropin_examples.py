"""The reference's own example call sites (source files unchanged) on the engine's host side -- CPU edition of
tools/dropin_examples.py: tests/kernel_double.py stands behind the launch records, everything above it (shims, planner,
wrapper batching, buffer choreography, callbacks, dtype policy) is the product.  The GPU edition is the tool itself
(profiles/r05_dropin.json) and tests/test_gpu_extensions.py::test_dropin_examples_on_the_gpu when the tree travelled.

Needs the reference checkout ($DPM_REFERENCE_DIR or /root/reference); skipped where it does not exist (the GPU box)."""
import importlib.util
import os
import sys

import pytest

import dpm_solver_amd as D
import dpm_solver_amd.solver as S
import kernel_double as KD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("dropin_examples", os.path.join(ROOT, "tools", "dropin_examples.py"))
DE = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(DE)

needs_reference = pytest.mark.skipif(DE.reference_examples() is None, reason="no reference checkout with examples/")


@needs_reference
@pytest.mark.parametrize("site", range(len(DE.SITES)), ids=[s[0].split()[0] for s in DE.SITES])
def test_reference_call_site_runs_unchanged_on_the_engine(site, monkeypatch):
    KD.install_cpu_double(monkeypatch, S, D)
    name, where, fn = DE.SITES[site]
    before = dict(sys.modules)
    row = DE.run_site(name, where, fn, "cpu", DE.reference_examples())
    assert row["network_trace_equal"], row["network_trace"]
    assert row["integers_equal"], {k: v for k, v in row["results"].items() if "equal" in v}
    assert row["max_rel_err"] <= DE.TOL, {k: v["rel_err"] for k, v in row["results"].items() if "rel_err" in v}
    assert row["shim"]["solver_file"].startswith("dpm_solver_amd")
    # the application modules the tool imported are gone again (they shadow names like `models`, `datasets`, `utils`)
    leaked = [k for k in sys.modules if k not in before and k.split(".")[0] in DE._APP_ROOTS]
    assert not leaked, leaked
