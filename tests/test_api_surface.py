"""The drop-in boundary of SURVEY 8(b): the engine keeps every public name and signature of the reference
(dpm_solver_pytorch.py: NoiseScheduleVP, model_wrapper, DPM_Solver, interpolate_fn, expand_dims).

Two checks: against `tests/golden/api_signatures.json` (a snapshot `tests/golden/make_golden.py api` takes from the
unmodified reference; runs everywhere) and, where the reference checkout exists, against the live module -- so neither
side can drift unnoticed.  Plus `NoiseScheduleVP.numerical_clip_alpha` (ref :114-125) against goldens on the raw tables of
the SD / DDPM-linear / cosine-4000 / cosine-1000 schedules.
"""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

import dpm_solver_amd as D
from make_golden_api import api_snapshot

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.environ.get("DPM_REFERENCE_DIR", "/root/reference")

# the additive differences the engine is allowed (documented in DESIGN.md section 1): trailing parameters with
# defaults that the reference does not have
ALLOWED_EXTRA = {
    "DPM_Solver.__init__": [["state_dtype", "POSITIONAL_OR_KEYWORD", "None"]],
    # the reference's dynamic_thresholding_fn(x0, t) never reads t; the engine makes it optional
    "DPM_Solver.dynamic_thresholding_fn": [],
}
RELAXED_DEFAULT = {("DPM_Solver.dynamic_thresholding_fn", "t"): "None"}


def _compare(want, got):
    missing = sorted(set(want) - set(got))
    assert not missing, "public names of the reference missing from the engine: %s" % missing
    for name, wsig in want.items():
        gsig = [list(p) for p in got[name]]
        extra = ALLOWED_EXTRA.get(name, [])
        if extra:
            assert gsig[len(wsig):] == extra, "%s: unexpected extra parameters %s" % (name, gsig[len(wsig):])
            gsig = gsig[:len(wsig)]
        assert len(gsig) == len(wsig), "%s: %s vs the reference's %s" % (name, gsig, wsig)
        for w, g in zip(wsig, gsig):
            if (name, w[0]) in RELAXED_DEFAULT:
                assert g[:2] == w[:2] and g[2] == RELAXED_DEFAULT[(name, w[0])]
                continue
            assert g == w, "%s: parameter %s vs the reference's %s" % (name, g, w)


def test_public_signatures_match_the_reference_snapshot():
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "api_signatures.json")))
    assert len(want) >= 29
    _compare(want, api_snapshot(D))


def test_root_level_shim_exports_the_same_objects():
    import dpm_solver_pytorch as M          # README.md:380 import path
    for name in ("NoiseScheduleVP", "model_wrapper", "DPM_Solver", "interpolate_fn", "expand_dims"):
        assert getattr(M, name) is getattr(D, name)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "dpm_solver_pytorch.py")),
                    reason="the reference checkout is only present in the build container")
def test_public_signatures_match_the_live_reference():
    spec = importlib.util.spec_from_file_location("_dpm_reference_api", os.path.join(REF_DIR, "dpm_solver_pytorch.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    live = api_snapshot(R)
    # the committed snapshot is the live reference's
    assert live == json.load(open(os.path.join(ROOT, "tests", "golden", "api_signatures.json")))
    _compare(live, api_snapshot(D))


@pytest.mark.parametrize("name", ["sd", "ddpm", "cosine4000", "cosine1000"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_numerical_clip_alpha_matches_reference_goldens(golden, name, prec):
    ns = D.NoiseScheduleVP("linear")        # the method does not depend on the instance (ref :114)
    la = torch.from_numpy(golden.get("clip", "clip/%s/%s/log_alphas" % (name, prec)))
    for cl in (-5.1, -3.0, 0.0, -20.0):
        want = int(golden.get("clip", "clip/%s/%s/len/%g" % (name, prec, cl)))
        got = ns.numerical_clip_alpha(la, cl) if cl != -5.1 else ns.numerical_clip_alpha(la)
        assert got.shape[0] == want, (name, prec, cl, got.shape[0], want)
        assert got.dtype == la.dtype and torch.equal(got, la[:want])
        if want:
            assert got.data_ptr() == la.data_ptr()       # a slice of the argument, like the reference's


def test_numerical_clip_alpha_pins_total_N(golden):
    """the constructor's clip and the public method agree: total_N = 1000 / 1000 / 3984 / 996"""
    from engine_cases import make_schedule
    for name, total in (("sd", 1000), ("ddpm", 1000), ("cosine4000", 3984), ("cosine1000", 996)):
        ns = make_schedule(name)
        assert ns.total_N == total == int(golden.get("schedules", "sched/%s/total_N" % name))
        la = torch.from_numpy(golden.get("clip", "clip/%s/f32/log_alphas" % name))
        assert ns.numerical_clip_alpha(la).shape[0] == total
        # torch's vectorised log rounds a few entries 1 ulp apart from the planner's
        np.testing.assert_allclose(ns.numerical_clip_alpha(la).numpy(), ns.log_alpha_array.reshape(-1).numpy(), rtol=3e-7)
