"""The module-level helpers the reference exports next to its classes (dpm_solver_pytorch.py:1253-1305) against golden
vectors generated from the unmodified reference (tests/golden/make_golden.py, group `utils`)."""
import numpy as np
import torch

import dpm_solver_amd as D
from dpm_solver_amd.utils import expand_dims, interpolate_fn


def test_interpolate_fn_matches_reference_goldens(golden):
    for tag in "abc":
        g = lambda k: torch.from_numpy(golden.get("utils", "utils/%s/%s" % (tag, k)))
        y = interpolate_fn(g("x"), g("xp"), g("yp"))
        want = g("y")
        assert y.shape == want.shape and y.dtype == want.dtype
        # same segment, same formula: the only freedom is the association of one product (<= 2 ulp)
        np.testing.assert_allclose(y.numpy(), want.numpy(), rtol=3e-7, atol=3e-7)
        assert float((y - want).abs().max()) <= 4 * float(np.finfo(np.float32).eps) * float(want.abs().max())


def test_interpolate_fn_on_a_schedule_table_is_bit_equal_to_the_planner(golden):
    """the use the reference makes of it: marginal_log_mean_coeff of a discrete schedule (ref :129-131) -- the helper,
    the C planner's binary search and the reference's golden agree"""
    la = torch.from_numpy(golden.get("schedules", "sched/sd/log_alpha_array"))
    ta = torch.from_numpy(golden.get("schedules", "sched/sd/t_array"))
    t = torch.from_numpy(golden.get("schedules", "sched/sd/t"))
    want = golden.get("schedules", "sched/sd/log_mean_coeff")
    got = interpolate_fn(t.reshape(-1, 1), ta, la).reshape(-1).numpy()
    np.testing.assert_allclose(got, want.reshape(-1), rtol=2e-6, atol=1e-7)


def test_expand_dims_matches_reference_goldens(golden):
    v = torch.from_numpy(golden.get("utils", "utils/expand/v"))
    for dims in (1, 2, 4):
        out = expand_dims(v, dims)
        assert tuple(out.shape) == tuple(golden.get("utils", "utils/expand/shape%d" % dims))
        np.testing.assert_array_equal(out.numpy(), golden.get("utils", "utils/expand/val%d" % dims))
        assert out.data_ptr() == v.data_ptr()          # a view, as in the reference


def test_helpers_are_exported_like_the_reference():
    import dpm_solver_pytorch as M          # the root-level import path of README.md:380
    assert M.interpolate_fn is interpolate_fn and M.expand_dims is expand_dims
    assert D.interpolate_fn is interpolate_fn
