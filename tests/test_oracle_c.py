"""The C restatement of the hot path's arithmetic (oracle/dpm_oracle_kernels.c, test infrastructure) against the numpy oracle
-- bit for bit: IEEE + - * / in fp32, no contraction -- and, through whole trajectories, against the golden fixtures the
unmodified reference produced (tests/golden/e2e.npz).  Runs without a GPU."""
import numpy as np
import pytest

import cases as C
from conftest import rel_err
from oracle import dpm_oracle as O
from oracle import dpm_oracle_c as OC
import test_oracle_golden as TO

F32 = np.float32


@pytest.fixture(scope="module", autouse=True)
def _built():
    OC.build()
    assert OC.lib().dpmo_version() >= 1 and OC.lib().dpmo_max_threads() >= 1


def _sched(name="sd"):
    si = C.schedule_inputs(name)
    if si["kind"] == "linear":
        return O.Schedule.linear(si["beta_0"], si["beta_1"])
    return O.Schedule.from_betas(si["betas"]) if "betas" in si else O.Schedule.from_alphas_cumprod(si["alphas_cumprod"])


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(F32)


def test_model_conversions_and_cfg_blend_bitwise():
    sch = _sched("ddpm")
    x, out = _rand((3, 3, 8, 8), 0), _rand((3, 3, 8, 8), 1)
    t = F32(0.37)
    tv = np.full((3,), t, dtype=F32)
    sc = O.Solver._sc
    a, s = sc(sch.alpha(t)), sc(sch.std(t))
    for kind in ("x_start", "v", "score"):
        want = O.wrap_model(lambda xx, ti: out, sch, model_type=kind)(x, tv)
        assert np.array_equal(OC.K.to_noise(kind, x, out, a, s), want), kind
    nu, nc = _rand((3, 3, 8, 8), 2), _rand((3, 3, 8, 8), 3)
    want = O.wrap_model(lambda xx, ti, c: np.concatenate([nu, nc]), sch, guidance_type="classifier-free", condition=np.ones(3, F32),
                        unconditional_condition=np.zeros(3, F32), guidance_scale=7.5)(x, tv)
    assert np.array_equal(OC.K.cfg_blend(nu, nc, 7.5), want)
    want = O.Solver(O.wrap_model(lambda xx, ti: out, sch), sch).data_pred(x, t)
    assert np.array_equal(OC.K.eps_to_x0(x, out, a, s), want)


@pytest.mark.parametrize("shape", [(4, 3, 64, 64), (2, 3, 7, 9), (1, 1, 1, 2)])
def test_dynamic_thresholding_bitwise(shape):
    x0 = _rand(shape, 5) * F32(2.5)
    for ratio, mv in ((0.995, 1.0), (0.9, 0.5), (0.5, 3.0)):
        got, s = OC.K.dynamic_threshold(x0, ratio, mv)
        assert np.array_equal(got, O.dynamic_threshold(x0, ratio, mv)), (ratio, mv)
        assert np.all(s >= F32(mv))


@pytest.mark.parametrize("algo", ["dpmsolver++", "dpmsolver"])
@pytest.mark.parametrize("sname", ["sd", "ddpm", "vp_linear"])
def test_update_formulas_bitwise(algo, sname):
    sch = _sched(sname)
    pp = algo == "dpmsolver++"
    sol = O.Solver(None, sch, algorithm_type=algo)
    x, m0, m1, m2 = (_rand((2, 4, 8, 8), k) for k in range(10, 14))
    ts = [F32(0.9), F32(0.8), F32(0.72), F32(0.6)]
    want, _ = sol.first_update(x, ts[2], ts[3], model_s=m0)
    assert np.array_equal(OC.K.update_first(x, m0, *OC.coef_first(sch, pp, ts[2], ts[3])), want)
    for st in ("dpmsolver", "taylor"):
        want = sol.ms2_update(x, [m1, m0], [ts[1], ts[2]], ts[3], st)
        assert np.array_equal(OC.K.update_ms2(x, m0, m1, *OC.coef_ms2(sch, pp, ts[1], ts[2], ts[3], st)), want), st
    want = sol.ms3_update(x, [m2, m1, m0], ts[:3], ts[3])
    assert np.array_equal(OC.K.update_ms3(x, m0, m1, m2, *OC.coef_ms3(sch, pp, ts[0], ts[1], ts[2], ts[3])), want)


@pytest.mark.parametrize("algo", ["dpmsolver++", "dpmsolver"])
@pytest.mark.parametrize("steps,order", [(20, 2), (20, 3), (10, 1), (6, 3), (5, 2), (25, 2)])
def test_trajectories_equal_the_numpy_oracle_bitwise(algo, steps, order):
    sch = _sched("sd")
    x = _rand((3, 4, 8, 8), steps * 10 + order)
    net = O.wrap_model(lambda xx, ti: np.asarray(C.model_tdep(xx, ti), dtype=F32), sch)
    want = O.Solver(net, sch, algorithm_type=algo).sample(x, steps=steps, order=order)
    got = OC.Stepper(sch, algo).sample(net, x, steps=steps, order=order)
    assert np.array_equal(got, want)
    if algo == "dpmsolver++":
        wt = O.Solver(net, sch, correcting_x0_fn="dynamic_thresholding").sample(x, steps=steps, order=order)
        gt = OC.Stepper(sch, algo, thresholding=True).sample(net, x, steps=steps, order=order)
        assert np.array_equal(gt, wt)


def test_fused_2m_stage_equals_the_unfused_trajectory_and_is_thread_invariant():
    sch = _sched("sd")
    x, eps = _rand((8, 4, 16, 16), 1), _rand((8, 4, 16, 16), 2)
    want = O.Solver(O.wrap_model(lambda xx, ti: eps, sch), sch).sample(x, steps=20, order=2)
    for th in (1, 3, 8):
        assert np.array_equal(OC.Stepper(sch).sample_2m_fused(eps, x, 20, threads=th), want), th


def test_c_oracle_against_the_reference_goldens(golden):
    """the multistep, time_uniform, unguided noise-network cases of tests/golden/e2e.npz (outputs of the unmodified
    reference): the C restatement is pinned to the reference itself, not only to the numpy oracle"""
    n = 0
    for case in C.E2E_CASES:
        if (case["method"] != "multistep" or case["model_type"] != "noise" or case["guidance_type"] != "uncond"
                or case["skip_type"] != "time_uniform" or case["denoise_to_zero"] or case["t_start"] is not None
                or case["t_end"] is not None or case["call"] != "sample"):
            continue
        sch = TO.make_schedule(case["schedule"])
        base = C.MODELS[case["model"]]
        net = O.wrap_model(lambda xx, ti: np.asarray(base(xx, ti), dtype=F32), sch)
        got = OC.Stepper(sch, case["algorithm_type"], thresholding=case["thresholding"]).sample(
            net, C.x_T_for(case).astype(F32), steps=case["steps"], order=case["order"],
            lower_order_final=case["lower_order_final"], solver_type=case["solver_type"])
        ref = golden.get("e2e", "e2e/%s/final" % case["name"])
        assert rel_err(got, ref) < TO.E2E_TOL, (case["name"], rel_err(got, ref))
        n += 1
    assert n >= 5, n
