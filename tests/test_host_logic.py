"""Host side of the engine on CPU: C planner + Python mirror classes, with the HIP kernel replaced by the
numpy test double (tests/kernel_double.py).  Compared against goldens generated from the real reference.

What this pins without a GPU: every coefficient the planner produces (through their effect on full
trajectories), the stage list / buffer-role choreography of sample(), wrapper batching for CFG and
classifier guidance, time labels handed to the network, callbacks, dtype promotion, error conventions.
"""
import ctypes as C_

import numpy as np
import pytest
import torch

import cases as C
import dpm_solver_amd as D
import dpm_solver_amd.solver as S
import dpm_solver_amd.wrapper as W
from conftest import rel_err
from dpm_solver_amd import _lib as L
from engine_cases import build_solver, make_schedule, run_case, sample_kwargs, tt
from kernel_double import (add_noise_double, adaptive_error_double, install_cpu_double, launch_stage_double,
                           maskblend_apply_double)
from oracle import dpm_oracle as O

F32 = np.float32
TOL = 1e-5


@pytest.fixture(autouse=True)
def cpu_double(monkeypatch):
    install_cpu_double(monkeypatch, S, D)


def test_linspace_and_time_grids_bitwise_vs_torch_and_golden(golden):
    ns = make_schedule("sd")
    dpm = D.DPM_Solver(lambda x, t: x, ns)
    for (a, b, n) in [(1.0, 1e-3, 20), (1.0, 1e-3, 15), (1.0, 1e-4, 10), (0.8, 0.0123, 6), (1.0, 0.001, 1000)]:
        got = dpm.get_time_steps("time_uniform", a, b, n, "cpu")
        assert torch.equal(got, torch.linspace(a, b, n + 1))
    for key in golden.keys("timesteps"):
        parts = key.split("/")
        if parts[0] == "tsteps":
            _, name, skip, spec = parts
            tT, t0, N = spec.split("_")
            dpm = D.DPM_Solver(lambda x, t: x, make_schedule(name))
            got = dpm.get_time_steps(skip, float(tT), float(t0), int(N), "cpu").numpy()
            np.testing.assert_allclose(got, golden.get("timesteps", key), rtol=2e-5, atol=2e-7)
        elif parts[0] == "ssgrid" and parts[-1] == "t":
            _, name, skip, spec, _ = parts
            order, steps = map(int, spec.split("_"))
            dpm = D.DPM_Solver(lambda x, t: x, make_schedule(name))
            outer, orders = dpm.get_orders_and_timesteps_for_singlestep_solver(steps, order, skip, 1.0, 1e-3, "cpu")
            assert orders == list(golden.get("timesteps", key[:-2] + "/orders"))
            np.testing.assert_allclose(outer.numpy(), golden.get("timesteps", key), rtol=2e-5, atol=2e-7)


@pytest.mark.parametrize("name", C.SCHEDULE_NAMES)
def test_schedule_matches_oracle_bitwise_and_golden(golden, name):
    """C planner (fp32, reference op order) == numpy oracle bit for bit; both within ulps of the reference."""
    ns = make_schedule(name)
    g = lambda k: golden.get("schedules", "sched/%s/%s" % (name, k))
    import test_oracle_golden as T
    osch = T.make_schedule(name)
    assert ns.total_N == int(g("total_N")) == osch.total_N
    if ns.schedule == "discrete":
        np.testing.assert_array_equal(ns.log_alpha_array.numpy()[0], osch.log_alpha)
        np.testing.assert_array_equal(ns.t_array.numpy(), g("t_array"))
        assert ns.log_alpha_array.shape == (1, ns.total_N) and ns.t_array.shape == (1, ns.total_N)
    t = torch.from_numpy(g("t"))
    ok = np.isfinite(g("lambda"))
    for meth, ofn, key in [("marginal_log_mean_coeff", osch.log_alpha_t, "log_mean_coeff"),
                           ("marginal_alpha", osch.alpha, "alpha"), ("marginal_std", osch.std, "std"),
                           ("marginal_lambda", osch.lam, "lambda")]:
        got = getattr(ns, meth)(t).numpy()
        np.testing.assert_array_equal(got[ok], ofn(g("t"))[ok])
        np.testing.assert_allclose(got, g(key), rtol=2e-5, atol=2e-6)
    lq = g("lambda_q")
    fin = np.isfinite(lq)
    np.testing.assert_array_equal(ns.inverse_lambda(torch.from_numpy(lq)).numpy()[fin], osch.inv_lam(lq)[fin])
    # shape conventions of the reference: flattened for discrete, preserved for linear
    q = torch.tensor(0.4321)
    assert ns.marginal_lambda(q).shape == (torch.Size([1]) if ns.schedule == "discrete" else torch.Size([]))


@pytest.mark.parametrize("name", [c["name"] for c in C.E2E_CASES])
def test_e2e_host_logic_against_reference_goldens(golden, name):
    case = C.E2E_BY_NAME[name]
    trace = []
    xf, inter = run_case(case, "cpu", trace)
    g = lambda k: golden.get("e2e", "e2e/%s/%s" % (name, k))
    assert xf.dtype == torch.float32
    assert len(inter) == int(g("n_intermediates"))
    # time labels and batch sizes the network saw (CFG doubles the batch)
    np.testing.assert_array_equal(np.array([b for b, _ in trace]), g("trace_b"))
    np.testing.assert_allclose(np.array([float(t.reshape(-1)[0]) for _, t in trace], dtype=F32), g("trace_t"),
                               rtol=1e-5, atol=2e-3)
    for b, t in trace:
        assert t.shape == (b,) and t.dtype == torch.float32
    e = rel_err(xf.numpy(), g("final"))
    assert e < TOL, e
    if case["intermediates"]:
        ri = g("intermediates")
        for i, v in enumerate(inter):
            assert rel_err(v.float().numpy(), ri[i]) < TOL, i


def test_callbacks(golden):
    case = C.E2E_BY_NAME["cfg1_small"]
    ns = make_schedule("sd")
    x = tt(C.x_T_for(case), "cpu")
    mask = torch.from_numpy(golden.get("callbacks", "cb/mask"))
    seen = []

    def cxt(xt, t, step):
        seen.append((step, tuple(t.shape)))
        return xt * mask + (1.0 - mask) * (0.25 * step)

    cx0 = lambda x0, t: torch.clamp(x0, -1.5, 1.5)
    cx0_old = lambda x0: torch.clamp(x0, -1.5, 1.5)          # one-argument form of the older vendored revision
    fn = D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns)
    for tag, kw in [("xt", dict(correcting_xt_fn=cxt)), ("x0", dict(correcting_x0_fn=cx0)),
                    ("both", dict(correcting_xt_fn=cxt, correcting_x0_fn=cx0)), ("x0", dict(correcting_x0_fn=cx0_old))]:
        for method, order, steps in [("multistep", 2, 8), ("singlestep", 3, 8)]:
            seen.clear()
            dpm = D.DPM_Solver(fn, ns, **kw)
            xf, inter = dpm.sample(x, steps=steps, order=order, method=method, denoise_to_zero=True,
                                   return_intermediate=True)
            pre = "cb/%s/%s/" % (tag, method)
            assert rel_err(xf.numpy(), golden.get("callbacks", pre + "final")) < TOL
            ri = golden.get("callbacks", pre + "intermediates")
            assert len(inter) == ri.shape[0]
            for i, v in enumerate(inter):
                assert rel_err(v.numpy(), ri[i]) < TOL
            if "correcting_xt_fn" in kw:
                steps_seen = [s for s, _ in seen]
                assert steps_seen == (list(range(0, 10)) if method == "multistep" else list(range(0, 4)))   # orders [3,3,2] + denoise
                assert seen[-1][1] == (1,)          # denoise_to_zero hands a (1,)-shaped t (ref :1236)


@pytest.mark.parametrize("name,sname,order,algo", [("a12", "vp_linear", 2, "dpmsolver"), ("a23", "vp_linear", 3, "dpmsolver"),
                                                   ("a23pp", "sd", 3, "dpmsolver++")])
def test_adaptive_control_loop_against_reference_goldens(golden, capsys, name, sname, order, algo):
    """DPM-Solver-12 / -23 (ref :956-1010): same accept / reject sequence (NFE) and result as the reference"""
    ns = make_schedule(sname)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns), ns, algorithm_type=algo)
    g = lambda k: golden.get("adaptive", "adaptive/%s/%s" % (name, k))
    xf = dpm.sample(tt(g("x"), "cpu"), method="adaptive", order=order, t_end=1e-3)
    assert capsys.readouterr().out.strip() == "adaptive solver nfe %d" % int(g("nfe"))
    assert rel_err(xf.numpy(), g("final")) < TOL


def test_maskblend_against_reference_callback_goldens(golden):
    """The reference's correcting_xt_fn hook with the mask-blend closure of the goldens (cb/xt:
    xt*mask + (1-mask)*(0.25*step)) expressed as a MaskBlend object, i.e. folded into the stage epilogue."""
    case = C.E2E_BY_NAME["cfg1_small"]
    ns = make_schedule("sd")
    x = tt(C.x_T_for(case), "cpu")
    mask = torch.from_numpy(golden.get("callbacks", "cb/mask"))
    levels = [torch.full(x.shape, 0.25 * step) for step in range(12)]
    fn = D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns)
    for method, order, steps in [("multistep", 2, 8), ("singlestep", 3, 8)]:
        dpm = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, intermediates=levels))
        xf, inter = dpm.sample(x, steps=steps, order=order, method=method, denoise_to_zero=True, return_intermediate=True)
        pre = "cb/xt/%s/" % method
        assert rel_err(xf.numpy(), golden.get("callbacks", pre + "final")) < TOL
        ri = golden.get("callbacks", pre + "intermediates")
        assert len(inter) == ri.shape[0]
        for i, v in enumerate(inter):
            assert rel_err(v.numpy(), ri[i]) < TOL


def test_maskblend_host_logic_equals_closure():
    shape = (2, 4, 8, 8)
    ns = make_schedule("sd")
    rng = np.random.default_rng(2)
    x, x0, noise = [torch.from_numpy(rng.standard_normal(shape).astype(F32)) for _ in range(3)]
    mask = torch.from_numpy(rng.random((8, 8)).astype(F32))
    fn = D.model_wrapper(lambda xx, t: xx * 0.5, ns)
    a_s = lambda t: (F32(ns.marginal_alpha(t.reshape(1))[0].item()), F32(ns.marginal_std(t.reshape(1))[0].item()))

    def closure(xt, t, step):
        a, s = a_s(t)
        return xt * mask + (1 - mask) * (float(a) * x0 + float(s) * noise)

    for method, order, steps in [("multistep", 2, 8), ("singlestep", 3, 9)]:
        kw = dict(steps=steps, order=order, method=method, return_intermediate=True, denoise_to_zero=True)
        want, wi = D.DPM_Solver(fn, ns, correcting_xt_fn=closure).sample(x, **kw)
        got, gi = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, x0=x0, noise=noise)).sample(x, **kw)
        assert torch.equal(got, want)
        assert len(gi) == len(wi) and all(torch.equal(p, q) for p, q in zip(gi, wi))


def test_cfg_input_buffer_host_logic():
    """classifier-free guidance: every network call after the first receives the [2B,...] buffer the previous stage
    produced (both halves equal), torch.cat([x]*2) is used for the caller's x_T only."""
    case = C.E2E_BY_NAME["cfg3_pp"]
    seen = []

    def net(xx, t, c):
        B = xx.shape[0] // 2
        assert torch.equal(xx[:B], xx[B:])
        seen.append(xx)
        return C.model_cond(xx, t, c)

    ns = make_schedule(case["schedule"])
    cond, uncond = C.cond_for(case)
    fn = D.model_wrapper(net, ns, guidance_type="classifier-free", guidance_scale=case["guidance_scale"],
                         condition=tt(cond, "cpu"), unconditional_condition=tt(uncond, "cpu"))
    dpm = D.DPM_Solver(fn, ns, algorithm_type=case["algorithm_type"])
    x = tt(C.x_T_for(case), "cpu")
    xf, inter = dpm.sample(x, **sample_kwargs(case))
    xo, _ = run_case(case, "cpu")
    assert torch.equal(xf, xo)
    assert len(seen) == case["steps"]
    for v in inter[1:-1]:       # solver states are first halves of the network-input buffers
        assert v._base is not None and v._base.shape[0] == 2 * v.shape[0]


@pytest.mark.parametrize("kw", [dict(steps=8, order=2), dict(steps=9, order=3, method="singlestep"),
                                dict(steps=6, order=3, skip_type="logSNR", denoise_to_zero=True),
                                dict(steps=5, order=2, thr=True), dict(steps=6, order=2, cfg=True),
                                dict(steps=5, order=2, model_type="v"), dict(steps=4, order=2, half=True)])
def test_sample_requests_equals_sample_per_request(kw):
    """DPM_Solver.sample_requests (requests advanced together, one dpm_stage_launch_multi per stage) == sample() of every
    request; the network is called once per request and stage"""
    kw = dict(kw)
    ns = make_schedule("sd")
    thr, cfg, half = kw.pop("thr", False), kw.pop("cfg", False), kw.pop("half", False)
    mt = kw.pop("model_type", "noise")
    calls = []
    if cfg:
        def net(x, t, c):
            calls.append(x.shape[0])
            return torch.tanh(x * 0.7) * (0.5 + 0.1 * c.reshape(-1, 1, 1, 1)[:x.shape[0]])
        c = torch.ones(4)
        fn = D.model_wrapper(net, ns, model_type=mt, guidance_type="classifier-free", guidance_scale=3.0, condition=c,
                             unconditional_condition=c * 0)
    else:
        def net(x, t):
            calls.append(x.shape[0])
            return torch.tanh(x * 0.7).to(x.dtype)
        fn = D.model_wrapper(net, ns, model_type=mt)
    dpm = D.DPM_Solver(fn, ns, correcting_x0_fn="dynamic_thresholding" if thr else None,
                       state_dtype=torch.float16 if half else None)
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(4, 3, 8, 8, generator=g) for _ in range(5)]
    if half:
        xs = [x.half() for x in xs]
    want = [dpm.sample(x, **kw) for x in xs]
    n_single = len(calls)
    calls.clear()
    got = dpm.sample_requests(xs, **kw)
    assert len(calls) == n_single and len(got) == len(xs)
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and torch.equal(a, b)
    # a second call reuses the prebuilt records and must not hand out the same result tensors
    again = dpm.sample_requests(xs, **kw)
    for a, b, c in zip(again, want, got):
        assert torch.equal(a, b) and a.data_ptr() != c.data_ptr()


def test_sample_requests_falls_back():
    """one request, mixed shapes, Python correctors, intermediates or the adaptive method: the requests run one by one"""
    ns = make_schedule("sd")
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x * 0.3, ns), ns, correcting_xt_fn=lambda x, t, step: x * 0.99)
    xs = [torch.randn(2, 3, 4, 4), torch.randn(2, 3, 4, 4)]
    for a, x in zip(dpm.sample_requests(xs, steps=5), xs):
        assert torch.equal(a, dpm.sample(x, steps=5))
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x * 0.3, ns), ns)
    mixed = [torch.randn(2, 3, 4, 4), torch.randn(1, 3, 4, 4)]
    for a, x in zip(dpm.sample_requests(mixed, steps=5), mixed):
        assert torch.equal(a, dpm.sample(x, steps=5))
    out = dpm.sample_requests(xs, steps=5, return_intermediate=True)
    assert len(out) == 2 and len(out[0]) == 2 and len(out[0][1]) == 6
    assert torch.equal(dpm.sample_requests(xs[:1], steps=5)[0], dpm.sample(xs[0], steps=5))


@pytest.mark.parametrize("sname", ["sd", "vp_linear", "cosine4000"])
@pytest.mark.parametrize("algo", ["dpmsolver++", "dpmsolver"])
def test_public_update_methods(golden, sname, algo):
    g = lambda k: torch.from_numpy(golden.get("updates", k))
    x, m = g("upd/x"), [g("upd/m%d" % i) for i in range(3)]
    t = [torch.tensor([v]) for v in golden.get("updates", "upd/t")]
    ns = make_schedule(sname)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, tv: C.model_half(xx, tv), ns), ns, algorithm_type=algo)
    pre = "upd/%s/%s/" % (sname, algo)
    tol = 3e-6
    assert rel_err(dpm.dpm_solver_first_update(x, t[2], t[3], model_s=m[2]).numpy(), g(pre + "first").numpy()) < tol
    assert rel_err(dpm.multistep_dpm_solver_update(x, [m[2]], [t[2]], t[3], 1).numpy(), g(pre + "first").numpy()) < tol
    for st in ["dpmsolver", "taylor"]:
        got = dpm.multistep_dpm_solver_second_update(x, [m[1], m[2]], [t[1], t[2]], t[3], solver_type=st)
        assert rel_err(got.numpy(), g(pre + "ms2/" + st).numpy()) < tol
        got = dpm.multistep_dpm_solver_third_update(x, m, t[:3], t[3], solver_type=st)
        assert rel_err(got.numpy(), g(pre + "ms3/" + st).numpy()) < tol
        for (r1, r2, tag) in [(None, None, "def"), (0.3, 0.75, "cust")]:
            xt, im = dpm.singlestep_dpm_solver_second_update(x, t[2], t[3], r1=r1, return_intermediate=True, solver_type=st)
            assert rel_err(xt.numpy(), g(pre + "ss2/%s/%s/x_t" % (st, tag)).numpy()) < tol
            assert rel_err(im["model_s1"].numpy(), g(pre + "ss2/%s/%s/model_s1" % (st, tag)).numpy()) < tol
            xt, im = dpm.singlestep_dpm_solver_third_update(x, t[2], t[3], r1=r1, r2=r2, return_intermediate=True,
                                                            solver_type=st)
            assert rel_err(xt.numpy(), g(pre + "ss3/%s/%s/x_t" % (st, tag)).numpy()) < tol
            assert rel_err(im["model_s1"].numpy(), g(pre + "ss3/%s/%s/model_s1" % (st, tag)).numpy()) < tol
            assert rel_err(im["model_s2"].numpy(), g(pre + "ss3/%s/%s/model_s2" % (st, tag)).numpy()) < tol
            # supplying model_s / model_s1 (what the adaptive solver does, ref :997-998) gives the same x_t
            xt2 = dpm.singlestep_dpm_solver_third_update(x, t[2], t[3], r1=r1, r2=r2, solver_type=st,
                                                         model_s=im["model_s"], model_s1=im["model_s1"])
            assert rel_err(xt2.numpy(), xt.numpy()) < 1e-7
            assert rel_err(dpm.singlestep_dpm_solver_update(x, t[2], t[3], 3, solver_type=st, r1=r1, r2=r2).numpy(),
                           xt.numpy()) == 0.0


def test_model_evaluation_methods(golden):
    """noise_prediction_fn / data_prediction_fn / model_fn / WrappedModel.__call__ vs the oracle."""
    import test_oracle_golden as T
    monkey_launch = launch_stage_double
    W_launch = S._launch_stage
    assert W_launch is monkey_launch
    for name in ["mt_v", "cfg_ms2", "clsg_ms2", "cfg5_thresh_small"]:
        case = C.E2E_BY_NAME[name]
        dpm = build_solver(case, "cpu")
        osol = T.build_oracle_solver(case)
        x = tt(C.x_T_for(case), "cpu")
        t = torch.tensor([0.6172])
        want_eps = osol.noise_pred(x.numpy(), F32(0.6172))
        want_x0 = osol.data_pred(x.numpy(), F32(0.6172))
        assert rel_err(dpm.noise_prediction_fn(x, t).numpy(), want_eps) < 2e-6
        assert rel_err(dpm.data_prediction_fn(x, t).numpy(), want_x0) < 2e-6
        assert rel_err(dpm.model_fn(x, t).numpy(), want_x0 if case["algorithm_type"] == "dpmsolver++" else want_eps) < 2e-6
        # the wrapper object itself is the reference's model_fn(x, t_continuous) -> noise
        eps = dpm._wrapped(x, t.expand(x.shape[0]))
        assert rel_err(eps.numpy(), want_eps) < 2e-6


def test_plan_structure_2m():
    """The north-star path: 20-step DPM-Solver++(2M) = 20 stages, 18 of them the 5-stream steady state."""
    ns = make_schedule("sd")
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns)
    plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1e-3)
    st = plan.stages
    assert len(st) == 20 and plan.slots == 2
    assert st[0].form == L.FORM_LIN1 and st[0].flags & L.F_STORE_M and st[0].h1_slot == -1
    for i in range(1, 20):
        assert st[i].form == L.FORM_TWO and st[i].h1_slot == (i - 1) % 2 and not (st[i].flags & L.F_BASE_HIST)
        assert bool(st[i].flags & L.F_STORE_M) == (i < 19)      # the last stage writes no model value
        assert st[i].m_slot == (i % 2 if i < 19 else -1)
    assert all(s.flags & L.F_TO_X0 for s in st) and all(s.emits_state for s in st)
    assert abs(st[0].t_input - 999.0) < 1e-3 and abs(st[19].t_input - 49.95) < 1e-3   # SURVEY 8c anchors


def test_error_conventions():
    ns = make_schedule("sd")
    with pytest.raises(ValueError, match="Unsupported noise schedule"):
        D.NoiseScheduleVP("cosine_typo")
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns)
    x = torch.zeros(2, 4, 8, 8)
    with pytest.raises(ValueError, match="Unsupported skip_type"):
        dpm.sample(x, skip_type="bogus")
    with pytest.raises(ValueError, match="Got wrong method"):
        dpm.sample(x, method="bogus")
    with pytest.raises(ValueError, match="'solver_type' must be either"):
        dpm.sample(x, solver_type="bogus")
    with pytest.raises(ValueError, match="must be '1' or '2' or '3'"):
        dpm.sample(x, method="singlestep", order=4)
    with pytest.raises(ValueError, match="Solver order must be 1 or 2 or 3"):
        dpm.sample(x, method="multistep", order=4, steps=20)
    with pytest.raises(AssertionError):
        dpm.sample(x, method="multistep", order=3, steps=2)
    with pytest.raises(AssertionError):
        dpm.sample(x, t_end=0.0)
    with pytest.raises(AssertionError, match="Cannot use adaptive solver"):
        dpm.sample(x, method="adaptive", return_intermediate=True)
    with pytest.raises(AssertionError):
        D.DPM_Solver(lambda x, t: x, ns, algorithm_type="bogus")
    with pytest.raises(AssertionError):
        D.model_wrapper(lambda x, t: x, ns, model_type="bogus")
    with pytest.raises(ValueError, match="Solver order must be 1 or 2 or 3"):
        dpm.multistep_dpm_solver_update(x, [x], [torch.tensor([0.5])], torch.tensor([0.4]), 4)


def test_no_cpu_fallback(monkeypatch):
    """Without the test double the product refuses CPU tensors instead of silently computing on the host."""
    monkeypatch.undo()
    ns = make_schedule("sd")
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dpm.sample(torch.zeros(2, 4, 8, 8), steps=5)


# ------------------------------------------------------------------------------------------------
# the Stable-Diffusion adapter (sampler.py) against goldens produced by the reference's DPMSolverSampler
# ------------------------------------------------------------------------------------------------
def sampler_checks(golden, device, tol):
    from dpm_solver_amd.adapters import DPMSolverSampler
    g = lambda k: golden.get("sampler", "sampler/" + k)
    inp = C.sampler_inputs()
    model = C.FakeLatentDiffusion(torch, device)
    smp = DPMSolverSampler(model)
    T = lambda k: tt(inp[k], device)
    x_T, x0, noise, mask, cond, uncond = T("x_T"), T("x0"), T("noise"), T("mask"), T("cond"), T("uncond")
    B = x_T.shape[0]
    x, inter = smp.sample(10, B, x_T.shape[1:], conditioning=cond, unconditional_guidance_scale=7.5,
                          unconditional_conditioning=uncond, x_T=x_T, verbose=False)
    assert rel_err(x.cpu().numpy(), g("sample/final")) < tol
    ri = g("sample/intermediates")
    assert len(inter) == ri.shape[0] and all(rel_err(v.cpu().numpy(), ri[i]) < tol for i, v in enumerate(inter))
    np.testing.assert_allclose(np.array([t for _, t in model.calls]), g("sample/calls_t"), rtol=1e-6)
    assert all(s == (2 * B,) + tuple(x_T.shape[1:]) for s, _ in model.calls)       # one batched CFG call per step
    assert rel_err(smp.stochastic_encode(x0, 0.6, noise=noise.unsqueeze(0)).cpu().numpy(), g("stochastic_encode")) < tol
    enc, einter = smp.encode(10, x0, 0.6, conditioning=cond, unconditional_guidance_scale=7.5, unconditional_conditioning=uncond)
    assert rel_err(enc.cpu().numpy(), g("encode/final")) < tol
    ri = g("encode/intermediates")
    assert len(einter) == ri.shape[0] and all(rel_err(v.cpu().numpy(), ri[i]) < tol for i, v in enumerate(einter))
    tv = torch.tensor([0.001, 0.25, 0.6004, 1.0])
    got = np.stack([smp.time_discrete_to_continuous(tv * 999).numpy(), smp.time_continuous_to_discrete(tv).numpy(),
                    smp.ratio_to_time(tv).numpy(), smp.time_to_ratio(tv).numpy()])
    np.testing.assert_allclose(got, g("times"), rtol=1e-6)
    # DiffEdit (diffedit_inpaint.ipynb cell 6), both as the notebook's closures and as fused MaskBlend objects
    N = smp.noise_schedule.total_N
    rev = list(reversed(einter))
    noised = smp.stochastic_encode(x0, 0.6, noise=noise.unsqueeze(0))
    f32 = np.float32

    def remap(t):      # ratio_to_time(time_to_ratio(t)) in the reference's fp32 tensor arithmetic
        r = (f32(t) - f32(1. / N)) / f32(1. - N)
        return float(f32(f32(1. - 1. / N) * r) + f32(1. / N))

    det_c = lambda xt, t, step: xt * mask + (1 - mask) * rev[step]
    sto_c = lambda xt, t, step: xt * mask + (1 - mask) * smp.stochastic_encode(x0, smp.time_to_ratio(t), noise=noise.unsqueeze(0))
    det_f = D.MaskBlend(smp.noise_schedule, mask, intermediates=rev)
    sto_f = D.MaskBlend(smp.noise_schedule, mask, x0=x0, noise=noise, time_fn=remap)
    for key, start, fns in [("diffedit_det", enc, (det_c, det_f)), ("diffedit_sto", noised, (sto_c, sto_f))]:
        for fn in fns:
            x, _ = smp.sample(10, B, x_T.shape[1:], conditioning=cond * 0.5, unconditional_guidance_scale=7.5,
                              unconditional_conditioning=uncond, lower_order_final=False, t_start=smp.ratio_to_time(0.6),
                              x_T=start, correcting_xt_fn=fn)
            assert rel_err(x.cpu().numpy(), g(key)) < tol, (key, type(fn).__name__)


def test_stable_diffusion_adapter_against_reference_goldens(golden):
    sampler_checks(golden, "cpu", TOL)


# ------------------------------------------------------------------------------------------------
# the older vendored revision (examples/score_sde_pytorch/dpm_solver.py): 'cosine' schedule, unclipped tables
# ------------------------------------------------------------------------------------------------
def legacy_checks(golden, device):
    from test_oracle_golden import LEGACY_RUNS
    g = lambda k: golden.get("legacy", "legacy/" + k)
    with pytest.raises(ValueError, match="Unsupported noise schedule cosine"):
        D.NoiseScheduleVP("cosine")                          # the root revision does not know it (ref :94-95)
    ns = D.LegacyNoiseScheduleVP("cosine")
    assert ns.T == float(g("cosine/T")) and ns.total_N == 1000
    osch = O.Schedule.cosine()
    t = g("cosine/t")
    # planner == oracle bit for bit (both round the elementary functions correctly); oracle vs reference: test_oracle_golden
    np.testing.assert_array_equal(ns.marginal_log_mean_coeff(torch.from_numpy(t)).numpy(), osch.log_alpha_t(t))
    np.testing.assert_array_equal(ns.marginal_lambda(torch.from_numpy(t)).numpy(), osch.lam(t))
    lam = g("cosine/lambda")
    np.testing.assert_array_equal(ns.inverse_lambda(torch.from_numpy(lam)).numpy(), osch.inv_lam(lam))
    x = tt(g("x"), device)
    for tag, algo, kw in LEGACY_RUNS:
        dpm = D.DPM_Solver(D.model_wrapper(lambda xx, tv: C.model_tdep(xx, tv), ns), ns, algorithm_type=algo)
        assert rel_err(dpm.sample(x, t_end=1e-3, **kw).cpu().numpy(), g("cosine/" + tag)) < TOL, tag
    nd = D.LegacyNoiseScheduleVP("discrete", betas=torch.from_numpy(C.schedule_inputs("cosine4000")["betas"]))
    assert nd.total_N == int(g("noclip/total_N")) == 4000
    assert make_schedule("cosine4000").total_N == 3984          # the root revision clips (ref :114-125)
    np.testing.assert_allclose(nd.log_alpha_array.numpy()[0, -32:], g("noclip/log_alpha_tail"), rtol=2e-6)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, tv: C.model_tdep(xx, tv), nd), nd)
    assert rel_err(dpm.sample(x, steps=10, order=2, t_start=0.9).cpu().numpy(), g("noclip/ms2")) < TOL


def test_legacy_revision_cosine_schedule_and_unclipped_tables(golden):
    legacy_checks(golden, "cpu")


# ------------------------------------------------------------------------------------------------
# seeded random sweep over sample() configurations: C planner + host loop (kernel double) against the oracle
# ------------------------------------------------------------------------------------------------
def random_configs(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        method = str(rng.choice(["multistep", "singlestep", "singlestep_fixed"]))
        order = int(rng.integers(1, 4))
        steps = int(rng.integers(order if method == "multistep" else 1, 24))
        sname = str(rng.choice(["sd", "ddpm", "vp_linear", "cosine1000"]))
        skip = str(rng.choice(["time_uniform", "logSNR", "time_quadratic"]))
        out.append(dict(method=method, order=order, steps=steps, schedule=sname, skip_type=skip,
                        solver_type=str(rng.choice(["dpmsolver", "taylor"])),
                        algorithm_type=str(rng.choice(["dpmsolver++", "dpmsolver"])),
                        model_type=str(rng.choice(["noise", "x_start", "v", "score"])),
                        lower_order_final=bool(rng.integers(0, 2)), denoise_to_zero=bool(rng.integers(0, 2)),
                        t_end=float(rng.choice([1e-3, 1e-2, 0.05])), t_start=float(rng.choice([1.0, 0.8, 0.5])),
                        seed=int(rng.integers(0, 1 << 30))))
    return out


def run_random_config(cfg, device):
    """(engine result, oracle result) of one random configuration"""
    from test_oracle_golden import make_schedule as make_oracle_schedule
    ns, osch = make_schedule(cfg["schedule"]), make_oracle_schedule(cfg["schedule"])
    rng = np.random.default_rng(cfg["seed"])
    x = rng.standard_normal((2, 3, 6, 6)).astype(F32)
    kw = dict(steps=cfg["steps"], order=cfg["order"], method=cfg["method"], skip_type=cfg["skip_type"],
              solver_type=cfg["solver_type"], lower_order_final=cfg["lower_order_final"],
              denoise_to_zero=cfg["denoise_to_zero"], t_start=cfg["t_start"], t_end=cfg["t_end"])
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: C.model_tdep(xx, t), ns, model_type=cfg["model_type"]), ns,
                       algorithm_type=cfg["algorithm_type"])
    got = dpm.sample(tt(x, device), **kw).cpu().numpy()
    sol = O.Solver(O.wrap_model(lambda xx, t: C.model_tdep(xx, t), osch, model_type=cfg["model_type"]), osch,
                   algorithm_type=cfg["algorithm_type"])
    return got, sol.sample(x, **kw)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_configurations_against_oracle(seed):
    for cfg in random_configs(seed, 40):
        got, want = run_random_config(cfg, "cpu")
        assert np.all(np.isfinite(want)), cfg
        assert rel_err(got, want) < TOL, cfg


# ------------------------------------------------------------------------------------------------
# the guided-diffusion runner's DPM-Solver branch (runners/diffusion.py:594-640) against goldens from the reference
# ------------------------------------------------------------------------------------------------
def guided_checks(golden, device, tol):
    from dpm_solver_amd.adapters import guided_diffusion_sample_image
    inp = C.gd_inputs()
    x, y = tt(inp["x"], device), torch.from_numpy(inp["y"]).to(device)
    model = C.gd_network(torch, tt(inp["junk"], device))
    classifier = C.gd_classifier(torch, tt(inp["w"], device))
    betas = torch.from_numpy(C.schedule_inputs("ddpm")["betas"])
    for tag, kw in C.GD_RUNS:
        got, cls = guided_diffusion_sample_image(
            x, model, betas, classifier=classifier if kw["use_clf"] else None, classes=y,
            classifier_scale=kw["scale"], out_channels=6, sample_type=kw["sample_type"], thresholding=kw["thresholding"],
            timesteps=kw.get("timesteps", 12), denoise=kw["denoise"], dpm_solver_order=kw.get("order", 2),
            dpm_solver_method=kw.get("method", "multistep"))
        assert cls is y
        assert rel_err(got.cpu().numpy(), golden.get("guided", "guided/" + tag)) < tol, tag


def test_guided_diffusion_adapter_against_reference_goldens(golden):
    guided_checks(golden, "cpu", TOL)


# ------------------------------------------------------------------------------------------------
# round 3 host-side additions
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [False, True])
def test_fresh_time_tensors_for_networks_that_write_into_t(cfg):
    """The (batch,) time vectors are built once per plan and shared by every call; a network that edits its time argument
    in place corrupts them unless DPM_Solver.fresh_time_tensors hands out clones (the reference makes a fresh tensor per
    call, ref :404)."""
    ns = make_schedule("sd")
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((3, 4, 8, 8)).astype(F32))

    def make(mutate):
        def net(xx, t, c=None):
            scale = (t * 0.0005 + 0.25).reshape(-1, 1, 1, 1)
            out = xx * scale
            if mutate:
                t.mul_(0.0)                         # what a careless network might do
            return out
        if cfg:
            cond = torch.ones(3)
            return D.model_wrapper(net, ns, guidance_type="classifier-free", condition=cond, unconditional_condition=cond * 0,
                                   guidance_scale=2.0)
        return D.model_wrapper(net, ns)
    want = D.DPM_Solver(make(False), ns).sample(x, steps=6, order=2)
    dpm = D.DPM_Solver(make(True), ns)
    dpm.fresh_time_tensors = True
    for _ in range(2):                              # the second call would see the first call's damage
        assert torch.equal(dpm.sample(x, steps=6, order=2), want)
    # correctors / intermediates take the general loop: same guarantee
    got, inter = dpm.sample(x, steps=6, order=2, return_intermediate=True)
    assert torch.equal(got, want) and len(inter) == 7


@pytest.mark.parametrize("cfg", [False, True])
def test_a_network_that_writes_into_t_is_detected_and_gets_clones(cfg):
    """without anyone setting fresh_time_tensors, a network that edits its time argument in place must not
    corrupt later calls: the version counters of the shared vectors give the write away, the vectors are rebuilt from the host
    plan and the solver hands out clones from then on -- every call equals the reference-style result."""
    ns = make_schedule("sd")
    x = torch.from_numpy(np.random.default_rng(4).standard_normal((2, 4, 8, 8)).astype(F32))

    def make(mutate):
        def net(xx, t, c=None):
            out = xx * (t * 0.0005 + 0.25).reshape(-1, 1, 1, 1)
            if mutate:
                t.mul_(1000.0)
            return out
        if cfg:
            cond = torch.ones(2)
            return D.model_wrapper(net, ns, guidance_type="classifier-free", condition=cond, unconditional_condition=cond * 0,
                                   guidance_scale=2.0)
        return D.model_wrapper(net, ns)
    want = D.DPM_Solver(make(False), ns).sample(x, steps=6, order=2)
    dpm = D.DPM_Solver(make(True), ns)
    assert not dpm.fresh_time_tensors
    for _ in range(3):
        assert torch.equal(dpm.sample(x, steps=6, order=2), want)
    assert dpm.fresh_time_tensors                     # switched on by the detection, not by the caller
    got, inter = dpm.sample(x, steps=6, order=2, return_intermediate=True)
    assert torch.equal(got, want)
    # requests in flight share one row per stage: the write is caught after the first request's first call
    dpm2 = D.DPM_Solver(make(True), ns)
    xs = [x, x * 0.5, x + 0.25]
    ref = D.DPM_Solver(make(False), ns)
    for _ in range(2):
        for got, xx in zip(dpm2.sample_requests(xs, steps=6, order=2), xs):
            assert torch.equal(got, ref.sample(xx, steps=6, order=2))
    # singlestep with a batch of one (expand would alias the plan's own table there)
    x1 = x[:1]
    want1 = D.DPM_Solver(make(False), ns).sample(x1, steps=7, order=3, method="singlestep")
    dpm3 = D.DPM_Solver(make(True), ns)
    for _ in range(3):
        assert torch.equal(dpm3.sample(x1, steps=7, order=3, method="singlestep"), want1)


def test_sampling_under_inference_mode():
    """tensors created under torch.inference_mode() do not track version counters: the write detection of the shared time
    vectors switches itself off there instead of raising (SD pipelines sample under inference_mode)"""
    ns = make_schedule("sd")
    x = torch.from_numpy(np.random.default_rng(8).standard_normal((2, 4, 8, 8)).astype(F32))
    model = D.model_wrapper(lambda xx, t: xx * (t * 0.0005 + 0.25).reshape(-1, 1, 1, 1), ns)
    want = D.DPM_Solver(model, ns).sample(x, steps=6, order=2)
    dpm = D.DPM_Solver(model, ns)
    with torch.inference_mode():
        for _ in range(2):
            got = dpm.sample(x, steps=6, order=2)
            assert torch.equal(got, want)
        outs = dpm.sample_requests([x, x * 0.5], steps=6, order=2)
        assert torch.equal(outs[0], want)
    assert torch.equal(dpm.sample(x, steps=6, order=2), want)       # and the same solver keeps working outside


def test_abi_version_and_load_time_checks():
    """the structs grew (thr_hint) -- the version says so, and a binding can verify its layout at load time
    (dpm_sizeof); DPM_ERR_FAULT is retired: no hook machinery is left in the binding."""
    assert L.lib.dpm_version() >= 200
    import ctypes
    for i, t in enumerate((L.Stage, L.Buffers, L.PlanDesc, L.RunBuffers, L.AdaptiveDesc, L.LaunchOpts)):
        assert L.lib.dpm_sizeof(i) == ctypes.sizeof(t)
    assert not hasattr(L, "fault_hooks")
    assert L.cluster_timeout_poll() is False            # no clustered launch has run in this process


@pytest.mark.lab
def test_tuning_knobs_validate_their_values():
    """the run-time tuning hooks are host state (no device needed): set / get round trip, bad values refused with a text"""
    for knob, good, bad in [(L.TUNE_BLOCK_THREADS, (256, 512, 0), (1024, 100, -1)),
                            (L.TUNE_THR_DEBUG_FAULT, (1, 2, 3, 0), (4, -1)),
                            (L.TUNE_THR_SPIN_LIMIT, (0, 32, 4096), (-1,))]:
        old = L.lib.dpm_tuning_get(knob)
        try:
            for v in good:
                assert L.lib.dpm_tuning_set(knob, v) == 0 and L.lib.dpm_tuning_get(knob) == v
            for v in bad:
                assert L.lib.dpm_tuning_set(knob, v) != 0
                assert L.lib.dpm_last_error().decode()
                assert L.lib.dpm_tuning_get(knob) == good[-1]
        finally:
            L.lib.dpm_tuning_set(knob, old)
    assert L.lib.dpm_tuning_set(999, 0) != 0 and L.lib.dpm_tuning_get(999) == -1


@pytest.mark.lab
def test_every_device_ordinal_has_a_context_of_its_own():
    """8-GPU readiness that needs no GPU: one process per GPU, so 7 of 8 ranks launch on a device ordinal != 0.  The only
    state of the library that outlives a call is its per-device context (chain of clustered thresholding launches,
    diagnostics word): ordinals 0..7 must not share one, the same ordinal must get the same one from every thread."""
    import threading
    ctx = [L.lib.dpm_lab_device_context(d) for d in range(8)]
    assert all(ctx) and len(set(ctx)) == 8
    seen = []
    th = [threading.Thread(target=lambda d=d: seen.append((d, L.lib.dpm_lab_device_context(d)))) for d in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert sorted(seen) == sorted(enumerate(ctx))
    assert L.lib.dpm_lab_device_context(-1) == ctx[0]      # "no current device" falls back to ordinal 0


def test_lab_suite_on_cpu():
    """the lab-marked host tests (tuning knob validation, per-device contexts), on the lab build in a subprocess"""
    from conftest import run_lab_suite
    assert run_lab_suite("lab and not gpu", timeout=600) >= 2


def test_launch_options_travel_per_call_not_per_process(monkeypatch):
    """dpm_launch_opts (DPM_Solver.cluster_in_graph / thr_spin_limit): the prebuilt launch records of a solver carry a
    pointer to ITS options; another solver in the same process keeps the defaults (a null pointer)."""
    ns = make_schedule("ddpm")
    install_cpu_double(monkeypatch, S, D)
    seen = []
    real = S._stage_launch_raw

    def spy(st, b, stream):
        bb = b._obj
        seen.append((bool(bb.opts), bb.opts.contents.cluster_in_graph if bb.opts else 0,
                     bb.opts.contents.thr_spin_limit if bb.opts else 0))
        return real(st, b, stream)
    monkeypatch.setattr(S, "_stage_launch_raw", spy)
    x = torch.randn(2, 3, 8, 8)
    a = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns, correcting_x0_fn="dynamic_thresholding")
    b = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns, correcting_x0_fn="dynamic_thresholding")
    a.cluster_in_graph, a.thr_spin_limit = True, 77
    ya = a.sample(x, steps=4, order=2)
    na = len(seen)
    yb = b.sample(x, steps=4, order=2)
    assert na == 4 and all(s == (True, 1, 77) for s in seen[:na])
    assert all(s == (False, 0, 0) for s in seen[na:]) and len(seen) == 8
    assert torch.equal(ya, yb)


def test_plan_and_adaptive_caches_are_bounded():
    ns = make_schedule("sd")
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns)
    plan = dpm._get_plan(method="multistep", order=2, steps=5, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1e-3)
    for b in range(1, 14):
        plan.time_views("cpu", b, False)
    assert len(plan._views) <= 8


def test_numerical_clip_alpha_method_and_c_entry_agree():
    ns = make_schedule("cosine1000")
    raw = np.linspace(-1e-4, -9.0, 400).astype(np.float32)
    keep = C_.c_int()
    L.check(L.lib.dpm_numerical_clip_len_f32(raw.ctypes.data_as(C_.POINTER(C_.c_float)), 400, -5.1, C_.byref(keep)))
    out = ns.numerical_clip_alpha(torch.from_numpy(raw))
    assert out.shape[0] == keep.value and 0 < keep.value < 400
    assert L.lib.dpm_numerical_clip_len_f32(None, 4, -5.1, C_.byref(keep)) == L.ERR_ARG


# ------------------------------------------------------------------------------------------------
# the ScoreSDE example's sampler (examples/score_sde_pytorch/sampling.py:505-555)
# ------------------------------------------------------------------------------------------------
class VPSDE:
    """what get_dpm_solver_sampler reads of sde_lib.VPSDE: beta_0, beta_1, T, prior_sampling (the adapter recognises the
    class by its name, like the reference's isinstance(sde, sde_lib.VPSDE))"""
    beta_0, beta_1, T = 0.1, 20., 1

    def prior_sampling(self, shape):
        return torch.randn(*shape)


_VPSDE = VPSDE


class subVPSDE:
    """sde_lib.subVPSDE has the same attributes but another marginal distribution: not a VP model"""
    beta_0, beta_1, T = 0.1, 20., 1

    def prior_sampling(self, shape):
        return torch.randn(*shape)


def test_score_sde_adapter_rejects_sdes_that_are_not_vp():
    """the reference raises NotImplementedError for anything but a VPSDE (models/utils.py:143-153); an SDE
    that merely carries beta_0 / beta_1 (subVPSDE) must not be sampled with the VP noise parametrisation"""
    from dpm_solver_amd.adapters import score_sde_get_dpm_solver_sampler
    fn = score_sde_get_dpm_solver_sampler(subVPSDE(), C.SCORE_SDE_SHAPE, lambda x: x, device="cpu")
    with pytest.raises(NotImplementedError, match="subVPSDE"):
        fn(C.score_sde_model(torch))

    class Derived(VPSDE):
        pass
    y, nfe = score_sde_get_dpm_solver_sampler(Derived(), C.SCORE_SDE_SHAPE, lambda x: x, device="cpu")(C.score_sde_model(torch))
    assert nfe == 10 and torch.isfinite(y).all()


def run_score_sde_adapter(device, golden, thresholding_too=False):
    from dpm_solver_amd.adapters import score_sde_get_dpm_solver_sampler
    worst = 0.0
    for i, (tag, kw) in enumerate(C.SCORE_SDE_RUNS):
        torch.manual_seed(100 + i)
        fn = score_sde_get_dpm_solver_sampler(_VPSDE(), C.SCORE_SDE_SHAPE, lambda x: (x + 1.) / 2., device=device, **kw)
        y, nfe = fn(C.score_sde_model(torch).to(device))
        assert nfe == int(golden.get("score_sde", "score_sde/%s/nfe" % tag))
        assert y.dtype == torch.float32 and str(y.device).startswith(str(device).split(":")[0])
        worst = max(worst, rel_err(y.cpu().numpy(), golden.get("score_sde", "score_sde/%s/x" % tag)))
    if thresholding_too:    # the reference's vendored revision raises TypeError with thresholding=True; the engine runs it
        torch.manual_seed(7)
        fn = score_sde_get_dpm_solver_sampler(_VPSDE(), C.SCORE_SDE_SHAPE, lambda x: x, device=device, thresholding=True,
                                              algorithm_type="dpmsolver++", method="multistep", order=2, steps=8)
        y, _ = fn(C.score_sde_model(torch).to(device))
        assert torch.isfinite(y).all() and y.shape == C.SCORE_SDE_SHAPE
    return worst


def test_score_sde_sampler_against_reference_goldens(golden, capsys):
    """goldens produced by the example's own sampling.get_dpm_solver_sampler (tests/golden/make_golden.py score_sde)"""
    assert run_score_sde_adapter("cpu", golden, thresholding_too=True) < TOL


# ------------------------------------------------------------------------------------------------
# round 4: channels_last (NHWC) networks run on their own storage
# ------------------------------------------------------------------------------------------------
class _CopySpy:
    """counts the layout / dtype conversions the solver makes (calls of S._conv that return a new tensor)"""

    def __init__(self, monkeypatch):
        self.copies = 0
        real = S._conv

        def conv(t, dt, mf=None):
            out = real(t, dt, mf)
            if t is not None and out is not t:
                self.copies += 1
            return out
        monkeypatch.setattr(S, "_conv", conv)


def _nhwc_model(ns, cfg, thr_scale=False, fmt=torch.channels_last):
    def net(xx, t, c=None):
        scale = (t * 0.0005 + 0.25).reshape(-1, 1, 1, 1)
        if c is not None:
            scale = scale * (1.0 + 0.1 * c.reshape(-1, 1, 1, 1))
        out = xx * scale
        return out.contiguous(memory_format=fmt) if fmt is not None else out.contiguous()
    if cfg:
        cond = torch.ones(3)
        return D.model_wrapper(net, ns, guidance_type="classifier-free", condition=cond, unconditional_condition=cond * 0,
                               guidance_scale=3.0)
    return D.model_wrapper(net, ns)


@pytest.mark.parametrize("cfg", [False, True])
@pytest.mark.parametrize("thr", [False, True])
@pytest.mark.parametrize("x_nhwc", [False, True])
def test_channels_last_network_runs_on_its_own_storage(cfg, thr, x_nhwc, monkeypatch):
    """A network that answers in channels_last: the launch records point at its outputs as they are, the scratch states
    are NHWC too (the network is handed the layout it works in), x_T is converted at most once and the result comes back
    in x_T's layout -- bit-identical to the default-layout run (the kernels are elementwise over the flat storage, and the
    thresholding quantile is a per-sample order statistic)."""
    ns = make_schedule("sd")
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((3, 4, 8, 8)).astype(F32))
    kw = dict(correcting_x0_fn="dynamic_thresholding") if thr else {}
    want = D.DPM_Solver(_nhwc_model(ns, cfg, fmt=None), ns, **kw).sample(x, steps=8, order=2)
    xin = x.contiguous(memory_format=torch.channels_last) if x_nhwc else x
    seen = []
    model = _nhwc_model(ns, cfg)
    inner = model.model

    def spy_net(xx, t, *a):
        seen.append(xx.is_contiguous(memory_format=torch.channels_last))
        return inner(xx, t, *a)
    model.model = spy_net
    dpm = D.DPM_Solver(model, ns, **kw)
    spy = _CopySpy(monkeypatch)
    got = dpm.sample(xin, steps=8, order=2)
    assert torch.equal(got, want)
    assert got.is_contiguous(memory_format=torch.channels_last) == x_nhwc and (x_nhwc or got.is_contiguous())
    # conversions: x_T into the network's layout and the result back into x_T's -- none per stage
    assert spy.copies == (0 if x_nhwc else 2), spy.copies
    # from the second evaluation on the network is handed NHWC states (the first one sees the caller's x_T)
    assert all(seen[1:]) and seen[0] == x_nhwc
    # requests in flight and the general loop (intermediates) keep the guarantee
    outs = dpm.sample_requests([xin, xin * 0.5], steps=8, order=2)
    assert torch.equal(outs[0], want) and outs[0].is_contiguous(memory_format=torch.channels_last) == x_nhwc
    got2, inter = dpm.sample(xin, steps=8, order=2, return_intermediate=True)
    assert torch.equal(got2, want) and len(inter) == 9
    assert all(v.is_contiguous(memory_format=torch.channels_last) == x_nhwc for v in inter[1:])


def test_channels_last_x_T_with_a_default_layout_network(monkeypatch):
    """the other mixed case: x_T in NHWC, the network answers in the default layout -> the run stays in the network's
    layout, x_T is converted once, the result returns in NHWC"""
    ns = make_schedule("sd")
    x = torch.from_numpy(np.random.default_rng(6).standard_normal((2, 4, 8, 8)).astype(F32))
    want = D.DPM_Solver(_nhwc_model(ns, False, fmt=None), ns).sample(x, steps=5, order=2)
    spy = _CopySpy(monkeypatch)
    got = D.DPM_Solver(_nhwc_model(ns, False, fmt=None), ns).sample(x.contiguous(memory_format=torch.channels_last), steps=5, order=2)
    assert torch.equal(got, want) and got.is_contiguous(memory_format=torch.channels_last)
    assert spy.copies == 2


def test_adaptive_host_loop_raises_on_a_nan_estimate_instead_of_spinning():
    """A NaN error estimate is never accepted and turns the step size into NaN: the reference's loop (ref :1002-1008) then never
    ends.  The engine raises where the reference would hang (the device-side controller is bounded by adaptive_max_iterations)."""
    ns = make_schedule("vp_linear")
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((2, 3, 4, 4)).astype(F32))
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * float("nan"), ns), ns)
    dpm.adaptive_on_device = False
    with pytest.raises(FloatingPointError, match="error estimate is NaN"):
        dpm.sample(x, method="adaptive", order=2)
