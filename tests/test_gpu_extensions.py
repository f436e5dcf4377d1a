"""GPU tests of the SURVEY 8f rows built around the stage kernel: the CFG input producer (x_out written into both
halves of the [2B,...] network input), in-place reads of channel-sliced network outputs (learned-variance models),
the fused mask-blend corrector (DiffEdit / inpainting) and hipGraph capture.  Each is checked bit for bit against
the same computation done the reference's way (torch.cat / .contiguous() / a Python closure / eager launches), and
the end-to-end cases against the oracle.  Run on an MI355X:  pytest -m gpu
"""
import ctypes as C_

import numpy as np
import pytest
import torch

import cases as C
import dpm_solver_amd as D
import dpm_solver_amd.solver as S
from conftest import rel_err
from dpm_solver_amd import _lib as L
from engine_cases import build_solver, make_schedule, run_case, sample_kwargs, tt
import test_oracle_golden as TO

pytestmark = pytest.mark.gpu
F32 = np.float32
TOL = 1e-5
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need a GPU; run with -m 'not gpu' elsewhere"
    yield
    torch.cuda.synchronize()


class LaunchSpy:
    """records the dpm_buffers of every dpm_stage_launch"""

    def __init__(self, monkeypatch):
        self.calls = []
        real = L.lib.dpm_stage_launch

        def spy(st, b, stream):
            bo, so = b._obj, st._obj
            self.calls.append(dict(x_out2=bo.x_out2, eps_stride=bo.eps_stride, mask=bo.mask, flags=so.flags, form=so.form))
            return real(st, b, stream)

        monkeypatch.setattr(S.L.lib, "dpm_stage_launch", spy)
        monkeypatch.setattr(S, "_stage_launch_raw", spy)          # the prebuilt launch records of sample()


# ------------------------------------------------------------------------------------------------
# classifier-free guidance: the stage kernel writes the network's [2B,...] input itself
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cfg_ms2", "cfg3_dpmsolver", "cfg3_pp", "cfg_v"])
def test_cfg_network_input_is_produced_by_the_stage_kernel(name, monkeypatch):
    case = C.E2E_BY_NAME[name]
    seen = []
    cats = []
    real_cat = torch.cat

    def counting_cat(ts, *a, **k):
        if len(ts) == 2 and ts[0] is ts[1]:
            cats.append(1)
        return real_cat(ts, *a, **k)

    ns = make_schedule(case["schedule"])
    base = C.MODELS[case["model"]]

    def net(x, t, cond=None):
        B = x.shape[0] // 2
        assert torch.equal(x[:B], x[B:])                       # both halves hold the same state
        seen.append((x.data_ptr(), x.is_contiguous()))
        return base(x, t, cond)

    cond, uncond = C.cond_for(case)
    fn = D.model_wrapper(net, ns, model_type=case["model_type"], guidance_type="classifier-free",
                         guidance_scale=case["guidance_scale"], condition=tt(cond, DEV), unconditional_condition=tt(uncond, DEV))
    dpm = D.DPM_Solver(fn, ns, algorithm_type=case["algorithm_type"])
    spy = LaunchSpy(monkeypatch)
    monkeypatch.setattr(torch, "cat", counting_cat)
    x = tt(C.x_T_for(case), DEV)
    xf, inter = dpm.sample(x, **sample_kwargs(case))
    monkeypatch.setattr(torch, "cat", real_cat)
    assert len(cats) == 1, "only x_T (the caller's tensor) is duplicated with torch.cat"
    assert all(c for _, c in seen)
    n_dup = sum(1 for c in spy.calls if c["x_out2"])
    assert n_dup == len(spy.calls) - 1                         # every stage but the last feeds a network call
    xo, _ = TO.run_oracle_case(case)
    assert rel_err(xf.cpu().numpy(), xo) < TOL
    # and bit-identical to the torch.cat path (an opaque corrector forces it)
    dpm_cat = D.DPM_Solver(fn, ns, algorithm_type=case["algorithm_type"], correcting_xt_fn=lambda xt, t, step: xt)
    seen.clear()
    xc, _ = dpm_cat.sample(x, **sample_kwargs(case))
    assert torch.equal(xf, xc)


# ------------------------------------------------------------------------------------------------
# learned-variance networks: out[:, :C] of a [B,2C,H,W] output is read in place (runners/diffusion.py:596-603)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("guidance,thresh,shape", [("uncond", False, (4, 3, 32, 32)), ("classifier-free", False, (4, 3, 32, 32)),
                                                   ("uncond", True, (4, 3, 32, 32)), ("classifier", True, (3, 3, 16, 16)),
                                                   ("uncond", True, (2, 3, 256, 256)), ("uncond", False, (3, 3, 5, 7))])
def test_channel_sliced_network_output_is_read_in_place(guidance, thresh, shape, monkeypatch):
    ns = make_schedule("ddpm")
    B, Cc = shape[0], shape[1]
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV)
    cond = torch.arange(B, device=DEV, dtype=torch.float32) * 0.25
    junk = torch.from_numpy(rng.standard_normal((2 * B,) + shape[1:]).astype(F32)).to(DEV)

    def mean_of(xx, t, c=None):
        out = xx * 0.5 if c is None else xx * (c * 0.1 + 1.0).reshape(-1, 1, 1, 1)
        return out

    def net6(xx, t, c=None, **kw):        # [B, 2C, H, W]: mean and variance channels; the solver uses the mean
        out = torch.cat([mean_of(xx, t, c), junk[:xx.shape[0]]], dim=1)
        return torch.split(out, Cc, dim=1)[0]

    def net3(xx, t, c=None, **kw):
        return mean_of(xx, t, c).contiguous()

    def solver(net):
        kw = dict(guidance_type=guidance)
        if guidance == "classifier-free":
            kw.update(condition=cond, unconditional_condition=torch.zeros_like(cond), guidance_scale=3.0)
        elif guidance == "classifier":
            kw.update(condition=cond, guidance_scale=2.0, classifier_fn=C.classifier_logp_torch)
        return D.DPM_Solver(D.model_wrapper(net, ns, **kw), ns,
                            correcting_x0_fn="dynamic_thresholding" if thresh else None)

    want = solver(net3).sample(x, steps=8, order=2)
    spy = LaunchSpy(monkeypatch)
    got = solver(net6).sample(x, steps=8, order=2)
    assert torch.equal(got, want)
    per_sample = int(np.prod(shape[1:]))
    assert all(c["eps_stride"] == 2 * per_sample for c in spy.calls), [c["eps_stride"] for c in spy.calls]


# ------------------------------------------------------------------------------------------------
# MaskBlend: the DiffEdit / inpainting corrector folded into the stage kernel's epilogue
# ------------------------------------------------------------------------------------------------
def _blend_setup(shape, mask_shape, seed=3):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV)
    x0 = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV)
    noise = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV)
    mask = torch.from_numpy(rng.random(mask_shape).astype(F32)).to(DEV)        # soft mask: exercises the arithmetic
    return x, x0, noise, mask


@pytest.mark.parametrize("method,order,steps", [("multistep", 2, 10), ("singlestep", 3, 9), ("multistep", 3, 8)])
@pytest.mark.parametrize("mask_shape", [(16, 16), (1, 4, 16, 16), (2, 4, 16, 16), (2, 1, 16, 16)])
def test_maskblend_fused_equals_closure_stochastic(method, order, steps, mask_shape, monkeypatch):
    shape = (2, 4, 16, 16)
    ns = make_schedule("sd")
    x, x0, noise, mask = _blend_setup(shape, mask_shape)
    fn = D.model_wrapper(lambda xx, t: xx * 0.5, ns)
    helper = D.DPM_Solver(fn, ns)

    def closure(xt, t, step):            # what the notebook passes (diffedit_inpaint.ipynb cell 6)
        return xt * mask + (1 - mask) * helper.add_noise(x0, t.reshape(1), noise=noise.unsqueeze(0))

    kw = dict(steps=steps, order=order, method=method, return_intermediate=True, denoise_to_zero=True)
    want, wi = D.DPM_Solver(fn, ns, correcting_xt_fn=closure).sample(x, **kw)
    spy = LaunchSpy(monkeypatch)
    got, gi = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, x0=x0, noise=noise)).sample(x, **kw)
    assert torch.equal(got, want)
    assert len(gi) == len(wi) and all(torch.equal(a, b) for a, b in zip(gi, wi))
    assert sum(1 for c in spy.calls if c["flags"] & L.F_BLEND) >= steps // order     # folded into the stage kernels
    # the object is also a plain callable (stand-alone kernel)
    t = torch.tensor(0.37, device=DEV)
    assert torch.equal(D.MaskBlend(ns, mask, x0=x0, noise=noise)(x, t, 3), closure(x, t, 3))


@pytest.mark.parametrize("mask_shape", [(64, 64), (1, 4, 64, 64), (8, 4, 64, 64)])
@pytest.mark.parametrize("sdt", [torch.float32, torch.float16])
@pytest.mark.parametrize("cfg", [False, True])
def test_maskblend_first_stage_takes_the_vector_kernel(mask_shape, sdt, cfg, monkeypatch):
    """The first stage of a multistep run with a corrector has a separate evaluation state (the network saw the raw x_T,
    the update starts from the blended state, ref :1179-1183): first-order form + xe != x.  At whole-tile sizes it runs
    the streaming kernel (split-tile layout for fp32) -- bit-equal to the Python closure and to the kernel double."""
    from kernel_double import install_cpu_double
    shape = (8, 4, 64, 64)
    ns = make_schedule("sd")
    x, x0, noise, mask = _blend_setup(shape, mask_shape, seed=11)
    xs = x.to(sdt)

    def mk(dev, corr):
        if cfg:
            c = torch.ones(shape[0], device=dev)
            fn = D.model_wrapper(lambda xx, t, cc: xx * (0.4 + 0.1 * cc.reshape(-1, 1, 1, 1)).to(xx.dtype), ns,
                                 guidance_type="classifier-free", condition=c, unconditional_condition=c * 0, guidance_scale=2.0)
        else:
            fn = D.model_wrapper(lambda xx, t: xx * 0.5, ns)
        return D.DPM_Solver(fn, ns, correcting_xt_fn=corr, state_dtype=sdt)

    helper = mk(DEV, None)
    closure = lambda xt, t, step: (xt * mask.to(sdt) + (1 - mask.to(sdt)) * helper.add_noise(
        x0.to(sdt), t.reshape(1), noise=noise.to(sdt).unsqueeze(0))).to(sdt)
    spy = LaunchSpy(monkeypatch)
    got, gi = mk(DEV, D.MaskBlend(ns, mask, x0=x0, noise=noise)).sample(xs, steps=6, order=2, return_intermediate=True)
    first = spy.calls[0]
    assert first["form"] == L.FORM_LIN1 and (first["flags"] & L.F_BLEND)
    if sdt is torch.float32:          # the closure's half-precision products round differently; fp32 is exact
        want, wi = mk(DEV, closure).sample(xs, steps=6, order=2, return_intermediate=True)
        assert torch.equal(got, want) and all(torch.equal(a, b) for a, b in zip(gi, wi))
    with monkeypatch.context() as m:
        install_cpu_double(m, S, D)
        dbl, di = mk("cpu", D.MaskBlend(ns, mask.cpu(), x0=x0.cpu(), noise=noise.cpu())).sample(
            xs.cpu(), steps=6, order=2, return_intermediate=True)
    assert got.dtype == sdt and torch.equal(got.cpu(), dbl)
    assert all(torch.equal(a.cpu(), b) for a, b in zip(gi, di))


def test_maskblend_against_reference_callback_goldens(golden):
    """goldens produced by the real reference with the closure xt*mask + (1-mask)*(0.25*step) as correcting_xt_fn"""
    case = C.E2E_BY_NAME["cfg1_small"]
    ns = make_schedule("sd")
    x = tt(C.x_T_for(case), DEV)
    mask = torch.from_numpy(golden.get("callbacks", "cb/mask")).to(DEV)
    levels = [torch.full(x.shape, 0.25 * step, device=DEV) for step in range(12)]
    fn = D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns)
    for method, order, steps in [("multistep", 2, 8), ("singlestep", 3, 8)]:
        dpm = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, intermediates=levels))
        xf, inter = dpm.sample(x, steps=steps, order=order, method=method, denoise_to_zero=True, return_intermediate=True)
        pre = "cb/xt/%s/" % method
        assert rel_err(xf.cpu().numpy(), golden.get("callbacks", pre + "final")) < TOL
        ri = golden.get("callbacks", pre + "intermediates")
        assert len(inter) == ri.shape[0]
        for i, v in enumerate(inter):
            assert rel_err(v.cpu().numpy(), ri[i]) < TOL


def test_maskblend_fresh_noise_follows_the_same_random_stream():
    shape = (2, 4, 16, 16)
    ns = make_schedule("sd")
    x, x0, _, mask = _blend_setup(shape, (16, 16))
    fn = D.model_wrapper(lambda xx, t: xx * 0.5, ns)
    helper = D.DPM_Solver(fn, ns)
    closure = lambda xt, t, step: xt * mask + (1 - mask) * helper.add_noise(x0, t.reshape(1))
    torch.manual_seed(7)
    want = D.DPM_Solver(fn, ns, correcting_xt_fn=closure).sample(x, steps=10, t_start=0.6)
    torch.manual_seed(7)
    got = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, x0=x0)).sample(x, steps=10, t_start=0.6)
    assert torch.equal(got, want)


def test_maskblend_deterministic_intermediates_and_time_fn():
    """the notebook's 'deterministic' variant: encode with inverse(), then blend reversed intermediates back in"""
    shape = (2, 4, 16, 16)
    ns = make_schedule("sd")
    x, x0, noise, mask = _blend_setup(shape, (16, 16), seed=5)
    fn = D.model_wrapper(lambda xx, t: xx * 0.5, ns)
    dpm = D.DPM_Solver(fn, ns)
    enc, inter = dpm.inverse(x0, steps=10, t_end=0.6, lower_order_final=False, return_intermediate=True)
    inter = list(reversed(inter))
    closure = lambda xt, t, step: xt * mask + (1 - mask) * inter[step]
    kw = dict(steps=10, t_start=0.6, lower_order_final=False)
    want = D.DPM_Solver(fn, ns, correcting_xt_fn=closure).sample(enc, **kw)
    got = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, intermediates=inter)).sample(enc, **kw)
    assert torch.equal(got, want)
    # time_fn: the noise level is looked up at a remapped time
    tf = lambda t: 0.5 * t + 0.1
    helper = D.DPM_Solver(fn, ns)
    closure2 = lambda xt, t, step: xt * mask + (1 - mask) * helper.add_noise(
        x0, torch.tensor([tf(float(t))], device=DEV), noise=noise.unsqueeze(0))
    want2 = D.DPM_Solver(fn, ns, correcting_xt_fn=closure2).sample(x, steps=6)
    got2 = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, x0=x0, noise=noise, time_fn=tf)).sample(x, steps=6)
    assert torch.equal(got2, want2)


def test_maskblend_with_cfg_thresholding_and_half_state():
    shape = (4, 3, 32, 32)
    ns = make_schedule("ddpm")
    x, x0, noise, mask = _blend_setup(shape, (32, 32), seed=9)
    cond = torch.arange(4, device=DEV, dtype=torch.float32)
    fn = D.model_wrapper(lambda xx, t, c: xx * (c * 0.1 + 1.0).reshape(-1, 1, 1, 1), ns, guidance_type="classifier-free",
                         condition=cond, unconditional_condition=torch.zeros_like(cond), guidance_scale=3.0)
    helper = D.DPM_Solver(fn, ns)
    closure = lambda xt, t, step: xt * mask + (1 - mask) * helper.add_noise(x0, t.reshape(1), noise=noise.unsqueeze(0))
    kw = dict(correcting_x0_fn="dynamic_thresholding")
    want = D.DPM_Solver(fn, ns, correcting_xt_fn=closure, **kw).sample(x, steps=8)
    got = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, x0=x0, noise=noise), **kw).sample(x, steps=8)
    assert torch.equal(got, want)
    # fp16 state: same values up to the half-precision rounding of the intermediate products
    gh = D.DPM_Solver(fn, ns, correcting_xt_fn=D.MaskBlend(ns, mask, x0=x0, noise=noise), state_dtype=torch.float16,
                      **kw).sample(x.half(), steps=8)
    assert gh.dtype == torch.float16
    assert rel_err(gh.float().cpu().numpy(), want.cpu().numpy()) < 4e-3


# ------------------------------------------------------------------------------------------------
# hipGraph capture
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cfg1_small", "cfg_ms2", "cfg3_dpmsolver", "cfg5_thresh_small"])
def test_captured_sample_equals_eager(name):
    case = C.E2E_BY_NAME[name]
    dpm = build_solver(case, DEV)
    x = tt(C.x_T_for(case), DEV)
    kw = sample_kwargs(case, False)
    want = dpm.sample(x, **kw)
    g = dpm.capture(x, **kw)
    assert torch.equal(g(x), want)
    x2 = x * 0.5 + 0.25
    want2 = dpm.sample(x2, **kw)
    assert torch.equal(g(x2), want2)          # replay on new input
    xo, _ = TO.run_oracle_case(case)
    assert rel_err(g(x).cpu().numpy(), xo) < TOL


@pytest.mark.parametrize("name", ["cfg1_small", "cfg_ms2", "cfg5_thresh_small"])
def test_auto_capture_replays_after_n_identical_calls(name):
    """DPM_Solver.auto_capture = N (opt-in): N eager calls, then the call is captured and later calls replay the graph --
    same bits as the eager loop, a fresh tensor per call, other shapes / arguments keep running eagerly; Python callbacks
    and return_intermediate are never captured."""
    case = C.E2E_BY_NAME[name]
    dpm = build_solver(case, DEV)
    x = tt(C.x_T_for(case), DEV)
    kw = sample_kwargs(case, False)
    want = dpm.sample(x, **kw)
    x2 = x * 0.5 + 0.25
    want2 = dpm.sample(x2, **kw)
    dpm.auto_capture = 2
    outs = [dpm.sample(x, **kw) for _ in range(2)]
    assert all(e[1] is None for e in dpm._auto.values()) and len(dpm._auto) == 1       # still eager
    outs += [dpm.sample(x, **kw) for _ in range(3)]                                     # captured at the third call
    assert [e[1] is not None for e in dpm._auto.values()] == [True]
    assert all(torch.equal(o, want) for o in outs)
    assert len({o.data_ptr() for o in outs}) == len(outs)                               # never the graph's static buffer
    assert torch.equal(dpm.sample(x2, **kw), want2)                                     # replay on new values
    kw3 = dict(kw, steps=kw.get("steps", 20) - 1)
    eager3 = dpm.sample(x, **kw3)                                                       # other arguments: a new, eager entry
    assert len(dpm._auto) == 2 and torch.isfinite(eager3).all()
    got_i, inter = dpm.sample(x, **dict(kw, return_intermediate=True))                  # never captured
    assert torch.equal(got_i, want) and len(inter) > 1
    dpm.auto_capture = 0
    assert torch.equal(dpm.sample(x, **kw), want)


def test_capture_rejects_the_host_side_adaptive_loop():
    """only the host-side control loop (adaptive_on_device = False: one synchronisation per iteration) cannot be
    captured; the device-side controller can (tests/test_gpu_parity.py::test_adaptive_captured_into_a_graph)"""
    ns = make_schedule("sd")
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns)
    dpm.adaptive_on_device = False
    with pytest.raises(NotImplementedError, match="adaptive"):
        dpm.capture(torch.zeros(2, 4, 8, 8, device=DEV), method="adaptive")


def test_auto_capture_stays_eager_where_the_adaptive_solver_takes_its_host_loop():
    """auto_capture (and capture()) use the SAME predicate dpm_solver_adaptive uses to choose the device
    controller.  With dynamic thresholding, a double state or a half x on a continuous schedule the adaptive solver runs its
    host loop (one .item() per iteration): such calls are never recorded -- they run eagerly call after call, no exception."""
    ns = make_schedule("sd")
    x = torch.randn((2, 3, 16, 16), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.3, ns), ns, correcting_x0_fn="dynamic_thresholding")
    kw = dict(method="adaptive", order=2, atol=0.05, rtol=0.1)
    want = dpm.sample(x, **kw)
    dpm.auto_capture = 1
    outs = [dpm.sample(x, **kw) for _ in range(4)]
    assert all(torch.equal(o, want) for o in outs) and not dpm._auto          # never entered the capture bookkeeping
    with pytest.raises(NotImplementedError, match="adaptive"):
        dpm.capture(x, **kw)
    d64 = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.3, ns), ns)
    d64.auto_capture = 1
    xd = x.double()
    w64 = d64.sample(xd, **kw)
    assert all(torch.equal(d64.sample(xd, **kw), w64) for _ in range(3)) and not d64._auto
    # ... while the device-side controller IS captured
    dev = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.3, ns), ns)
    dev.auto_capture = 1
    wdev = dev.sample(x, **kw)
    for _ in range(3):
        assert rel_err(dev.sample(x, **kw).cpu().numpy(), wdev.cpu().numpy()) < TOL
    assert [e[1] is not None and e[1] is not False for e in dev._auto.values()] == [True]


def test_auto_capture_key_follows_solver_and_wrapper_settings():
    """a captured call bakes in the solver's settings; changing one between calls (thresholding ratio,
    guidance scale, algorithm type, state dtype) must miss the cache and be computed with the NEW setting -- as the eager
    path's plan cache does -- not replay the old graph"""
    ns = make_schedule("ddpm")
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((4, 3, 32, 32), device=DEV, generator=g) * 1.5
    eps = torch.randn((8, 3, 32, 32), device=DEV, generator=g)
    cond = torch.ones(4, device=DEV)
    fn = D.model_wrapper(lambda xx, t, c: eps, ns, guidance_type="classifier-free", condition=cond, unconditional_condition=cond * 0,
                         guidance_scale=2.0)
    dpm = D.DPM_Solver(fn, ns, correcting_x0_fn="dynamic_thresholding")
    kw = dict(steps=8, order=2)

    def eager():
        saved, dpm.auto_capture = dpm.auto_capture, 0
        try:
            return dpm.sample(x, **kw)
        finally:
            dpm.auto_capture = saved
    dpm.auto_capture = 1
    w0 = eager()
    for _ in range(3):
        assert torch.equal(dpm.sample(x, **kw), w0)
    n0 = len(dpm._auto)
    dpm.dynamic_thresholding_ratio = 0.9
    w1 = eager()
    assert not torch.equal(w1, w0)
    assert torch.equal(dpm.sample(x, **kw), w1) and len(dpm._auto) > n0          # a new entry, the new ratio
    fn.guidance_scale = 5.0
    w2 = eager()
    assert not torch.equal(w2, w1)
    for _ in range(3):
        assert torch.equal(dpm.sample(x, **kw), w2)
    dpm.dynamic_thresholding_ratio, fn.guidance_scale = 0.995, 2.0
    assert torch.equal(dpm.sample(x, **kw), w0)                                  # back to the first settings


def test_auto_capture_falls_back_when_the_network_cannot_be_captured():
    """a network that synchronises with the host cannot be recorded: the failed capture is remembered, the call is served
    eagerly from then on (one warning), results unchanged"""
    ns = make_schedule("sd")
    x = torch.randn((2, 4, 16, 16), device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))

    def net(xx, t):
        float(t[0].item())                    # a device -> host synchronisation: illegal under stream capture
        return xx * 0.25
    dpm = D.DPM_Solver(D.model_wrapper(net, ns), ns)
    want = dpm.sample(x, steps=6, order=2)
    dpm.auto_capture = 1
    with pytest.warns(UserWarning, match="auto_capture"):
        outs = [dpm.sample(x, steps=6, order=2) for _ in range(3)]
    assert all(torch.equal(o, want) for o in outs)
    assert [e[1] for e in dpm._auto.values()] == [False]
    torch.cuda.synchronize()
    assert torch.equal(dpm.sample(x, steps=6, order=2), want)


@pytest.mark.parametrize("x_dtype,t_dtype,ns_dtype", [(torch.float32, torch.float64, torch.float32),
                                                      (torch.float32, torch.float32, torch.float64),
                                                      (torch.float64, torch.float64, torch.float64)])
def test_add_noise_with_double_scalars_on_the_gpu(x_dtype, t_dtype, ns_dtype):
    """dpm_add_noise_launch_f64 (version 201): a double t or double tables make the result float64 with the schedule
    evaluated in double (ref :1012-1030 under torch's type promotion) -- against the same expression in torch double"""
    betas = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64)
    ns = D.NoiseScheduleVP("discrete", betas=betas, dtype=ns_dtype)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn((3, 4, 8, 8), device=DEV, generator=g, dtype=torch.float32).to(x_dtype)
    t = torch.tensor([0.9, 0.31, 0.02], dtype=t_dtype, device=DEV)
    noise = torch.randn((3, 3, 4, 8, 8), device=DEV, generator=g, dtype=torch.float32).to(x_dtype)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, tt_: xx, ns), ns)
    got = dpm.add_noise(x, t, noise=noise)
    assert got.dtype == torch.float64 and got.shape == (3, 3, 4, 8, 8)
    a = ns.marginal_alpha(t.cpu()).double().to(DEV).reshape(3, 1, 1, 1, 1)
    s_ = ns.marginal_std(t.cpu()).double().to(DEV).reshape(3, 1, 1, 1, 1)
    assert ns.marginal_alpha(t.cpu()).dtype == torch.float64
    want = a * x.double().unsqueeze(0) + s_ * noise.double()
    assert float((got - want).abs().max()) <= 1e-15 * float(want.abs().max()) + 1e-300
    assert torch.equal(dpm.add_noise(x, t[:1], noise=noise[:1]), got[0])


def test_multistep_order_four_runs_where_the_reference_runs_it():
    """sample(order=4, steps=5, lower_order_final=True): step orders 1, 2, 3, 2, 1 (ref :1185-1201) -- the same trajectory as
    order=3 at these sizes -- on the fast path and in the general loop, against the oracle; steps=7 raises the reference's error"""
    from oracle import dpm_oracle as O
    ac = np.cumprod(1.0 - np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2).astype(F32)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(ac))
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 4, 16, 16)).astype(F32)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns)
    got = dpm.sample(torch.from_numpy(x).to(DEV), steps=5, order=4)
    osch = O.Schedule.from_alphas_cumprod(ac)
    want = O.Solver(O.wrap_model(lambda xx, t: xx * F32(0.5), osch), osch).sample(x, steps=5, order=4)
    assert rel_err(got.cpu().numpy(), want) < TOL
    assert torch.equal(dpm.sample(torch.from_numpy(x).to(DEV), steps=5, order=3), got)
    gi, inter = dpm.sample(torch.from_numpy(x).to(DEV), steps=5, order=4, return_intermediate=True)
    assert torch.equal(gi, got) and len(inter) == 6
    with pytest.raises(ValueError, match="Solver order must be 1 or 2 or 3, got 4"):
        dpm.sample(torch.from_numpy(x).to(DEV), steps=7, order=4)


@pytest.mark.parametrize("sdt", [torch.float32, torch.float16])
def test_native_graph_equals_native_loop(sdt):
    """C ABI: dpm_graph_create / dpm_graph_launch replay the 20 launches of dpm_plan_run"""
    ns = make_schedule("sd")
    shape = (16, 4, 64, 64)
    rng = np.random.default_rng(13)
    x = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV).to(sdt)
    eps = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV).to(sdt)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: eps, ns), ns, state_dtype=sdt)
    want = dpm.sample(x, steps=20, order=2)
    plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    xb = [x.clone()] + [torch.empty_like(x) for _ in range(3)]
    hb = [torch.empty_like(x) for _ in range(3)]
    rb = L.RunBuffers()
    for i in range(4):
        rb.xbuf[i] = xb[i].data_ptr()
    for i in range(3):
        rb.hist[i] = hb[i].data_ptr()
    rb.e0 = eps.data_ptr()
    dt = {torch.float32: L.DTYPE_F32, torch.float16: L.DTYPE_F16}[sdt]
    rb.n, rb.batch, rb.state_dtype, rb.eps_dtype = x.numel(), shape[0], dt, dt
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = C_.c_void_p()
    with pytest.raises(ValueError, match="non-null stream"):
        L.check(L.lib.dpm_graph_create(plan.handle, C_.byref(rb), None, None, None, C_.byref(g)))
    L.check(L.lib.dpm_graph_create(plan.handle, C_.byref(rb), None, None, C_.c_void_p(side.cuda_stream), C_.byref(g)))
    assert L.lib.dpm_graph_num_nodes(g) == len(plan.stages)
    for _ in range(3):
        L.check(L.lib.dpm_graph_launch(g, C_.c_void_p(side.cuda_stream)))
    side.synchronize()
    assert torch.equal(xb[L.lib.dpm_graph_result(g)], want)
    assert torch.equal(xb[0], x)
    L.lib.dpm_graph_destroy(g)


def test_native_loop_with_cfg_callback_and_duplicated_state():
    """C ABI for non-Python hosts: dpm_plan_run with a model callback under classifier-free guidance; dup_state makes
    every stage write the callback's [2B,...] input itself.  Must equal the Python loop bit for bit."""
    case = C.E2E_BY_NAME["cfg3_pp"]
    ns = make_schedule(case["schedule"])
    cond, uncond = C.cond_for(case)
    c_in = torch.cat([tt(uncond, DEV), tt(cond, DEV)])
    dpm = build_solver(case, DEV)
    x = tt(C.x_T_for(case), DEV)
    want = dpm.sample(x, **sample_kwargs(case, False))
    plan = dpm._get_plan(method=case["method"], order=case["order"], steps=case["steps"], skip_type=case["skip_type"],
                         solver_type=case["solver_type"], lower_order_final=case["lower_order_final"],
                         denoise_to_zero=case["denoise_to_zero"], t_T=1.0, t_0=1.0 / ns.total_N)
    B = x.shape[0]
    xb = [torch.cat([x, x])] + [torch.empty((2 * B,) + tuple(x.shape[1:]), device=DEV) for _ in range(3)]
    hb = [torch.empty_like(x) for _ in range(3)]
    out2 = torch.empty((2 * B,) + tuple(x.shape[1:]), device=DEV)            # [uncond half | cond half]
    by_ptr = {t.data_ptr(): t for t in xb}
    calls = []

    def model_cb(user, st, xptr, e0, e1, stream):
        xin = by_ptr[xptr]
        assert torch.equal(xin[:B], xin[B:])
        t_in = torch.full((2 * B,), st.contents.t_input, device=DEV)
        out2.copy_(C.model_cond(xin, t_in, c_in))
        calls.append(st.contents.index)
        return 0

    cb = L.MODEL_CB(model_cb)
    rb = L.RunBuffers()
    for i in range(4):
        rb.xbuf[i] = xb[i].data_ptr()
    for i in range(3):
        rb.hist[i] = hb[i].data_ptr()
    rb.e1, rb.e0 = out2[:B].data_ptr(), out2[B:].data_ptr()
    rb.n, rb.batch, rb.state_dtype, rb.eps_dtype = x.numel(), B, L.DTYPE_F32, L.DTYPE_F32
    rb.dup_state = 1
    res = C_.c_int(-1)
    L.check(L.lib.dpm_plan_run(plan.handle, C_.byref(rb), C_.cast(cb, C_.c_void_p), None,
                               C_.c_void_p(torch.cuda.current_stream().cuda_stream), C_.byref(res)))
    torch.cuda.synchronize()
    assert calls == list(range(len(plan.stages)))
    assert torch.equal(xb[res.value][:B], want)


def test_stable_diffusion_adapter_against_reference_goldens(golden):
    """DPMSolverSampler (sampler.py) on the HIP path vs goldens produced by the reference's class: txt2img-style
    sampling, stochastic / deterministic encoding, both DiffEdit variants (closures and fused MaskBlend objects)"""
    import test_host_logic as TH
    TH.sampler_checks(golden, DEV, TOL)


def test_legacy_revision_cosine_schedule_and_unclipped_tables(golden):
    """LegacyNoiseScheduleVP (the revision vendored under examples/score_sde_pytorch) on the HIP path"""
    import test_host_logic as TH
    TH.legacy_checks(golden, DEV)


def test_clustered_thresholding_launched_from_two_streams():
    """Small batches / large samples run the thresholding kernel as clusters that synchronise through spin barriers;
    launches from different streams are chained device-wide so two of them never starve each other's peers."""
    ns = make_schedule("ddpm")
    dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding")
    rng = np.random.default_rng(4)
    xs = [torch.from_numpy((rng.standard_normal(shape) * 2).astype(F32)).to(DEV)
          for shape in [(8, 3, 64, 64), (2, 3, 256, 256)]]
    want = [dpm.dynamic_thresholding_fn(x, None) for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for it in range(40):
        for j, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[j].append(dpm.dynamic_thresholding_fn(xs[j], None))
    torch.cuda.synchronize()
    for j in range(2):
        assert all(torch.equal(o, want[j]) for o in outs[j])


@pytest.mark.parametrize("seed", [21, 22])
def test_random_configurations_against_oracle(seed):
    """seeded random sweep over sample() configurations (method, order, steps, schedule, skip type, solver type,
    algorithm, parameterisation, time range): HIP path vs oracle"""
    import test_host_logic as TH
    for cfg in TH.random_configs(seed, 40):
        got, want = TH.run_random_config(cfg, DEV)
        assert rel_err(got, want) < TOL, cfg


def test_plain_c_host_without_python_or_torch(tmp_path):
    """examples/native_host.c (gcc, links libdpm_hip.so + the system HIP runtime) runs the 2M trajectory through
    dpm_plan_run with a C model callback; its result must equal the Python host's bit for bit on the same inputs."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "native_host")
    if not os.path.exists(exe):          # normally built by __graft_entry__.build(); gcc and ROCm are on the box too
        import __graft_entry__ as G
        G.build_native_example()
    assert os.path.exists(exe)
    out = tmp_path / "native.bin"
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "gfx950" in r.stdout and "20 network calls" in r.stdout, r.stdout
    got = np.fromfile(out, dtype=F32).reshape(8, 4, 64, 64)
    # the C program's inputs: sum of four 24-bit LCG uniforms, seed 12345, x first then eps
    n = 8 * 4 * 64 * 64
    raw = np.empty(2 * n * 4, dtype=np.uint32)
    s = 12345
    for i in range(raw.shape[0]):
        s = (s * 1664525 + 1013904223) & 0xffffffff
        raw[i] = s
    u = (raw >> 8).astype(F32) * F32(1.0 / 16777216.0)
    acc = np.zeros(2 * n, dtype=F32)
    for k in range(4):
        acc = (acc + u[k::4]).astype(F32)
    vals = ((acc - F32(2.0)) * F32(1.7320508)).astype(F32)
    x = torch.from_numpy(vals[:n].reshape(8, 4, 64, 64)).to(DEV)
    eps = torch.from_numpy(vals[n:].reshape(8, 4, 64, 64)).to(DEV)
    b0, b1 = np.sqrt(0.00085), np.sqrt(0.012)                       # the C program's double arithmetic, op for op
    v = b0 + (b1 - b0) * np.arange(1000, dtype=np.float64) / 999.0
    ns = D.NoiseScheduleVP("discrete", betas=torch.from_numpy(v * v))
    want = D.DPM_Solver(D.model_wrapper(lambda xx, t: eps, ns), ns).sample(x, steps=20, order=2)
    np.testing.assert_array_equal(got, want.cpu().numpy())


def test_sharded_sampling_over_rccl_single_rank():
    """dpm_solver_amd.distributed on the real backend (RCCL via torch.distributed 'nccl'), world size 1: the all-gather of
    the finished shards and the MAX all-reduce of the adaptive solver run through RCCL and change nothing"""
    import os
    import torch.distributed as dist
    from dpm_solver_amd import distributed as DD
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        ns = make_schedule("vp_linear")
        mk = lambda: D.DPM_Solver(D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns), ns, algorithm_type="dpmsolver")
        rng = np.random.default_rng(8)
        x = torch.from_numpy(rng.standard_normal((5, 3, 8, 8)).astype(F32)).to(DEV)
        assert torch.equal(DD.sample_sharded(mk(), x, steps=10, order=2), mk().sample(x, steps=10, order=2))
        a = DD.sample_sharded(mk(), x, method="adaptive", order=2, t_end=1e-3)
        b = mk().sample(x, method="adaptive", order=2, t_end=1e-3)
        assert torch.equal(a, b)
        assert DD.rank_seed(3) == 3
    finally:
        dist.destroy_process_group()
        # the communicator's teardown frees device memory: let it happen here, not inside a later test's stream capture
        import gc
        gc.collect()
        torch.cuda.synchronize()


@pytest.mark.parametrize("shape,in_graph", [((2, 3, 64, 64), 0), ((2, 3, 64, 64), 1), ((2, 3, 160, 160), 0)])
def test_captured_sample_with_clustered_thresholding(shape, in_graph):
    """hipGraph capture of a trajectory with dynamic thresholding.  Eagerly a small batch runs as workgroup clusters; under
    capture a sample that fits one workgroup takes the cluster-free shape (a replayed graph is outside the library's
    device-wide chain of clustered launches) unless dpm_launch_opts.cluster_in_graph (DPM_Solver.cluster_in_graph) opts in; larger samples keep their
    clusters.  Replays must keep matching eager runs in every case."""
    case = dict(C.E2E_BY_NAME["cfg5_thresh"], shape=shape, steps=8)
    dpm = build_solver(case, DEV)
    x = tt(C.x_T_for(case), DEV)
    kw = sample_kwargs(case, False)
    want = dpm.sample(x, **kw)
    dpm.cluster_in_graph = bool(in_graph)          # a per-call option of the C ABI: no process-wide switch
    g = dpm.capture(x, **kw)
    for _ in range(3):
        assert torch.equal(g(x), want)
    x2 = x * 0.75
    assert torch.equal(g(x2), dpm.sample(x2, **kw))


def test_guided_diffusion_adapter_against_reference_goldens(golden, monkeypatch):
    """runners/diffusion.py:594-640 on the HIP path: 6-channel network read in place, classifier guidance through
    log_softmax + autograd, thresholding, denoise -- vs goldens from the reference solver with the same wiring"""
    import test_host_logic as TH
    spy = LaunchSpy(monkeypatch)
    TH.guided_checks(golden, DEV, 2 * TOL)
    assert all(c["eps_stride"] == 2 * 3 * 8 * 8 for c in spy.calls)       # the mean half is never copied out


def test_adaptive_sharded_with_an_empty_shard_on_the_gpu():
    """batch < world on the GPU: the rank with the empty shard runs the device-side controller too (no stage launches), so
    every rank issues the same all-reduces; results equal the unsharded run's slices (solver.py adaptive
    path choice).  Two gloo ranks share cuda:0."""
    import os
    import torch.multiprocessing as mp
    from adaptive_shard_worker import worker
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=worker, args=(r, 2, port, 1, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in range(4)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    first = sorted(r for r in res if r[4] is None)
    second = sorted(r for r in res if r[4] is not None)
    # pass 1: every rank was handed the full 1-sample batch -- both took the device path, same number of all-reduces
    assert [r[1] for r in first] == [1, 1] and first[0][2] == first[1][2] > 0
    # pass 2: rank 0 owns the sample, rank 1 an empty shard
    assert [r[3][0] for r in second] == [1, 0] and all(r[4] for r in second), second


# ------------------------------------------------------------------------------------------------
# measurement entry points of the C ABI (round 3): event-bracketed launches without synchronisation, the prefetch kernel
# ------------------------------------------------------------------------------------------------
@pytest.mark.lab
def test_traced_launches_time_the_kernel_and_change_nothing():
    """dpm_stage_launch_traced = dpm_stage_launch with a start / stop event pair attached to the kernel itself; nothing
    synchronises until dpm_trace_read.  Same results as the plain launches, positive durations for the slots used, -1 for
    the others, argument errors for slots outside the trace."""
    ns = make_schedule("sd")
    rng = np.random.default_rng(41)
    x = torch.from_numpy(rng.standard_normal((16, 4, 64, 64)).astype(F32)).to(DEV)
    e = torch.from_numpy(rng.standard_normal((16, 4, 64, 64)).astype(F32)).to(DEV)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: e, ns), ns)
    want = dpm.sample(x, steps=10, order=2)
    trace = C_.c_void_p()
    L.check(L.lib.dpm_trace_create(16, C_.byref(trace)))
    count = [0]
    real = S._stage_launch_raw

    def traced(st, b, stream):
        k = count[0]
        count[0] += 1
        return L.lib.dpm_stage_launch_traced(st, b, stream, trace, k)
    try:
        S._stage_launch_raw = traced
        got = dpm.sample(x, steps=10, order=2)
    finally:
        S._stage_launch_raw = real
    ms = (C_.c_float * 16)()
    L.check(L.lib.dpm_trace_read(trace, C_.c_void_p(torch.cuda.current_stream().cuda_stream), ms, 16))
    v = np.frombuffer(ms, dtype=np.float32)
    assert count[0] == 10 and torch.equal(got, want)
    assert np.all(v[:10] > 0) and np.all(v[:10] < 5.0) and np.all(v[10:] == -1.0), v
    st, b = L.Stage(), L.Buffers()
    assert L.lib.dpm_stage_launch_traced(C_.byref(st), C_.byref(b), None, trace, 16) == L.ERR_ARG
    assert L.lib.dpm_trace_read(trace, None, ms, 16) == L.DPM_OK and np.all(np.frombuffer(ms, dtype=np.float32) == -1.0)
    L.lib.dpm_trace_destroy(trace)
    assert L.lib.dpm_trace_create(0, C_.byref(trace)) == L.ERR_ARG


@pytest.mark.lab
def test_prefetch_launch_reads_and_leaves_the_buffers_alone():
    """dpm_prefetch_launch (the rejected experiment of DESIGN.md 4.3 stays callable): reads up to 8 buffers with either load
    policy, writes nothing, rejects unaligned buffers and more than 8."""
    a = torch.arange(1 << 20, dtype=torch.float32, device=DEV)
    b = torch.ones(12345, dtype=torch.float16, device=DEV)
    ca, cb = a.clone(), b.clone()
    stream = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    for policy in (0, 1):
        ptrs = (C_.c_void_p * 3)(a.data_ptr(), None, b.data_ptr())
        nbytes = (C_.c_int64 * 3)(a.numel() * 4, 0, b.numel() * 2)
        L.check(L.lib.dpm_prefetch_launch(ptrs, nbytes, 3, policy, stream))
    torch.cuda.synchronize()
    assert torch.equal(a, ca) and torch.equal(b, cb)
    ptrs = (C_.c_void_p * 1)(a.data_ptr() + 4)
    nbytes = (C_.c_int64 * 1)(1024)
    assert L.lib.dpm_prefetch_launch(ptrs, nbytes, 1, 0, stream) == L.ERR_ALIGN
    many = (C_.c_void_p * 9)(*[a.data_ptr()] * 9)
    assert L.lib.dpm_prefetch_launch(many, (C_.c_int64 * 9)(*[64] * 9), 9, 0, stream) == L.ERR_ARG
    assert L.lib.dpm_prefetch_launch(None, None, 0, 0, stream) == L.ERR_ARG


def test_score_sde_sampler_against_reference_goldens(golden, capsys):
    """dpm_solver_amd.adapters.score_sde_get_dpm_solver_sampler on the HIP path vs goldens from the ScoreSDE example's own
    sampler (singlestep-3 / logSNR default, denoise, dpmsolver++ multistep, adaptive)"""
    from test_host_logic import run_score_sde_adapter
    assert run_score_sde_adapter(DEV, golden, thresholding_too=True) < TOL


# ------------------------------------------------------------------------------------------------
# the reference's own example call sites, source files unchanged (tools/dropin_examples.py).  The reference tree is not
# part of this repository: the test runs when it travelled to the box as scratch ($DPM_REFERENCE_DIR, tools/ref_scratch.sh)
# ------------------------------------------------------------------------------------------------
def _dropin_tool():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dropin_examples", os.path.join(root, "tools", "dropin_examples.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("site", [0, 1, 2], ids=["stable-diffusion", "score-sde", "guided-diffusion"])
def test_dropin_examples_on_the_gpu(site):
    DE = _dropin_tool()
    ex = DE.reference_examples()
    if ex is None:
        pytest.skip("no reference checkout on this box (DPM_REFERENCE_DIR)")
    name, where, fn = DE.SITES[site]
    row = DE.run_site(name, where, fn, DEV, ex)
    assert row["passed"], {k: row[k] for k in ("max_rel_err", "integers_equal", "network_trace")}


# ------------------------------------------------------------------------------------------------
# channels_last (NHWC) networks: the stage kernels consume their outputs in place
# ------------------------------------------------------------------------------------------------
class _ContiguousSpy:
    """counts the torch `.contiguous()` calls that really copy (a layout-changing kernel launch)"""

    def __init__(self, monkeypatch):
        self.copies = 0
        real = torch.Tensor.contiguous

        def contiguous(t, *a, **k):
            out = real(t, *a, **k)
            if out.data_ptr() != t.data_ptr():
                self.copies += 1
            return out
        monkeypatch.setattr(torch.Tensor, "contiguous", contiguous)


def _layout_net(ns, cfg, fmt, B, eps_dtype=None):
    def net(xx, t, c=None):
        scale = (t * 0.0005 + 0.25).reshape(-1, 1, 1, 1)
        if c is not None:
            scale = scale * (1.0 + 0.1 * c.reshape(-1, 1, 1, 1))
        out = (xx.float() * scale).to(eps_dtype or xx.dtype)
        return out.to(memory_format=fmt)
    if cfg:
        cond = torch.ones(B, device=DEV)
        return D.model_wrapper(net, ns, guidance_type="classifier-free", condition=cond, unconditional_condition=cond * 0,
                               guidance_scale=7.5)
    return D.model_wrapper(net, ns)


@pytest.mark.parametrize("cfg", [False, True])
@pytest.mark.parametrize("thr", [False, True])
@pytest.mark.parametrize("x_nhwc,net_nhwc", [(False, True), (True, True), (True, False)])
@pytest.mark.parametrize("sdt,edt", [(torch.float32, None), (torch.float16, None), (torch.float32, torch.float16)])
def test_channels_last_runs_without_a_copy_per_stage(cfg, thr, x_nhwc, net_nhwc, sdt, edt, monkeypatch):
    """x_T and / or the network in channels_last: bit-identical to the default-layout run, the result in x_T's layout, at
    most two layout conversions per trajectory (x_T in, result out) instead of one per stage; with classifier-free guidance
    the duplicate store feeds the network NHWC halves of one [2B,...] buffer."""
    ns = make_schedule("sd" if not thr else "ddpm")
    shape = (6, 3, 32, 32) if thr else (6, 4, 32, 32)
    B = shape[0]
    x = torch.randn(shape, generator=torch.Generator().manual_seed(7)).to(DEV, sdt)
    kw = dict(correcting_x0_fn="dynamic_thresholding") if thr else {}
    if sdt is not torch.float32:
        kw["state_dtype"] = sdt
    steps = 9
    want = D.DPM_Solver(_layout_net(ns, cfg, torch.contiguous_format, B, edt), ns, **kw).sample(x, steps=steps, order=2)
    xin = x.to(memory_format=torch.channels_last) if x_nhwc else x
    dpm = D.DPM_Solver(_layout_net(ns, cfg, torch.channels_last if net_nhwc else torch.contiguous_format, B, edt), ns, **kw)
    dpm.sample(xin, steps=steps, order=2)                      # builds the launch records
    spy = _ContiguousSpy(monkeypatch)
    got = dpm.sample(xin, steps=steps, order=2)
    assert torch.equal(got, want)
    assert got.is_contiguous(memory_format=torch.channels_last) == x_nhwc
    assert spy.copies == (0 if x_nhwc == net_nhwc else 2), spy.copies
    outs = dpm.sample_requests([xin, xin * 0.5, xin + 0.125], steps=steps, order=2)
    assert torch.equal(outs[0], want)
    # the general loop (SD's sampler always asks for the intermediates)
    got2, inter = dpm.sample(xin, steps=steps, order=2, return_intermediate=True)
    assert torch.equal(got2, want) and len(inter) == steps + 1


def test_channels_last_conv_network_in_both_layouts():
    """a real convolution (MIOpen) as the network, weights and activations in channels_last vs the default layout: the two
    trajectories agree to the convolution's own rounding (different MIOpen kernels), and the NHWC one makes no copy per stage"""
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(4, 4, 3, padding=1).to(DEV)
    ns = make_schedule("sd")
    x = torch.randn((4, 4, 32, 32), generator=torch.Generator().manual_seed(8)).to(DEV)
    outs = {}
    for name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        net = conv.to(memory_format=fmt)
        seen = []

        def model(xx, t, net=net, seen=seen):
            y = net(xx) * 0.1
            seen.append(y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous())
            return y
        with torch.no_grad():
            outs[name] = D.DPM_Solver(D.model_wrapper(model, ns), ns).sample(x.to(memory_format=fmt), steps=10, order=2)
        if name == "nhwc":
            assert all(seen)
    assert rel_err(outs["nhwc"].cpu().numpy(), outs["nchw"].cpu().numpy()) < 1e-4
    assert outs["nhwc"].is_contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------------------------------------
# the escape hatch of the assembly stores: the library built with -DDPM_STORE_WRITE_THROUGH=0 (compiler-generated stores)
# must give the same bits.  __graft_entry__.build() keeps that build under tools/_variants/nowt.
# ------------------------------------------------------------------------------------------------
def test_escape_hatch_build_without_write_through_stores():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tools", "_variants", "nowt", "libdpm_hip.so")
    if not os.path.exists(lib):
        pytest.skip("no -DDPM_STORE_WRITE_THROUGH=0 build next to the library (__graft_entry__.build() makes it)")
    assert os.path.getmtime(lib) >= os.path.getmtime(L.LIB_PATH) - 3600, "the escape-hatch build is older than the library"
    env = dict(os.environ, DPM_SOLVER_AMD_LIB=lib)
    probe = subprocess.run([sys.executable, "-c", "import dpm_solver_amd._lib as L; print(L.LIB_PATH)"], env=env, cwd=root,
                           stdout=subprocess.PIPE, text=True, check=True)
    assert probe.stdout.strip() == lib
    # bit-exact tests: full-size cfg2 / cfg5 / cfg3 against the kernel double, the rounding of the packed half stores,
    # fused launches against single ones
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_multi.py", "-m", "gpu", "-q", "-x",
                        "-k", "cfg2 or cfg5 or cfg3 or half_precision_stores or fused"],
                       env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], r.stdout[-500:]


def test_lab_suite_on_the_lab_build():
    """The tests that need what the product library does not have -- forced cluster faults, the general / predicted route
    switches, forced workgroup sizes, event-bracketed launches, the side-stream helpers -- run on the LAB build of the same
    sources (tools/_variants/lab/libdpm_lab.so, -DDPM_LAB=1) in a subprocess."""
    from conftest import run_lab_suite
    assert run_lab_suite("lab and gpu") >= 30


def test_quickstart_example_runs():
    """examples/quickstart.py end to end (reference calling convention, capture, requests in flight, channels_last, inpainting)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "quickstart.py")], cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    for needle in ("sample:", "captured", "sample_requests: 8 requests", "channels_last: result in channels_last = True", "inpaint:"):
        assert needle in r.stdout, r.stdout[-2000:]


# ------------------------------------------------------------------------------------------------
# round 6, found by tools/fuzz_gpu.py (the drop-in fuzz's cases, engine on the GPU vs the engine's host code on the numpy double)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("edt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("method,order", [("multistep", 2), ("singlestep", 3), ("multistep", 1)])
def test_classifier_gradient_stays_fp32_next_to_a_half_precision_network(edt, method, order, monkeypatch):
    """Classifier guidance with a network that answers in half precision (fp32 state: guided diffusion under autocast).  The
    reference's `noise - scale * sigma_t * cond_grad` (ref :320-321) promotes to fp32 -- the gradient is an fp32 tensor and is
    never rounded to the network's dtype.  The stage kernel reads the output and the gradient in ONE dtype: the launch must
    widen the output (exact), not narrow the gradient (1e-4 of the state, the bug this test pins).  Bit-equal to the double,
    which the CPU differential holds to the live reference for exactly this combination (tools/fuzz_dropin.py: net_dt)."""
    from kernel_double import install_cpu_double
    ns = make_schedule("ddpm")
    x = torch.from_numpy(np.random.default_rng(17).standard_normal((3, 3, 16, 16)).astype(np.float32))

    def mk(dev):
        cond = torch.arange(1, 4, dtype=torch.float32, device=dev) * 0.5
        net = lambda xx, t, c=None: (xx * (t.reshape(-1, 1, 1, 1) * 0.0005 + 0.25)).to(edt)
        clf = lambda xx, t, c: -0.5 * (xx.reshape(xx.shape[0], -1) ** 2).sum(dim=1) * 0.013
        fn = D.model_wrapper(net, ns, guidance_type="classifier", condition=cond, guidance_scale=7.5, classifier_fn=clf)
        return D.DPM_Solver(fn, ns, algorithm_type="dpmsolver++")
    kw = dict(steps=6, order=order, method=method, return_intermediate=True)
    got, gi = mk(DEV).sample(x.to(DEV), **kw)
    with monkeypatch.context() as m:
        install_cpu_double(m, S, D)
        want, wi = mk("cpu").sample(x, **kw)
    assert got.dtype is torch.float32 and torch.equal(got.cpu(), want)
    assert len(gi) == len(wi) and all(torch.equal(a.cpu(), b) for a, b in zip(gi, wi))
    # ... and the fast path (no intermediates: prebuilt launch records, launch_list._bind_outputs) binds the same way
    assert torch.equal(mk(DEV).sample(x.to(DEV), steps=6, order=order, method=method).cpu(), want)


def test_drop_in_fuzz_slice_on_the_gpu(monkeypatch, capsys):
    """300 random cases of tools/fuzz_gpu.py (the generator of the CPU drop-in fuzz plus larger shapes and networks that
    answer in another dtype): the engine on the GPU against the engine's host code on the numpy double -- raised or returned,
    exception, dtype, shape, network-call trace, values (fp32 and double results bit-identical up to the adaptive solver's
    accept / reject decisions).  The CPU suite holds the double to the live reference over the same kind of cases."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gpu as FG
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu.py", "--cases", "300", "--seed", "7"])
    undo = []

    class MP:
        def setattr(self, o, n, v):
            undo.append((o, n, getattr(o, n)))
            setattr(o, n, v)
    monkeypatch.setattr(FG, "_MP", MP)
    try:
        n_bad = FG.main()
    finally:
        for o, n, v in reversed(undo):
            setattr(o, n, v)
    out = capsys.readouterr().out
    assert n_bad == 0, out[-3000:]
    assert '"cases": 300' in out


@pytest.mark.parametrize("hdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("order", [2, 3])
def test_adaptive_host_loop_with_a_half_state_on_a_continuous_schedule(hdt, order, monkeypatch, capsys):
    """A half x_T on a 'linear' schedule takes the reference's host loop (the state dtype is only known after the first
    network output): its lower-order estimate is still a half tensor, the higher-order one already fp32 (the inner node
    promotes it, ref :161).  The error-norm kernel reads its three operands in one dtype -- the launch must widen them, not
    reinterpret the fp32 tensor's bytes as half (an infinite / NaN estimate and a loop that never ends: found by
    tools/fuzz_gpu.py).  Same accept / reject sequence and result as the host code on the numpy double."""
    from kernel_double import install_cpu_double
    ns = make_schedule("vp_linear")
    x = torch.from_numpy(np.random.default_rng(23).standard_normal((4, 3, 8, 8)).astype(np.float32)).to(hdt)
    net = lambda xx, t: (xx.float() * (t.float().reshape(-1, 1, 1, 1) * 0.0005 + 0.25)).to(xx.dtype)

    def mk():
        dpm = D.DPM_Solver(D.model_wrapper(net, ns), ns, algorithm_type="dpmsolver")
        dpm.adaptive_on_device = False
        return dpm
    kw = dict(method="adaptive", order=order, atol=0.05, rtol=0.1, solver_type="taylor")
    got = mk().sample(x.to(DEV), **kw)
    nfe_gpu = capsys.readouterr().out
    with monkeypatch.context() as m:
        install_cpu_double(m, S, D)
        want = mk().sample(x, **kw)
    nfe_cpu = capsys.readouterr().out
    assert nfe_gpu == nfe_cpu and "adaptive solver nfe" in nfe_gpu
    assert got.dtype is want.dtype and bool(torch.isfinite(got).all())
    assert rel_err(got.float().cpu().numpy(), want.float().numpy()) < 1e-4


def test_adaptive_host_loop_raises_on_a_nan_estimate_instead_of_spinning():
    ns = make_schedule("vp_linear")
    x = torch.randn(2, 3, 4, 4, device=DEV)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * float("nan"), ns), ns)
    dpm.adaptive_on_device = False
    with pytest.raises(FloatingPointError, match="error estimate is NaN"):
        dpm.sample(x, method="adaptive", order=2)


@pytest.mark.parametrize("order", [2, 3])
def test_adaptive_host_loop_behind_a_channels_last_network(order, monkeypatch, capsys):
    """The host-side adaptive loop with a network that answers in channels_last: its states are in the network's layout, and
    the error-norm launch makes dense default-order copies of them.  Those copies must outlive the launch call -- taken as
    `_ptr(t.contiguous())` the temporary was freed at once and the next copy landed in the same block: two operands aliasing,
    a zero estimate, every step accepted (found by tools/fuzz_gpu_api.py: 18 network calls where the double makes 81)."""
    from kernel_double import install_cpu_double
    ns = make_schedule("cosine1000")
    x = torch.from_numpy(np.random.default_rng(29).standard_normal((3, 3, 32, 32)).astype(np.float32))

    def mk(nhwc):
        def net(xx, t):
            out = xx * (t.reshape(-1, 1, 1, 1) * 0.0005 + 0.25)
            return out.contiguous(memory_format=torch.channels_last) if nhwc else out
        dpm = D.DPM_Solver(D.model_wrapper(net, ns, model_type="score"), ns, algorithm_type="dpmsolver")
        dpm.adaptive_on_device = False
        return dpm
    kw = dict(method="adaptive", order=order, atol=0.05, rtol=0.1, solver_type="taylor", t_end=0.01)
    got = mk(True).sample(x.to(DEV), **kw)
    nfe_nhwc = capsys.readouterr().out
    plain = mk(False).sample(x.to(DEV), **kw)
    nfe_plain = capsys.readouterr().out
    with monkeypatch.context() as m:
        install_cpu_double(m, S, D)
        want = mk(False).sample(x, **kw)
    nfe_cpu = capsys.readouterr().out
    assert nfe_nhwc == nfe_plain == nfe_cpu and "adaptive solver nfe" in nfe_cpu
    assert torch.equal(got, plain) and rel_err(got.cpu().numpy(), want.numpy()) < 1e-5


def test_drop_in_fuzz_slice_of_the_extensions_on_the_gpu(monkeypatch, capsys):
    """240 random cases of tools/fuzz_gpu_api.py: sample_requests, capture, auto_capture, a channels_last network, a non-default
    stream, explicit half states, MaskBlend and the device-side adaptive controller on the GPU against plain sample() of the
    engine's host code on the numpy double -- bit-identical results (6000 cases recorded: profiles/r06_fuzz_gpu_api.json)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gpu as FG
    import fuzz_gpu_api as FA
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu_api.py", "--cases", "240", "--seed", "9"])
    undo = []

    class MP:
        def setattr(self, o, n, v):
            undo.append((o, n, getattr(o, n)))
            setattr(o, n, v)
    monkeypatch.setattr(FG, "_MP", MP)
    try:
        n_bad = FA.main()
    finally:
        for o, n, v in reversed(undo):
            setattr(o, n, v)
    out = capsys.readouterr().out
    assert n_bad == 0, out[-3000:]
    assert '"cases": 240' in out


def test_fuzz_slice_of_the_native_sample_loop(monkeypatch, capsys):
    """300 random plans of tools/fuzz_gpu_capi.py: the C ABI's native loop -- dpm_plan_run with an enqueue-only model callback,
    dpm_plan_run_multi over frozen outputs, dpm_graph_create / dpm_graph_launch -- against DPM_Solver.sample() on the same GPU,
    bit for bit: every method, order, skip / solver type, parameterisation, classifier-free guidance with and without
    dup_state, thresholding, fp32 / fp16 / bf16 states (10 000 cases recorded: profiles/r06_fuzz_gpu_capi.json)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gpu_capi as FC
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu_capi.py", "--cases", "300", "--seed", "5"])
    n_bad = FC.main()
    out = capsys.readouterr().out
    assert n_bad == 0, out[-3000:]
    assert '"cases": 300' in out


def test_fuzz_slice_of_single_stage_launches(monkeypatch, capsys):
    """500 random cases of tools/fuzz_gpu_kernel.py --extreme: ONE dpm_stage_launch per case -- random stage record (form,
    eps -> x0, parameterisation, guidance, thresholding, random coefficients), geometry around every tiling boundary, all six
    dtype pairs, unaligned views, channel-sliced outputs, NHWC, the duplicated store; one case in six with magnitudes 1e-45 ..
    1e38, inf and NaN -- against the numpy double from the same inputs, bit for bit (12 000 recorded:
    profiles/r06_fuzz_gpu_kernel.json; found the v_fma_mixlo_f16 fold of the fp16 classifier-free blend)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gpu_kernel as FK
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu_kernel.py", "--cases", "500", "--seed", "17", "--extreme"])
    n_bad = FK.main()
    out = capsys.readouterr().out
    assert n_bad == 0, out[-3000:]
    assert '"cases": 500' in out


def test_fuzz_slice_of_the_public_methods_on_the_gpu(monkeypatch, capsys):
    """600 random calls of tools/fuzz_gpu_methods.py (the generator of fuzz_dropin.py --mode methods): the per-update methods,
    model evaluations, add_noise, time grids, thresholding, the schedule's functions and interpolate_fn on the GPU against the
    engine's host code on the numpy double -- same exception or bit-identical tensors (12 000 recorded:
    profiles/r06_fuzz_gpu_methods.json)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gpu_methods as FM
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu_methods.py", "--cases", "600", "--seed", "13"])
    undo = []

    class MP:
        def setattr(self, o, n, v):
            undo.append((o, n, getattr(o, n)))
            setattr(o, n, v)
    monkeypatch.setattr(FM, "_MP", MP)
    monkeypatch.setattr(FM.FD, "WIDE_NET", True)
    try:
        n_bad = FM.main()
    finally:
        for o, n, v in reversed(undo):
            setattr(o, n, v)
    out = capsys.readouterr().out
    assert n_bad == 0, out[-3000:]
    assert '"calls": 600' in out
