"""Every stage-kernel family the library ships is launched by the GPU suite, and agrees bit for bit with the one-element-per-
lane kernel (same arithmetic, no 16-byte accesses): the cross product of algorithm x method / order / solver type x guidance x
network parameterisation x KExt extension (strided 6-channel network output, mask blend) on a small state, once with an
aligned network output (the vector kernels: compile-time or run-time prologue, separate evaluation state, duplicate store,
device-resident coefficients for the adaptive solver) and once with the same values in a view that is 4 bytes off (the
scalar kernels).  profiles/r05_kernels_launched_suite.md is the rocprofv3 name set of the suite.

The values themselves are pinned elsewhere (goldens, oracle, differential tests); this file pins the families to each other."""
import numpy as np
import pytest
import torch

import dpm_solver_amd as D
from engine_cases import make_schedule

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32 = np.float32
SHAPE = (4, 3, 16, 16)

METHODS = [("multistep", 1, "dpmsolver"), ("multistep", 2, "dpmsolver"), ("multistep", 3, "dpmsolver"),
           ("singlestep", 2, "dpmsolver"), ("singlestep", 3, "dpmsolver"), ("singlestep", 2, "taylor"), ("singlestep", 3, "taylor")]


def _unaligned(t):
    buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
    buf[1:] = t.reshape(-1)
    v = buf[1:].reshape(t.shape)
    assert v.data_ptr() % 16 != 0
    return v


def _solver(ns, algo, guidance, mt, ext, unaligned, x_known, noise, edt=torch.float32, shape=SHAPE, state_dtype=None):
    B = shape[0]
    c = torch.linspace(0.5, 1.5, B, device=DEV)
    six = ext == "6ch"

    def out_of(xx, scale):
        o = (xx * scale).to(edt)
        if six:                                     # learned-variance network: 6 channels, the solver reads the first 3 in place
            full = torch.cat([o, o * 0.25], dim=1)
            if unaligned:
                full = _unaligned(full)
            return full[:, :3]
        return _unaligned(o) if unaligned else o
    kw = {}
    if guidance == "cfg":
        net = lambda xx, t, cc: out_of(xx, 0.4 + 0.1 * cc.reshape(-1, 1, 1, 1))
        kw = dict(guidance_type="classifier-free", condition=c, unconditional_condition=torch.zeros_like(c), guidance_scale=2.5)
    elif guidance == "classifier":
        net = lambda xx, t: out_of(xx, 0.5)
        kw = dict(guidance_type="classifier", condition=c, guidance_scale=1.5,
                  classifier_fn=lambda xx, t, cc: -(xx * xx).sum(dim=(1, 2, 3)) * 0.01 * cc)
    else:
        net = lambda xx, t: out_of(xx, 0.5)
    xt = None
    if ext == "blend":
        mask = (torch.arange(shape[2] * shape[3], device=DEV).reshape(shape[2], shape[3]) % 3 != 0).float()
        xt = D.MaskBlend(ns, mask, x0=x_known, noise=noise)
    return D.DPM_Solver(D.model_wrapper(net, ns, model_type=mt, **kw), ns, algorithm_type=algo, correcting_xt_fn=xt,
                        state_dtype=state_dtype)


@pytest.mark.parametrize("ext", [None, "6ch", "blend"])
@pytest.mark.parametrize("guidance", ["uncond", "cfg", "classifier"])
@pytest.mark.parametrize("algo", ["dpmsolver++", "dpmsolver"])
def test_vector_kernels_equal_the_scalar_kernel(algo, guidance, ext):
    ns = make_schedule("ddpm")
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(SHAPE, device=DEV, generator=g)
    x_known = torch.randn(SHAPE, device=DEV, generator=g)
    noise = torch.randn(SHAPE, device=DEV, generator=g)
    for mt in ("noise", "v"):
        for method, order, st in METHODS:
            for dz in (False, True):
                if dz and not (order == 2 and st == "dpmsolver"):
                    continue                     # denoise_to_zero adds one DENOISE launch: once per method is enough
                kw = dict(steps=6, order=order, method=method, solver_type=st, denoise_to_zero=dz)
                got = _solver(ns, algo, guidance, mt, ext, False, x_known, noise).sample(x, **kw)
                ref = _solver(ns, algo, guidance, mt, ext, True, x_known, noise).sample(x, **kw)
                assert torch.isfinite(got).all()
                assert torch.equal(got, ref), (algo, guidance, ext, mt, kw)


@pytest.mark.parametrize("edt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("guidance", ["uncond", "cfg"])
def test_vector_kernels_equal_the_scalar_kernel_for_a_low_precision_network(guidance, edt):
    """an fp32 state next to a 2-byte network output (autocast), and 2-byte states: the other four dtype pairs"""
    ns = make_schedule("sd")
    g = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn((4, 4, 16, 16), device=DEV, generator=g)
    for state in (None, edt):
        for algo in ("dpmsolver++", "dpmsolver"):
            for method, order, st in METHODS:
                kw = dict(steps=5, order=order, method=method, solver_type=st)
                res = []
                for un in (False, True):
                    dpm = _solver(ns, algo, guidance, "noise", None, un, None, None, edt=edt, shape=(4, 4, 16, 16), state_dtype=state)
                    res.append(dpm.sample(x if state is None else x.to(state), **kw))
                assert torch.equal(res[0], res[1]), (guidance, edt, state, algo, kw)


@pytest.mark.parametrize("guidance", ["uncond", "cfg"])
@pytest.mark.parametrize("algo", ["dpmsolver", "dpmsolver++"])
def test_device_resident_coefficient_kernels_equal_the_scalar_kernel(algo, guidance, capsys):
    """the adaptive solver with its controller on the device: DYN vector kernels (LIN1, TWO, SS3T) against the DYN scalar kernel"""
    ns = make_schedule("sd")
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.randn((3, 4, 16, 16), device=DEV, generator=g)
    for order, st in ((2, "dpmsolver"), (3, "dpmsolver"), (3, "taylor")):
        res = []
        for un in (False, True):
            dpm = _solver(ns, algo, guidance, "noise", None, un, None, None, shape=(3, 4, 16, 16))
            dpm.adaptive_on_device = True
            res.append(dpm.sample(x, method="adaptive", order=order, t_end=5e-3, solver_type=st))
        o = capsys.readouterr().out.strip().splitlines()
        assert o[0] == o[1], (algo, guidance, order, st, o)            # same NFE
        assert torch.equal(res[0], res[1]), (algo, guidance, order, st)


@pytest.mark.parametrize("mt,algo", [("noise", "dpmsolver++"), ("noise", "dpmsolver"), ("v", "dpmsolver++")])
def test_third_order_multistep_under_cfg_ends_without_the_duplicate_store(mt, algo):
    """10 steps or more: `lower_order_final` does not apply, the last stage is a third-order one -- under classifier-free
    guidance the only MS3 launch without the [2B] network input of a next stage (no KExt)"""
    ns = make_schedule("sd")
    x = torch.randn((4, 4, 16, 16), device=DEV, generator=torch.Generator(device=DEV).manual_seed(14))
    res = [_solver(ns, algo, "cfg", mt, None, un, None, None, shape=(4, 4, 16, 16)).sample(x, steps=11, order=3) for un in (False, True)]
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("guidance,mt", [("uncond", "v"), ("cfg", "noise"), ("cfg", "v"), ("uncond", "noise")])
def test_third_order_requests_in_flight_equal_single_requests(guidance, mt):
    """the fused multi-request kernels of the third-order form (compile-time and run-time prologue, unguided and CFG)"""
    ns = make_schedule("sd")
    g = torch.Generator(device=DEV).manual_seed(15)
    for dt in (torch.float32, torch.float16):
        xs = [torch.randn((4, 4, 16, 16), device=DEV, generator=g).to(dt) for _ in range(3)]
        mk = lambda: _solver(ns, "dpmsolver++", guidance, mt, None, False, None, None, edt=dt, shape=(4, 4, 16, 16), state_dtype=dt)
        outs = mk().sample_requests(xs, steps=12, order=3)
        for x, o in zip(xs, outs):
            assert torch.equal(o, mk().sample(x, steps=12, order=3)), (guidance, mt, dt)


@pytest.mark.parametrize("guidance", ["cfg", "classifier"])
def test_third_order_thresholded_stages_equal_the_catch_all_kernel(guidance):
    """3M++ with dynamic thresholding under guidance (the reference's ImageNet examples sample with both): the specialised
    thresholding kernels of form MS3 against the catch-all kernel"""
    ns = make_schedule("ddpm")
    g = torch.Generator(device=DEV).manual_seed(16)
    for shape in ((32, 3, 64, 64), (600, 3, 16, 16)):
        x = torch.randn(shape, device=DEV, generator=g) * 1.5
        res = []
        for un in (False, True):
            dpm = _solver(ns, "dpmsolver++", guidance, "noise", None, un, None, None, shape=shape)
            dpm = D.DPM_Solver(dpm._wrapped, ns, correcting_x0_fn="dynamic_thresholding")
            res.append(dpm.sample(x, steps=12, order=3))
        assert torch.equal(res[0], res[1]), (guidance, shape)
