#!/usr/bin/env python3
"""Drop-in fuzz: random corners of the public API against the LIVE reference (build container only: needs /root/reference).

The engine's host side (C planner + Python loops) runs on CPU tensors with the stage kernel replaced by its numpy double
(tests/kernel_double.py -- the GPU suite shows kernel == double bit for bit); the unmodified dpm_solver_pytorch.py runs beside
it.  Every case compares: did both raise (same exception type and text) or both return; dtype, shape, values (1e-5 of the
trajectory's peak for fp32, 1e-10 for double, the half format's resolution for half results), the intermediates, and the
network-call trace (shapes, dtypes and times the network saw).  Wider than tests/test_differential_reference.py: state
shapes of 1 to 5 dimensions, batch 1, non-contiguous inputs, half / double x_T, steps below the order, singlestep grids that
degenerate, every method incl. adaptive, networks that answer in half precision or fp32 whatever the state's dtype (Stable
Diffusion under autocast), public update methods with tensor / float / None arguments.

    python tools/fuzz_dropin.py [--cases 1500] [--seed 0]        # prints every disagreement and a summary
"""
import argparse
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
REF_DIR = os.environ.get("DPM_REFERENCE_DIR", "/root/reference")     # (never put on sys.path: the reference is loaded by file)

import cases as C  # noqa: E402
import dpm_solver_amd as D  # noqa: E402
import dpm_solver_amd.solver as S  # noqa: E402
import importlib.util  # noqa: E402
# the reference BY FILE under a name of its own: the repository root holds a drop-in shim called dpm_solver_pytorch too (the
# engine), and whichever of the two was imported first owns that name in sys.modules
_ref_path = os.path.join(REF_DIR, "dpm_solver_pytorch.py")
if os.path.exists(_ref_path):
    _spec = importlib.util.spec_from_file_location("_the_reference_dpm_solver_pytorch", _ref_path)
    R = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(R)
    assert R.DPM_Solver is not D.DPM_Solver, "the reference module resolved to the engine's shim"
else:
    R = None          # (the GPU box: tools/fuzz_gpu*.py import this module for its case generator only)
from engine_cases import make_schedule  # noqa: E402
from kernel_double import install_cpu_double  # noqa: E402


class _MP:
    def setattr(self, o, n, v):
        setattr(o, n, v)


F32 = np.float32


def install(mp=None):
    """route the engine's device entry points to the numpy double of the kernels (CPU tensors); `mp`: a pytest monkeypatch (undone
    at the end of the test), default: for the life of the process (the command-line tool)"""
    install_cpu_double(mp or _MP(), S, D)


def _table_input(arr, dtype):
    """the betas / alphas_cumprod array handed to NoiseScheduleVP: for dtype=float64 a DOUBLE array, so that the log table is
    computed in double on both sides (from an fp32 array the reference takes torch's vectorised fp32 log -- SLEEF, whose last
    place differs between its AVX2 and AVX512 builds, and from a correctly rounded log, in a few entries of a thousand -- and
    widens the result: an fp32-ulp table difference is 1e-8 in a double run and says nothing about the plans)"""
    return np.asarray(arr, dtype=np.float64) if dtype is torch.float64 else np.asarray(arr)


def ref_schedule(name, dtype=torch.float32):
    si = C.schedule_inputs(name)
    if si["kind"] == "linear":
        return R.NoiseScheduleVP("linear", continuous_beta_0=si["beta_0"], continuous_beta_1=si["beta_1"])
    key = "betas" if "betas" in si else "alphas_cumprod"
    return R.NoiseScheduleVP("discrete", dtype=dtype, **{key: torch.from_numpy(_table_input(si[key], dtype))})


def eng_schedule(name, dtype=torch.float32):
    if dtype is torch.float32:
        return make_schedule(name)
    si = C.schedule_inputs(name)
    key = "betas" if "betas" in si else "alphas_cumprod"
    return D.NoiseScheduleVP("discrete", dtype=dtype, **{key: torch.from_numpy(_table_input(si[key], dtype))})


SHAPES = [(3,), (2, 5), (2, 3, 4), (2, 3, 4, 4), (1, 3, 4, 4), (1, 2, 3, 2, 2), (4, 1, 1, 1), (2, 12)]


def random_case(rng):
    method = str(rng.choice(["multistep", "singlestep", "singlestep_fixed", "adaptive"], p=[0.4, 0.3, 0.2, 0.1]))
    order = int(rng.choice([1, 2, 3, 3, 2, 4, 0])) if rng.random() < 0.15 else int(rng.integers(1, 4))
    steps = int(rng.integers(1, 13))
    guidance = str(rng.choice(["uncond", "uncond", "classifier-free", "classifier"]))
    return dict(method=method, order=order, steps=steps, shape=SHAPES[int(rng.integers(0, len(SHAPES)))],
                schedule=str(rng.choice(["sd", "ddpm", "vp_linear", "cosine1000"])),
                skip_type=str(rng.choice(["time_uniform", "logSNR", "time_quadratic", "bogus"], p=[0.4, 0.3, 0.27, 0.03])),
                solver_type=str(rng.choice(["dpmsolver", "taylor", "bogus"], p=[0.55, 0.42, 0.03])),
                algorithm_type=str(rng.choice(["dpmsolver++", "dpmsolver"])),
                model_type=str(rng.choice(["noise", "x_start", "v", "score"])), guidance=guidance,
                scale=float(rng.choice([1.0, 2.5, 7.5])), thresholding=bool(rng.integers(0, 4) == 0),
                cxt=bool(rng.integers(0, 5) == 0), cx0=bool(rng.integers(0, 6) == 0),
                lower_order_final=bool(rng.integers(0, 2)), denoise_to_zero=bool(rng.integers(0, 3) == 0),
                t_end=(None if rng.random() < 0.5 else float(rng.choice([1e-3, 1e-2, 0.05]))),
                t_start=(None if rng.random() < 0.5 else float(rng.choice([1.0, 0.8, 0.5]))),
                call=str(rng.choice(["sample", "sample", "sample", "inverse"])),
                ret_inter=bool(rng.integers(0, 2)), xdt=str(rng.choice(["f32", "f32", "f32", "f64", "f16", "bf16"])),
                noncontig=bool(rng.integers(0, 6) == 0), seed=int(rng.integers(0, 1 << 30)),
                net_dt=str(rng.choice(["same", "same", "same", "f16", "f32"])))


WIDE_NET = False       # tools/fuzz_gpu_methods.py: the stand-in network computes half inputs in fp32 and rounds once -- torch's half
                       # operations with Python scalars do not round alike on the CPU and on the GPU, and that tool compares the two


def build(mod, ns, cfg, x, trace):
    B = x.shape[0]
    cond = torch.arange(1, B + 1, dtype=torch.float32, device=x.device) * 0.5
    kw = dict(model_type=cfg["model_type"], guidance_type=cfg["guidance"], guidance_scale=cfg["scale"])

    def base(xx, t, c=None):
        trace.append((tuple(xx.shape), str(xx.dtype), str(t.dtype), tuple(t.shape), round(float(t.reshape(-1)[0]), 4)))
        wd = torch.float32 if (WIDE_NET and xx.dtype in (torch.float16, torch.bfloat16)) else xx.dtype
        tt = t.to(wd).reshape((-1,) + (1,) * (xx.dim() - 1))
        out = xx.to(wd) * (tt * 0.0005 + 0.25)
        if c is not None:
            out = out * (c.to(wd).reshape((-1,) + (1,) * (xx.dim() - 1)) * 0.1 + 1.0)
        out = out.to(xx.dtype)
        # (a network that answers in its own dtype: Stable Diffusion under autocast hands fp16 to an fp32 state)
        return out if (ndt is None or xx.dtype is torch.float64) else out.to(ndt)
    ndt = {"same": None, "f16": torch.float16, "f32": torch.float32}[cfg.get("net_dt", "same")]
    if cfg["guidance"] == "classifier-free":
        net = lambda xx, t, c: base(xx, t, c)
        kw.update(condition=cond, unconditional_condition=torch.zeros(B, device=x.device))
    elif cfg["guidance"] == "classifier":
        net = lambda xx, t, c=None: base(xx, t)
        kw.update(condition=cond, classifier_fn=lambda xx, t, c: -0.5 * (xx.reshape(xx.shape[0], -1) ** 2).sum(dim=1) * 0.01)
    else:
        net = lambda xx, t: base(xx, t)
    fn = mod.model_wrapper(net, ns, **kw)
    skw = dict(algorithm_type=cfg["algorithm_type"])
    if cfg["thresholding"]:
        skw["correcting_x0_fn"] = "dynamic_thresholding"
    elif cfg["cx0"]:
        skw["correcting_x0_fn"] = lambda x0, t: torch.clamp(x0, -2.0, 2.0)
    if cfg["cxt"]:
        skw["correcting_xt_fn"] = lambda xt, t, step: xt * 0.99 + 0.001 * step
    return mod.DPM_Solver(fn, ns, **skw)


def run(mod, ns, cfg, x):
    trace = []
    try:
        dpm = build(mod, ns, cfg, x, trace)
        kw = dict(steps=cfg["steps"], order=cfg["order"], method=cfg["method"], skip_type=cfg["skip_type"],
                  solver_type=cfg["solver_type"], lower_order_final=cfg["lower_order_final"],
                  denoise_to_zero=cfg["denoise_to_zero"], return_intermediate=cfg["ret_inter"])
        if cfg["method"] == "adaptive":
            kw.update(atol=0.05, rtol=0.1)
        if cfg["call"] == "sample":
            out = dpm.sample(x, t_start=cfg["t_start"], t_end=cfg["t_end"], **kw)
        else:
            out = dpm.inverse(x, t_start=cfg["t_end"], t_end=cfg["t_start"], **kw)
        return ("ok", out, trace)
    except Exception as e:                              # noqa: BLE001 -- the comparison is about what is raised
        return ("raise", (type(e).__name__, str(e)[:160]), trace, traceback.format_exc(limit=3))


def compare(cfg, r, e, yardstick=None, half_yardstick=None):
    """list of disagreements of one case; yardstick(): |fp32 reference - its own double run| / peak, or None"""
    bad = []
    if r[0] == "raise" and r[1][0] in ("UnboundLocalError", "RuntimeError", "IndexError", "TypeError"):
        # the reference crashed on its own terms -- `step` unbound when a loop ran zero times (ref :1233-1237), torch.quantile on
        # a half tensor, torch.linspace with a negative count: the engine is not asked to crash the same way
        return bad
    if r[0] != e[0]:
        return ["reference %s, engine %s: %s | %s" % (r[0], e[0], r[1] if r[0] == "raise" else "", e[1] if e[0] == "raise" else "")]
    if r[0] == "raise":
        if r[1][0] != e[1][0]:
            bad.append("exception type %s vs %s (%s | %s)" % (r[1][0], e[1][0], r[1][1], e[1][1]))
        elif r[1][1] != e[1][1] and r[1][0] in ("ValueError", "AssertionError") and r[1][1] and "unpack" not in r[1][1]:
            bad.append("exception text %r vs %r" % (r[1][1], e[1][1]))
        return bad
    ro, eo = r[1], e[1]
    ri, ei = ([], [])
    if cfg["ret_inter"] and cfg["method"] != "adaptive":
        ro, ri = ro
        eo, ei = eo
    pinned_half = (cfg["schedule"] == "vp_linear" and cfg["xdt"] in ("f16", "bf16") and cfg.get("net_dt") in ("f16", "bf16")
                   and cfg["algorithm_type"] == "dpmsolver" and cfg["model_type"] == "noise" and cfg["guidance"] != "classifier"
                   and cfg["method"] in ("singlestep", "singlestep_fixed") and cfg["order"] >= 2)
    if pinned_half and ro.dtype != eo.dtype and eo.dtype == torch.float32:
        # documented: a network PINNED to half precision keeps the reference's singlestep run half in the
        # noise-prediction form (only the intermediate state meets a (1,)-shaped fp32 coefficient); the engine's run is fp32
        # from its first update, like the reference's with a network that follows its input's dtype
        eo, ei = eo.to(ro.dtype), [t.to(a.dtype) for t, a in zip(ei, ri)] if len(ei) == len(ri) else ei
    if ro.dtype != eo.dtype:
        bad.append("result dtype %s vs %s" % (ro.dtype, eo.dtype))
    if tuple(ro.shape) != tuple(eo.shape):
        bad.append("result shape %s vs %s" % (tuple(ro.shape), tuple(eo.shape)))
        return bad
    if not bool(torch.isfinite(ro.float()).all()):
        return bad                                      # the reference itself diverged
    peak = max([float(ro.double().abs().max())] + [float(t.double().abs().max()) for t in ri]) or 1.0
    half = cfg["xdt"] in ("f16", "bf16") and cfg["schedule"] == "vp_linear"
    tol = (0.15 if cfg["xdt"] == "bf16" else 2e-2) if half else 1e-5   # (a double state on an fp32 schedule is fp32 scalars on both sides: the fp32 bar; largest seen 7.2e-6)
    if cfg["method"] == "adaptive" and cfg["xdt"] in ("f16", "bf16"):
        tol = max(tol, 5e-2)        # the reference's loop scalars (t, h, atol) are half until the first accepted step: INTEGRATION.md
    net_half = cfg.get("net_dt") if cfg.get("net_dt") in ("f16", "bf16") else (cfg["xdt"] if cfg.get("net_dt") == "same" and cfg["xdt"] in ("f16", "bf16") else None)
    if (net_half and cfg["algorithm_type"] == "dpmsolver" and cfg["model_type"] == "noise" and cfg["solver_type"] == "taylor"
            and cfg["order"] == 3 and cfg["method"] in ("singlestep", "singlestep_fixed")):
        # the one formula whose half arithmetic is not reproduced (the singlestep third-order 'taylor'
        # combination in the noise-prediction form of a half-precision network stays fp32): the half format's resolution
        tol = max(tol, 3e-2 if net_half == "bf16" else 4e-3)
    if "_tol" in cfg:                # --double-tables: every scalar is a double on both sides
        tol, yardstick, half_yardstick = cfg["_tol"], None, None
    err = float((ro.double() - eo.double()).abs().max()) / peak
    if err > tol and yardstick is not None and (ro.dtype == torch.float32 or cfg["xdt"] == "f64"):
        # the judge's yardstick: how far is the fp32 reference from ITS OWN double-precision run?  A case
        # where that distance is of the order of the disagreement is ill-conditioned (cancelling O(100) terms), not a delta
        own = yardstick()
        if own is not None and own >= 0.2 * err:
            return bad + ["conditioning: %.3g of the peak, the fp32 reference is %.3g from its own double run" % (err, own)]
    if err > tol and half and half_yardstick is not None:
        # a half state on a continuous schedule: the reference rounds after every operation, the engine once per stage --
        # the engine must be at least as near the fp32 trajectory as the reference's own half arithmetic is
        f32 = half_yardstick()
        if f32 is not None:
            e_eng = float((eo.double() - f32.double()).abs().max()) / peak
            e_ref = float((ro.double() - f32.double()).abs().max()) / peak
            if e_eng <= 1.5 * e_ref + 1e-3:
                return bad + ["conditioning: half arithmetic, engine %.3g / reference %.3g from the fp32 trajectory" % (e_eng, e_ref)]
    if err > tol:
        bad.append("values: %.3g of the peak (tolerance %.2g)" % (err, tol))
    if len(ri) != len(ei):
        bad.append("%d vs %d intermediates" % (len(ri), len(ei)))
    else:
        for k, (a, b) in enumerate(zip(ri, ei)):
            if a.dtype != b.dtype:
                bad.append("intermediate %d dtype %s vs %s" % (k, a.dtype, b.dtype))
                break
            ierr = float((a.double() - b.double()).abs().max()) / peak
            if ierr > tol:
                own = yardstick(True) if (yardstick is not None and a.dtype == torch.float32) else None
                if own is not None and own >= 0.2 * ierr:     # an early state of an ill-conditioned trajectory (see above)
                    return bad + ["conditioning: intermediate %d %.3g of the peak, the fp32 reference is %.3g from its own double run" % (k, ierr, own)]
                if half and half_yardstick is not None:       # half arithmetic: the same rule as for the result, state by state
                    f32s = half_yardstick(True)
                    if f32s is not None and len(f32s) == len(ri) + 1:
                        e_eng = float((b.double() - f32s[k + 1].double()).abs().max()) / peak
                        e_ref = float((a.double() - f32s[k + 1].double()).abs().max()) / peak
                        if e_eng <= 1.5 * e_ref + 1e-3:
                            return bad + ["conditioning: half arithmetic, intermediate %d engine %.3g / reference %.3g from the fp32 trajectory" % (k, e_eng, e_ref)]
                bad.append("intermediate %d values: %.3g" % (k, ierr))
                break
    if not half and cfg["method"] != "adaptive":
        rt, et = r[2], e[2]
        if len(rt) != len(et):
            bad.append("network calls %d vs %d" % (len(rt), len(et)))
        else:
            for k, (a, b) in enumerate(zip(rt, et)):
                if a[0] != b[0] or a[2:4] != b[2:4] or abs(a[4] - b[4]) > 2e-3 or a[1] != b[1]:
                    bad.append("network call %d: %s vs %s" % (k, a, b))
                    break
    return bad


# ------------------------------------------------------------------------------------------------
# mode "methods": the public per-update methods, the schedule's functions and the helpers, one random call each
# ------------------------------------------------------------------------------------------------
def _tens(rng, shape, dt):
    return torch.from_numpy(rng.standard_normal(shape)).to(dt)


def random_method_call(rng):
    """(name, builder): builder(mod, ns, dpm) -> the call's result; the same random arguments for either module"""
    sched = str(rng.choice(["sd", "ddpm", "vp_linear", "cosine1000"]))
    shape = SHAPES[int(rng.integers(0, len(SHAPES)))]
    xdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16}[str(rng.choice(["f32", "f32", "f32", "f64", "f16"]))]
    tdt = torch.float64 if rng.random() < 0.15 else torch.float32
    algo = str(rng.choice(["dpmsolver++", "dpmsolver"]))
    mt = str(rng.choice(["noise", "noise", "x_start", "v", "score"]))
    guid = str(rng.choice(["uncond", "uncond", "classifier-free", "classifier"]))
    thr = bool(rng.integers(0, 5) == 0) and algo == "dpmsolver++"
    seed = int(rng.integers(0, 1 << 30))
    tshape = [(1,), (), (1,)][int(rng.integers(0, 3))]
    ts = sorted(rng.uniform(0.02, 0.98, size=4).tolist(), reverse=True)
    solver_type = str(rng.choice(["dpmsolver", "taylor", "bogus"], p=[0.5, 0.45, 0.05]))
    what = str(rng.choice(["first", "ss2", "ss3", "ms2", "ms3", "ss_disp", "ms_disp", "noise_fn", "data_fn", "model_fn", "denoise", "add_noise",
                           "time_steps", "orders", "thresh", "marginals", "inv_lambda", "interp"]))
    r_kind = str(rng.choice(["default", "float", "tensor", "none"]))
    given = int(rng.integers(0, 4))
    ret_i = bool(rng.integers(0, 2))
    order = int(rng.choice([1, 2, 3, 0, 4], p=[0.3, 0.3, 0.3, 0.05, 0.05]))
    nt = int(rng.integers(1, 4))
    cfg = dict(what=what, schedule=sched, shape=shape, xdt=str(xdt)[6:], tdt=str(tdt)[6:], algorithm_type=algo, model_type=mt, guidance=guid, thr=thr,
               tshape=tshape, solver_type=solver_type, r=r_kind, given=given, ret_inter=ret_i, order=order, nt=nt)

    def call(mod, ns, util, eps=0.0, device="cpu"):
        """eps: relative perturbation of every time argument (the conditioning yardstick of fuzz_methods); device: where the
        tensors live (tools/fuzz_gpu_methods.py runs the engine's side of this on the GPU)"""
        g = np.random.default_rng(seed)
        x = _tens(g, shape, xdt).to(device)
        tt = lambda v: torch.full(tshape, v * (1.0 + eps), dtype=tdt, device=device)
        c = dict(method="multistep", order=2, steps=5, shape=shape, schedule=sched, skip_type="time_uniform", solver_type="dpmsolver",
                 algorithm_type=algo, model_type=mt, guidance=guid, scale=2.5, thresholding=thr, cxt=False, cx0=False)
        trace = []
        dpm = build(mod, ns, c, x, trace)
        s_, t_ = tt(ts[1]), tt(ts[2])
        r1 = {"default": 0.5, "float": 0.37, "tensor": torch.tensor(0.41), "none": None}[r_kind]
        r2 = {"default": 2. / 3., "float": 0.71, "tensor": torch.tensor(0.77), "none": None}[r_kind]
        if what == "first":
            ms = dpm.model_fn(x, s_) if given & 1 else None
            return dpm.dpm_solver_first_update(x, s_, t_, model_s=ms, return_intermediate=ret_i)
        if what == "ss2":
            ms = dpm.model_fn(x, s_) if given & 1 else None
            kw = {} if r_kind == "default" else dict(r1=r1)
            return dpm.singlestep_dpm_solver_second_update(x, s_, t_, model_s=ms, return_intermediate=ret_i, solver_type=solver_type, **kw)
        if what == "ss3":
            ms = dpm.model_fn(x, s_) if given & 1 else None
            ms1 = dpm.model_fn((x.double() * 0.9).to(x.dtype), tt(ts[1] * 0.9 + ts[2] * 0.1)) if given & 2 else None   # (one rounding: the same on CPU and GPU)
            kw = {} if r_kind == "default" else dict(r1=(r1 if r1 is None else r1 * 0.6), r2=r2)
            return dpm.singlestep_dpm_solver_third_update(x, s_, t_, model_s=ms, model_s1=ms1, return_intermediate=ret_i,
                                                          solver_type=solver_type, **kw)
        if what in ("ms2", "ms3", "ms_disp"):
            tl = [tt(ts[0]), tt(ts[0] * 0.5 + ts[1] * 0.5), tt(ts[1])]
            ml = [dpm.model_fn(x, v) for v in tl]
            if what == "ms2":
                return dpm.multistep_dpm_solver_second_update(x, ml[1:], tl[1:], t_, solver_type=solver_type)
            if what == "ms3":
                return dpm.multistep_dpm_solver_third_update(x, ml, tl, t_, solver_type=solver_type)
            return dpm.multistep_dpm_solver_update(x, ml, tl, t_, order, solver_type=solver_type)
        if what == "ss_disp":
            return dpm.singlestep_dpm_solver_update(x, s_, t_, order, return_intermediate=ret_i, solver_type=solver_type)
        if what == "noise_fn":
            return dpm.noise_prediction_fn(x, s_)
        if what == "data_fn":
            return dpm.data_prediction_fn(x, s_)
        if what == "model_fn":
            return dpm.model_fn(x, s_)
        if what == "denoise":
            return dpm.denoise_to_zero_fn(x, s_)
        if what == "add_noise":
            tv = torch.tensor(ts[:nt], dtype=tdt, device=device)
            return dpm.add_noise(x, tv, noise=_tens(g, (nt,) + tuple(shape), xdt).to(device))
        if what == "time_steps":
            sk = str(np.random.default_rng(seed).choice(["time_uniform", "logSNR", "time_quadratic", "bogus"], p=[0.35, 0.3, 0.3, 0.05]))
            return dpm.get_time_steps(sk, ts[0], ts[3] * 0.1 + 1e-3, int(seed % 12) + 1, device)
        if what == "orders":
            sk = str(np.random.default_rng(seed).choice(["time_uniform", "logSNR", "time_quadratic"]))
            t_, o_ = dpm.get_orders_and_timesteps_for_singlestep_solver(int(seed % 14) + 1, order, sk, ts[0], ts[3] * 0.1 + 1e-3, device)
            return (t_, torch.tensor(o_))
        if what == "thresh":
            return mod.DPM_Solver(dpm.model if False else (lambda a, b: a), ns, correcting_x0_fn="dynamic_thresholding",
                                  dynamic_thresholding_ratio=float(0.9 + 0.099 * g.random()), thresholding_max_val=float(g.choice([1.0, 0.5, 2.0]))
                                  ).dynamic_thresholding_fn(x.float() * 2, None)
        tv = torch.tensor(g.uniform(0.001, 0.999, size=tuple(int(v) for v in g.integers(1, 4, size=int(g.integers(1, 3))))), dtype=tdt, device=device)
        if what == "marginals":
            return (ns.marginal_log_mean_coeff(tv), ns.marginal_alpha(tv), ns.marginal_std(tv), ns.marginal_lambda(tv))
        if what == "inv_lambda":
            return ns.inverse_lambda(ns.marginal_lambda(tv))
        if what == "interp":
            xp = torch.sort(torch.from_numpy(g.uniform(0, 1, size=(1, 9))).float(), dim=1)[0].to(device)
            yp = torch.from_numpy(g.standard_normal((1, 9))).float().to(device)
            xq = torch.from_numpy(g.uniform(-0.2, 1.2, size=(5, 1))).float().to(device)
            return util.interpolate_fn(xq, xp, yp)
        raise AssertionError(what)
    return cfg, call


def _flatten(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, dict):
        return [v for k in sorted(o) for v in _flatten(o[k])]
    if isinstance(o, (tuple, list)):
        return [v for it in o for v in _flatten(it)]
    return [] if o is None else [torch.as_tensor(o)]


def fuzz_methods(args):
    import contextlib
    import io
    import dpm_solver_amd.utils as U
    rng = np.random.default_rng(args.seed)
    n_bad = n_raise = n_cond = 0
    kinds = {}
    for i in range(args.cases):
        cfg, call = random_method_call(rng)
        if args.only is not None and i != args.only:
            continue
        res = []
        for mod, mk, util in ((R, ref_schedule, R), (D, eng_schedule, U)):
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    res.append(("ok", call(mod, mk(cfg["schedule"]), util)))
            except Exception as ex:                     # noqa: BLE001
                res.append(("raise", (type(ex).__name__, str(ex)[:160]), traceback.format_exc(limit=4)))
        r, e = res
        n_raise += r[0] == "raise"
        bad = []
        if r[0] == "raise" and r[1][0] in ("RuntimeError", "IndexError", "TypeError", "UnboundLocalError", "AttributeError"):
            continue                                    # the reference crashed on its own terms
        if r[0] != e[0]:
            bad.append("reference %s, engine %s: %s | %s" % (r[0], e[0], r[1] if r[0] == "raise" else "", e[1] if e[0] == "raise" else ""))
        elif r[0] == "raise":
            if r[1][0] != e[1][0] or (r[1][1] != e[1][1] and r[1][0] == "ValueError" and "unpack" not in r[1][1]):
                bad.append("exception %s vs %s" % (r[1], e[1]))
        else:
            ra, ea = _flatten(r[1]), _flatten(e[1])
            if len(ra) != len(ea):
                bad.append("%d vs %d tensors returned" % (len(ra), len(ea)))
            for k, (a, b) in enumerate(zip(ra, ea)):
                if k > 0 and isinstance(r[1], tuple) and isinstance(r[1][-1], dict) and a.dtype != b.dtype and tuple(a.shape) == tuple(b.shape):
                    # a model value handed back by return_intermediate: the reference returns the raw network output in the
                    # network's dtype, the engine the stored copy in the state's dtype -- same values
                    # (a half x on a continuous schedule: the reference's model value is a half tensor, the engine's fp32 copy
                    # carries the value before that rounding)
                    wtol = 8e-3 if cfg["xdt"] == "float16" else 1e-6
                    if float((a.double() - b.double()).abs().max()) <= wtol * (float(a.double().abs().max()) or 1.0):
                        continue
                if a.dtype != b.dtype or tuple(a.shape) != tuple(b.shape):
                    bad.append("tensor %d: %s %s vs %s %s" % (k, a.dtype, tuple(a.shape), b.dtype, tuple(b.shape)))
                    break
                if not bool(torch.isfinite(a.double()).all()):
                    continue
                pk = float(a.double().abs().max()) or 1.0
                tol = 1e-10 if a.dtype == torch.float64 and cfg["tdt"] == "float64" else (5e-6 if a.dtype == torch.float64 else (1e-5 if a.dtype == torch.float32 else 8e-3))
                if cfg["xdt"] == "float16":
                    tol = max(tol, 8e-3)        # a half x: the reference's partial sums are half operations
                elif a.dtype == torch.float64 and cfg["xdt"] == "float32":
                    tol = max(tol, 2e-7)        # a double result from an fp32 x: the reference's `coefficient * x` is still an fp32
                                                # product where the coefficient is 0-dim; the engine's whole expression is double
                elif a.dtype == torch.float64:
                    tol = max(tol, 1e-9)
                if a.dtype == torch.float64 and cfg["schedule"] in ("ddpm", "cosine1000"):
                    # tables built from fp32 betas go through ATen's vectorised fp32 log (SLEEF, <= 1 ulp, and not the same
                    # bits on an AVX2 and an AVX-512 host or on the GPU); the planner's is the correctly rounded one: a few
                    # table entries differ by one fp32 ulp (12 of 1000 / 3 of 996 here), which a double evaluation resolves
                    tol = max(tol, 2e-7)
                err = float((a.double() - b.double()).abs().max()) / pk
                if err > tol:
                    # conditioning yardstick: the reference against ITSELF with every time argument moved by one part in 1e7
                    # (an fp32 ulp -- the size of the table differences noted below): update formulas between nearly equal
                    # times divide by tiny logSNR steps and amplify that as much as they amplify anything else
                    try:
                        with contextlib.redirect_stdout(io.StringIO()):
                            own = _flatten(call(R, ref_schedule(cfg["schedule"]), R, eps=1e-7))[k]
                        moved = float((a.double() - own.double()).abs().max()) / pk
                    except Exception:                   # noqa: BLE001
                        moved = 0.0
                    if moved >= 0.2 * err:
                        n_cond += 1
                        if args.verbose:
                            print("call %d: conditioning: %.3g, the reference moves %.3g when its times move by 1e-7" % (i, err, moved))
                        break
                    bad.append("tensor %d values: %.3g (tolerance %.1g)" % (k, err, tol))
                    break
        if bad:
            n_bad += 1
            kinds[cfg["what"]] = kinds.get(cfg["what"], 0) + 1
            print("call %d: %s\n    %s" % (i, cfg, "\n    ".join(bad)), flush=True)
            if e[0] == "raise" and r[0] != "raise":
                print("    " + e[2].replace("\n", "\n    "))
    print("%d method calls, %d where the reference raised, %d ill-conditioned (the reference moves as far when its times move by "
          "1e-7), %d disagreements %s" % (args.cases, n_raise, n_cond, n_bad, kinds))
    return n_bad


def main():
    assert R is not None, "no reference at %s (DPM_REFERENCE_DIR)" % _ref_path
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--case-timeout", type=int, default=60)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--mode", default="sample", choices=["sample", "methods"])
    ap.add_argument("--double-tables", action="store_true", help="sample mode: discrete schedules declared dtype=torch.float64")
    ap.add_argument("--only", type=int, default=None, help="replay one case of the seeded sequence (the earlier ones are drawn and skipped)")
    args = ap.parse_args()
    if args.mode == "methods":
        install()
        torch.set_num_threads(1)
        return fuzz_methods(args)
    rng = np.random.default_rng(args.seed)
    n_bad = n_raise = n_slow = n_cond = 0
    install()
    torch.set_num_threads(1)
    kinds = {}
    import contextlib
    import io
    for i in range(args.cases):
        cfg = random_case(rng)
        if args.only is not None and i != args.only:
            continue
        if cfg["thresholding"] and cfg["algorithm_type"] == "dpmsolver":
            cfg["thresholding"] = False
        if cfg["method"] == "adaptive":
            cfg["call"] = "sample"              # the reference's adaptive loop does not terminate on an inversion (t increasing)
        g = torch.Generator().manual_seed(cfg["seed"])
        x = torch.randn(cfg["shape"], generator=g)
        x = x.to({"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[cfg["xdt"]])
        if cfg["noncontig"] and x.dim() >= 2:
            x = x.transpose(0, 1).contiguous().transpose(0, 1)
        import signal
        import time

        class _Slow(BaseException):
            pass

        def _alarm(sig, frm):
            raise _Slow()
        signal.signal(signal.SIGALRM, _alarm)
        t0 = time.perf_counter()
        who = "reference"
        try:
            signal.alarm(args.case_timeout)
            with contextlib.redirect_stdout(io.StringIO()):
                ns_dt = torch.float32
                if args.double_tables and cfg["schedule"] != "vp_linear" and cfg["method"] != "adaptive":
                    # NoiseScheduleVP(dtype=torch.float64): double tables, the run is a double run from its first update whatever
                    # x_T's dtype (the planner's double-precision plans); one fp32 step survives in the reference -- the logSNR
                    # grid's logaddexp on an fp32 linspace (1e-7 at the grid times)
                    ns_dt = torch.float64
                    cfg["_tol"] = 2e-7 if cfg["skip_type"] == "logSNR" else 1e-11
                r = run(R, ref_schedule(cfg["schedule"], ns_dt), cfg, x)
                t1 = time.perf_counter()
                who = "engine"
                e = run(D, eng_schedule(cfg["schedule"], ns_dt), cfg, x)
            signal.alarm(0)
        except _Slow:
            signal.alarm(0)
            print("case %d: %s\n    TIMEOUT after %d s inside the %s" % (i, {k: v for k, v in cfg.items() if k != "seed"}, args.case_timeout, who), flush=True)
            n_slow += 1
            continue
        if args.verbose:
            print("case %d %.2f s (reference %.2f) %s %s steps=%d" % (i, time.perf_counter() - t0, t1 - t0, cfg["method"], cfg["schedule"], cfg["steps"]), flush=True)
        n_raise += r[0] == "raise"

        def yardstick(with_intermediates=False):
            if (cfg["xdt"] == "f64" and cfg["schedule"] == "vp_linear") or (cfg["xdt"] not in ("f32", "f64") and cfg["schedule"] == "vp_linear"):
                return None
            with contextlib.redirect_stdout(io.StringIO()):
                dtype = torch.float64 if cfg["schedule"] != "vp_linear" else torch.float32
                r64 = run(R, ref_schedule(cfg["schedule"], dtype), cfg, x.double())
                # (a half x_T on a discrete schedule is promoted at the first update: the fp32 run from the same values)
                # (a double x_T on fp32 tables: the run under test itself -- double tensors, fp32 scalars -- against double tables)
                r32 = r if cfg["xdt"] in ("f32", "f64") else run(R, ref_schedule(cfg["schedule"]), cfg, x.float())
            if r64[0] != "ok" or r32[0] != "ok":
                return None
            o64 = r64[1][0] if (cfg["ret_inter"] and cfg["method"] != "adaptive") else r64[1]
            o32 = r32[1][0] if (cfg["ret_inter"] and cfg["method"] != "adaptive") else r32[1]
            pk = float(o32.double().abs().max()) or 1.0
            if with_intermediates and cfg["ret_inter"] and cfg["method"] != "adaptive" and len(r32[1][1]) == len(r64[1][1]):
                pk = max([pk] + [float(t.double().abs().max()) for t in r32[1][1]])
                return max(float((a.double() - b).abs().max()) for a, b in zip([o32] + list(r32[1][1]), [o64] + list(r64[1][1]))) / pk
            return float((o32.double() - o64).abs().max()) / pk
        def half_yardstick(with_intermediates=False):
            with contextlib.redirect_stdout(io.StringIO()):
                r32 = run(R, ref_schedule(cfg["schedule"]), cfg, x.float())
            if r32[0] != "ok":
                return None
            if with_intermediates:
                return ([r32[1][0]] + list(r32[1][1])) if (cfg["ret_inter"] and cfg["method"] != "adaptive") else None
            return r32[1][0] if (cfg["ret_inter"] and cfg["method"] != "adaptive") else r32[1]
        bad = compare(cfg, r, e, yardstick, half_yardstick)
        if bad and bad[-1].startswith("conditioning"):
            n_cond += 1
            if args.verbose:
                print("case %d: %s" % (i, bad[-1]), flush=True)
            bad = bad[:-1]
        if bad:
            n_bad += 1
            kinds[bad[0].split(":")[0][:40]] = kinds.get(bad[0].split(":")[0][:40], 0) + 1
            print("case %d: %s\n    %s" % (i, {k: v for k, v in cfg.items() if k != "seed"}, "\n    ".join(bad)), flush=True)
            if e[0] == "raise" and r[0] != "raise":
                print("    " + e[3].replace("\n", "\n    "))
    print("%d cases, %d where the reference raised, %d timed out, %d ill-conditioned (the fp32 reference as far from its own double "
          "run), %d disagreements %s" % (args.cases, n_raise, n_slow, n_cond, n_bad, kinds))
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
