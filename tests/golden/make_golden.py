#!/usr/bin/env python3
"""Generate the golden fixtures by running the REAL reference implementation.

Run in the build container only (it needs /root/reference, which does not exist on the GPU
box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference module is imported unmodified from /root/reference/dpm_solver_pytorch.py and
executed on CPU tensors (torch CPU kernels).  Nothing from the reference is copied into the
repository: only its *outputs* on the seeded inputs defined in tests/golden/cases.py are
stored.  The fixtures pin (a) the numpy oracle in oracle/ and (b) the HIP engine.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF_DIR = os.environ.get("DPM_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, REF_DIR)

import cases as C  # noqa: E402
from make_golden_api import api_snapshot  # noqa: E402
import dpm_solver_pytorch as R  # noqa: E402  (the reference)

torch.set_num_threads(1)
torch.set_grad_enabled(True)


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_ref_schedule(name):
    si = C.schedule_inputs(name)
    if si["kind"] == "linear":
        return R.NoiseScheduleVP("linear", continuous_beta_0=si["beta_0"], continuous_beta_1=si["beta_1"])
    if "betas" in si:
        return R.NoiseScheduleVP("discrete", betas=tt(si["betas"]))
    return R.NoiseScheduleVP("discrete", alphas_cumprod=tt(si["alphas_cumprod"]))


# ----------------------------------------------------------------------------------------
def gen_schedules(out):
    rng = np.random.default_rng(7)
    for name in C.SCHEDULE_NAMES:
        ns = make_ref_schedule(name)
        pre = "sched/%s/" % name
        out[pre + "total_N"] = np.int64(ns.total_N)
        if ns.schedule == "discrete":
            out[pre + "log_alpha_array"] = ns.log_alpha_array.numpy()
            out[pre + "t_array"] = ns.t_array.numpy()
            K = ns.total_N
            grid = ns.t_array[0, [0, 1, 2, K // 2, K - 2, K - 1]].numpy()
        else:
            grid = np.array([1e-3, 0.5, 1.0], dtype=np.float32)
        t = np.concatenate([
            rng.uniform(1e-4, 1.0, size=48).astype(np.float32),
            grid,
            np.array([1e-5, 1e-4, 1e-3, 0.0123, 0.25, 0.999, 1.0], dtype=np.float32),
        ]).astype(np.float32)
        tq = tt(t)
        out[pre + "t"] = t
        out[pre + "log_mean_coeff"] = ns.marginal_log_mean_coeff(tq).numpy()
        out[pre + "alpha"] = ns.marginal_alpha(tq).numpy()
        out[pre + "std"] = ns.marginal_std(tq).numpy()
        lam = ns.marginal_lambda(tq)
        out[pre + "lambda"] = lam.numpy()
        lam_q = np.concatenate([lam.numpy(), rng.uniform(-6.0, 8.0, size=32).astype(np.float32)])
        out[pre + "lambda_q"] = lam_q
        out[pre + "inverse_lambda"] = ns.inverse_lambda(tt(lam_q)).numpy()
        # 0-dim query (what sample() feeds): shape behaviour matters for dtype promotion
        t0 = torch.tensor(0.4321)
        out[pre + "lambda_0dim"] = ns.marginal_lambda(t0).numpy().reshape(-1)


def gen_timesteps(out):
    for name in ["sd", "vp_linear", "cosine4000"]:
        ns = make_ref_schedule(name)
        dpm = R.DPM_Solver(lambda x, t: x, ns)
        for skip in ["time_uniform", "logSNR", "time_quadratic"]:
            for (tT, t0, N) in [(1.0, 1e-3, 20), (1.0, 1e-4, 7), (0.8, 0.05, 9), (1.0, 1.0 / ns.total_N, 15)]:
                key = "tsteps/%s/%s/%g_%g_%d" % (name, skip, tT, t0, N)
                out[key] = dpm.get_time_steps(skip, tT, t0, N, "cpu").numpy()
        for order in [1, 2, 3]:
            for steps in [5, 6, 7, 8, 9, 15]:
                for skip in ["time_uniform", "logSNR"]:
                    ts, orders = dpm.get_orders_and_timesteps_for_singlestep_solver(
                        steps, order, skip, 1.0, 1e-3, "cpu")
                    key = "ssgrid/%s/%s/%d_%d" % (name, skip, order, steps)
                    out[key + "/t"] = ts.numpy()
                    out[key + "/orders"] = np.array(orders, dtype=np.int64)


def gen_updates(out):
    rng = np.random.default_rng(11)
    shape = (2, 3, 4, 4)
    x = rng.standard_normal(shape).astype(np.float32)
    m = [rng.standard_normal(shape).astype(np.float32) for _ in range(3)]  # m[0]=oldest .. m[2]=newest
    out["upd/x"] = x
    for i in range(3):
        out["upd/m%d" % i] = m[i]
    tl = [0.91, 0.78, 0.7, 0.55]  # t_prev_2, t_prev_1, t_prev_0, t
    out["upd/t"] = np.array(tl, dtype=np.float32)
    T = [torch.tensor([v], dtype=torch.float32) for v in tl]
    for sname in ["sd", "vp_linear", "cosine4000"]:
        ns = make_ref_schedule(sname)
        for algo in ["dpmsolver++", "dpmsolver"]:
            dpm = R.DPM_Solver(lambda xx, t: C.model_half(xx, t), ns, algorithm_type=algo)
            pre = "upd/%s/%s/" % (sname, algo)
            with torch.no_grad():
                out[pre + "first"] = dpm.dpm_solver_first_update(tt(x), T[2], T[3], model_s=tt(m[2])).numpy()
                for st in ["dpmsolver", "taylor"]:
                    out[pre + "ms2/" + st] = dpm.multistep_dpm_solver_second_update(
                        tt(x), [tt(m[1]), tt(m[2])], [T[1], T[2]], T[3], solver_type=st).numpy()
                    out[pre + "ms3/" + st] = dpm.multistep_dpm_solver_third_update(
                        tt(x), [tt(m[0]), tt(m[1]), tt(m[2])], [T[0], T[1], T[2]], T[3], solver_type=st).numpy()
                    for (r1, r2, tag) in [(None, None, "def"), (0.3, 0.75, "cust")]:
                        xt, inter = dpm.singlestep_dpm_solver_second_update(
                            tt(x), T[2], T[3], r1=r1, return_intermediate=True, solver_type=st)
                        out[pre + "ss2/%s/%s/x_t" % (st, tag)] = xt.numpy()
                        out[pre + "ss2/%s/%s/model_s1" % (st, tag)] = inter["model_s1"].numpy()
                        xt, inter = dpm.singlestep_dpm_solver_third_update(
                            tt(x), T[2], T[3], r1=r1, r2=r2, return_intermediate=True, solver_type=st)
                        out[pre + "ss3/%s/%s/x_t" % (st, tag)] = xt.numpy()
                        out[pre + "ss3/%s/%s/model_s1" % (st, tag)] = inter["model_s1"].numpy()
                        out[pre + "ss3/%s/%s/model_s2" % (st, tag)] = inter["model_s2"].numpy()


def gen_quantile(out):
    """dynamic_thresholding_fn (dpm_solver_pytorch.py:416-425) on raw x0 tensors."""
    rng = np.random.default_rng(3)
    ns = make_ref_schedule("ddpm")
    for (tag, shape, scale, p, mv) in [
        ("a", (4, 3, 64, 64), 3.0, 0.995, 1.0),   # BASELINE cfg5 per-sample size, clamp active
        ("b", (3, 3, 8, 8), 0.5, 0.995, 1.0),     # quantile < max_val -> s = max_val
        ("c", (2, 1, 5, 7), 2.0, 0.9, 1.5),       # odd size, other ratio / max_val
        ("d", (2, 3, 16, 16), 1.0, 1.0, 0.1),     # p = 1 -> max
        ("e", (2, 2, 2, 2), 4.0, 0.5, 1.0),       # tiny, median
        ("f", (2, 3, 256, 256), 2.5, 0.995, 1.0), # large sample (config-3 sized rows)
    ]:
        x0 = (rng.standard_normal(shape) * scale).astype(np.float32)
        if tag == "e":
            x0[0] = np.float32(2.0)               # all-equal row (ties)
        dpm = R.DPM_Solver(lambda xx, t: xx, ns, correcting_x0_fn="dynamic_thresholding",
                           thresholding_max_val=mv, dynamic_thresholding_ratio=p)
        y = dpm.dynamic_thresholding_fn(tt(x0), None)
        s = torch.quantile(torch.abs(tt(x0)).reshape((shape[0], -1)), p, dim=1)
        pre = "quant/%s/" % tag
        if tag != "f":
            out[pre + "x0"] = x0
            out[pre + "y"] = y.numpy()
        else:
            out[pre + "seed_scale"] = np.array([3, scale])  # regenerate: see tests
            out[pre + "y_sum"] = np.float64(y.double().sum().item())
            out[pre + "y_head"] = y.numpy().reshape(shape[0], -1)[:, :256].copy()
        out[pre + "s"] = s.numpy()
        out[pre + "p_mv"] = np.array([p, mv], dtype=np.float64)


def gen_add_noise(out):
    rng = np.random.default_rng(5)
    for sname in ["sd", "vp_linear"]:
        ns = make_ref_schedule(sname)
        dpm = R.DPM_Solver(lambda xx, t: xx, ns)
        x = rng.standard_normal((2, 4, 8, 8)).astype(np.float32)
        for tag, tv in [("one", [0.37]), ("three", [0.9, 0.5, 0.013])]:
            t = np.array(tv, dtype=np.float32)
            noise = rng.standard_normal((len(tv),) + x.shape).astype(np.float32)
            y = dpm.add_noise(tt(x), tt(t), noise=tt(noise))
            pre = "addnoise/%s/%s/" % (sname, tag)
            out[pre + "x"] = x
            out[pre + "t"] = t
            out[pre + "noise"] = noise
            out[pre + "y"] = y.numpy()


# ----------------------------------------------------------------------------------------
def build_ref_solver(case, trace):
    ns = make_ref_schedule(case["schedule"])
    base = C.MODELS[case["model"]]

    def net(x, t, cond=None):
        trace.append((tuple(x.shape), float(t.reshape(-1)[0]), str(x.dtype)))
        return base(x, t, cond)

    cond, uncond = C.cond_for(case)
    kw = dict(model_type=case["model_type"], guidance_type=case["guidance_type"],
              guidance_scale=case["guidance_scale"])
    if case["guidance_type"] == "classifier-free":
        kw.update(condition=tt(cond), unconditional_condition=tt(uncond))
    elif case["guidance_type"] == "classifier":
        kw.update(condition=tt(cond), classifier_fn=C.classifier_logp_torch)
    model_fn = R.model_wrapper(net, ns, **kw)
    dpm = R.DPM_Solver(model_fn, ns, algorithm_type=case["algorithm_type"],
                       correcting_x0_fn="dynamic_thresholding" if case["thresholding"] else None)
    return dpm


def run_ref_case(case):
    trace = []
    dpm = build_ref_solver(case, trace)
    x = tt(C.x_T_for(case))
    fn = dpm.sample if case["call"] == "sample" else dpm.inverse
    kw = dict(steps=case["steps"], order=case["order"], skip_type=case["skip_type"], method=case["method"],
              lower_order_final=case["lower_order_final"], denoise_to_zero=case["denoise_to_zero"],
              solver_type=case["solver_type"], return_intermediate=True)
    if case["t_start"] is not None:
        kw["t_start"] = case["t_start"]
    if case["t_end"] is not None:
        kw["t_end"] = case["t_end"]
    xf, inter = fn(x, **kw)
    return xf, inter, trace


def gen_e2e(out):
    for case in C.E2E_CASES:
        xf, inter, trace = run_ref_case(case)
        pre = "e2e/%s/" % case["name"]
        out[pre + "final"] = xf.numpy()
        out[pre + "final_dtype"] = np.array(str(xf.dtype))
        if case["intermediates"]:
            out[pre + "intermediates"] = np.stack([v.float().numpy() for v in inter])
        out[pre + "n_intermediates"] = np.int64(len(inter))
        out[pre + "trace_t"] = np.array([t for (_, t, _) in trace], dtype=np.float32)
        out[pre + "trace_b"] = np.array([s[0] for (s, _, _) in trace], dtype=np.int64)
        print("  %-22s nfe=%3d  sum=%.9g absmax=%.9g" % (
            case["name"], len(trace), xf.double().sum().item(), xf.abs().max().item()))


def gen_callbacks(out):
    """correcting_xt_fn / callable correcting_x0_fn hooks (dpm_solver_pytorch.py:370-394,
    :1180-1181,:1203-1204)."""
    case = dict(C.E2E_BY_NAME["cfg1_small"])
    ns = make_ref_schedule(case["schedule"])
    x = tt(C.x_T_for(case))
    mask = (torch.arange(x.numel()).reshape(x.shape) % 3 == 0).float()

    def cxt(xt, t, step):
        return xt * mask + (1.0 - mask) * (0.25 * step)

    def cx0(x0, t):
        return torch.clamp(x0, -1.5, 1.5)

    model_fn = R.model_wrapper(lambda xx, t: C.model_half(xx, t), ns)
    for tag, kw in [("xt", dict(correcting_xt_fn=cxt)), ("x0", dict(correcting_x0_fn=cx0)),
                    ("both", dict(correcting_xt_fn=cxt, correcting_x0_fn=cx0))]:
        for method, order, steps in [("multistep", 2, 8), ("singlestep", 3, 8)]:
            dpm = R.DPM_Solver(model_fn, ns, **kw)
            xf, inter = dpm.sample(x, steps=steps, order=order, method=method, denoise_to_zero=True,
                                   return_intermediate=True)
            pre = "cb/%s/%s/" % (tag, method)
            out[pre + "final"] = xf.numpy()
            out[pre + "intermediates"] = np.stack([v.numpy() for v in inter])
    out["cb/mask"] = mask.numpy()


def gen_adaptive(out):
    import contextlib
    import io
    for (name, sname, order, algo) in [("a12", "vp_linear", 2, "dpmsolver"), ("a23", "vp_linear", 3, "dpmsolver"),
                                       ("a23pp", "sd", 3, "dpmsolver++")]:
        ns = make_ref_schedule(sname)
        model_fn = R.model_wrapper(lambda xx, t: C.model_half(xx, t), ns)
        dpm = R.DPM_Solver(model_fn, ns, algorithm_type=algo)
        rng = np.random.default_rng(21)
        x = rng.standard_normal((2, 3, 8, 8)).astype(np.float32)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            xf = dpm.sample(tt(x), method="adaptive", order=order, t_end=1e-3, atol=0.0078, rtol=0.05)
        nfe = int(buf.getvalue().strip().split()[-1])
        pre = "adaptive/%s/" % name
        out[pre + "x"] = x
        out[pre + "final"] = xf.numpy()
        out[pre + "nfe"] = np.int64(nfe)
        print("  adaptive %-6s nfe=%d" % (name, nfe))


def gen_sampler(out):
    """The reference's Stable-Diffusion adapter (DPMSolverSampler) driven by a stand-in model on CPU: txt2img-style
    sampling, stochastic / deterministic encoding and both DiffEdit variants of scripts/diffedit_inpaint.ipynb
    (cell 6).  The class is imported unmodified; only its `register_buffer` -- which insists on a CUDA device --
    is replaced at run time so that the goldens can be produced in this CPU-only container."""
    sd_dir = os.path.join(REF_DIR, "examples", "stable-diffusion", "ldm", "models", "diffusion")
    sys.path.insert(0, sd_dir)
    import dpm_solver.sampler as RS  # the reference package ldm/models/diffusion/dpm_solver
    RS.DPMSolverSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    inp = C.sampler_inputs()
    model = C.FakeLatentDiffusion(torch, "cpu")
    smp = RS.DPMSolverSampler(model)
    x_T, x0, noise, mask = tt(inp["x_T"]), tt(inp["x0"]), tt(inp["noise"]), tt(inp["mask"])
    cond, uncond = tt(inp["cond"]), tt(inp["uncond"])
    B = x_T.shape[0]
    x, inter = smp.sample(10, B, x_T.shape[1:], conditioning=cond, unconditional_guidance_scale=7.5,
                          unconditional_conditioning=uncond, x_T=x_T, verbose=False)
    out["sampler/sample/final"] = x.numpy()
    out["sampler/sample/intermediates"] = np.stack([v.numpy() for v in inter])
    out["sampler/sample/calls_t"] = np.array([t for _, t in model.calls], dtype=np.float64)
    out["sampler/stochastic_encode"] = smp.stochastic_encode(x0, 0.6, noise=noise.unsqueeze(0)).numpy()
    enc, einter = smp.encode(10, x0, 0.6, conditioning=cond, unconditional_guidance_scale=7.5,
                             unconditional_conditioning=uncond)
    out["sampler/encode/final"] = enc.numpy()
    out["sampler/encode/intermediates"] = np.stack([v.numpy() for v in einter])
    tv = torch.tensor([0.001, 0.25, 0.6004, 1.0])
    out["sampler/times"] = np.stack([smp.time_discrete_to_continuous(tv * 999).numpy(),
                                     smp.time_continuous_to_discrete(tv).numpy(), smp.ratio_to_time(tv).numpy(),
                                     smp.time_to_ratio(tv).numpy()])
    # DiffEdit, deterministic: reversed encode() intermediates are blended back in at every step
    rev = list(reversed(einter))
    det = lambda xt, t, step: xt * mask + (1 - mask) * rev[step]
    x, _ = smp.sample(10, B, x_T.shape[1:], conditioning=cond * 0.5, unconditional_guidance_scale=7.5,
                      unconditional_conditioning=uncond, lower_order_final=False, t_start=smp.ratio_to_time(0.6),
                      x_T=enc, correcting_xt_fn=det)
    out["sampler/diffedit_det"] = x.numpy()
    # DiffEdit, stochastic (noise fixed so that the golden does not depend on a random stream)
    noised = smp.stochastic_encode(x0, 0.6, noise=noise.unsqueeze(0))

    def sto(xt, t, step):
        return xt * mask + (1 - mask) * smp.stochastic_encode(x0, smp.time_to_ratio(t), noise=noise.unsqueeze(0))
    x, _ = smp.sample(10, B, x_T.shape[1:], conditioning=cond * 0.5, unconditional_guidance_scale=7.5,
                      unconditional_conditioning=uncond, lower_order_final=False, t_start=smp.ratio_to_time(0.6),
                      x_T=noised, correcting_xt_fn=sto)
    out["sampler/diffedit_sto"] = x.numpy()
    print("  sampler: %d network calls" % len(model.calls))


def gen_legacy(out):
    """The older revision the reference vendors for the ScoreSDE example (examples/score_sde_pytorch/dpm_solver.py),
    imported unmodified: its continuous-time 'cosine' schedule (T = 0.9946) and its unclipped discrete schedules."""
    import importlib.util
    path = os.path.join(REF_DIR, "examples", "score_sde_pytorch", "dpm_solver.py")
    spec = importlib.util.spec_from_file_location("dpm_solver_legacy", path)
    LG = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(LG)
    rng = np.random.default_rng(17)
    ns = LG.NoiseScheduleVP("cosine")
    out["legacy/cosine/T"] = np.float64(ns.T)
    t = np.concatenate([rng.uniform(1e-4, ns.T, size=64).astype(np.float32), np.array([1e-3, 0.5, ns.T], dtype=np.float32)])
    out["legacy/cosine/t"] = t
    tt_ = tt(t)
    out["legacy/cosine/log_alpha"] = ns.marginal_log_mean_coeff(tt_).numpy()
    out["legacy/cosine/alpha"] = ns.marginal_alpha(tt_).numpy()
    out["legacy/cosine/std"] = ns.marginal_std(tt_).numpy()
    lam = ns.marginal_lambda(tt_)
    out["legacy/cosine/lambda"] = lam.numpy()
    out["legacy/cosine/inv_lambda"] = ns.inverse_lambda(lam).numpy()
    x = rng.standard_normal((2, 3, 8, 8)).astype(np.float32)
    out["legacy/x"] = x
    for tag, kw in [("ms2", dict(steps=10, order=2, method="multistep")),
                    ("ss3_logsnr", dict(steps=9, order=3, method="singlestep", skip_type="logSNR")),
                    ("ms3_noise", dict(steps=8, order=3, method="multistep"))]:
        algo = "dpmsolver" if tag == "ms3_noise" else "dpmsolver++"
        fn = LG.model_wrapper(lambda xx, t: C.model_tdep(xx, t), ns)
        xf = LG.DPM_Solver(fn, ns, algorithm_type=algo).sample(tt(x), t_end=1e-3, **kw)
        out["legacy/cosine/%s" % tag] = xf.numpy()
    # unclipped discrete schedule: cosine-4000 betas keep all 4000 entries (the root file clips to 3984)
    si = C.schedule_inputs("cosine4000")
    nd = LG.NoiseScheduleVP("discrete", betas=tt(si["betas"]))
    out["legacy/noclip/total_N"] = np.int64(nd.total_N)
    out["legacy/noclip/log_alpha_tail"] = nd.log_alpha_array.numpy()[0, -32:]
    fn = LG.model_wrapper(lambda xx, t: C.model_tdep(xx, t), nd)
    out["legacy/noclip/ms2"] = LG.DPM_Solver(fn, nd).sample(tt(x), steps=10, order=2, t_start=0.9).numpy()
    print("  legacy: cosine T=%.4f, noclip total_N=%d" % (ns.T, nd.total_N))


def gen_guided(out):
    """The DPM-Solver branch of the guided-diffusion runner (runners/diffusion.py:594-640) wired by hand around the
    reference solver classes: 6-channel network, classifier guidance through log_softmax + autograd, thresholding,
    denoise.  (The Runner class itself needs the whole app's args/config machinery; the wiring is eight lines.)"""
    inp = C.gd_inputs()
    x, y = tt(inp["x"]), torch.from_numpy(inp["y"])
    model = C.gd_network(torch, tt(inp["junk"]))
    classifier = C.gd_classifier(torch, tt(inp["w"]))
    betas = tt(C.schedule_inputs("ddpm")["betas"])

    def run(sample_type, use_clf, thresholding, denoise, scale, method="multistep", order=2, timesteps=12):
        def model_fn(xx, t, **kw):
            return torch.split(model(xx, t, **kw), 3, dim=1)[0]

        def classifier_fn(xx, t, yy, **kw):
            log_probs = torch.nn.functional.log_softmax(classifier(xx, t), dim=-1)
            return log_probs[range(len(log_probs)), yy.view(-1)]
        ns = R.NoiseScheduleVP(schedule="discrete", betas=betas)
        fn = R.model_wrapper(model_fn, ns, model_type="noise", model_kwargs={"y": y},
                             guidance_type="classifier" if use_clf else "uncond", condition=y, guidance_scale=scale,
                             classifier_fn=classifier_fn, classifier_kwargs={})
        dpm = R.DPM_Solver(fn, ns, algorithm_type=sample_type,
                           correcting_x0_fn="dynamic_thresholding" if thresholding else None)
        return dpm.sample(x, steps=(timesteps - 1 if denoise else timesteps), order=order, skip_type="time_uniform",
                          method=method, lower_order_final=True, denoise_to_zero=denoise, solver_type="dpmsolver")
    for tag, kw in C.GD_RUNS:
        out["guided/" + tag] = run(**kw).detach().numpy()


def gen_score_sde(out):
    """`get_dpm_solver_sampler` of the ScoreSDE example, imported unmodified together with its `sde_lib` and `models.utils`
    (examples/score_sde_pytorch/sampling.py:505-555; its `from dpm_solver import ...` resolves to the older revision vendored
    next to it): VP SDE, stand-in score model, prior samples from a seeded torch.randn"""
    ex = os.path.join(REF_DIR, "examples", "score_sde_pytorch")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k in ("dpm_solver", "sampling", "sde_lib", "models") or k.startswith("models.")}
    sys.path.insert(0, ex)
    try:
        import sampling as SM
        import sde_lib
        sde = sde_lib.VPSDE(beta_min=0.1, beta_max=20., N=1000)
        inverse_scaler = lambda x: (x + 1.) / 2.
        for i, (tag, kw) in enumerate(C.SCORE_SDE_RUNS):
            torch.manual_seed(100 + i)
            fn = SM.get_dpm_solver_sampler(sde, C.SCORE_SDE_SHAPE, inverse_scaler, device="cpu", **kw)
            y, nfe = fn(C.score_sde_model(torch))
            out["score_sde/%s/x" % tag] = y.numpy()
            out["score_sde/%s/nfe" % tag] = np.int64(nfe)
    finally:
        sys.path.remove(ex)
        for k in [k for k in sys.modules if k in ("dpm_solver", "sampling", "sde_lib", "models") or k.startswith("models.")]:
            sys.modules.pop(k)
        sys.modules.update(saved)


def gen_utils(out):
    """the module-level helpers interpolate_fn / expand_dims (ref :1253-1305) on seeded inputs: queries inside the
    keypoint range, on keypoints, beyond both ends (linear extrapolation), several channels"""
    rng = np.random.default_rng(21)
    for tag, (n, c, k) in dict(a=(37, 1, 9), b=(5, 3, 2), c=(64, 2, 1000)).items():
        xp = np.sort(rng.uniform(-2.0, 3.0, size=(c, k)).astype(np.float32), axis=1)
        yp = rng.standard_normal((c, k)).astype(np.float32)
        x = rng.uniform(-3.0, 4.0, size=(n, c)).astype(np.float32)
        x[0, :] = xp[:, 0]               # exactly on the first / last keypoint, and on an inner one
        x[1, :] = xp[:, -1]
        x[2, :] = xp[:, k // 2]
        out["utils/%s/x" % tag], out["utils/%s/xp" % tag], out["utils/%s/yp" % tag] = x, xp, yp
        out["utils/%s/y" % tag] = R.interpolate_fn(tt(x), tt(xp), tt(yp)).numpy()
    v = rng.standard_normal(6).astype(np.float32)
    out["utils/expand/v"] = v
    for dims in (1, 2, 4):
        out["utils/expand/shape%d" % dims] = np.array(R.expand_dims(tt(v), dims).shape, dtype=np.int64)
        out["utils/expand/val%d" % dims] = R.expand_dims(tt(v), dims).numpy()


def gen_clip(out):
    """NoiseScheduleVP.numerical_clip_alpha (ref :114-125) called directly: the raw (unclipped) log-alpha tables of the
    discrete schedules as ref :100 / :103 compute them, in fp32 and fp64, clipped at several half-logSNR bounds"""
    any_ns = R.NoiseScheduleVP("linear")
    for name in ["sd", "ddpm", "cosine4000", "cosine1000"]:
        si = C.schedule_inputs(name)
        for prec in ("f32", "f64"):
            dt = torch.float32 if prec == "f32" else torch.float64
            if "betas" in si:
                la = 0.5 * torch.log(1 - tt(si["betas"]).to(dt)).cumsum(dim=0)
            else:
                la = 0.5 * torch.log(tt(si["alphas_cumprod"]).to(dt))
            pre = "clip/%s/%s/" % (name, prec)
            out[pre + "log_alphas"] = la.numpy()
            for cl in (-5.1, -3.0, 0.0, -20.0):
                got = any_ns.numerical_clip_alpha(la, cl) if cl != -5.1 else any_ns.numerical_clip_alpha(la)
                out[pre + "len/%g" % cl] = np.int64(got.shape[0])
                assert torch.equal(got, la[:got.shape[0]])


def gen_api(out):
    """signatures of the reference's public API -> tests/golden/api_signatures.json (the engine must keep them)"""
    import json
    snap = api_snapshot(R)
    with open(os.path.join(HERE, "api_signatures.json"), "w") as f:
        json.dump(snap, f, indent=1, sort_keys=True)
    out["api/entries"] = np.int64(len(snap))


def main():
    groups = dict(score_sde=gen_score_sde, api=gen_api, clip=gen_clip, utils=gen_utils, guided=gen_guided, legacy=gen_legacy, schedules=gen_schedules, timesteps=gen_timesteps, updates=gen_updates,
                  quantile=gen_quantile, add_noise=gen_add_noise, e2e=gen_e2e,
                  callbacks=gen_callbacks, adaptive=gen_adaptive, sampler=gen_sampler)
    only = sys.argv[1:]
    for gname, fn in groups.items():
        if only and gname not in only:
            continue
        print("[golden] %s" % gname)
        out = {}
        fn(out)
        path = os.path.join(HERE, gname + ".npz")
        np.savez_compressed(path, **{k.replace("/", "|"): v for k, v in out.items()})
        print("  -> %s (%d arrays, %.1f KiB)" % (path, len(out), os.path.getsize(path) / 1024))
    with open(os.path.join(HERE, "PROVENANCE.txt"), "w") as f:
        f.write("generated by tests/golden/make_golden.py\n")
        f.write("reference: %s/dpm_solver_pytorch.py (LuChengTHU/dpm-solver @ v1, unmodified)\n" % REF_DIR)
        f.write("torch %s (CPU kernels), numpy %s, 1 thread\n" % (torch.__version__, np.__version__))


if __name__ == "__main__":
    main()
