"""Build the engine's (dpm_solver_amd) objects for the cases of tests/golden/cases.py.
Shared by the CPU host-logic tests (kernel replaced by tests/kernel_double.py) and the GPU parity tests."""
import numpy as np
import torch

import cases as C
import dpm_solver_amd as D


def tt(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def make_schedule(name):
    si = C.schedule_inputs(name)
    if si["kind"] == "linear":
        return D.NoiseScheduleVP("linear", continuous_beta_0=si["beta_0"], continuous_beta_1=si["beta_1"])
    if "betas" in si:
        return D.NoiseScheduleVP("discrete", betas=torch.from_numpy(si["betas"]))
    return D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(si["alphas_cumprod"]))


def build_solver(case, device, trace=None, **solver_kw):
    ns = make_schedule(case["schedule"])
    base = C.MODELS[case["model"]]

    def net(x, t, cond=None):
        if trace is not None:
            trace.append((int(x.shape[0]), t))
        return base(x, t, cond)

    cond, uncond = C.cond_for(case)
    kw = dict(model_type=case["model_type"], guidance_type=case["guidance_type"],
              guidance_scale=case["guidance_scale"])
    if case["guidance_type"] == "classifier-free":
        kw.update(condition=tt(cond, device), unconditional_condition=tt(uncond, device))
    elif case["guidance_type"] == "classifier":
        kw.update(condition=tt(cond, device), classifier_fn=C.classifier_logp_torch)
    model_fn = D.model_wrapper(net, ns, **kw)
    return D.DPM_Solver(model_fn, ns, algorithm_type=case["algorithm_type"],
                        correcting_x0_fn="dynamic_thresholding" if case["thresholding"] else None, **solver_kw)


def sample_kwargs(case, return_intermediate=True):
    kw = dict(steps=case["steps"], order=case["order"], skip_type=case["skip_type"], method=case["method"],
              lower_order_final=case["lower_order_final"], denoise_to_zero=case["denoise_to_zero"],
              solver_type=case["solver_type"], return_intermediate=return_intermediate)
    if case["t_start"] is not None:
        kw["t_start"] = case["t_start"]
    if case["t_end"] is not None:
        kw["t_end"] = case["t_end"]
    return kw


def run_case(case, device, trace=None, **solver_kw):
    dpm = build_solver(case, device, trace, **solver_kw)
    x = tt(C.x_T_for(case), device)
    fn = dpm.sample if case["call"] == "sample" else dpm.inverse
    return fn(x, **sample_kwargs(case))
