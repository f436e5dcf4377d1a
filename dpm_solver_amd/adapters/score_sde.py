"""`get_dpm_solver_sampler` of the ScoreSDE example (examples/score_sde_pytorch/sampling.py:505-555) on top of the MI355X
engine: same name, arguments and return convention, so `sampling.get_sampling_fn`'s `dpm_solver` branch can call it as is.

What the example does around the solver, kept with the same meaning:
  * the schedule is the continuous-time VP SDE's: `NoiseScheduleVP('linear', beta_0, beta_1)` from `sde.beta_0 / beta_1`;
  * the network is handed over as a BARE noise-prediction function (no `model_wrapper`): `get_noise_fn(sde, model,
    continuous=True)` = `model(x, t * 999)` in eval mode (models/utils.py:129-155; only a continuously trained VP model is
    supported there, here as well);
  * the prior sample comes from `sde.prior_sampling(shape)`, the run goes from `sde.T` to `eps` with
    `lower_order_final=False`; `denoise` trades the last step for `denoise_to_zero`; `thresholding` selects dynamic
    thresholding; the result passes through `inverse_scaler` and is returned with the step count as NFE.
"""
import torch

from ..schedule import NoiseScheduleVP
from ..solver import DPM_Solver


def get_noise_fn(sde, model, train=False, continuous=True):
    """models/utils.py:129-155 for a continuously trained VP model: time labels are t * 999."""
    # the reference accepts `isinstance(sde, sde_lib.VPSDE)` only (models/utils.py:143) and raises otherwise.  sde_lib is the
    # application's module, so the class is recognised by name along the MRO: subVPSDE -- which also carries beta_0 / beta_1
    # but has another marginal std -- derives from SDE, not from VPSDE, and must be rejected, not sampled as a VP model
    is_vp = any(k.__name__ == "VPSDE" for k in type(sde).__mro__)
    if not continuous or not is_vp or not (hasattr(sde, "beta_0") and hasattr(sde, "beta_1")):
        raise NotImplementedError("SDE class %s not yet supported." % sde.__class__.__name__)
    if hasattr(model, "train"):
        model.train(bool(train))

    def noise_fn(x, t):
        return model(x, t * 999)
    return noise_fn


def get_dpm_solver_sampler(sde, shape, inverse_scaler, steps=10, eps=1e-3, skip_type="logSNR", method="singlestep", order=3,
                           denoise=False, algorithm_type="dpmsolver", thresholding=False, rtol=0.05, atol=0.0078,
                           device='cuda'):
    """Returns `dpm_solver_sampler(model) -> (samples, nfe)` like sampling.py:505-555."""
    ns = NoiseScheduleVP('linear', continuous_beta_0=sde.beta_0, continuous_beta_1=sde.beta_1)

    def dpm_solver_sampler(model):
        with torch.no_grad():
            noise_pred_fn = get_noise_fn(sde, model, train=False, continuous=True)
            dpm_solver = DPM_Solver(noise_pred_fn, ns, algorithm_type=algorithm_type,
                                    correcting_x0_fn="dynamic_thresholding" if thresholding else None)
            x = sde.prior_sampling(shape).to(device)
            x = dpm_solver.sample(x, steps=steps - 1 if denoise else steps, t_start=sde.T, t_end=eps, order=order,
                                  skip_type=skip_type, method=method, denoise_to_zero=denoise, atol=atol, rtol=rtol,
                                  lower_order_final=False)
            return inverse_scaler(x), steps
    return dpm_solver_sampler
