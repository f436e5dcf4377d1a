"""The DPM-Solver branch of the guided-diffusion / DDPM example runner
(examples/ddpm_and_guided-diffusion/runners/diffusion.py:594-640, `Diffusion.sample_image`) as a free function on
top of the MI355X engine.

The runner wires three things around the solver, all kept here with the same meaning:
  * networks with a learned variance return `[B, 2C, H, W]`; only the mean half `out[:, :C]` is used (:599-603) --
    the engine reads that channel slice in place (`dpm_buffers.eps_stride`), no copy;
  * classifier guidance: `classifier_fn(x, t, y) = log_softmax(classifier(x, t))[range(B), y]` (:605-608), its
    autograd gradient stays an opaque PyTorch call, the `noise - scale * sigma_t * grad` term is fused into the stage
    kernel;
  * `--thresholding` -> `correcting_x0_fn="dynamic_thresholding"`, `--denoise` -> `denoise_to_zero` with one step
    fewer (:626, :629-636).
"""
import torch

from ..schedule import NoiseScheduleVP
from ..solver import DPM_Solver
from ..wrapper import model_wrapper


def sample_image(x, model, betas, classifier=None, classes=None, classifier_scale=1.0, out_channels=None,
                 sample_type="dpmsolver++", thresholding=False, timesteps=20, denoise=False, dpm_solver_order=2,
                 skip_type="time_uniform", dpm_solver_method="multistep", lower_order_final=True,
                 dpm_solver_type="dpmsolver", dpm_solver_atol=0.0078, dpm_solver_rtol=0.05, model_kwargs=None):
    """Sample with DPM-Solver / DPM-Solver++ exactly as the runner does; the keyword names follow its `args.*` /
    `config.*` fields.  `classes` are the labels `y` (the runner draws them itself, :532-538); returns `(x, classes)`."""
    assert sample_type in ("dpmsolver", "dpmsolver++")
    model_kwargs = dict(model_kwargs or {})
    if classes is not None and "y" not in model_kwargs:
        model_kwargs["y"] = classes

    def model_fn(x, t, **kw):
        out = model(x, t, **kw)
        if out_channels == 6:               # mean and variance: DPM-Solver integrates the ODE, it needs the mean only
            out = torch.split(out, 3, dim=1)[0]
        return out

    def classifier_fn(x, t, y, **kw):
        logits = classifier(x, t)
        log_probs = torch.nn.functional.log_softmax(logits, dim=-1)
        return log_probs[range(len(logits)), y.view(-1)]

    noise_schedule = NoiseScheduleVP(schedule='discrete', betas=betas)
    model_fn_continuous = model_wrapper(
        model_fn, noise_schedule, model_type="noise", model_kwargs=model_kwargs,
        guidance_type="uncond" if classifier is None else "classifier",
        condition=model_kwargs["y"] if "y" in model_kwargs else None,
        guidance_scale=classifier_scale, classifier_fn=classifier_fn, classifier_kwargs={})
    dpm_solver = DPM_Solver(model_fn_continuous, noise_schedule, algorithm_type=sample_type,
                            correcting_x0_fn="dynamic_thresholding" if thresholding else None)
    x = dpm_solver.sample(
        x, steps=(timesteps - 1 if denoise else timesteps), order=dpm_solver_order, skip_type=skip_type,
        method=dpm_solver_method, lower_order_final=lower_order_final, denoise_to_zero=denoise,
        solver_type=dpm_solver_type, atol=dpm_solver_atol, rtol=dpm_solver_rtol)
    return x, classes
