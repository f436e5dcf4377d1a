"""DPMSolverSampler -- the Stable-Diffusion adapter of the reference
(examples/stable-diffusion/ldm/models/diffusion/dpm_solver/sampler.py) on top of the MI355X engine.

Same constructor and methods, so `scripts/txt2img.py` (:251-311) and `scripts/diffedit_inpaint.ipynb` (cells 2-6)
run unchanged with `from dpm_solver_amd.adapters import DPMSolverSampler`.  The latent-diffusion `model` is only
used through `model.alphas_cumprod`, `model.betas.device` and `model.apply_model(x, t, c)`; it stays an opaque
PyTorch-ROCm module.  What the engine adds underneath:

  * classifier-free guidance: the UNet's [2B,...] input is written by the previous stage kernel (no torch.cat),
    its two output halves are blended inside the next stage kernel (no chunk / sub / mul / add passes);
  * `correcting_xt_fn` may be a `dpm_solver_amd.MaskBlend` (DiffEdit / inpainting), folded into the stage kernel;
  * `stochastic_encode` is one `add_noise` kernel.
"""
import torch

from ..schedule import NoiseScheduleVP
from ..solver import DPM_Solver
from ..wrapper import model_wrapper


class DPMSolverSampler(object):
    def __init__(self, model, **kwargs):
        """sampler.py:9-15: the discrete-time schedule comes from the model's alphas_cumprod (fp32)."""
        super().__init__()
        self.model = model
        dev = getattr(model, "device", None)
        ac = model.alphas_cumprod.clone().detach().to(torch.float32)
        if dev is not None:
            ac = ac.to(dev)
        self.register_buffer('alphas_cumprod', ac)
        self.noise_schedule = NoiseScheduleVP('discrete', alphas_cumprod=self.alphas_cumprod)

    def register_buffer(self, name, attr):
        """sampler.py:17-21: tensors are kept on the GPU"""
        if isinstance(attr, torch.Tensor) and not attr.is_cuda and torch.cuda.is_available():
            attr = attr.to(torch.device("cuda"))
        setattr(self, name, attr)

    # ---- solver construction shared by sample() and encode() ------------------------------------
    def _solver(self, conditioning, unconditional_conditioning, unconditional_guidance_scale, correcting_xt_fn=None):
        model_fn = model_wrapper(
            lambda x, t, c: self.model.apply_model(x, t, c),
            self.noise_schedule,
            model_type="noise",
            guidance_type="classifier-free",
            condition=conditioning,
            unconditional_condition=unconditional_conditioning,
            guidance_scale=unconditional_guidance_scale,
        )
        return DPM_Solver(model_fn, self.noise_schedule, algorithm_type="dpmsolver++", correcting_xt_fn=correcting_xt_fn)

    @staticmethod
    def _check_batch(conditioning, batch_size):
        """sampler.py:55-62: a mismatch is a warning, not an error"""
        if conditioning is None:
            return
        cbs = (conditioning[list(conditioning.keys())[0]] if isinstance(conditioning, dict) else conditioning).shape[0]
        if cbs != batch_size:
            print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, skip_type="time_uniform", method="multistep", order=2,
               lower_order_final=True, correcting_xt_fn=None, t_start=None, t_end=None, **kwargs):
        """sampler.py:23-89.  The DDIM-sampler arguments (callback ... log_every_t) are accepted and ignored, as in the
        reference.  Returns (x, intermediates)."""
        self._check_batch(conditioning, batch_size)
        C_, H, W = shape
        size = (batch_size, C_, H, W)
        device = self.model.betas.device
        img = torch.randn(size, device=device) if x_T is None else x_T
        dpm_solver = self._solver(conditioning, unconditional_conditioning, unconditional_guidance_scale, correcting_xt_fn)
        x, intermediates = dpm_solver.sample(img, t_start=t_start, t_end=t_end, steps=S, skip_type=skip_type, method=method,
                                             order=order, lower_order_final=lower_order_final, return_intermediate=True)
        return x.to(device), intermediates

    @torch.no_grad()
    def stochastic_encode(self, x0, encode_ratio, noise=None):
        """sampler.py:91-96: noise x0 to the level of `encode_ratio` (one add_noise kernel)"""
        t_end = self.ratio_to_time(encode_ratio)
        t_end = torch.tensor([t_end], device=x0.device, dtype=x0.dtype)
        return DPM_Solver(None, self.noise_schedule).add_noise(x0, t_end, noise=noise)

    @torch.no_grad()
    def encode(self, S, x, encode_ratio, conditioning=None, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, skip_type="time_uniform", method="multistep", order=2,
               lower_order_final=False, **kwargs):
        """sampler.py:98-138: deterministic encoding by running the ODE backwards (DPM_Solver.inverse)"""
        self._check_batch(conditioning, x.shape[0])
        t_end = self.ratio_to_time(encode_ratio)
        dpm_solver = self._solver(conditioning, unconditional_conditioning, unconditional_guidance_scale)
        return dpm_solver.inverse(x, steps=S, t_end=t_end, skip_type=skip_type, method=method, order=order,
                                  lower_order_final=lower_order_final, return_intermediate=True)

    # ---- time conventions (sampler.py:140-162) -----------------------------------------------------
    def time_discrete_to_continuous(self, t_discrete):
        """[0, 999] -> [0.001, 1]"""
        return (t_discrete + 1.) / self.noise_schedule.total_N

    def time_continuous_to_discrete(self, t_continuous):
        """[0.001, 1] -> [0, 999]"""
        return t_continuous * self.noise_schedule.total_N - 1.

    def ratio_to_time(self, ratio):
        """[0, 1] -> [0.001, 1]"""
        return (1. - 1. / self.noise_schedule.total_N) * ratio + 1. / self.noise_schedule.total_N

    def time_to_ratio(self, t_continuous):
        """Reproduces the reference's expression (sampler.py:162), including its denominator `1 - total_N` (the
        inverse of ratio_to_time would divide by `1 - 1/total_N`): callers such as the DiffEdit notebook depend on
        what the reference computes, not on what it meant."""
        return (t_continuous - 1. / self.noise_schedule.total_N) / (1. - self.noise_schedule.total_N)
