"""model_wrapper -- host mirror of the reference function (dpm_solver_pytorch.py:170-334).

Same signature and semantics.  The returned object is callable like the reference's `model_fn(x,
t_continuous) -> noise`, but it also exposes the *raw* network outputs so that DPM_Solver can fuse the
parameterisation conversion (x_start / v / score -> noise, ref :288-298), the classifier-free-guidance
blend (ref :326-330) and the classifier-guidance term (ref :315-321) into the prologue of the stage
kernel instead of running them as separate elementwise passes.  The network itself (and the autograd
call through the classifier) stays an opaque PyTorch-ROCm call on the current stream.
"""
import torch


class WrappedModel:
    def __init__(self, model, noise_schedule, model_type, model_kwargs, guidance_type, condition,
                 unconditional_condition, guidance_scale, classifier_fn, classifier_kwargs):
        self.model = model
        self.noise_schedule = noise_schedule
        self.model_type = model_type
        self.model_kwargs = model_kwargs
        self.guidance_type = guidance_type
        self.condition = condition
        self.unconditional_condition = unconditional_condition
        self.guidance_scale = guidance_scale
        self.classifier_fn = classifier_fn
        self.classifier_kwargs = classifier_kwargs
        self._c_in = None

    # ---- what the fused kernel must do for this wrapper -------------------------------------
    @property
    def effective_guidance(self):
        if self.guidance_type == "classifier":
            return "classifier"
        if self.guidance_type == "classifier-free" and not (
                self.guidance_scale == 1. or self.unconditional_condition is None):
            return "classifier-free"
        return "uncond"

    def get_model_input_time(self, t_continuous):
        """ref :271-280"""
        if self.noise_schedule.schedule == 'discrete':
            return (t_continuous - 1. / self.noise_schedule.total_N) * 1000.
        return t_continuous

    def _cond_grad(self, x, t_input):
        """grad_x log p_t(cond | x_t) through the (opaque) classifier, ref :300-307"""
        with torch.enable_grad():
            x_in = x.detach().requires_grad_(True)
            log_prob = self.classifier_fn(x_in, t_input, self.condition, **self.classifier_kwargs)
            return torch.autograd.grad(log_prob.sum(), x_in)[0]

    def raw_outputs(self, x, t_continuous, t_input=None, t_input2=None, x_in2=None):
        """Run the network(s) exactly as the reference does and return (e0, e1, g):
        e0 raw output (conditional half under CFG), e1 raw unconditional output or None, g classifier
        gradient or None.  No conversion or blending is applied here -- the stage kernel does that."""
        if t_input is None:
            t_input = self.get_model_input_time(t_continuous)
        kw = self.model_kwargs
        if self.guidance_type == "uncond":
            return self.model(x, t_input, **kw), None, None
        if self.guidance_type == "classifier":
            assert self.classifier_fn is not None
            g = self._cond_grad(x, t_input)
            return self.model(x, t_input, **kw), None, g
        if self.guidance_type == "classifier-free":
            if self.guidance_scale == 1. or self.unconditional_condition is None:
                return self.model(x, t_input, self.condition, **kw), None, None
            # x_in2: the stage kernel that produced x already wrote it into both halves of one [2B,...] buffer
            x_in = x_in2 if x_in2 is not None else torch.cat([x] * 2)
            t_in = t_input2 if t_input2 is not None else torch.cat([t_input] * 2)
            if self._c_in is None:  # the conditioning does not change between steps: concatenate once
                self._c_in = torch.cat([self.unconditional_condition, self.condition])
            out = self.model(x_in, t_in, self._c_in, **kw)
            e1, e0 = out.chunk(2)
            return e0, e1, None
        raise AssertionError(self.guidance_type)

    # ---- reference-compatible call: noise prediction ------------------------------------------
    def __call__(self, x, t_continuous):
        """The reference's model_fn(x, t_continuous) -> noise (ref :282-330) for callers OTHER than DPM_Solver (which fuses
        these conversions into its stage kernels and never comes here): the raw network call(s), then the reference's own
        tensor expressions in torch on x's device -- per-sample times (`t_continuous` of shape (B,) with distinct entries)
        included, no host synchronisation (the schedule is evaluated on the device, NoiseScheduleVP.device_alpha_sigma)."""
        from .utils import expand_dims
        from ._device import _require_gpu
        _require_gpu(x)
        e0, e1, g = self.raw_outputs(x, t_continuous)
        mt = self.model_type
        dims = x.dim()

        def to_noise(out):                                        # noise_pred_fn, ref :288-298
            if mt == "noise":
                return out
            alpha_t, sigma_t = self.noise_schedule.device_alpha_sigma(t_continuous)
            if mt == "x_start":
                return (x - expand_dims(alpha_t, dims) * out) / expand_dims(sigma_t, dims)
            if mt == "v":
                return expand_dims(alpha_t, dims) * out + expand_dims(sigma_t, dims) * x
            return -expand_dims(sigma_t, dims) * out              # "score"
        eff = self.effective_guidance
        if eff == "classifier":                                   # ref :315-321
            _, sigma_t = self.noise_schedule.device_alpha_sigma(t_continuous)
            return to_noise(e0) - self.guidance_scale * expand_dims(sigma_t, dims) * g
        if eff == "classifier-free":                              # ref :326-330
            nu, nc = to_noise(e1), to_noise(e0)
            return nu + self.guidance_scale * (nc - nu)
        return to_noise(e0)


def model_wrapper(model, noise_schedule, model_type="noise", model_kwargs={}, guidance_type="uncond",
                  condition=None, unconditional_condition=None, guidance_scale=1., classifier_fn=None,
                  classifier_kwargs={}):
    """Create a wrapper function for the noise prediction model (signature of ref :170-181).

    model_type: "noise" | "x_start" | "v" | "score";  guidance_type: "uncond" | "classifier" |
    "classifier-free".  Returns `model_fn(x, t_continuous) -> noise`.
    """
    assert model_type in ["noise", "x_start", "v", "score"]
    assert guidance_type in ["uncond", "classifier", "classifier-free"]
    return WrappedModel(model, noise_schedule, model_type, model_kwargs, guidance_type, condition,
                        unconditional_condition, guidance_scale, classifier_fn, classifier_kwargs)
