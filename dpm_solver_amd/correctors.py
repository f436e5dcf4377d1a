"""Fusable `correcting_xt_fn` objects.

The reference calls `correcting_xt_fn(x, t, step)` after every solver update (dpm_solver_pytorch.py:1180,
:1188,:1203,:1229,:1238).  The callers that use the hook -- DiffEdit / inpainting on Stable Diffusion
(examples/stable-diffusion/scripts/diffedit_inpaint.ipynb cell 6, through sampler.py:75-87) -- all pass the
same elementwise mask blend

    stochastic      x <- x*mask + (1 - mask) * add_noise(x0, t')      add_noise: alpha_t'*x0 + sigma_t'*noise (ref :1028)
    deterministic   x <- x*mask + (1 - mask) * intermediates[step]

as a Python closure: one `add_noise` pass plus four elementwise passes per step.  `MaskBlend` is the same function
as an object.  It is callable like the closure (one stand-alone kernel), and `DPM_Solver` recognises it and folds
the blend into the epilogue of the stage kernel (`DPM_F_BLEND`): the update and the blend become one pass that reads
mask, x0 and noise next to the update's own streams.  Any other callable keeps working through the generic hook.

Arithmetic: fp32, the reference's association `x*mask + (1 - mask)*(alpha*x0 + sigma*noise)`, no fused
multiply-adds -- bit-identical to the closure on an fp32 state.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

_DT = {torch.float32: L.DTYPE_F32, torch.float16: L.DTYPE_F16, torch.bfloat16: L.DTYPE_BF16, torch.float64: L.DTYPE_F64}


class MaskBlend:
    def __init__(self, noise_schedule, mask, x0=None, intermediates=None, noise=None, time_fn=None, generator=None):
        """mask: tensor broadcastable against the state from the right ([H,W], [C,H,W], [1,C,H,W] or full size);
        1 keeps the solver's x, 0 takes the known content.

        Exactly one of
          x0            known clean image: blended in at the noise level of time `time_fn(t)` (default: t itself).
                        `noise` fixes the noise tensor; None draws torch.randn per call like
                        `DPM_Solver.add_noise(x0, t)` does (ref :1026), from `generator` if given;
          intermediates list of states indexed by `step` (e.g. reversed `DPM_Solver.inverse(...,
                        return_intermediate=True)[1]`).

        time_fn: host function float -> float applied to the solver time before alpha/sigma are looked up (the
        notebook routes t through `sampler.time_to_ratio` / `ratio_to_time`)."""
        assert (x0 is None) != (intermediates is None), "give exactly one of x0 / intermediates"
        self.noise_schedule = noise_schedule
        self.mask = mask
        self.x0 = x0
        self.intermediates = intermediates
        self.noise = noise
        self.time_fn = time_fn
        self.generator = generator
        self._mask_cache = {}

    # ---- pieces shared by the fused and the stand-alone path ----------------------------------
    def _mask_for(self, shape, dtype, device):
        """contiguous mask in the state's dtype + its period in elements"""
        key = (dtype, str(device), tuple(shape))
        hit = self._mask_cache.get(key)
        if hit is None:
            m = self.mask.to(device=device, dtype=dtype)
            while m.dim() > 1 and m.shape[0] == 1:                     # leading broadcast dims carry no data
                m = m[0]
            if tuple(m.shape) != tuple(shape[len(shape) - m.dim():]):
                m = m.expand(shape)                                    # general broadcasting: materialise once
            m = m.contiguous()
            hit = (m, max(int(m.numel()), 1))
            self._mask_cache[key] = hit
        return hit

    def _alpha_sigma(self, t):
        tt = float(t) if self.time_fn is None else float(self.time_fn(float(t)))
        tin = np.array([tt], dtype=np.float32)
        a = self.noise_schedule._eval_np(L.EVAL_ALPHA, tin)[0]
        s = self.noise_schedule._eval_np(L.EVAL_STD, tin)[0]
        return float(a), float(s)

    def operands(self, shape, dtype, device, t, step):
        """(mask, period, a, b, alpha, sigma) for a state of the given shape / dtype at solver time t (host float)
        and step index.  No device synchronisation: alpha / sigma come from the host schedule."""
        shape = tuple(shape)
        m, period = self._mask_for(shape, dtype, device)
        if self.intermediates is not None:
            a = self.intermediates[step].to(device=device, dtype=dtype).contiguous()
            return m, period, a, None, 1.0, 0.0
        a = self.x0.to(device=device, dtype=dtype)
        if tuple(a.shape) != shape:
            a = a.expand(shape)
        if self.noise is not None:
            b = self.noise.to(device=device, dtype=dtype)
        else:
            # DPM_Solver.add_noise draws randn((t.shape[0], *x.shape)) (ref :1026); same shape => same stream of
            # random numbers as the closure under one seed
            b = torch.randn((1, *shape), device=device, generator=self.generator)[0].to(dtype)
        if tuple(b.shape) != shape:
            b = b.expand(shape)
        alpha, sigma = self._alpha_sigma(t)
        return m, period, a.contiguous(), b.contiguous(), alpha, sigma

    # ---- the closure's interface ---------------------------------------------------------------
    def __call__(self, x, t, step):
        if not x.is_cuda:
            raise RuntimeError("dpm_solver_amd.MaskBlend runs on the GPU (no CPU fallback)")
        tf = float(t.detach().reshape(-1)[0].float().item()) if torch.is_tensor(t) else float(t)
        return self.apply(x, tf, step)

    def apply(self, x, t_host, step):
        """stand-alone launch with the time already on the host (no synchronisation)"""
        xc = x.contiguous()
        m, period, a, b, alpha, sigma = self.operands(xc.shape, xc.dtype, xc.device, t_host, step)
        out = torch.empty_like(xc)
        p = lambda v: None if v is None else C.c_void_p(v.data_ptr())
        with torch.cuda.device(x.device):
            L.check(L.lib.dpm_blend_launch(p(xc), p(m), p(a), p(b), alpha, sigma, p(out), xc.numel(), period,
                                           _DT[xc.dtype], C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return out
