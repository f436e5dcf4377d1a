"""`dpm_solver.dpm_solver`: the module the Stable-Diffusion adapter imports with `from .dpm_solver import …`
(ldm/models/diffusion/dpm_solver/sampler.py:5)."""
from dpm_solver_amd import DPM_Solver, NoiseScheduleVP, expand_dims, interpolate_fn, model_wrapper  # noqa: F401
