"""Host-side mirrors of the reference's caller adapters (the code on the caller's side of the solver path)."""
from .stable_diffusion import DPMSolverSampler  # noqa: F401
from .guided_diffusion import sample_image as guided_diffusion_sample_image  # noqa: F401
from .score_sde import get_dpm_solver_sampler as score_sde_get_dpm_solver_sampler  # noqa: F401
