"""Host-side mirrors of the reference's caller adapters (the code on the caller's side of the solver path)."""
from .stable_diffusion import DPMSolverSampler  # noqa: F401
