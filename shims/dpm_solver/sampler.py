"""`dpm_solver.sampler`: guided-diffusion imports the solver classes from here (runners/diffusion.py:595), the
Stable-Diffusion tree its DPMSolverSampler (ldm/models/diffusion/dpm_solver/sampler.py)."""
from dpm_solver_amd import DPM_Solver, NoiseScheduleVP, model_wrapper  # noqa: F401
from dpm_solver_amd.adapters import DPMSolverSampler  # noqa: F401
