"""The older revision vendored under examples/score_sde_pytorch ('cosine' schedule, unclipped discrete tables)."""
from dpm_solver_amd import DPM_Solver, model_wrapper  # noqa: F401
from dpm_solver_amd import LegacyNoiseScheduleVP as NoiseScheduleVP  # noqa: F401
