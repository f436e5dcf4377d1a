"""Drop-in module: `from dpm_solver_pytorch import NoiseScheduleVP, model_wrapper, DPM_Solver`
(the import line of the reference's README.md:380) now resolves to the MI355X-native engine."""
from dpm_solver_amd import DPM_Solver, NoiseScheduleVP, expand_dims, interpolate_fn, model_wrapper  # noqa: F401
