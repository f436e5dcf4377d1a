"""`dpm_solver` as the reference's example applications import it, backed by dpm_solver_amd."""
from dpm_solver_amd import DPM_Solver, NoiseScheduleVP, expand_dims, interpolate_fn, model_wrapper  # noqa: F401
from dpm_solver_amd.adapters import DPMSolverSampler  # noqa: F401
