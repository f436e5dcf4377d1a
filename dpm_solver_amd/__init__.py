"""dpm_solver_amd -- MI355X-native DPM-Solver / DPM-Solver++ sampling engine.

Drop-in for the reference module `dpm_solver_pytorch` (LuChengTHU/dpm-solver):

    from dpm_solver_amd import NoiseScheduleVP, model_wrapper, DPM_Solver

The classes keep the reference's signatures; the per-step update runs as one fused HIP kernel per
solver stage behind the C ABI of include/dpm_hip.h (dpm_solver_amd/libdpm_hip.so).
"""
from ._lib import LIB_PATH, DpmError  # noqa: F401  (import fails loudly if the HIP library is missing)
from .schedule import LegacyNoiseScheduleVP, NoiseScheduleVP
from .wrapper import WrappedModel, model_wrapper
from .solver import DPM_Solver, GraphedSample
from .correctors import MaskBlend
from .utils import expand_dims, interpolate_fn

__all__ = ["NoiseScheduleVP", "model_wrapper", "DPM_Solver", "WrappedModel", "interpolate_fn", "expand_dims",
           "MaskBlend", "GraphedSample", "LegacyNoiseScheduleVP"]
__version__ = "0.1.0"
