"""The C-ABI library loads without a GPU and exports every function include/dpm_hip.h declares."""
import ctypes
import os
import re

import dpm_solver_amd
from dpm_solver_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "dpm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|size_t|char\s*\*|const char\s*\*)\s+\**(dpm_[A-Za-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_functions_are_exported_and_bound():
    decl = declared_functions()
    assert len(decl) >= 30, decl
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), "libdpm_hip.so does not export %s" % name
    assert sorted(L.SYMBOLS) == decl, (set(decl) ^ set(L.SYMBOLS))


def test_struct_layouts_match_header():
    # sizes are part of the ABI: the ctypes mirrors against sizeof() as compiled, and against the header by count
    for i, t in enumerate((L.Stage, L.Buffers, L.PlanDesc, L.RunBuffers, L.AdaptiveDesc)):
        assert L.lib.dpm_sizeof(i) == ctypes.sizeof(t), t.__name__
    assert ctypes.sizeof(L.Stage) == 12 * 4 + 20 * 4                       # 12 int32 + 20 float
    assert ctypes.sizeof(L.Buffers) == 15 * 8 + 4 * 8 + 4 * 4              # 15 pointers, 4 int64, 4 int32
    assert ctypes.sizeof(L.PlanDesc) == 12 * 4 + 5 * 8
    assert ctypes.sizeof(L.RunBuffers) == 10 * 8 + 2 * 8 + 2 * 4 + 8 + 2 * 4 + 8      # + thr_hint
    assert ctypes.sizeof(L.AdaptiveDesc) == 6 * 4 + 8 * 8


def test_version_and_error_text():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'dpm_hip.h')).read()
    import re
    assert L.lib.dpm_version() == int(re.search(r'#define DPM_HIP_VERSION (\d+)', hdr).group(1)) >= 102
    rc = L.lib.dpm_time_steps(None, 0, 1.0, 0.001, 5, None)
    assert rc == L.ERR_ARG and b"time_steps" in L.lib.dpm_last_error()


def test_package_reports_library_path():
    assert os.path.exists(dpm_solver_amd.LIB_PATH)


def test_reference_import_paths_resolve_to_the_engine():
    """README.md:380 and the vendored paths of the example apps"""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "shims"))
    try:
        for name in ("dpm_solver", "dpm_solver.sampler", "dpm_solver.dpm_solver", "dpm_solver.legacy"):
            sys.modules.pop(name, None)
        m = importlib.import_module("dpm_solver")
        assert m.DPM_Solver is dpm_solver_amd.DPM_Solver and m.NoiseScheduleVP is dpm_solver_amd.NoiseScheduleVP
        s = importlib.import_module("dpm_solver.sampler")
        assert s.model_wrapper is dpm_solver_amd.model_wrapper and s.DPMSolverSampler.__name__ == "DPMSolverSampler"
        assert importlib.import_module("dpm_solver.dpm_solver").DPM_Solver is dpm_solver_amd.DPM_Solver
        assert importlib.import_module("dpm_solver.legacy").NoiseScheduleVP is dpm_solver_amd.LegacyNoiseScheduleVP
        root = importlib.import_module("dpm_solver_pytorch")
        assert root.DPM_Solver is dpm_solver_amd.DPM_Solver
    finally:
        sys.path.remove(os.path.join(ROOT, "shims"))
        for name in ("dpm_solver", "dpm_solver.sampler", "dpm_solver.dpm_solver", "dpm_solver.legacy"):
            sys.modules.pop(name, None)
