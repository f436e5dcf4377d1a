"""tools/ run on the LAB build of the library (tools/_variants/lab/libdpm_lab.so: tuning knobs, event-bracketed launches,
calibration / floor kernels -- include/dpm_lab.h); the product library has none of that.  Import this module BEFORE
dpm_solver_amd: it points DPM_SOLVER_AMD_LIB at the lab build unless the caller chose a library already."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB_LIB = os.path.join(ROOT, "tools", "_variants", "lab", "libdpm_lab.so")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if "dpm_solver_amd" in sys.modules:
    raise ImportError("import tools/_lab.py before dpm_solver_amd")
if not os.environ.get("DPM_SOLVER_AMD_LIB"):
    if not os.path.exists(LAB_LIB):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % LAB_LIB)
    os.environ["DPM_SOLVER_AMD_LIB"] = LAB_LIB
