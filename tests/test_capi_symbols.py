"""The C-ABI library loads without a GPU and exports every function include/dpm_hip.h declares -- and nothing else: the
tuning knobs, fault injection, event-bracketed launches, calibration kernels and experiments live in the LAB build
(include/dpm_lab.h, tools/_variants/lab/libdpm_lab.so), which exports both headers' functions."""
import ctypes
import os
import re
import subprocess

import pytest

import dpm_solver_amd
from dpm_solver_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# The product ABI, pinned: a new entry point is a decision (add it here and to include/dpm_hip.h in the same change).
PRODUCT_ABI = """
dpm_adaptive_begin dpm_adaptive_create dpm_adaptive_destroy dpm_adaptive_done_at dpm_adaptive_error
dpm_adaptive_error_launch dpm_adaptive_poll dpm_adaptive_reset dpm_adaptive_stage_launch dpm_adaptive_stage_template
dpm_add_noise_launch dpm_add_noise_launch_f64 dpm_blend_launch dpm_cluster_timeout_poll dpm_coef_first dpm_coef_multistep dpm_coef_multistep_f64
dpm_coef_prologue dpm_coef_prologue_f64 dpm_coef_singlestep dpm_coef_singlestep_f64 dpm_device_info dpm_graph_create dpm_graph_destroy dpm_graph_launch dpm_graph_num_nodes
dpm_graph_result dpm_last_error dpm_numerical_clip_len_f32 dpm_numerical_clip_len_f64 dpm_plan_create dpm_plan_destroy
dpm_plan_num_slots dpm_plan_num_stages dpm_plan_run dpm_plan_run_multi dpm_plan_stage dpm_plan_stage_f64 dpm_plan_timesteps
dpm_schedule_create_alphas_cumprod_f32 dpm_schedule_create_alphas_cumprod_f64 dpm_schedule_create_betas_f32
dpm_schedule_create_betas_f64 dpm_schedule_create_cosine dpm_schedule_create_linear dpm_schedule_create_log_alpha
dpm_schedule_destroy dpm_schedule_eval dpm_schedule_eval_f64 dpm_schedule_is_discrete dpm_schedule_set_table_dtype
dpm_schedule_tables dpm_schedule_tables_f64 dpm_schedule_total_N
dpm_singlestep_grid dpm_singlestep_orders dpm_sizeof dpm_stage_launch dpm_stage_launch_multi
dpm_threshold_workspace_bytes dpm_time_steps dpm_version
""".split()
PRODUCT_LIB = os.path.join(ROOT, "dpm_solver_amd", "libdpm_hip.so")


def declared_functions(header="dpm_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:DPM_API\s+)?(?:const\s+)?(?:int|void|size_t|char\s*\*|const char\s*\*|const void\s*\*)\s+\**(dpm_[A-Za-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def exported(lib):
    """the library's WHOLE dynamic table (defined symbols): with -fvisibility=hidden + DPM_API + csrc/dpm_exports.map that is
    the C ABI name by name -- a mangled internal (`_Z...`), a libstdc++ instantiation or a data symbol fails the comparison"""
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, stdout=subprocess.PIPE, text=True).stdout
    rows = [l.split() for l in out.splitlines() if l.strip()]
    assert all(r[-2] == "T" for r in rows), [r for r in rows if r[-2] != "T"][:5]
    return sorted(r[-1] for r in rows)


def test_header_functions_are_exported_and_bound():
    decl = declared_functions()
    assert len(decl) >= 30, decl
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), "libdpm_hip.so does not export %s" % name
    assert sorted(L.SYMBOLS) == decl, (set(decl) ^ set(L.SYMBOLS))
    assert decl == sorted(PRODUCT_ABI), (set(decl) ^ set(PRODUCT_ABI))


def test_product_library_exports_the_pinned_abi_and_nothing_else():
    """no tuning knob, no fault-injection switch, no calibration kernel, no experiment in the shipped library"""
    got = exported(PRODUCT_LIB)
    assert got == sorted(PRODUCT_ABI), (set(got) ^ set(PRODUCT_ABI))
    assert len(got) == 62 and not [n for n in got if n.startswith("_Z")]
    lab_only = declared_functions("dpm_lab.h")
    assert lab_only and not (set(got) & set(lab_only))
    # ... and no process-global tuning state either: the lab build's knobs live in a variable the product does not have
    syms = subprocess.run(["nm", "-D", PRODUCT_LIB], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "g_lab_tuning" not in syms and "g_tuning" not in syms


def test_lab_library_exports_both_headers():
    if not os.path.exists(L.LAB_LIB_PATH):
        pytest.skip("no lab build next to the library (__graft_entry__.build() makes it)")
    got = exported(L.LAB_LIB_PATH)
    want = sorted(set(declared_functions()) | set(declared_functions("dpm_lab.h")))
    assert got == want, (set(got) ^ set(want))
    assert sorted(L.LAB_SYMBOLS) == declared_functions("dpm_lab.h")


def test_struct_layouts_match_header():
    # sizes are part of the ABI: the ctypes mirrors against sizeof() as compiled, and against the header by count
    for i, t in enumerate((L.Stage, L.Buffers, L.PlanDesc, L.RunBuffers, L.AdaptiveDesc, L.LaunchOpts, L.StageF64)):
        assert L.lib.dpm_sizeof(i) == ctypes.sizeof(t), t.__name__
    assert ctypes.sizeof(L.Stage) == 12 * 4 + 20 * 4                       # 12 int32 + 20 float
    assert ctypes.sizeof(L.Buffers) == 17 * 8 + 4 * 8 + 4 * 4              # 17 pointers (+ opts, coef64), 4 int64, 4 int32
    assert ctypes.sizeof(L.StageF64) == 20 * 8 + 2 * 4
    assert ctypes.sizeof(L.PlanDesc) == 12 * 4 + 5 * 8
    assert ctypes.sizeof(L.RunBuffers) == 10 * 8 + 2 * 8 + 2 * 4 + 8 + 2 * 4 + 8 + 8  # + thr_hint, opts
    assert ctypes.sizeof(L.LaunchOpts) == 8 * 4
    assert ctypes.sizeof(L.AdaptiveDesc) == 6 * 4 + 8 * 8


def test_version_and_error_text():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'dpm_hip.h')).read()
    import re
    assert L.lib.dpm_version() == int(re.search(r'#define DPM_HIP_VERSION (\d+)', hdr).group(1)) >= 201
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
