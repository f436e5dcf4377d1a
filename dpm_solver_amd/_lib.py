"""ctypes binding of the C ABI declared in include/dpm_hip.h.

The shared library is built in-tree by `__graft_entry__.build()` (hipcc, --offload-arch=gfx950) as
dpm_solver_amd/libdpm_hip.so.  There is no fallback: if it is missing the import fails loudly.
"""
import ctypes as C
import os

# torch FIRST: PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
# libdpm_hip.so needs "libamdhip64.so.7"; with torch's copy already mapped the dynamic loader binds to it by
# SONAME, so kernels, streams and allocations of both sides live in ONE runtime.  Loaded the other way round
# the process ends up with two runtimes and every launch fails with hipErrorNoDevice.
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPM_SOLVER_AMD_LIB: another build of the same sources -- the LAB build (tools/_variants/lab/libdpm_lab.so: tuning knobs,
# fault injection, event-bracketed launches, include/dpm_lab.h) that tools/ and the lab-marked tests run on, or the
# escape-hatch build; unset in normal use
LIB_PATH = os.environ.get("DPM_SOLVER_AMD_LIB") or os.path.join(_HERE, "libdpm_hip.so")
LAB_LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "_variants", "lab", "libdpm_lab.so")

# ---- enumerations (mirror include/dpm_hip.h) --------------------------------------------------
DPM_OK = 0
ERR_ARG, ERR_UNSUPPORTED, ERR_ALIGN, ERR_NOMEM, ERR_CALLBACK, ERR_FAULT = -1, -2, -3, -4, -5, -6
ALGO = {"dpmsolver": 0, "dpmsolver++": 1}
SOLVER = {"dpmsolver": 0, "taylor": 1}
METHOD = {"multistep": 0, "singlestep": 1, "singlestep_fixed": 2}
SKIP = {"time_uniform": 0, "logSNR": 1, "time_quadratic": 2}
MODEL = {"noise": 0, "x_start": 1, "v": 2, "score": 3}
GUIDE = {"uncond": 0, "classifier-free": 1, "classifier": 2}
DTYPE_F32, DTYPE_F16, DTYPE_BF16, DTYPE_F64 = 0, 1, 2, 3
EVAL_LOG_ALPHA, EVAL_ALPHA, EVAL_STD, EVAL_LAMBDA, EVAL_INV_LAMBDA = 0, 1, 2, 3, 4
FORM_LIN1, FORM_TWO, FORM_MS3, FORM_SS3T, FORM_DENOISE = 0, 1, 2, 3, 4
F_TO_X0, F_STORE_M, F_BASE_HIST, F_THRESH, F_USER_X0, F_BLEND = 1, 2, 4, 8, 16, 32
SRC_STATE, SRC_TMP = 0, 1
# knobs of the LAB build (include/dpm_lab.h: dpm_tuning_set / dpm_tuning_get; the product library has none)
TUNE_UNROLL, TUNE_NONTEMPORAL, TUNE_BLOCKS_PER_CU, TUNE_ASSUME_RESIDENT = 0, 1, 2, 3
TUNE_MULTI_FUSE, TUNE_MULTI_BLOCKS_PER_CU, TUNE_CLUSTER_IN_GRAPH, TUNE_CLUSTER_ONE_HOP = 4, 5, 6, 7
TUNE_MULTI_XCD_REMAP = 8
TUNE_THR_PREDICT = 9
TUNE_THR_SPIN_LIMIT, TUNE_THR_DEBUG_FAULT, TUNE_BLOCK_THREADS = 10, 11, 12
TUNE_FORCE_GENERIC, TUNE_THR_ELECT, TUNE_LDS_DMA, TUNE_THR_STAGGER, TUNE_BIG_TILES = 13, 14, 15, 16, 17
MULTI_MAX = 32
THR_HINT_WORDS = 4


class Stage(C.Structure):
    _fields_ = [
        ("index", C.c_int32), ("form", C.c_int32), ("flags", C.c_uint32), ("model_type", C.c_int32),
        ("guidance", C.c_int32), ("outer_step", C.c_int32), ("emits_state", C.c_int32),
        ("x_src", C.c_int32), ("xe_src", C.c_int32), ("h1_slot", C.c_int32), ("h2_slot", C.c_int32),
        ("m_slot", C.c_int32),
        ("t_eval", C.c_float), ("t_input", C.c_float), ("t_out", C.c_float),
        ("alpha_e", C.c_float), ("sigma_e", C.c_float), ("cfg_scale", C.c_float), ("cg_scale", C.c_float),
        ("cx", C.c_float), ("c0", C.c_float), ("c1", C.c_float), ("c2", C.c_float),
        ("k", C.c_float * 5), ("thr_ratio", C.c_float), ("thr_max", C.c_float),
        ("blend_alpha", C.c_float), ("blend_sigma", C.c_float),
    ]

    def copy(self):
        s = Stage()
        C.memmove(C.byref(s), C.byref(self), C.sizeof(Stage))
        return s


class StageF64(C.Structure):
    """dpm_stage_f64: the float fields of a stage in double (double-precision plans)"""
    _fields_ = [(n, C.c_double) for n in ("t_eval", "t_input", "t_out", "alpha_e", "sigma_e", "cfg_scale", "cg_scale",
                                          "cx", "c0", "c1", "c2")] + [("k", C.c_double * 5)] + \
               [(n, C.c_double) for n in ("thr_ratio", "thr_max", "blend_alpha", "blend_sigma")] + \
               [("time_f64", C.c_int32), ("reserved", C.c_int32)]


class LaunchOpts(C.Structure):
    """dpm_launch_opts: what a caller may choose per call (zero = defaults)"""
    _fields_ = [("cluster_in_graph", C.c_int32), ("no_fuse", C.c_int32), ("thr_spin_limit", C.c_int32),
                ("reserved", C.c_int32 * 5)]


class Buffers(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("xe", C.c_void_p), ("e0", C.c_void_p), ("e1", C.c_void_p), ("g", C.c_void_p),
        ("h1", C.c_void_p), ("h2", C.c_void_p), ("x_out", C.c_void_p), ("m_out", C.c_void_p),
        ("workspace", C.c_void_p), ("n", C.c_int64), ("batch", C.c_int64),
        ("state_dtype", C.c_int32), ("eps_dtype", C.c_int32),
        ("x_out2", C.c_void_p), ("eps_stride", C.c_int64), ("mask", C.c_void_p), ("blend_a", C.c_void_p),
        ("blend_b", C.c_void_p), ("mask_period", C.c_int64),
        ("inputs_resident", C.c_int32), ("reserved", C.c_int32), ("thr_hint", C.c_void_p),
        ("opts", C.POINTER(LaunchOpts)), ("coef64", C.POINTER(StageF64)),
    ]


class PlanDesc(C.Structure):
    _fields_ = [
        ("algorithm_type", C.c_int32), ("method", C.c_int32), ("order", C.c_int32), ("steps", C.c_int32),
        ("skip_type", C.c_int32), ("solver_type", C.c_int32), ("lower_order_final", C.c_int32),
        ("denoise_to_zero", C.c_int32), ("model_type", C.c_int32), ("guidance", C.c_int32),
        ("thresholding", C.c_int32), ("precision", C.c_int32),
        ("t_start", C.c_double), ("t_end", C.c_double), ("guidance_scale", C.c_double),
        ("thr_ratio", C.c_double), ("thr_max", C.c_double),
    ]


class RunBuffers(C.Structure):
    _fields_ = [
        ("xbuf", C.c_void_p * 4), ("hist", C.c_void_p * 3), ("e0", C.c_void_p), ("e1", C.c_void_p),
        ("workspace", C.c_void_p), ("n", C.c_int64), ("batch", C.c_int64),
        ("state_dtype", C.c_int32), ("eps_dtype", C.c_int32),
        ("eps_stride", C.c_int64), ("dup_state", C.c_int32), ("reserved", C.c_int32), ("thr_hint", C.c_void_p),
        ("opts", C.POINTER(LaunchOpts)),
    ]


class AdaptiveDesc(C.Structure):
    _fields_ = [
        ("algorithm_type", C.c_int32), ("solver_type", C.c_int32), ("order", C.c_int32), ("model_type", C.c_int32),
        ("guidance", C.c_int32), ("reserved", C.c_int32),
        ("guidance_scale", C.c_double), ("t_start", C.c_double), ("t_end", C.c_double), ("h_init", C.c_double),
        ("atol", C.c_double), ("rtol", C.c_double), ("theta", C.c_double), ("t_err", C.c_double),
    ]


MODEL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Stage), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)

# every symbol include/dpm_hip.h declares: (name, restype, argtypes)
_P = C.POINTER
_SIGNATURES = [
    ("dpm_schedule_create_betas_f32", C.c_int, [_P(C.c_float), C.c_int, C.c_int, _P(C.c_void_p)]),
    ("dpm_schedule_create_betas_f64", C.c_int, [_P(C.c_double), C.c_int, C.c_int, _P(C.c_void_p)]),
    ("dpm_schedule_create_alphas_cumprod_f32", C.c_int, [_P(C.c_float), C.c_int, C.c_int, _P(C.c_void_p)]),
    ("dpm_schedule_create_alphas_cumprod_f64", C.c_int, [_P(C.c_double), C.c_int, C.c_int, _P(C.c_void_p)]),
    ("dpm_schedule_create_log_alpha", C.c_int, [_P(C.c_float), C.c_int, _P(C.c_void_p)]),
    ("dpm_numerical_clip_len_f32", C.c_int, [_P(C.c_float), C.c_int, C.c_double, _P(C.c_int)]),
    ("dpm_numerical_clip_len_f64", C.c_int, [_P(C.c_double), C.c_int, C.c_double, _P(C.c_int)]),
    ("dpm_schedule_create_linear", C.c_int, [C.c_double, C.c_double, _P(C.c_void_p)]),
    ("dpm_schedule_create_cosine", C.c_int, [_P(C.c_void_p)]),
    ("dpm_schedule_set_table_dtype", C.c_int, [C.c_void_p, C.c_int]),
    ("dpm_schedule_tables_f64", C.c_int, [C.c_void_p, _P(_P(C.c_double)), _P(_P(C.c_double)), _P(C.c_int)]),
    ("dpm_schedule_eval_f64", C.c_int, [C.c_void_p, C.c_int, _P(C.c_double), C.c_int, _P(C.c_double)]),
    ("dpm_schedule_destroy", None, [C.c_void_p]),
    ("dpm_schedule_is_discrete", C.c_int, [C.c_void_p]),
    ("dpm_schedule_total_N", C.c_int, [C.c_void_p]),
    ("dpm_schedule_tables", C.c_int, [C.c_void_p, _P(_P(C.c_float)), _P(_P(C.c_float)), _P(C.c_int)]),
    ("dpm_schedule_eval", C.c_int, [C.c_void_p, C.c_int, _P(C.c_float), C.c_int, _P(C.c_float)]),
    ("dpm_time_steps", C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, _P(C.c_float)]),
    ("dpm_singlestep_orders", C.c_int, [C.c_int, C.c_int, _P(C.c_int), _P(C.c_int)]),
    ("dpm_singlestep_grid", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                      _P(C.c_float), _P(C.c_int), _P(C.c_int)]),
    ("dpm_plan_create", C.c_int, [C.c_void_p, _P(PlanDesc), _P(C.c_void_p)]),
    ("dpm_plan_destroy", None, [C.c_void_p]),
    ("dpm_plan_num_stages", C.c_int, [C.c_void_p]),
    ("dpm_plan_num_slots", C.c_int, [C.c_void_p]),
    ("dpm_plan_stage", C.c_int, [C.c_void_p, C.c_int, _P(Stage)]),
    ("dpm_plan_stage_f64", C.c_int, [C.c_void_p, C.c_int, _P(StageF64)]),
    ("dpm_plan_timesteps", C.c_int, [C.c_void_p, _P(C.c_float), C.c_int, _P(C.c_int)]),
    ("dpm_coef_first", C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, _P(Stage)]),
    ("dpm_coef_multistep", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_float), C.c_float, _P(Stage)]),
    ("dpm_coef_singlestep", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_double,
                                      C.c_double, C.c_int, _P(Stage)]),
    ("dpm_coef_prologue", C.c_int, [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_double, _P(Stage)]),
    ("dpm_coef_multistep_f64", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(C.c_double), C.c_double, C.c_int, _P(Stage),
                                         _P(StageF64)]),
    ("dpm_coef_singlestep_f64", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double,
                                          C.c_double, C.c_int, _P(Stage), _P(StageF64)]),
    ("dpm_coef_prologue_f64", C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, _P(Stage), _P(StageF64)]),
    ("dpm_stage_launch", C.c_int, [_P(Stage), _P(Buffers), C.c_void_p]),
    ("dpm_stage_launch_multi", C.c_int, [_P(Stage), _P(Buffers), C.c_int, C.c_void_p]),
    ("dpm_threshold_workspace_bytes", C.c_size_t, [C.c_int64, C.c_int64]),
    ("dpm_add_noise_launch", C.c_int, [C.c_void_p, _P(C.c_float), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_int, C.c_void_p]),
    ("dpm_add_noise_launch_f64", C.c_int, [C.c_void_p, _P(C.c_double), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p]),
    ("dpm_blend_launch", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                   C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    ("dpm_adaptive_error_launch", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                            C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    ("dpm_adaptive_create", C.c_int, [C.c_void_p, _P(AdaptiveDesc), _P(C.c_void_p)]),
    ("dpm_adaptive_destroy", None, [C.c_void_p]),
    ("dpm_adaptive_stage_template", C.c_int, [C.c_void_p, C.c_int, _P(Stage)]),
    ("dpm_adaptive_reset", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dpm_adaptive_begin", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("dpm_adaptive_stage_launch", C.c_int, [C.c_void_p, C.c_int, _P(Stage), _P(Buffers), C.c_void_p]),
    ("dpm_adaptive_error", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    ("dpm_adaptive_done_at", C.c_int, [C.c_void_p, C.c_int]),
    ("dpm_adaptive_poll", C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    ("dpm_plan_run", C.c_int, [C.c_void_p, _P(RunBuffers), C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_int)]),
    ("dpm_plan_run_multi", C.c_int, [C.c_void_p, _P(RunBuffers), C.c_int, C.c_void_p, _P(C.c_float), _P(C.c_int)]),
    ("dpm_graph_create", C.c_int, [C.c_void_p, _P(RunBuffers), C.c_void_p, C.c_void_p, C.c_void_p, _P(C.c_void_p)]),
    ("dpm_graph_launch", C.c_int, [C.c_void_p, C.c_void_p]),
    ("dpm_graph_result", C.c_int, [C.c_void_p]),
    ("dpm_graph_num_nodes", C.c_int, [C.c_void_p]),
    ("dpm_graph_destroy", None, [C.c_void_p]),
    ("dpm_cluster_timeout_poll", C.c_int, []),
    ("dpm_version", C.c_int, []),
    ("dpm_sizeof", C.c_size_t, [C.c_int]),
    ("dpm_last_error", C.c_char_p, []),
    ("dpm_device_info", C.c_int, [_P(C.c_int), _P(C.c_int), C.c_char_p, C.c_int]),
]
SYMBOLS = [s[0] for s in _SIGNATURES]


class FloorDesc(C.Structure):
    """dpm_floor_desc (include/dpm_lab.h)"""
    _fields_ = [("load_path", C.c_int32), ("rows", C.c_int32), ("block", C.c_int32), ("blocks_per_cu", C.c_int32),
                ("nt", C.c_int32), ("prio", C.c_int32), ("store", C.c_int32), ("reserved", C.c_int32)]


# what include/dpm_lab.h declares: exported by the LAB build only (bound when the loaded library is one)
_LAB_SIGNATURES = [
    ("dpm_lab_build", C.c_int, []),
    ("dpm_lab_device_context", C.c_void_p, [C.c_int]),
    ("dpm_stage_launch_timed", C.c_int, [_P(Stage), _P(Buffers), C.c_void_p, _P(C.c_float)]),
    ("dpm_trace_create", C.c_int, [C.c_int, _P(C.c_void_p)]),
    ("dpm_stage_launch_traced", C.c_int, [_P(Stage), _P(Buffers), C.c_void_p, C.c_void_p, C.c_int]),
    ("dpm_trace_read", C.c_int, [C.c_void_p, C.c_void_p, _P(C.c_float), C.c_int]),
    ("dpm_trace_destroy", None, [C.c_void_p]),
    ("dpm_prefetch_launch", C.c_int, [_P(C.c_void_p), _P(C.c_int64), C.c_int, C.c_int, C.c_void_p]),
    ("dpm_tuning_set", C.c_int, [C.c_int, C.c_int]),
    ("dpm_tuning_get", C.c_int, [C.c_int]),
    ("dpm_calib_launch", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p, _P(C.c_float)]),
    ("dpm_resident_create", C.c_int, [_P(Stage), _P(Buffers), C.c_int, C.c_int, C.c_int, _P(C.c_void_p)]),
    ("dpm_resident_start", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("dpm_resident_signal", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("dpm_resident_destroy", None, [C.c_void_p]),
    ("dpm_floor_launch", C.c_int, [_P(FloorDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p, _P(C.c_float)]),
    ("dpm_floor_launch_traced", C.c_int, [_P(FloorDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_int]),
    ("dpm_pagetouch_launch", C.c_int, [_P(C.c_void_p), _P(C.c_int64), C.c_int, C.c_int64, C.c_void_p]),
]
LAB_SYMBOLS = [s[0] for s in _LAB_SIGNATURES]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "dpm_solver_amd: %s not found.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the library is stale
        fn.restype = res
        fn.argtypes = args
    if hasattr(lib, "dpm_lab_build"):       # the lab build: everything above + include/dpm_lab.h
        for name, res, args in _LAB_SIGNATURES:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    return lib


lib = _load()
IS_LAB = hasattr(lib, "dpm_lab_build")
for _i, _t in enumerate((Stage, Buffers, PlanDesc, RunBuffers, AdaptiveDesc, LaunchOpts, StageF64)):  # the ctypes mirrors must match the compiled structs
    if lib.dpm_sizeof(_i) != C.sizeof(_t):
        raise ImportError("dpm_solver_amd: %s is %d bytes in _lib.py but %d in libdpm_hip.so -- stale library, rebuild"
                          % (_t.__name__, C.sizeof(_t), lib.dpm_sizeof(_i)))


if lib.dpm_version() < 201:
    raise ImportError("dpm_solver_amd: libdpm_hip.so reports version %d, this binding needs >= 201 -- stale library, rebuild"
                      % lib.dpm_version())


if IS_LAB and os.environ.get("DPM_LAB_TUNE"):
    # lab build only: knobs for a whole tool run, e.g. DPM_LAB_TUNE="thr_elect=1,thr_predict=0" (A/B runs under rocprofv3)
    for _kv in os.environ["DPM_LAB_TUNE"].split(","):
        _k, _v = _kv.split("=")
        if lib.dpm_tuning_set(globals()["TUNE_" + _k.strip().upper()], int(_v)) != 0:
            raise ImportError("DPM_LAB_TUNE: %s refused: %s" % (_kv, lib.dpm_last_error().decode()))


class DpmError(RuntimeError):
    pass


def require_lab(what="this tool"):
    """tools/ and the lab-marked tests: fail with the recipe when the loaded library is the product one"""
    if not IS_LAB:
        raise RuntimeError("%s needs the LAB build of the library (tuning knobs / fault injection / event-bracketed launches, "
                           "include/dpm_lab.h): run with DPM_SOLVER_AMD_LIB=%s (built by __graft_entry__.build())"
                           % (what, LAB_LIB_PATH))


def cluster_timeout_poll():
    """True when a wait between the workgroups of a thresholding cluster timed out since the last call (the kernel
    recovered: results are unaffected; see include/dpm_hip.h).  Diagnostics for shared-GPU deployments."""
    return bool(lib.dpm_cluster_timeout_poll())


def check(rc):
    """Map a C status to the reference's exception convention: argument errors are
    ValueError, everything else RuntimeError."""
    if rc == DPM_OK:
        return
    msg = lib.dpm_last_error().decode("utf-8", "replace")
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise DpmError("dpm_hip error %d: %s" % (rc, msg))
