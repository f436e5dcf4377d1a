/*
 * dpm_hip.h -- C ABI of the MI355X-native DPM-Solver / DPM-Solver++ sampling engine.
 *
 * The reference (LuChengTHU/dpm-solver, dpm_solver_pytorch.py) has no FFI boundary: its "plugin
 * interface" is the Python API  NoiseScheduleVP / model_wrapper / DPM_Solver  (README.md:380).
 * This header is the boundary a host in any language binds to *under* that API; the Python
 * mirror in dpm_solver_amd/ is one such host (ctypes).
 *
 * Design:
 *   - every scalar of the noise schedule (alpha_t, sigma_t, lambda_t, h, r, phi_k) is computed
 *     ONCE on the host by the planner, in the reference's own fp32 operation order, and frozen
 *     into a list of `dpm_stage` records -- one per network evaluation;
 *   - the device runs exactly one fused streaming kernel per stage:
 *        raw network output(s) -> [CFG blend | classifier term] -> [x_start/v/score -> eps]
 *        -> [eps -> x0] -> [dynamic thresholding] -> exponential-integrator update
 *     reading the state x, the fresh output(s) and 0-2 cached model values, writing x_next and
 *     (if a later stage needs it) the new model value;
 *   - the library never allocates or frees user-visible device memory: every buffer is the
 *     caller's, every launch is asynchronous on the caller's HIP stream, nothing synchronises.
 *
 * Conventions: plain C types only.  Return value 0 = DPM_OK, negative = argument/state error
 * (dpm_last_error() has the text), positive = a hipError_t from the HIP runtime.  No function
 * throws or aborts.  Handles are immutable after creation and may be shared between threads.
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).
 */
#ifndef DPM_HIP_H
#define DPM_HIP_H

#include <stddef.h>
#include <stdint.h>

/* The library is compiled with -fvisibility=hidden: DPM_API marks the entry points, which are then the ONLY symbols in its
   dynamic table (tests/test_capi_symbols.py compares `nm -D --defined-only` with this header name by name) -- no mangled
   internal can collide with a symbol of the host process or of a second library it links. */
#ifndef DPM_API
#define DPM_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* major*10000 + minor*100 + patch.  History: 100 rounds 1-2; 101 dpm_buffers / dpm_run_buffers gained the trailing
   `thr_hint` pointer; 102 dpm_cluster_timeout_poll, DPM_TUNE_THR_SPIN_LIMIT / _DEBUG_FAULT, DPM_ERR_FAULT retired;
   103 DPM_TUNE_BLOCK_THREADS, dpm_calib_launch kind 2 (no struct changed);
   200 (round 5) the measurement / tuning / experiment entry points left this header and the product library: dpm_tuning_*,
   dpm_calib_launch, dpm_prefetch_launch, dpm_resident_*, dpm_trace_*, dpm_stage_launch_timed and dpm_plan_run_timed are
   declared in include/dpm_lab.h and exported by the lab build only (libdpm_lab.so, the same sources with -DDPM_LAB=1);
   what a caller may legitimately choose per call travels in dpm_launch_opts (dpm_buffers.opts / dpm_run_buffers.opts);
   DPM_DTYPE_F64 (double-precision state, NoiseScheduleVP(dtype=torch.float64) callers).  The product library has no
   process-global mutable state: per-device contexts only (the chain of clustered launches, the diagnostics word).
   201 (round 6) dpm_add_noise_launch_f64 (double times on double tensors); -fvisibility=hidden + DPM_API: the dynamic symbol
   table is this header's functions and nothing else.
   The structs grow at their END only.  A host MUST zero-initialise every struct it passes (memset / = {0}: new trailing
   fields then read as "absent") and SHOULD check at load time that dpm_version() >= the version it was built against and
   that dpm_sizeof(DPM_SIZEOF_*) == its own sizeof() -- a host compiled against an older header passes shorter structs,
   and the library would read past their end (examples/native_host.c and dpm_solver_amd/_lib.py do both checks). */
#define DPM_HIP_VERSION 201

/* ---- status --------------------------------------------------------------------------- */
enum {
  DPM_OK = 0,
  DPM_ERR_ARG = -1,         /* bad argument value (ValueError at the Python layer)            */
  DPM_ERR_UNSUPPORTED = -2, /* valid in the reference, not (yet) built here -- never silent   */
  DPM_ERR_ALIGN = -3,       /* a buffer is not aligned as the entry point requires (lab entry points only)   */
  DPM_ERR_NOMEM = -4,
  DPM_ERR_CALLBACK = -5,    /* the model callback of dpm_plan_run returned non-zero           */
  DPM_ERR_FAULT = -6        /* retired (version 102): a clustered thresholding launch whose wait on a peer
                               workgroup times out now recovers inside the kernel -- the workgroup computes
                               the sample's order statistics alone, results unchanged -- and no entry point
                               returns this code any more; see dpm_cluster_timeout_poll                      */
};

/* ---- enumerations (values are ABI) ----------------------------------------------------- */
enum { DPM_ALGO_DPMSOLVER = 0, DPM_ALGO_DPMSOLVERPP = 1 };        /* algorithm_type, ref :342,:406 */
enum { DPM_SOLVER_DPMSOLVER = 0, DPM_SOLVER_TAYLOR = 1 };         /* solver_type,    ref :611      */
enum { DPM_METHOD_MULTISTEP = 0, DPM_METHOD_SINGLESTEP = 1, DPM_METHOD_SINGLESTEP_FIXED = 2 }; /* ref :1171,:1214 */
enum { DPM_SKIP_TIME_UNIFORM = 0, DPM_SKIP_LOGSNR = 1, DPM_SKIP_TIME_QUADRATIC = 2 };          /* ref :468-478    */
enum { DPM_MODEL_NOISE = 0, DPM_MODEL_X_START = 1, DPM_MODEL_V = 2, DPM_MODEL_SCORE = 3 };     /* ref :288-298    */
enum { DPM_GUIDE_NONE = 0, DPM_GUIDE_CFG = 1, DPM_GUIDE_CLASSIFIER = 2 };                      /* ref :313-330    */
enum { DPM_DTYPE_F32 = 0, DPM_DTYPE_F16 = 1, DPM_DTYPE_BF16 = 2,
       DPM_DTYPE_F64 = 3 /* state AND network output in double, double arithmetic: the reference run on x.double()
                            (ref :14, :105-107); one run-time dispatched kernel, not a performance path */ };
enum { DPM_EVAL_LOG_ALPHA = 0, DPM_EVAL_ALPHA = 1, DPM_EVAL_STD = 2, DPM_EVAL_LAMBDA = 3, DPM_EVAL_INV_LAMBDA = 4 };

/* update forms.  `mn` = model value produced by this stage's prologue, h1/h2 = cached values.  */
enum {
  DPM_FORM_LIN1 = 0,    /* out = cx*x - c0*mn                                         (ref :573,:585)        */
  DPM_FORM_TWO = 1,     /* D = k0*(mn-h1); out = (cx*x - c0*P) - c1*D, P = BASE_HIST ? h1 : mn
                           (ref :636-646,:659-669,:728-739,:767-778,:827-851)                               */
  DPM_FORM_MS3 = 2,     /* multistep third order (ref :879-903)                                              */
  DPM_FORM_SS3T = 3,    /* singlestep third order, 'taylor' final combination (ref :741-750,:780-789)        */
  DPM_FORM_DENOISE = 4, /* out = mn  (denoise_to_zero, ref :541-545,:1235-1237)                              */
  DPM_FORM_COUNT = 5
};

/* dpm_stage.flags */
#define DPM_F_TO_X0 1u      /* prologue converts eps -> x0 = (xe - sigma*eps)/alpha   (ref :439)              */
#define DPM_F_STORE_M 2u    /* write mn to m_out (a later stage reads it as h1/h2)                            */
#define DPM_F_BASE_HIST 4u  /* FORM_TWO: first-order term uses h1 (singlestep) instead of mn (multistep)      */
#define DPM_F_THRESH 8u     /* dynamic thresholding of x0 (ref :416-425)                                      */
#define DPM_F_USER_X0 16u   /* host applies a callable correcting_x0_fn: stage is split by the host shim       */
#define DPM_F_BLEND 32u     /* epilogue: x_out <- x_out*mask + (1-mask)*(blend_alpha*blend_a + blend_sigma*blend_b),
                               the mask blend DiffEdit / inpainting callers run as correcting_xt_fn after every
                               update (scripts/diffedit_inpaint.ipynb cell 6; hook: ref :1180,:1188,:1203,:1229)     */

/* buffer roles for the host-side loop */
enum { DPM_SRC_STATE = 0, DPM_SRC_TMP = 1 };

/* ---- one stage = one network evaluation + one fused kernel ----------------------------- */
typedef struct dpm_stage {
  int32_t index;       /* position in the plan                                                   */
  int32_t form;        /* DPM_FORM_*                                                             */
  uint32_t flags;      /* DPM_F_*                                                                */
  int32_t model_type;  /* DPM_MODEL_*  : conversion applied to the raw network output            */
  int32_t guidance;    /* DPM_GUIDE_*                                                            */
  int32_t outer_step;  /* `step` handed to correcting_xt_fn(x, t, step) for this stage's output  */
  int32_t emits_state; /* 1: x_out is a solver state x_i (goes to `intermediates`), 0: mid-stage */
  int32_t x_src;       /* DPM_SRC_*: which buffer is the update's x                              */
  int32_t xe_src;      /* DPM_SRC_*: which buffer the network was evaluated on                   */
  int32_t h1_slot;     /* history slot read as h1, -1 = none                                     */
  int32_t h2_slot;     /* history slot read as h2, -1 = none                                     */
  int32_t m_slot;      /* history slot written with mn, -1 = none                                */
  float t_eval;        /* continuous time of the network evaluation (fp32 as the reference)      */
  float t_input;       /* model time label: (t - 1/N)*1000 for discrete schedules (ref :278)     */
  float t_out;         /* continuous time of x_out                                               */
  float alpha_e;       /* alpha(t_eval)                                                          */
  float sigma_e;       /* sigma(t_eval)                                                          */
  float cfg_scale;     /* guidance_scale as fp32 (ref :330)                                      */
  float cg_scale;      /* fl(guidance_scale * sigma(t_eval))  (ref :321)                         */
  float cx, c0, c1, c2; /* update coefficients, reference association (see kernels)              */
  float k[5];          /* difference-quotient scalars: 1/r0, 1/r1, ...                           */
  float thr_ratio;     /* dynamic_thresholding_ratio                                             */
  float thr_max;       /* thresholding_max_val                                                   */
  float blend_alpha;   /* DPM_F_BLEND: alpha at the time the known image is noised to (add_noise, ref :1028)  */
  float blend_sigma;   /* DPM_F_BLEND: sigma at that time                                        */
} dpm_stage;

/* The float fields of dpm_stage in double: a double-precision run (state DPM_DTYPE_F64; dpm_plan_desc.precision = 1).
   The reference computes in whatever dtype torch's type promotion yields (ref :14, :105-107, :573-576): with a double state
   and NoiseScheduleVP(dtype=torch.float64) every scalar of a step is a double.  dpm_plan_stage_f64 returns these next to
   the dpm_stage of the same index (whose float fields are the doubles rounded); a launch finds them through
   dpm_buffers.coef64 (NULL: the launch converts the dpm_stage's floats exactly -- a double state on an fp32 schedule, where
   the reference's scalars ARE fp32 tensors). */
typedef struct dpm_stage_f64 {
  double t_eval, t_input, t_out, alpha_e, sigma_e, cfg_scale, cg_scale, cx, c0, c1, c2, k[5], thr_ratio, thr_max, blend_alpha,
      blend_sigma;
  int32_t time_f64;  /* dtype of the reference's time TENSORS in this run: bit 0 set = the tensor the network is called with
                        at t_eval is a double (singlestep inner nodes, logSNR grids: they come out of inverse_lambda on double
                        tables), clear = an fp32 tensor (torch.linspace grids, ref :472-477) and t_input was computed in fp32
                        (ref :278); bit 1: the same for t_out (the time handed to correcting_xt_fn)                       */
  int32_t reserved;
} dpm_stage_f64;

/* ---- per-call options (optional; NULL or all-zero = the defaults) ------------------------ */
typedef struct dpm_launch_opts {
  int32_t cluster_in_graph; /* 1: dynamic thresholding keeps its workgroup clusters under stream capture also for samples
                               that fit one workgroup (default 0: one workgroup per sample there -- a replayed graph runs
                               outside the library's per-device chain of clustered launches)                          */
  int32_t no_fuse;          /* 1: dpm_stage_launch_multi / dpm_plan_run_multi launch request by request (results are
                               identical; what a single request's stage costs next to the fused launch)               */
  int32_t thr_spin_limit;   /* > 0: polls (a microsecond or two each) before a wait on a cluster peer gives up and the
                               workgroup finishes its sample alone; 0 = the default, 4096                             */
  int32_t reserved[5];      /* zero                                                                                   */
} dpm_launch_opts;

/* ---- buffers of one launch ------------------------------------------------------------- */
typedef struct dpm_buffers {
  const void* x;    /* state the update starts from                          [n] state dtype     */
  const void* xe;   /* state the network saw (NULL = same as x)              [n] state dtype     */
  const void* e0;   /* raw network output (conditional half under CFG)       [n] eps dtype       */
  const void* e1;   /* raw unconditional output (CFG only)                   [n] eps dtype       */
  const void* g;    /* classifier gradient (classifier guidance only)        [n] eps dtype       */
  const void* h1;   /* cached model value                                    [n] state dtype     */
  const void* h2;   /* cached model value                                    [n] state dtype     */
  void* x_out;      /* result of the update                                  [n] state dtype     */
  void* m_out;      /* new model value (only if DPM_F_STORE_M)               [n] state dtype     */
  void* workspace;  /* thresholding scratch, dpm_threshold_workspace_bytes() bytes (may be NULL) */
  int64_t n;          /* total elements = batch * per_sample                                     */
  int64_t batch;      /* number of independent samples                                           */
  int32_t state_dtype; /* DPM_DTYPE_* of x, xe, h1, h2, x_out, m_out                             */
  int32_t eps_dtype;   /* DPM_DTYPE_* of e0, e1, g                                               */
  /* ---- optional extensions (zero / NULL = off) ---- */
  void* x_out2;        /* second copy of x_out: the other half of the [2B,...] network input under
                          classifier-free guidance (replaces torch.cat([x]*2), ref :326)   [n] state dtype */
  int64_t eps_stride;  /* elements between consecutive samples of e0 / e1 (0 = contiguous): the network output
                          is a channel slice out[:, :C] of a learned-variance model's [B,2C,H,W] output
                          (runners/diffusion.py:596-603); a sample's C*H*W elements stay contiguous      */
  const void* mask;    /* DPM_F_BLEND: mask, state dtype, mask_period elements, indexed i % mask_period  */
  const void* blend_a; /* DPM_F_BLEND: known image (or, with blend_b NULL, the state to blend in)  [n]   */
  const void* blend_b; /* DPM_F_BLEND: noise [n], NULL = blend_a is used as is                          */
  int64_t mask_period; /* n for a full mask, H*W for a [H,W] mask, C*H*W for a [1,C,H,W] mask           */
  int32_t inputs_resident; /* cache-policy hint.  0 (default): a network ran since x / the cached values were
                          written, the streams come from HBM -> streaming loads.  1: the immediately preceding launch
                          wrote them (frozen-model loops; dpm_plan_run sets it when there is no model callback) ->
                          default cache policy, they are expected in the 256 MiB Infinity Cache                  */
  int32_t reserved;
  float* thr_hint;     /* DPM_F_THRESH, optional (NULL = off): DPM_THR_HINT_WORDS floats per sample that persist from stage
                          to stage of ONE trajectory -- the kernel's private state (content undefined to the caller; no
                          initialisation needed: the stage with index 0 resets it).  [0], [1]: the selected order statistic
                          of |x0| in the previous two thresholded stages, from which a clustered launch predicts this
                          stage's select bound (a correct prediction saves the bound search and shrinks the exchange; a
                          wrong one is detected and costs one extra exchange, never exactness); [2]: route taken
                          (diagnostics: 1 predicted, 2 prediction rejected, 3 single exchange, 4 general); [3]: entries of
                          the last union gathered (diagnostics)                                                      */
  const dpm_launch_opts* opts; /* per-call options, NULL = defaults (version 200; dpm_stage_launch_multi reads bs[0].opts) */
  const dpm_stage_f64* coef64; /* DPM_DTYPE_F64 launches: the stage's scalars in double (NULL: the dpm_stage's floats, exactly) */
} dpm_buffers;

/* ---- noise schedule (NoiseScheduleVP, ref :6-167) --------------------------------------- */
typedef struct dpm_schedule dpm_schedule;

/* discrete-time schedules; `clip` != 0 applies numerical_clip_alpha at lambda = -5.1 (ref :114-125). */
DPM_API int dpm_schedule_create_betas_f32(const float* betas, int n, int clip, dpm_schedule** out);           /* ref :100 */
DPM_API int dpm_schedule_create_betas_f64(const double* betas, int n, int clip, dpm_schedule** out);
DPM_API int dpm_schedule_create_alphas_cumprod_f32(const float* ac, int n, int clip, dpm_schedule** out);      /* ref :103 */
DPM_API int dpm_schedule_create_alphas_cumprod_f64(const double* ac, int n, int clip, dpm_schedule** out);
DPM_API int dpm_schedule_create_log_alpha(const float* log_alpha, int n, dpm_schedule** out);                  /* ready table */
/* numerical_clip_alpha on its own (ref :114-125): *out_len = number of leading entries of `log_alphas` whose
   half-logSNR is >= clipped_lambda (the reference returns log_alphas[:out_len]); arithmetic in the array's type */
DPM_API int dpm_numerical_clip_len_f32(const float* log_alphas, int n, double clipped_lambda, int* out_len);
DPM_API int dpm_numerical_clip_len_f64(const double* log_alphas, int n, double clipped_lambda, int* out_len);
DPM_API int dpm_schedule_create_linear(double beta_0, double beta_1, dpm_schedule** out);                      /* ref :109-112 */
/* the continuous-time 'cosine' schedule of the older vendored revision (examples/score_sde_pytorch/dpm_solver.py
   :114-124,:134-137,:171-175): s = 0.008, T = 0.9946 (the caller's default end time) */
DPM_API int dpm_schedule_create_cosine(dpm_schedule** out);
/* NoiseScheduleVP(dtype=torch.float64) (ref :14, :105-107): the tables keep the double values they were computed in (the
   *_f64 constructors) instead of their fp32 roundings -- what double-precision plans and dpm_schedule_eval_f64 read.  Part of
   construction: call it right after dpm_schedule_create_*, before the handle is shared. */
DPM_API int dpm_schedule_set_table_dtype(dpm_schedule* s, int dtype /* DPM_DTYPE_F32 | DPM_DTYPE_F64 */);
DPM_API int dpm_schedule_tables_f64(const dpm_schedule* s, const double** log_alpha, const double** t_array, int* K);
DPM_API void dpm_schedule_destroy(dpm_schedule* s);
DPM_API int dpm_schedule_is_discrete(const dpm_schedule* s);
DPM_API int dpm_schedule_total_N(const dpm_schedule* s);                                                       /* ref :106,:110 */
/* borrowed pointers into the handle: log_alpha_array / t_array of ref :105,:107 (K floats each). */
DPM_API int dpm_schedule_tables(const dpm_schedule* s, const float** log_alpha, const float** t_array, int* K);
/* marginal_log_mean_coeff / marginal_alpha / marginal_std / marginal_lambda / inverse_lambda (ref :127-167) */
DPM_API int dpm_schedule_eval(const dpm_schedule* s, int what, const float* in, int n, float* out);
DPM_API int dpm_schedule_eval_f64(const dpm_schedule* s, int what, const double* in, int n, double* out);  /* double times / tables */

/* ---- time grids (ref :453-539) ---------------------------------------------------------- */
DPM_API int dpm_time_steps(const dpm_schedule* s, int skip_type, double t_T, double t_0, int N, float* out /* N+1 */);
DPM_API int dpm_singlestep_orders(int steps, int order, int* orders /* >= steps */, int* n_orders);
DPM_API int dpm_singlestep_grid(const dpm_schedule* s, int steps, int order, int skip_type, double t_T, double t_0,
                        float* outer /* >= steps+1 */, int* orders /* >= steps */, int* n_orders);

/* ---- plan: DPM_Solver.sample() unrolled into stages (ref :1047-1245) ---------------------- */
typedef struct dpm_plan_desc {
  int32_t algorithm_type;    /* DPM_ALGO_*                        */
  int32_t method;            /* DPM_METHOD_*                      */
  int32_t order;             /* 1..3                              */
  int32_t steps;             /* NFE                               */
  int32_t skip_type;         /* DPM_SKIP_*                        */
  int32_t solver_type;       /* DPM_SOLVER_*                      */
  int32_t lower_order_final; /* bool                              */
  int32_t denoise_to_zero;   /* bool                              */
  int32_t model_type;        /* DPM_MODEL_*                       */
  int32_t guidance;          /* DPM_GUIDE_*                       */
  int32_t thresholding;      /* bool: correcting_x0_fn == "dynamic_thresholding" */
  int32_t precision;         /* 0: scalars in fp32, the reference's default (fp32 schedule tables and times); 1: in double --
                                a double state on a schedule declared dtype=float64 (dpm_schedule_set_table_dtype), or whose
                                times are doubles: dpm_plan_stage_f64 then returns the doubles (version 200; was `reserved`) */
  double t_start;            /* t_T                               */
  double t_end;              /* t_0                               */
  double guidance_scale;
  double thr_ratio;          /* dynamic_thresholding_ratio        */
  double thr_max;            /* thresholding_max_val              */
} dpm_plan_desc;

typedef struct dpm_plan dpm_plan;
DPM_API int dpm_plan_create(const dpm_schedule* s, const dpm_plan_desc* d, dpm_plan** out);
DPM_API void dpm_plan_destroy(dpm_plan* p);
DPM_API int dpm_plan_num_stages(const dpm_plan* p);
DPM_API int dpm_plan_num_slots(const dpm_plan* p);                 /* history buffers the loop needs      */
DPM_API int dpm_plan_stage(const dpm_plan* p, int i, dpm_stage* out);
DPM_API int dpm_plan_stage_f64(const dpm_plan* p, int i, dpm_stage_f64* out);   /* double-precision plans only */
DPM_API int dpm_plan_timesteps(const dpm_plan* p, float* out, int cap, int* n); /* solver grid t_0..t_K    */

/* ---- coefficient builders for the reference's public per-update methods -------------------- */
/* dpm_solver_first_update (ref :547-592) */
DPM_API int dpm_coef_first(const dpm_schedule* s, int algo, float t_s, float t_t, dpm_stage* out);
/* multistep_dpm_solver_update (ref :932-954): t_prev[0..order-1] oldest..newest */
DPM_API int dpm_coef_multistep(const dpm_schedule* s, int algo, int solver_type, int order, const float* t_prev, float t_t,
                       dpm_stage* out);
/* singlestep_dpm_solver_update (ref :906-930): fills `order` stages.  r_mode 0: r1/r2 are the reference's
   Python-float defaults or user floats (double arithmetic then one fp32 rounding), 1: fp32 tensors. */
DPM_API int dpm_coef_singlestep(const dpm_schedule* s, int algo, int solver_type, int order, float t_s, float t_t,
                        double r1, double r2, int r_mode, dpm_stage* out /* [order] */);
/* fill the prologue scalars (alpha_e, sigma_e, t_input, guidance) of a stage evaluated at t */
DPM_API int dpm_coef_prologue(const dpm_schedule* s, float t_eval, int model_type, int guidance, double guidance_scale,
                      dpm_stage* inout);

/* the same in double (a double-precision evaluation: double time tensors, or a schedule declared dtype=float64): out = the
   integer fields + the doubles rounded, out64 = the doubles.  time_f64: the caller's time tensors are doubles (else fp32
   tensors whose values arrive converted exactly) -- decides the dtype the model time label is computed in (ref :278).
   dpm_coef_first is dpm_coef_multistep_f64 with order 1. */
DPM_API int dpm_coef_multistep_f64(const dpm_schedule* s, int algo, int solver_type, int order, const double* t_prev, double t_t,
                           int time_f64, dpm_stage* out, dpm_stage_f64* out64);
DPM_API int dpm_coef_singlestep_f64(const dpm_schedule* s, int algo, int solver_type, int order, double t_s, double t_t, int time_f64,
                            double r1, double r2, int r_mode, dpm_stage* out /* [order] */, dpm_stage_f64* out64 /* [order] */);
DPM_API int dpm_coef_prologue_f64(const dpm_schedule* s, double t_eval, int time_f64, int model_type, int guidance, double guidance_scale,
                          dpm_stage* inout, dpm_stage_f64* inout64);

/* ---- device side --------------------------------------------------------------------------- */
/* one fused stage kernel, asynchronous on `stream` */
DPM_API int dpm_stage_launch(const dpm_stage* st, const dpm_buffers* b, void* stream);
/* The same stage of n_req independent requests (same plan position: one dpm_stage; same n, batch and dtypes; each its
   own buffers) as ONE fused launch per group of DPM_MULTI_MAX requests: a server that keeps R sampling requests in
   flight pays a launch's ramp-up and drain once per R x (5 n s) bytes instead of once per 5 n s -- with inputs coming
   from HBM (a network ran in between) that is 8.5 -> ~6.7 us per [256,4,64,64] fp16 request-stage.  Stages with
   dynamic thresholding become one thresholding launch over all requests' samples (a batch of n_req * batch: smaller
   clusters or none -- 32 requests of [32,3,64,64] cost about what one [1024,3,64,64] does), provided clustered shapes
   find a DIFFERENT workspace in every request; classifier-free guidance keeps its duplicate store (x_out2).  Stages
   neither family covers (mask blend, classifier guidance, strided or
   unaligned buffers, the singlestep mid-stages, thresholding with a shared workspace) are launched request by
   request; results are identical either way. */
#define DPM_MULTI_MAX 32
DPM_API int dpm_stage_launch_multi(const dpm_stage* st, const dpm_buffers* bs, int n_req, void* stream);
/* scratch needed by stages with DPM_F_THRESH on the current device: 0 when one workgroup per sample is the plan (the
   sample lives in that workgroup's LDS), else ~45-80 KiB per sample of slots, histograms and counters through which the
   workgroup cluster of a sample exchanges its candidates (small batches, samples beyond 12288 elements).
   Pass it as dpm_buffers.workspace.  Contract: the caller ZERO-FILLS the workspace once (hipMemset) before its first use;
   every launch leaves it zero-filled again (the last workgroup of a cluster cleans up), so no launch pays for a clear.
   Launches that share a workspace must be ordered (same stream).
   Cluster waits are bounded (dpm_launch_opts.thr_spin_limit polls, milliseconds).  When the peers of a cluster are kept off the
   chip that long -- another process running clusters on the same GPU, two clustered graphs replayed concurrently -- the
   waiting workgroup gives up, computes the order statistics of its sample alone from global memory and carries on:
   the launch's results are the same bits, the workspace is left zero-filled as always, nothing is reported as an error. */
DPM_API size_t dpm_threshold_workspace_bytes(int64_t batch, int64_t per_sample);
/* diagnostics: 1 when a cluster wait of any clustered thresholding launch of this process timed out (and was recovered
   from) since the last call, else 0.  Reads and clears a host-mapped word; meaningful after the launches in question
   have completed (no synchronisation here). */
DPM_API int dpm_cluster_timeout_poll(void);
#define DPM_THR_HINT_WORDS 4   /* floats per sample of dpm_buffers.thr_hint / dpm_run_buffers.thr_hint */
/* x_t = alpha_t*x + sigma_t*noise for nt times (add_noise, ref :1012-1030); out is [nt, n] */
DPM_API int dpm_add_noise_launch(const dpm_schedule* s, const float* t_host, int nt, const void* x, const void* noise,
                         void* out, int64_t n, int dtype, void* stream);
/* the same with DOUBLE times on double tensors (version 201): alpha_t / sigma_t evaluated in double at the double times -- what
   the reference's type promotion makes of add_noise(x, t) when t is a double tensor or the schedule's tables are
   (NoiseScheduleVP(dtype=torch.float64)): the result is float64 whatever x was (the host converts x / noise first) */
DPM_API int dpm_add_noise_launch_f64(const dpm_schedule* s, const double* t_host, int nt, const void* x, const void* noise,
                             void* out, int64_t n, void* stream);
/* stand-alone mask blend  out = x*mask + (1-mask)*(alpha*a + sigma*b)  (b NULL: (1-mask)*a): the DPM_F_BLEND
   epilogue as its own launch, for callable use of the corrector and for x_T before the first update (ref :1180) */
DPM_API int dpm_blend_launch(const void* x, const void* mask, const void* a, const void* b, float alpha, float sigma, void* out,
                     int64_t n, int64_t mask_period, int dtype, void* stream);
/* error norm of the adaptive solver (ref :999-1001): E_b = sqrt(mean(((xh-xl)/delta)^2)) per sample in e_out[0..batch),
   and their maximum over the batch (the value the step-size controller reads) in e_out[batch] */
DPM_API int dpm_adaptive_error_launch(const void* x_lower, const void* x_higher, const void* x_prev, float atol, float rtol,
                              float* e_out /* [batch + 1] device */, int64_t batch, int64_t per_sample, int dtype,
                              void* stream);

/* ---- adaptive step-size solver with the controller ON THE DEVICE (dpm_solver_adaptive, ref :956-1010) --------------
   The reference's loop decides accept / reject and the next step size on the host from E = max_b ||(x_hi - x_lo)/delta||,
   i.e. one device -> host synchronisation per iteration.  Here the controller state (s, lambda_s, h, nfe, done) and the
   coefficients of the iteration's stages live in device memory: a one-thread kernel takes the decision of the previous
   iteration, evaluates inverse_lambda and every coefficient of the next one with the planner's own code
   (csrc/dpm_coef.hpp, compiled for the device) and fills the time vectors the network is called with; the stage kernels
   read their coefficients from there; accepted states are committed by a device-side copy.  After `done` every kernel
   of the handle is a no-op, so the host may enqueue a fixed number of iterations (hipGraph capture) or poll the
   host-mapped status words without synchronising.  One iteration =
       dpm_adaptive_begin -> [network at t_vectors[0]] -> stages ... -> dpm_adaptive_error
   with the stages of DPM-Solver-12 (order 2: which = 0 lower, 2..3 higher) or -23 (order 3: 0..1 lower, 3..4 higher;
   stage 2 equals stage 0).  The caller owns x, x_prev, x_lower, x_higher, the scratch states, the time vectors and the
   error word; the handle owns ~2 KB of device state and a copy of the schedule tables. */
typedef struct dpm_adaptive_desc {
  int32_t algorithm_type; /* DPM_ALGO_*   */
  int32_t solver_type;    /* DPM_SOLVER_* */
  int32_t order;          /* 2 or 3       */
  int32_t model_type;     /* DPM_MODEL_*  */
  int32_t guidance;       /* DPM_GUIDE_*  */
  int32_t reserved;
  double guidance_scale;
  double t_start, t_end;  /* t_T, t_0 */
  double h_init, atol, rtol, theta, t_err; /* ref :956 defaults 0.05, 0.0078, 0.05, 0.9, 1e-5 */
} dpm_adaptive_desc;
typedef struct dpm_adaptive dpm_adaptive;
DPM_API int dpm_adaptive_create(const dpm_schedule* s, const dpm_adaptive_desc* d, dpm_adaptive** out);
DPM_API void dpm_adaptive_destroy(dpm_adaptive* a);
/* static fields (form, flags, slots) of stage `which` (0..4); the float fields are placeholders */
DPM_API int dpm_adaptive_stage_template(const dpm_adaptive* a, int which, dpm_stage* out);
/* start of a run: s = t_T, h = h_init, nfe = 0 */
DPM_API int dpm_adaptive_reset(dpm_adaptive* a, void* stream);
/* decision on the previous iteration (E read from *e_dev, then cleared), commit of an accepted step
   (x <- x_higher, x_prev <- x_lower), plan of the next iteration, t_vectors[j][0 | 1][0..tv_len) <- t_eval | t_input of
   network evaluation j */
DPM_API int dpm_adaptive_begin(dpm_adaptive* a, void* x, void* x_prev, const void* x_lower, const void* x_higher, int64_t n,
                       int dtype, float* e_dev, float* t_vectors, int64_t tv_len, void* stream);
/* stage `which` of the current iteration: `st` = the caller's copy of the template (it may edit flags / model_type /
   guidance, e.g. to feed a known model value), float coefficients come from the device */
DPM_API int dpm_adaptive_stage_launch(dpm_adaptive* a, int which, const dpm_stage* st, const dpm_buffers* b, void* stream);
/* *e_dev <- max(*e_dev, max_b E_b) (ref :999-1001); several workgroups per sample, fp32 / fp16 / bf16 */
DPM_API int dpm_adaptive_error(dpm_adaptive* a, const void* x_lower, const void* x_higher, const void* x_prev, int64_t batch,
                       int64_t per_sample, int dtype, float* e_dev, void* stream);
/* host-mapped status as of the last dpm_adaptive_begin the device has executed: no synchronisation */
DPM_API int dpm_adaptive_poll(const dpm_adaptive* a, int* done, int* nfe, int* iterations, int* accepted);
/* `done` as recorded by the begin_index-th dpm_adaptive_begin since the last reset (a ring of the last 32): the value to
   look at after waiting for THAT launch -- identical on every rank of a batch-sharded run, whatever has run since */
DPM_API int dpm_adaptive_done_at(const dpm_adaptive* a, int begin_index);

/* native sample loop for non-Python hosts and for the solver-only benchmark.
   model(user, stage, x, t_input, t_eval, e0_out, e1_out): evaluate the network on x, write the raw output(s);
   NULL means the outputs are already staged in e0/e1 (frozen model).  All buffers are the caller's:
   xbuf[0] holds x_T and is only read; xbuf[1..3] are state-sized scratch the loop rotates through;
   hist[num_slots] cache model values.  On return *result is the index of the xbuf holding the final sample. */
typedef int (*dpm_model_cb)(void* user, const dpm_stage* st, const void* x, void* e0, void* e1, void* stream);
typedef struct dpm_run_buffers {
  void* xbuf[4];
  void* hist[3];
  void* e0;
  void* e1;
  void* workspace;
  int64_t n, batch;
  int32_t state_dtype, eps_dtype;
  /* ---- optional (zero = off) ---- */
  int64_t eps_stride;  /* as dpm_buffers.eps_stride: e0 / e1 are channel slices of a wider network output        */
  int32_t dup_state;   /* 1: every xbuf holds 2n elements and each state is written to both halves, so the model
                          callback of a classifier-free-guidance network receives its [2B,...] input ready-made
                          (xbuf[0] must already hold x_T twice)                                                  */
  int32_t reserved;
  float* thr_hint;     /* as dpm_buffers.thr_hint (DPM_THR_HINT_WORDS floats per sample, or NULL)                */
  const dpm_launch_opts* opts; /* per-call options handed to every launch of the run, NULL = defaults (version 200;
                                  dpm_plan_run_multi reads rbs[0].opts)                                            */
} dpm_run_buffers;
DPM_API int dpm_plan_run(const dpm_plan* p, const dpm_run_buffers* rb, dpm_model_cb model, void* user, void* stream,
                 int* result);
/* several independent sampling requests advanced stage by stage (all requests stage s, then all stage s+1, ...)
   through dpm_stage_launch_multi: what a server holding n requests in flight does, and -- with a frozen model -- the
   HBM-cold measurement mode of bench.py: between two stages of one request the other n-1 requests stream their buffers
   through the 256 MiB Infinity Cache.  All requests must share n, batch and dtypes.  ms (optional,
   [n_req * num_stages], request-major) receives kernel-only durations; a fused launch's duration is divided evenly
   over the requests it advanced (with n_req = 1: the kernel-only durations of one frozen-model trajectory).
   dpm_launch_opts.no_fuse launches request by request instead. */
DPM_API int dpm_plan_run_multi(const dpm_plan* p, const dpm_run_buffers* rbs, int n_req, void* stream, float* ms, int* results);

/* ---- hipGraph capture of a whole trajectory ------------------------------------------------------------
   The step loop is launch-bound between network calls (a stage kernel runs for microseconds): capture the plan's
   launches over fixed buffers once, replay with a single hipGraphLaunch per trajectory.  `stream` must be a real
   (non-null) stream; `model` as in dpm_plan_run -- everything it does must be capturable (enqueue-only on `stream`);
   NULL = frozen outputs already staged in e0/e1.  The buffers named by `rb` are baked into the graph. */
typedef struct dpm_graph dpm_graph;
DPM_API int dpm_graph_create(const dpm_plan* p, const dpm_run_buffers* rb, dpm_model_cb model, void* user, void* stream,
                     dpm_graph** out);
DPM_API int dpm_graph_launch(dpm_graph* g, void* stream);
DPM_API int dpm_graph_result(const dpm_graph* g);    /* index of the xbuf that holds the final sample            */
DPM_API int dpm_graph_num_nodes(const dpm_graph* g); /* kernel / memset nodes captured                           */
DPM_API void dpm_graph_destroy(dpm_graph* g);

/* ---- misc ---------------------------------------------------------------------------------- */
DPM_API int dpm_version(void);
/* sizeof() of the ABI structs as compiled, so a binding can verify its own layout at load time */
enum { DPM_SIZEOF_STAGE = 0, DPM_SIZEOF_BUFFERS = 1, DPM_SIZEOF_PLAN_DESC = 2, DPM_SIZEOF_RUN_BUFFERS = 3,
       DPM_SIZEOF_ADAPTIVE_DESC = 4, DPM_SIZEOF_LAUNCH_OPTS = 5, DPM_SIZEOF_STAGE_F64 = 6 };
DPM_API size_t dpm_sizeof(int which);
DPM_API const char* dpm_last_error(void); /* thread-local text of the last non-zero return */
DPM_API int dpm_device_info(int* n_cu, int* lds_bytes, char* arch, int arch_len);

#ifdef __cplusplus
}
#endif
#endif /* DPM_HIP_H */
