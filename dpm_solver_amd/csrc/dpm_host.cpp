// dpm_host.cpp -- host side of the C ABI in include/dpm_hip.h: noise schedule, time grids and the
// planner that unrolls DPM_Solver.sample() into per-stage coefficient records.
//
// No device code here.  Everything is computed ONCE per sample() call, on the host, in the
// reference's own fp32 operation order (the reference carries every schedule scalar as a 0-dim or
// (1,)-shaped fp32 tensor), so that the coefficients agree with the reference to the last bit except
// where an elementary function (exp/log/expm1/log1p) rounds differently: here they are evaluated in
// double and rounded once, i.e. correctly rounded fp32.  Must be compiled with -ffp-contract=off.
//
// `ref :NNN` = line in the reference's dpm_solver_pytorch.py.
#include "dpm_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------
// error text
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

int dpm_set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

extern "C" const char* dpm_last_error(void) { return g_err.c_str(); }
extern "C" int dpm_version(void) { return DPM_HIP_VERSION; }
extern "C" size_t dpm_sizeof(int which) {
  switch (which) {
    case DPM_SIZEOF_STAGE: return sizeof(dpm_stage);
    case DPM_SIZEOF_BUFFERS: return sizeof(dpm_buffers);
    case DPM_SIZEOF_PLAN_DESC: return sizeof(dpm_plan_desc);
    case DPM_SIZEOF_RUN_BUFFERS: return sizeof(dpm_run_buffers);
    case DPM_SIZEOF_ADAPTIVE_DESC: return sizeof(dpm_adaptive_desc);
    case DPM_SIZEOF_LAUNCH_OPTS: return sizeof(dpm_launch_opts);
    case DPM_SIZEOF_STAGE_F64: return sizeof(dpm_stage_f64);
  }
  return 0;
}

#include "dpm_coef.hpp"
using namespace dpmc;

namespace {
// torch.linspace on CPU for fp32: fp32 step, filled from both ends, one fused multiply-add per point
// (checked bitwise against torch.linspace in tests/test_planner.py).
void linspace32(float start, float end, int n, float* out) {
  if (n <= 0) return;
  if (n == 1) {
    out[0] = start;
    return;
  }
  const float step = (end - start) / (float)(n - 1);
  const int half = n / 2;
  for (int i = 0; i < n; ++i)
    out[i] = i < half ? std::fma(step, (float)i, start) : std::fma(-step, (float)(n - i - 1), end);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// schedule
// ------------------------------------------------------------------------------------------------
struct dpm_schedule {
  typedef float F;  // the scalar type of the evaluation methods below (the coefficient builders of dpm_coef.hpp read it)
  bool discrete = true;
  int total_N = 1000;
  std::vector<float> la, t;        // log_alpha_array, t_array (ref :105,:107)
  std::vector<float> la_rev, t_rev;  // flipped copies for inverse_lambda (ref :166)
  // double-precision runs (a double state: torch's type promotion evaluates every scalar in double): the tables as the
  // reference holds them -- the double values they were computed in when the schedule was declared dtype=float64
  // (dpm_schedule_set_table_dtype, ref :105-107), else the fp32 tables above converted exactly
  std::vector<double> la_src;      // log_alpha as computed (double for the *_f64 constructors, (double)fp32 otherwise)
  std::vector<double> d_la, d_t, d_la_rev, d_t_rev;
  bool table_f64 = false;
  double beta0 = 0.1, beta1 = 20.0;
  // 'cosine' continuous-time schedule of the older vendored revision (examples/score_sde_pytorch/dpm_solver.py
  // :114-124, :134-137, :171-175)
  bool cosine = false;
  double cos_s = 0.008, cos_la0 = 0.;

  // the evaluation code lives in dpm_coef.hpp (shared with the device): a view over this object's tables
  dpmc::SchedView view() const {
    return dpmc::SchedView{discrete ? 1 : 0, cosine ? 1 : 0, total_N, la.data(), t.data(), la_rev.data(), t_rev.data(),
                           beta0, beta1, cos_s, cos_la0, 0};
  }
  float log_alpha(float tt) const { return view().log_alpha(tt); }   // marginal_log_mean_coeff, ref :127-134
  float alpha(float tt) const { return view().alpha(tt); }           // ref :140
  float std_(float tt) const { return view().std_(tt); }             // ref :146
  float lambda(float tt) const { return view().lambda(tt); }         // ref :152-154
  float inv_lambda(float lam) const { return view().inv_lambda(lam); }  // ref :156-167
  float inv_lambda_of_f32(float lam) const { return view().inv_lambda(lam); }
  dpmc::SchedView64 view64() const {
    return dpmc::SchedView64{discrete ? 1 : 0, cosine ? 1 : 0, total_N, d_la.data(), d_t.data(), d_la_rev.data(),
                             d_t_rev.data(), beta0, beta1, cos_s, cos_la0, table_f64 ? 0 : 1};
  }
  void build_double_tables() {
    if (!discrete) return;
    d_la.resize(total_N);
    d_t.resize(total_N);
    for (int i = 0; i < total_N; ++i) {
      d_la[i] = table_f64 && (int)la_src.size() == total_N ? la_src[i] : (double)la[i];
      d_t[i] = (double)t[i];  // linspace in fp32, then .to(dtype) (ref :107)
    }
    d_la_rev.assign(d_la.rbegin(), d_la.rend());
    d_t_rev.assign(d_t.rbegin(), d_t.rend());
  }
};

namespace {
int finish_discrete(std::vector<float>&& la, dpm_schedule** out, const std::vector<double>* src = nullptr) {
  if (la.size() < 2) return dpm_set_error(DPM_ERR_ARG, "discrete schedule needs >= 2 entries after clipping, got %zu", la.size());
  dpm_schedule* s = new (std::nothrow) dpm_schedule;
  if (!s) return dpm_set_error(DPM_ERR_NOMEM, "out of memory");
  s->discrete = true;
  if (src) s->la_src = *src;
  s->la = std::move(la);
  s->total_N = (int)s->la.size();
  std::vector<float> full(s->total_N + 1);
  linspace32(0.f, 1.f, s->total_N + 1, full.data());  // ref :107
  s->t.assign(full.begin() + 1, full.end());
  s->la_rev.assign(s->la.rbegin(), s->la.rend());
  s->t_rev.assign(s->t.rbegin(), s->t.rend());
  s->build_double_tables();
  *out = s;
  return DPM_OK;
}

// numerical_clip_alpha (ref :114-125), generic over the arithmetic type the caller's array came in
template <typename T>
size_t clip_len(const T* la, size_t n, double clipped_lambda = -5.1) {
  const T cl = (T)clipped_lambda;  // searchsorted casts the Python scalar to the tensor's dtype
  std::vector<T> lam(n);
  for (size_t i = 0; i < n; ++i) {
    T ls;
    if (sizeof(T) == 4)
      ls = (T)(0.5f * f_log(1.f - f_exp(2.f * (float)la[i])));
    else
      ls = (T)(0.5 * std::log(1. - std::exp(2. * (double)la[i])));
    lam[i] = la[i] - ls;
  }
  // searchsorted(flip(lam), cl, right=False) = #{lam_flipped < cl}, lam_flipped ascending
  std::vector<T> fl(lam.rbegin(), lam.rend());
  size_t idx = std::lower_bound(fl.begin(), fl.end(), cl) - fl.begin();
  return n - idx;
}
template <typename T>
size_t clip_len(const std::vector<T>& la) { return clip_len(la.data(), la.size()); }
}  // namespace

// NoiseScheduleVP.numerical_clip_alpha as a function of its own (ref :114-125): how many leading entries of a
// log-alpha table survive the clip of the half-logSNR at `clipped_lambda`
extern "C" int dpm_numerical_clip_len_f32(const float* log_alphas, int n, double clipped_lambda, int* out_len) {
  if (!log_alphas || !out_len || n < 0) return dpm_set_error(DPM_ERR_ARG, "numerical_clip_alpha: null pointer or n < 0");
  *out_len = (int)clip_len(log_alphas, (size_t)n, clipped_lambda);
  return DPM_OK;
}
extern "C" int dpm_numerical_clip_len_f64(const double* log_alphas, int n, double clipped_lambda, int* out_len) {
  if (!log_alphas || !out_len || n < 0) return dpm_set_error(DPM_ERR_ARG, "numerical_clip_alpha: null pointer or n < 0");
  *out_len = (int)clip_len(log_alphas, (size_t)n, clipped_lambda);
  return DPM_OK;
}

extern "C" int dpm_schedule_create_betas_f32(const float* betas, int n, int clip, dpm_schedule** out) {
  if (!betas || !out || n < 2) return dpm_set_error(DPM_ERR_ARG, "betas: need n >= 2 and non-null pointers");
  std::vector<float> la(n);
  double acc = 0.;  // torch CPU cumsum accumulates fp32 in double, rounding each prefix (ref :100)
  for (int i = 0; i < n; ++i) {
    acc += (double)f_log(1.f - betas[i]);
    la[i] = 0.5f * (float)acc;
  }
  if (clip) la.resize(clip_len(la));
  return finish_discrete(std::move(la), out);
}

extern "C" int dpm_schedule_create_betas_f64(const double* betas, int n, int clip, dpm_schedule** out) {
  if (!betas || !out || n < 2) return dpm_set_error(DPM_ERR_ARG, "betas: need n >= 2 and non-null pointers");
  std::vector<double> la(n);
  double acc = 0.;
  for (int i = 0; i < n; ++i) {
    acc += std::log(1. - betas[i]);
    la[i] = 0.5 * acc;
  }
  if (clip) la.resize(clip_len(la));
  return finish_discrete(std::vector<float>(la.begin(), la.end()), out, &la);  // .to(dtype=float32), ref :105
}

extern "C" int dpm_schedule_create_alphas_cumprod_f32(const float* ac, int n, int clip, dpm_schedule** out) {
  if (!ac || !out || n < 2) return dpm_set_error(DPM_ERR_ARG, "alphas_cumprod: need n >= 2 and non-null pointers");
  std::vector<float> la(n);
  for (int i = 0; i < n; ++i) la[i] = 0.5f * f_log(ac[i]);  // ref :103
  if (clip) la.resize(clip_len(la));
  return finish_discrete(std::move(la), out);
}

extern "C" int dpm_schedule_create_alphas_cumprod_f64(const double* ac, int n, int clip, dpm_schedule** out) {
  if (!ac || !out || n < 2) return dpm_set_error(DPM_ERR_ARG, "alphas_cumprod: need n >= 2 and non-null pointers");
  std::vector<double> la(n);
  for (int i = 0; i < n; ++i) la[i] = 0.5 * std::log(ac[i]);
  if (clip) la.resize(clip_len(la));
  return finish_discrete(std::vector<float>(la.begin(), la.end()), out, &la);
}

extern "C" int dpm_schedule_create_log_alpha(const float* log_alpha, int n, dpm_schedule** out) {
  if (!log_alpha || !out || n < 2) return dpm_set_error(DPM_ERR_ARG, "log_alpha: need n >= 2 and non-null pointers");
  return finish_discrete(std::vector<float>(log_alpha, log_alpha + n), out);
}

extern "C" int dpm_schedule_create_linear(double beta_0, double beta_1, dpm_schedule** out) {
  if (!out) return dpm_set_error(DPM_ERR_ARG, "null out");
  dpm_schedule* s = new (std::nothrow) dpm_schedule;
  if (!s) return dpm_set_error(DPM_ERR_NOMEM, "out of memory");
  s->discrete = false;
  s->total_N = 1000;  // ref :110
  s->beta0 = beta_0;
  s->beta1 = beta_1;
  *out = s;
  return DPM_OK;
}

extern "C" int dpm_schedule_create_cosine(dpm_schedule** out) {
  if (!out) return dpm_set_error(DPM_ERR_ARG, "null out");
  dpm_schedule* s = new (std::nothrow) dpm_schedule;
  if (!s) return dpm_set_error(DPM_ERR_NOMEM, "out of memory");
  s->discrete = false;
  s->cosine = true;
  s->total_N = 1000;  // legacy :111
  s->cos_s = 0.008;   // legacy :114
  s->cos_la0 = std::log(std::cos(s->cos_s / (1. + s->cos_s) * M_PI / 2.));  // legacy :117 (Python float arithmetic)
  *out = s;
  return DPM_OK;
}

extern "C" int dpm_schedule_set_table_dtype(dpm_schedule* s, int dtype) {
  if (!s || (dtype != DPM_DTYPE_F32 && dtype != DPM_DTYPE_F64)) return dpm_set_error(DPM_ERR_ARG, "table dtype must be DPM_DTYPE_F32 or DPM_DTYPE_F64");
  s->table_f64 = dtype == DPM_DTYPE_F64;
  s->build_double_tables();
  return DPM_OK;
}

int dpm_schedule_table_is_f64(const dpm_schedule* s) { return s && s->table_f64; }

extern "C" int dpm_schedule_tables_f64(const dpm_schedule* s, const double** log_alpha, const double** t_array, int* K) {
  if (!s || !s->discrete) return dpm_set_error(DPM_ERR_ARG, "tables exist only for discrete schedules");
  if (log_alpha) *log_alpha = s->d_la.data();
  if (t_array) *t_array = s->d_t.data();
  if (K) *K = s->total_N;
  return DPM_OK;
}

extern "C" int dpm_schedule_eval_f64(const dpm_schedule* s, int what, const double* in, int n, double* out) {
  if (!s || (n > 0 && (!in || !out))) return dpm_set_error(DPM_ERR_ARG, "schedule_eval: null pointer");
  const dpmc::SchedView64 v = s->view64();
  for (int i = 0; i < n; ++i) {
    switch (what) {
      case DPM_EVAL_LOG_ALPHA: out[i] = v.log_alpha(in[i]); break;
      case DPM_EVAL_ALPHA: out[i] = v.alpha(in[i]); break;
      case DPM_EVAL_STD: out[i] = v.std_(in[i]); break;
      case DPM_EVAL_LAMBDA: out[i] = v.lambda(in[i]); break;
      case DPM_EVAL_INV_LAMBDA: out[i] = v.inv_lambda(in[i]); break;
      default: return dpm_set_error(DPM_ERR_ARG, "schedule_eval: unknown quantity %d", what);
    }
  }
  return DPM_OK;
}

extern "C" void dpm_schedule_destroy(dpm_schedule* s) { delete s; }
// for dpm_kernels.hip (the device-side adaptive controller uploads the tables)
dpmc::SchedView dpm_schedule_view(const dpm_schedule* s) { return s->view(); }
extern "C" int dpm_schedule_is_discrete(const dpm_schedule* s) { return s && s->discrete; }
extern "C" int dpm_schedule_total_N(const dpm_schedule* s) { return s ? s->total_N : 0; }

extern "C" int dpm_schedule_tables(const dpm_schedule* s, const float** log_alpha, const float** t_array, int* K) {
  if (!s || !s->discrete) return dpm_set_error(DPM_ERR_ARG, "tables exist only for discrete schedules");
  if (log_alpha) *log_alpha = s->la.data();
  if (t_array) *t_array = s->t.data();
  if (K) *K = s->total_N;
  return DPM_OK;
}

extern "C" int dpm_schedule_eval(const dpm_schedule* s, int what, const float* in, int n, float* out) {
  if (!s || (n > 0 && (!in || !out))) return dpm_set_error(DPM_ERR_ARG, "schedule_eval: null pointer");
  for (int i = 0; i < n; ++i) {
    switch (what) {
      case DPM_EVAL_LOG_ALPHA: out[i] = s->log_alpha(in[i]); break;
      case DPM_EVAL_ALPHA: out[i] = s->alpha(in[i]); break;
      case DPM_EVAL_STD: out[i] = s->std_(in[i]); break;
      case DPM_EVAL_LAMBDA: out[i] = s->lambda(in[i]); break;
      case DPM_EVAL_INV_LAMBDA: out[i] = s->inv_lambda(in[i]); break;
      default: return dpm_set_error(DPM_ERR_ARG, "schedule_eval: unknown quantity %d", what);
    }
  }
  return DPM_OK;
}

// ------------------------------------------------------------------------------------------------
// time grids
// ------------------------------------------------------------------------------------------------
namespace {
// S::F = float: the reference's fp32 grids.  S::F = double (a double-precision run): torch.linspace still builds its grid
// in fp32 (the default dtype, ref :472-477) -- the values are fp32 numbers --, and only what is then evaluated on the
// schedule's double tables (the logSNR grid's inverse_lambda) is a double.
template <class S>
int time_steps(const S* s, int skip, double t_T, double t_0, int N, typename S::F* out) {
  typedef typename S::F F;
  std::vector<float> g((size_t)N + 1);
  switch (skip) {
    case DPM_SKIP_TIME_UNIFORM:  // ref :474
      linspace32((float)t_T, (float)t_0, N + 1, g.data());
      for (int i = 0; i <= N; ++i) out[i] = (F)g[i];
      return DPM_OK;
    case DPM_SKIP_LOGSNR: {  // ref :469-472
      // lambda(t_T).cpu().item(): a Python float; linspace rounds its end points to fp32
      const F lT = s->lambda((F)(float)t_T), l0 = s->lambda((F)(float)t_0);
      linspace32((float)lT, (float)l0, N + 1, g.data());
      for (int i = 0; i <= N; ++i) out[i] = s->inv_lambda_of_f32(g[i]);
      return DPM_OK;
    }
    case DPM_SKIP_TIME_QUADRATIC:  // ref :476-478
      linspace32((float)std::pow(t_T, 0.5), (float)std::pow(t_0, 0.5), N + 1, g.data());
      for (int i = 0; i <= N; ++i) out[i] = (F)(g[i] * g[i]);
      return DPM_OK;
  }
  return dpm_set_error(DPM_ERR_ARG, "Unsupported skip_type %d, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'", skip);
}

int singlestep_orders(int steps, int order, std::vector<int>& orders, int& K) {  // ref :514-533
  orders.clear();
  if (order == 3) {
    K = steps / 3 + 1;
    if (steps % 3 == 0) {
      orders.assign(std::max(K - 2, 0), 3);
      orders.push_back(2);
      orders.push_back(1);
    } else if (steps % 3 == 1) {
      orders.assign(K - 1, 3);
      orders.push_back(1);
    } else {
      orders.assign(K - 1, 3);
      orders.push_back(2);
    }
  } else if (order == 2) {
    if (steps % 2 == 0) {
      K = steps / 2;
      orders.assign(K, 2);
    } else {
      K = steps / 2 + 1;
      orders.assign(K - 1, 2);
      orders.push_back(1);
    }
  } else if (order == 1) {
    K = steps;
    orders.assign(steps, 1);
  } else {
    return dpm_set_error(DPM_ERR_ARG, "'order' must be '1' or '2' or '3'.");
  }
  return DPM_OK;
}

template <class S>
int singlestep_grid(const S* s, int steps, int order, int skip, double t_T, double t_0,
                    std::vector<typename S::F>& outer, std::vector<int>& orders) {
  int K = 0;
  int rc = singlestep_orders(steps, order, orders, K);
  if (rc) return rc;
  if (skip == DPM_SKIP_LOGSNR) {  // ref :536
    outer.resize(K + 1);
    return time_steps(s, skip, t_T, t_0, K, outer.data());
  }
  std::vector<typename S::F> full(steps + 1);  // ref :538
  rc = time_steps(s, skip, t_T, t_0, steps, full.data());
  if (rc) return rc;
  outer.clear();
  int pos = 0;
  outer.push_back(full[0]);
  for (int o : orders) {
    pos += o;
    if (pos > steps) return dpm_set_error(DPM_ERR_ARG, "singlestep: steps=%d too small for order %d", steps, order);
    outer.push_back(full[pos]);
  }
  return DPM_OK;
}
}  // namespace

extern "C" int dpm_time_steps(const dpm_schedule* s, int skip_type, double t_T, double t_0, int N, float* out) {
  if (!s || !out || N < 1) return dpm_set_error(DPM_ERR_ARG, "time_steps: bad arguments");
  return time_steps(s, skip_type, t_T, t_0, N, out);
}

extern "C" int dpm_singlestep_orders(int steps, int order, int* orders, int* n_orders) {
  std::vector<int> o;
  int K;
  int rc = singlestep_orders(steps, order, o, K);
  if (rc) return rc;
  if (orders) std::copy(o.begin(), o.end(), orders);
  if (n_orders) *n_orders = (int)o.size();
  return DPM_OK;
}

extern "C" int dpm_singlestep_grid(const dpm_schedule* s, int steps, int order, int skip_type, double t_T, double t_0,
                                   float* outer, int* orders, int* n_orders) {
  if (!s) return dpm_set_error(DPM_ERR_ARG, "null schedule");
  std::vector<float> g;
  std::vector<int> o;
  int rc = singlestep_grid(s, steps, order, skip_type, t_T, t_0, g, o);
  if (rc) return rc;
  if (outer) std::copy(g.begin(), g.end(), outer);
  if (orders) std::copy(o.begin(), o.end(), orders);
  if (n_orders) *n_orders = (int)o.size();
  return DPM_OK;
}

namespace {
int check_enum(int v, int lo, int hi, const char* what) {
  if (v < lo || v > hi) return dpm_set_error(DPM_ERR_ARG, "%s out of range: %d", what, v);
  return DPM_OK;
}
}  // namespace

extern "C" int dpm_coef_prologue(const dpm_schedule* s, float t_eval, int model_type, int guidance,
                                 double guidance_scale, dpm_stage* st) {
  if (!s || !st) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  if (check_enum(model_type, 0, 3, "model_type") || check_enum(guidance, 0, 2, "guidance")) return DPM_ERR_ARG;
  set_prologue(s, t_eval, model_type, guidance, guidance_scale, st);
  return DPM_OK;
}

extern "C" int dpm_coef_first(const dpm_schedule* s, int algo, float t_s, float t_t, dpm_stage* out) {
  if (!s || !out) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  if (check_enum(algo, 0, 1, "algorithm_type")) return DPM_ERR_ARG;
  stage_init(out);
  coef_first(s, algo == DPM_ALGO_DPMSOLVERPP, t_s, t_t, out);
  if (algo == DPM_ALGO_DPMSOLVERPP) out->flags |= DPM_F_TO_X0;
  set_prologue(s, t_s, DPM_MODEL_NOISE, DPM_GUIDE_NONE, 1., out);
  return DPM_OK;
}

extern "C" int dpm_coef_multistep(const dpm_schedule* s, int algo, int solver_type, int order, const float* t_prev,
                                  float t_t, dpm_stage* out) {
  if (!s || !out || !t_prev) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  if (check_enum(algo, 0, 1, "algorithm_type")) return DPM_ERR_ARG;
  if (order < 1 || order > 3) return dpm_set_error(DPM_ERR_ARG, "Solver order must be 1 or 2 or 3, got %d", order);
  if (order == 2 && (solver_type < 0 || solver_type > 1))
    return dpm_set_error(DPM_ERR_ARG, "'solver_type' must be either 'dpmsolver' or 'taylor', got %d", solver_type);
  const bool pp = algo == DPM_ALGO_DPMSOLVERPP;
  stage_init(out);
  if (order == 1)
    coef_first(s, pp, t_prev[0], t_t, out);
  else if (order == 2)
    coef_ms2(s, pp, solver_type, t_prev[0], t_prev[1], t_t, out);
  else
    coef_ms3(s, pp, t_prev[0], t_prev[1], t_prev[2], t_t, out);
  if (pp) out->flags |= DPM_F_TO_X0;
  set_prologue(s, t_prev[order - 1], DPM_MODEL_NOISE, DPM_GUIDE_NONE, 1., out);
  return DPM_OK;
}

extern "C" int dpm_coef_singlestep(const dpm_schedule* s, int algo, int solver_type, int order, float t_s, float t_t,
                                   double r1, double r2, int r_mode, dpm_stage* out) {
  if (!s || !out) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  if (check_enum(algo, 0, 1, "algorithm_type")) return DPM_ERR_ARG;
  if (order < 1 || order > 3) return dpm_set_error(DPM_ERR_ARG, "Solver order must be 1 or 2 or 3, got %d", order);
  if (order >= 2 && (solver_type < 0 || solver_type > 1))
    return dpm_set_error(DPM_ERR_ARG, "'solver_type' must be either 'dpmsolver' or 'taylor', got %d", solver_type);
  singlestep_fill(s, algo, solver_type, order, t_s, t_t, r1, r2, r_mode, out);
  return DPM_OK;
}

// ---- the same builders for a double-precision evaluation (double time tensors, or a schedule declared dtype=float64): the
// stage records come back as dpm_stage (integers + the doubles rounded) and dpm_stage_f64 (the doubles).  time_f64: the
// caller's time tensors are doubles (else fp32 tensors whose values arrive here converted exactly): decides the dtype
// get_model_input_time computes in (ref :278), see set_prologue.
namespace {
void split_stage64(const dpmc::Stage64& q, dpm_stage* st, dpm_stage_f64* d64);
}

extern "C" int dpm_coef_prologue_f64(const dpm_schedule* s, double t_eval, int time_f64, int model_type, int guidance,
                                     double guidance_scale, dpm_stage* st, dpm_stage_f64* st64) {
  if (!s || !st || !st64) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  if (check_enum(model_type, 0, 3, "model_type") || check_enum(guidance, 0, 2, "guidance")) return DPM_ERR_ARG;
  const dpmc::SchedView64 v = s->view64();
  dpmc::Stage64 q{};
  set_prologue(&v, t_eval, model_type, guidance, guidance_scale, &q, !time_f64);
  st->t_eval = (float)q.t_eval; st->t_input = (float)q.t_input; st->alpha_e = (float)q.alpha_e; st->sigma_e = (float)q.sigma_e;
  st->model_type = model_type; st->guidance = guidance; st->cfg_scale = (float)q.cfg_scale; st->cg_scale = (float)q.cg_scale;
  st64->t_eval = q.t_eval; st64->t_input = q.t_input; st64->alpha_e = q.alpha_e; st64->sigma_e = q.sigma_e;
  st64->cfg_scale = q.cfg_scale; st64->cg_scale = q.cg_scale;
  st64->time_f64 = (st64->time_f64 & ~1) | (time_f64 ? 1 : 0);
  return DPM_OK;
}

extern "C" int dpm_coef_multistep_f64(const dpm_schedule* s, int algo, int solver_type, int order, const double* t_prev,
                                      double t_t, int time_f64, dpm_stage* out, dpm_stage_f64* out64) {
  if (!s || !out || !out64 || !t_prev) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  if (check_enum(algo, 0, 1, "algorithm_type")) return DPM_ERR_ARG;
  if (order < 1 || order > 3) return dpm_set_error(DPM_ERR_ARG, "Solver order must be 1 or 2 or 3, got %d", order);
  if (order == 2 && (solver_type < 0 || solver_type > 1))
    return dpm_set_error(DPM_ERR_ARG, "'solver_type' must be either 'dpmsolver' or 'taylor', got %d", solver_type);
  const bool pp = algo == DPM_ALGO_DPMSOLVERPP;
  const dpmc::SchedView64 v = s->view64();
  dpmc::Stage64 q;
  stage_init(&q);
  if (order == 1)
    coef_first(&v, pp, t_prev[0], t_t, &q);
  else if (order == 2)
    coef_ms2(&v, pp, solver_type, t_prev[0], t_prev[1], t_t, &q);
  else
    coef_ms3(&v, pp, t_prev[0], t_prev[1], t_prev[2], t_t, &q);
  if (pp) q.flags |= DPM_F_TO_X0;
  set_prologue(&v, t_prev[order - 1], DPM_MODEL_NOISE, DPM_GUIDE_NONE, 1., &q, !time_f64);
  q.time_f64 = time_f64 ? 3 : 0;
  split_stage64(q, out, out64);
  return DPM_OK;
}

extern "C" int dpm_coef_singlestep_f64(const dpm_schedule* s, int algo, int solver_type, int order, double t_s, double t_t,
                                       int time_f64, double r1, double r2, int r_mode, dpm_stage* out, dpm_stage_f64* out64) {
  if (!s || !out || !out64) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  if (check_enum(algo, 0, 1, "algorithm_type")) return DPM_ERR_ARG;
  if (order < 1 || order > 3) return dpm_set_error(DPM_ERR_ARG, "Solver order must be 1 or 2 or 3, got %d", order);
  if (order >= 2 && (solver_type < 0 || solver_type > 1))
    return dpm_set_error(DPM_ERR_ARG, "'solver_type' must be either 'dpmsolver' or 'taylor', got %d", solver_type);
  const dpmc::SchedView64 v = s->view64();
  dpmc::Stage64 q[3];
  singlestep_fill(&v, algo, solver_type, order, t_s, t_t, r1, r2, r_mode, q);
  for (int i = 0; i < order; ++i) {
    // stage 0 is evaluated at the caller's time tensor; the inner nodes come out of inverse_lambda as doubles (ref :621)
    const int tf = (i == 0 ? (time_f64 ? 1 : 0) : 1) | ((i < order - 1 || time_f64) ? 2 : 0);
    set_prologue(&v, q[i].t_eval, DPM_MODEL_NOISE, DPM_GUIDE_NONE, 1., &q[i], !(tf & 1));
    q[i].time_f64 = tf;
    split_stage64(q[i], &out[i], &out64[i]);
  }
  return DPM_OK;
}

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
struct dpm_plan {
  std::vector<dpm_stage> stages;
  std::vector<dpm_stage_f64> stages64;  // double-precision plans (dpm_plan_desc.precision): the doubles behind `stages`
  std::vector<float> grid;
  std::vector<double> grid64;
  int slots = 0;
};

namespace {
// DPM_Solver.sample() unrolled into stages (ref :1047-1245), in the scalar type of the schedule view S (float: the
// reference's default; double: a double-precision run) with stage records ST (dpm_stage / dpmc::Stage64)
template <class S, class ST>
int plan_build(const S* s, const dpm_plan_desc* d, std::vector<ST>& stages, std::vector<typename S::F>& grid, int& slots_out) {
  typedef typename S::F F;
  const bool pp = d->algorithm_type == DPM_ALGO_DPMSOLVERPP;
  const double t_T = d->t_start, t_0 = d->t_end;
  int rc = DPM_OK;
  int last_step = 0;

  // the dtype of the reference's time tensors (double-precision runs; see set_prologue): bit 0 t_eval, bit 1 t_out
  std::vector<int> time_f64;
  const bool grid_f64 = d->skip_type == DPM_SKIP_LOGSNR;  // the whole grid comes out of inverse_lambda
  auto finish_stage = [&](ST& st, F t_eval, int tf64) {
    st.index = (int)stages.size();
    if (pp) st.flags |= DPM_F_TO_X0;
    if (pp && d->thresholding) st.flags |= DPM_F_THRESH;
    st.thr_ratio = (decltype(st.thr_ratio))d->thr_ratio;
    st.thr_max = (decltype(st.thr_max))d->thr_max;
    set_prologue(s, t_eval, d->model_type, d->guidance, d->guidance_scale, &st, !(tf64 & 1));
    stages.push_back(st);
    time_f64.push_back(tf64);
  };

  if (d->method == DPM_METHOD_MULTISTEP) {
    const int S_ = d->steps, P = d->order;
    // The reference validates the order where an update of that order is REACHED (multistep_dpm_solver_update, ref :948-954),
    // not up front: sample(order=4, steps=5, lower_order_final=True) runs -- its step orders are 1, 2, 3, 2, 1 (ref
    // :1185-1201) -- while order=4 with steps >= 7 raises at the first fourth-order update (and steps = 6 at the third-order
    // update of the main loop, see below).  Same here: the step orders
    // first, then the check on what they contain.  The history ring needs min(P, 3) slots.
    const int PS = std::min(P, 3);
    if (P < 1) rc = dpm_set_error(DPM_ERR_ARG, "Solver order must be 1 or 2 or 3, got %d", P);
    if (!rc && S_ < P) rc = dpm_set_error(DPM_ERR_ARG, "multistep needs steps >= order (steps=%d, order=%d)", S_, P);
    if (!rc) {
      grid.resize(S_ + 1);
      rc = time_steps(s, d->skip_type, t_T, t_0, S_, grid.data());  // ref :1173
    }
    std::vector<int> ord(rc ? 0 : S_);
    for (int i = 0; i < (int)ord.size(); ++i) {
      const int step = i + 1;  // the reference's loop variable: this stage produces x at ts[step]
      if (step < P)
        ord[i] = step;  // warm-up, ref :1185-1187
      else
        ord[i] = (d->lower_order_final && S_ < 10) ? std::min(P, S_ + 1 - step) : P;  // ref :1198-1201
      if (ord[i] > 3 && !rc) rc = dpm_set_error(DPM_ERR_ARG, "Solver order must be 1 or 2 or 3, got %d", ord[i]);
      // ... and its third-order update unpacks the history list into exactly three names (ref :869): with order > 3 the
      // list of the main loop is longer, and reaching a third-order update there is the ValueError Python raises for that
      if (P > 3 && step >= P && ord[i] == 3 && !rc) rc = dpm_set_error(DPM_ERR_ARG, "too many values to unpack (expected 3)");
    }
    if (!rc) {
      const F* ts = grid.data();
      for (int i = 0; i < S_; ++i) {
        ST st;
        stage_init(&st);
        if (ord[i] == 1)
          coef_first(s, pp, ts[i], ts[i + 1], &st);
        else if (ord[i] == 2)
          coef_ms2(s, pp, d->solver_type, ts[i - 1], ts[i], ts[i + 1], &st);
        else
          coef_ms3(s, pp, ts[i - 2], ts[i - 1], ts[i], ts[i + 1], &st);
        st.outer_step = i + 1;
        if (P >= 2) {
          if (ord[i] >= 2) st.h1_slot = (i - 1) % PS;
          if (ord[i] >= 3) st.h2_slot = (i - 2) % PS;
          bool needed = false;  // does a later stage read this stage's model value?
          for (int j = i + 1; j < S_ && j <= i + 2; ++j)
            if (ord[j] > j - i) needed = true;
          if (needed) {
            st.flags |= DPM_F_STORE_M;
            st.m_slot = i % PS;
          }
        }
        finish_stage(st, ts[i], grid_f64 ? 3 : 0);
      }
      slots_out = P >= 2 ? PS : 0;
      last_step = S_;
    }
  } else {
    std::vector<F> outer;
    std::vector<int> orders;
    if (d->order < 1 || d->order > 3) rc = dpm_set_error(DPM_ERR_ARG, "'order' must be '1' or '2' or '3'.");
    if (!rc) {
      if (d->method == DPM_METHOD_SINGLESTEP) {
        rc = singlestep_grid(s, d->steps, d->order, d->skip_type, t_T, t_0, outer, orders);  // ref :1216
      } else {
        const int K = d->steps / d->order;  // ref :1218-1220; K = 0 (steps < order) is a no-op in the reference too
        orders.assign(K, d->order);
        outer.resize(K + 1);
        rc = time_steps(s, d->skip_type, t_T, t_0, K, outer.data());
      }
    }
    if (!rc) {
      grid = outer;
      int slots = 0;
      for (size_t j = 0; j < orders.size() && !rc; ++j) {
        const int o = orders[j];
        const F ts_ = outer[j], tt_ = outer[j + 1];
        F inner[4], lam[4];
        rc = time_steps(s, d->skip_type, (double)ts_, (double)tt_, o, inner);  // ref :1223 (s.item(), t.item(): Python floats)
        if (rc) break;
        for (int i = 0; i <= o; ++i) lam[i] = s->lambda(inner[i]);
        F hh = lam[o] - lam[0];
        double r1 = o >= 2 ? (double)((lam[1] - lam[0]) / hh) : 0.;  // ref :1226-1227 (tensors of the run's scalar type)
        double r2 = o >= 3 ? (double)((lam[2] - lam[0]) / hh) : 0.;
        ST st[3];
        if (o >= 2 && (d->solver_type < 0 || d->solver_type > 1))
          rc = dpm_set_error(DPM_ERR_ARG, "'solver_type' must be either 'dpmsolver' or 'taylor', got %d", d->solver_type);
        if (rc) break;
        // r1 / r2 are tensors of the run's scalar type: fp32 tensors in an fp32 plan (their quotients `0.5 / r1`, `r2 / r1` are
        // fp32 operations), DOUBLE tensors in a double-precision plan -- which behave like Python floats there (r_mode 0).
        // (Round 6 passed 1 for both: 3e-8 off the reference on double tables, unseen while the differential tests compared the
        // engine with itself -- tests/test_differential_reference.py: load_reference.)
        singlestep_fill(s, d->algorithm_type, d->solver_type, o, ts_, tt_, r1, r2, sizeof(F) == 8 ? 0 : 1, st);
        for (int i = 0; i < o; ++i) {
          st[i].outer_step = (int)j;
          F te = (F)st[i].t_eval;
          if (st[i].m_slot >= 0) slots = std::max(slots, st[i].m_slot + 1);
          // inner nodes s1, s2 come out of inverse_lambda (ref :621, :706-707); an intermediate state's "t_out" is one too
          finish_stage(st[i], te, ((i > 0 || grid_f64) ? 1 : 0) | ((i < o - 1 || grid_f64) ? 2 : 0));
        }
      }
      slots_out = slots;
      last_step = (int)orders.size() - 1;
    }
  }
  if (!rc && d->denoise_to_zero) {  // ref :1235-1241, :541-545
    ST st;
    stage_init(&st);
    st.form = DPM_FORM_DENOISE;
    st.outer_step = last_step + 1;
    st.t_out = (F)(float)t_0;       // torch.ones((1,)).to(device) * t_0: an fp32 tensor
    finish_stage(st, (F)(float)t_0, 0);
    ST& b = stages.back();
    b.flags |= DPM_F_TO_X0;  // data_prediction_fn also under algorithm_type='dpmsolver'
    if (d->thresholding) b.flags |= DPM_F_THRESH;
  }
  if constexpr (sizeof(F) == 8)
    for (size_t i = 0; i < stages.size(); ++i) stages[i].time_f64 = time_f64[i];
  return rc;
}

// the doubles of a double-precision stage and its fp32 twin (integer fields + the doubles rounded: what inspection, the
// model time vectors and a host that ignores dpm_stage_f64 see)
void split_stage64(const dpmc::Stage64& q, dpm_stage* st, dpm_stage_f64* d64) {
  *st = dpm_stage{};
  st->index = q.index; st->form = q.form; st->flags = q.flags; st->model_type = q.model_type; st->guidance = q.guidance;
  st->outer_step = q.outer_step; st->emits_state = q.emits_state; st->x_src = q.x_src; st->xe_src = q.xe_src;
  st->h1_slot = q.h1_slot; st->h2_slot = q.h2_slot; st->m_slot = q.m_slot;
  st->t_eval = (float)q.t_eval; st->t_input = (float)q.t_input; st->t_out = (float)q.t_out;
  st->alpha_e = (float)q.alpha_e; st->sigma_e = (float)q.sigma_e; st->cfg_scale = (float)q.cfg_scale;
  st->cg_scale = (float)q.cg_scale; st->cx = (float)q.cx; st->c0 = (float)q.c0; st->c1 = (float)q.c1; st->c2 = (float)q.c2;
  for (int i = 0; i < 5; ++i) st->k[i] = (float)q.k[i];
  st->thr_ratio = (float)q.thr_ratio; st->thr_max = (float)q.thr_max;
  st->blend_alpha = (float)q.blend_alpha; st->blend_sigma = (float)q.blend_sigma;
  *d64 = dpm_stage_f64{};
  d64->t_eval = q.t_eval; d64->t_input = q.t_input; d64->t_out = q.t_out; d64->alpha_e = q.alpha_e; d64->sigma_e = q.sigma_e;
  d64->cfg_scale = q.cfg_scale; d64->cg_scale = q.cg_scale; d64->cx = q.cx; d64->c0 = q.c0; d64->c1 = q.c1; d64->c2 = q.c2;
  for (int i = 0; i < 5; ++i) d64->k[i] = q.k[i];
  d64->thr_ratio = q.thr_ratio; d64->thr_max = q.thr_max; d64->blend_alpha = q.blend_alpha; d64->blend_sigma = q.blend_sigma;
  d64->time_f64 = q.time_f64;
}
}  // namespace

extern "C" int dpm_plan_create(const dpm_schedule* s, const dpm_plan_desc* d, dpm_plan** out) {
  if (!s || !d || !out) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  if (check_enum(d->algorithm_type, 0, 1, "algorithm_type") || check_enum(d->model_type, 0, 3, "model_type") ||
      check_enum(d->guidance, 0, 2, "guidance_type"))
    return DPM_ERR_ARG;
  if (d->method < 0 || d->method > 2) return dpm_set_error(DPM_ERR_ARG, "Got wrong method %d", d->method);
  if (d->skip_type < 0 || d->skip_type > 2)
    return dpm_set_error(DPM_ERR_ARG, "Unsupported skip_type %d, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'", d->skip_type);
  if (d->solver_type < 0 || d->solver_type > 1)
    return dpm_set_error(DPM_ERR_ARG, "'solver_type' must be either 'dpmsolver' or 'taylor', got %d", d->solver_type);
  if (!(d->t_end > 0) || !(d->t_start > 0))
    return dpm_set_error(DPM_ERR_ARG, "Time range needs to be greater than 0. For discrete-time DPMs, it needs to be in [1 / N, 1], where N is the length of betas array");
  if (d->steps < 1) return dpm_set_error(DPM_ERR_ARG, "steps must be >= 1, got %d", d->steps);
  if (d->precision != 0 && d->precision != 1) return dpm_set_error(DPM_ERR_ARG, "precision must be 0 (fp32) or 1 (double)");
  dpm_plan* p = new (std::nothrow) dpm_plan;
  if (!p) return dpm_set_error(DPM_ERR_NOMEM, "out of memory");
  int rc;
  if (d->precision == 0) {
    rc = plan_build(s, d, p->stages, p->grid, p->slots);
    p->grid64.assign(p->grid.begin(), p->grid.end());
  } else {
    const dpmc::SchedView64 v = s->view64();
    std::vector<dpmc::Stage64> st64;
    rc = plan_build(&v, d, st64, p->grid64, p->slots);
    p->grid.assign(p->grid64.begin(), p->grid64.end());
    p->stages.resize(st64.size());
    p->stages64.resize(st64.size());
    for (size_t i = 0; i < st64.size(); ++i) split_stage64(st64[i], &p->stages[i], &p->stages64[i]);
  }
  if (rc) {
    delete p;
    return rc;
  }
  *out = p;
  return DPM_OK;
}

extern "C" void dpm_plan_destroy(dpm_plan* p) { delete p; }
extern "C" int dpm_plan_num_stages(const dpm_plan* p) { return p ? (int)p->stages.size() : 0; }
extern "C" int dpm_plan_num_slots(const dpm_plan* p) { return p ? p->slots : 0; }

extern "C" int dpm_plan_stage(const dpm_plan* p, int i, dpm_stage* out) {
  if (!p || !out || i < 0 || i >= (int)p->stages.size()) return dpm_set_error(DPM_ERR_ARG, "plan_stage: bad index %d", i);
  *out = p->stages[i];
  return DPM_OK;
}

extern "C" int dpm_plan_stage_f64(const dpm_plan* p, int i, dpm_stage_f64* out) {
  if (!p || !out || i < 0 || i >= (int)p->stages.size()) return dpm_set_error(DPM_ERR_ARG, "plan_stage_f64: bad index %d", i);
  if (p->stages64.empty()) return dpm_set_error(DPM_ERR_ARG, "plan_stage_f64: not a double-precision plan (dpm_plan_desc.precision)");
  *out = p->stages64[i];
  return DPM_OK;
}

extern "C" int dpm_plan_timesteps(const dpm_plan* p, float* out, int cap, int* n) {
  if (!p) return dpm_set_error(DPM_ERR_ARG, "null plan");
  if (n) *n = (int)p->grid.size();
  if (out) {
    if (cap < (int)p->grid.size()) return dpm_set_error(DPM_ERR_ARG, "timesteps: capacity %d < %zu", cap, p->grid.size());
    std::copy(p->grid.begin(), p->grid.end(), out);
  }
  return DPM_OK;
}

// ------------------------------------------------------------------------------------------------
// native sample loop (buffer choreography shared with the Python shim)
// ------------------------------------------------------------------------------------------------
// launch hooks implemented in dpm_kernels.hip
int dpm_stage_launch_ev(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop);
int dpm_timing_begin(int n, void*** starts, void*** stops);
int dpm_timing_end(int n, void** starts, void** stops, void* stream, float* ms, const unsigned char* recorded);

// buffer rotation of one stage, shared by the single- and multi-request loops: which xbuf the network saw (xe) and
// which one the stage writes (out), given where the state and the pending intermediate live.  Negative = plan error.
static int stage_rotation(const dpm_stage& st, int n_stages, int state, int tmp, int* xe, int* out) {
  if (st.index < 0 || st.index >= n_stages)
    return dpm_set_error(DPM_ERR_ARG, "plan_run: stage index %d outside the plan's %d stages", st.index, n_stages);
  *xe = st.xe_src == DPM_SRC_TMP ? tmp : state;
  if (*xe < 0) return dpm_set_error(DPM_ERR_ARG, "plan_run: stage %d reads TMP before it exists", st.index);
  int o = 1;  // xbuf[0] (the caller's x_T) is never written
  while (o == state || o == *xe) ++o;
  *out = o;
  return DPM_OK;
}

static int plan_run_impl(const dpm_plan* p, const dpm_run_buffers* rb, dpm_model_cb model, void* user, void* stream,
                         int* result, void** ev_start, void** ev_stop) {
  if (!p || !rb) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  for (int i = 0; i < 4; ++i)
    if (!rb->xbuf[i]) return dpm_set_error(DPM_ERR_ARG, "plan_run: xbuf[%d] is null", i);
  for (int i = 0; i < p->slots; ++i)
    if (!rb->hist[i]) return dpm_set_error(DPM_ERR_ARG, "plan_run: hist[%d] is null (plan needs %d slots)", i, p->slots);
  int state = 0, tmp = -1;
  for (const dpm_stage& st : p->stages) {
    int xe = 0, out = 0;
    if (int rc = stage_rotation(st, (int)p->stages.size(), state, tmp, &xe, &out)) return rc;
    if (model) {
      int rc = model(user, &st, rb->xbuf[xe], rb->e0, rb->e1, stream);
      if (rc) return dpm_set_error(DPM_ERR_CALLBACK, "model callback failed at stage %d (rc=%d)", st.index, rc);
    }
    dpm_buffers b;
    std::memset(&b, 0, sizeof b);
    b.x = rb->xbuf[state];
    b.xe = xe == state ? nullptr : rb->xbuf[xe];
    b.e0 = rb->e0;
    b.e1 = rb->e1;
    b.h1 = st.h1_slot >= 0 ? rb->hist[st.h1_slot] : nullptr;
    b.h2 = st.h2_slot >= 0 ? rb->hist[st.h2_slot] : nullptr;
    b.x_out = rb->xbuf[out];
    b.m_out = st.m_slot >= 0 ? rb->hist[st.m_slot] : nullptr;
    b.workspace = rb->workspace;
    b.thr_hint = rb->thr_hint;
    b.opts = rb->opts;
    b.coef64 = p->stages64.empty() ? nullptr : &p->stages64[(size_t)st.index];
    b.n = rb->n;
    b.batch = rb->batch;
    b.state_dtype = rb->state_dtype;
    b.eps_dtype = rb->eps_dtype;
    b.eps_stride = rb->eps_stride;
    // frozen outputs: nothing ran since the previous stage wrote x and m (the first stage reads the caller's x_T cold)
    b.inputs_resident = model == nullptr && st.index > 0;
    if (rb->dup_state && &st != &p->stages.back())  // the last stage's output feeds no network call
      b.x_out2 = static_cast<char*>(rb->xbuf[out]) + rb->n * (rb->state_dtype == DPM_DTYPE_F32 ? 4 : 2);
    int rc = dpm_stage_launch_ev(&st, &b, stream, ev_start ? ev_start[st.index] : nullptr,
                                 ev_stop ? ev_stop[st.index] : nullptr);
    if (rc) return rc;
    if (st.emits_state) {
      state = out;
      tmp = -1;
    } else {
      tmp = out;
    }
  }
  if (result) *result = state;
  return DPM_OK;
}

extern "C" int dpm_plan_run(const dpm_plan* p, const dpm_run_buffers* rb, dpm_model_cb model, void* user, void* stream,
                            int* result) {
  return plan_run_impl(p, rb, model, user, stream, result, nullptr, nullptr);
}

int dpm_stage_launch_multi_ev(const dpm_stage* st, const dpm_buffers* bs, int n_req, void* stream, void** ev_start,
                              void** ev_stop, int* fused_first);

extern "C" int dpm_plan_run_multi(const dpm_plan* p, const dpm_run_buffers* rbs, int n_req, void* stream, float* ms,
                                  int* results) {
  if (!p || !rbs || n_req < 1) return dpm_set_error(DPM_ERR_ARG, "plan_run_multi: bad arguments");
  const int ns = (int)p->stages.size();
  for (int r = 0; r < n_req; ++r) {
    for (int i = 0; i < 4; ++i)
      if (!rbs[r].xbuf[i]) return dpm_set_error(DPM_ERR_ARG, "plan_run_multi: request %d xbuf[%d] is null", r, i);
    for (int i = 0; i < p->slots; ++i)
      if (!rbs[r].hist[i]) return dpm_set_error(DPM_ERR_ARG, "plan_run_multi: request %d hist[%d] is null", r, i);
  }
  void **starts = nullptr, **stops = nullptr;
  if (ms) {
    int rc = dpm_timing_begin(n_req * ns, &starts, &stops);
    if (rc) return rc;
  }
  std::vector<int> state(n_req, 0), tmp(n_req, -1), first((size_t)n_req * ns, 0);
  std::vector<dpm_buffers> bs(n_req);
  int rc = DPM_OK;
  for (const dpm_stage& st : p->stages) {
    for (int r = 0; r < n_req; ++r) {
      const dpm_run_buffers& rb = rbs[r];
      int xe = 0, out = 0;
      if ((rc = stage_rotation(st, ns, state[r], tmp[r], &xe, &out)) != DPM_OK) break;
      dpm_buffers& b = bs[r];
      std::memset(&b, 0, sizeof b);
      b.x = rb.xbuf[state[r]];
      b.xe = xe == state[r] ? nullptr : rb.xbuf[xe];
      b.e0 = rb.e0;
      b.e1 = rb.e1;
      b.h1 = st.h1_slot >= 0 ? rb.hist[st.h1_slot] : nullptr;
      b.h2 = st.h2_slot >= 0 ? rb.hist[st.h2_slot] : nullptr;
      b.x_out = rb.xbuf[out];
      b.m_out = st.m_slot >= 0 ? rb.hist[st.m_slot] : nullptr;
      b.workspace = rb.workspace;
      b.thr_hint = rb.thr_hint;
      b.opts = rbs[0].opts;
      b.coef64 = p->stages64.empty() ? nullptr : &p->stages64[(size_t)st.index];
      b.n = rb.n;
      b.batch = rb.batch;
      b.state_dtype = rb.state_dtype;
      b.eps_dtype = rb.eps_dtype;
      b.eps_stride = rb.eps_stride;
      b.inputs_resident = n_req == 1;  // interleaved requests evict each other's buffers, like a network would
      if (rb.dup_state && &st != &p->stages.back())  // as in dpm_plan_run: the [2B, ...] network input of CFG
        b.x_out2 = static_cast<char*>(rb.xbuf[out]) + rb.n * (rb.state_dtype == DPM_DTYPE_F32 ? 4 : 2);
      if (st.emits_state) {
        state[r] = out;
        tmp[r] = -1;
      } else {
        tmp[r] = out;
      }
    }
    if (rc) break;
    // event k = st.index * n_req + r (stage-major, so a stage's events are one contiguous array)
    const size_t k0 = (size_t)st.index * n_req;
    rc = dpm_stage_launch_multi_ev(&st, bs.data(), n_req, stream, starts ? starts + k0 : nullptr,
                                   stops ? stops + k0 : nullptr, first.data() + k0);
    if (rc) break;
  }
  if (ms) {
    std::vector<float> raw((size_t)n_req * ns, 0.f);
    std::vector<unsigned char> recorded((size_t)n_req * ns, 0);
    for (int s = 0; s < ns; ++s)
      for (int r = 0; r < n_req; ++r) recorded[(size_t)s * n_req + r] = first[(size_t)s * n_req + r] == r;
    int rc2 = dpm_timing_end(n_req * ns, starts, stops, stream, rc ? nullptr : raw.data(), recorded.data());
    if (!rc) rc = rc2;
    if (!rc)
      for (int s = 0; s < ns; ++s)
        for (int r = 0; r < n_req;) {  // a fused group [r, e): its one measured duration, spread evenly
          int e = r + 1;
          while (e < n_req && first[(size_t)s * n_req + e] == first[(size_t)s * n_req + r]) ++e;
          const float each = raw[(size_t)s * n_req + r] / (float)(e - r);
          for (int q = r; q < e; ++q) ms[(size_t)q * ns + s] = each;
          r = e;
        }
  }
  if (!rc && results)
    for (int r = 0; r < n_req; ++r) results[r] = state[r];
  return rc;
}
