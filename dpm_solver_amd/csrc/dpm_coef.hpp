// dpm_coef.hpp -- the scalar mathematics of the engine, shared by the host planner (dpm_host.cpp) and the device-side
// controller of the adaptive solver (dpm_kernels.hip): noise-schedule evaluation (ref :127-167) and the coefficient
// builders of every update formula (ref :547-904).  Everything is `__host__ __device__` when compiled as HIP, plain
// C++ otherwise, and templated on the schedule type so that the host's dpm_schedule (std::vector tables) and the
// device's SchedView (raw pointers into device memory) run the SAME code: the reference's fp32 operation order,
// exp / log / expm1 / log1p evaluated in double and rounded once.  Compile with -ffp-contract=off.
//
// `ref :NNN` = line in the reference's dpm_solver_pytorch.py.
#pragma once
#include <cmath>
#include <cstring>

#include "dpm_hip.h"

#ifdef __HIP__
#define DPM_HD __host__ __device__
#else
#define DPM_HD
#endif

namespace dpmc {
// ---- elementary functions in the run's scalar type F ------------------------------------------------------------
// F = float (every run whose schedule and times are fp32 -- the reference's default): the reference's fp32 operation
// order, exp / log / expm1 / log1p evaluated in double and rounded once (correctly rounded fp32).  F = double (a run whose
// state is double and whose schedule tables or times are double: NoiseScheduleVP(dtype=torch.float64), ref :14,
// :105-107 -- torch's type promotion then evaluates every scalar in double): plain double arithmetic.
DPM_HD inline float f_exp(float x) { return (float)exp((double)x); }
DPM_HD inline float f_log(float x) { return (float)log((double)x); }
DPM_HD inline float f_expm1(float x) { return (float)expm1((double)x); }
DPM_HD inline float f_log1p(float x) { return (float)log1p((double)x); }
DPM_HD inline float f_sqrt(float x) { return sqrtf(x); }
DPM_HD inline float f_cos(float x) { return (float)cos((double)x); }
DPM_HD inline float f_acos(float x) { return (float)acos((double)x); }
DPM_HD inline float f_abs(float x) { return fabsf(x); }
DPM_HD inline double f_exp(double x) { return exp(x); }
DPM_HD inline double f_log(double x) { return log(x); }
DPM_HD inline double f_expm1(double x) { return expm1(x); }
DPM_HD inline double f_log1p(double x) { return log1p(x); }
DPM_HD inline double f_sqrt(double x) { return sqrt(x); }
DPM_HD inline double f_cos(double x) { return cos(x); }
DPM_HD inline double f_acos(double x) { return acos(x); }
DPM_HD inline double f_abs(double x) { return fabs(x); }
// torch.logaddexp
template <typename F>
DPM_HD inline F f_logaddexp(F a, F b) {
  F m = a > b ? a : b;
  return m + f_log1p(f_exp(-f_abs(a - b)));
}

// interpolate_fn (ref :1253-1292): piecewise-linear through (xp, yp), xp ascending, outermost segments
// extended.  The reference locates the segment by sorting [x, xp]; a binary search finds the same one.
// y_f32: a DOUBLE query on fp32 tables (a double time tensor on a schedule left at dtype=float32): `end_y - start_y`
// (ref :1290) is then the one operation of the expression between two fp32 tensors -- an fp32 subtraction -- while the
// x side went through the concatenation with the query (ref :1268) and is double.
template <typename F>
DPM_HD inline F interp(F x, const F* xp, const F* yp, int K, bool y_f32 = false) {
  int lo = 0, hi = K;  // idx = #{xp < x} (std::lower_bound)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (xp[mid] < x) lo = mid + 1; else hi = mid;
  }
  const int idx = lo;
  int i0 = idx == 0 ? 0 : (idx == K ? K - 2 : idx - 1);
  int i1 = i0 + 1;
  const F dy = y_f32 ? (F)((float)yp[i1] - (float)yp[i0]) : yp[i1] - yp[i0];
  return yp[i0] + (x - xp[i0]) * dy / (xp[i1] - xp[i0]);
}
DPM_HD inline float interp32(float x, const float* xp, const float* yp, int K) { return interp<float>(x, xp, yp, K); }


// ---- noise schedule (NoiseScheduleVP, ref :6-167) as a plain view: tables by pointer (host or device memory) ----
template <typename FT>
struct SchedViewT {
  typedef FT F;  // the scalar type every quantity of a run on this view is evaluated in
  int discrete, cosine, total_N;
  const F *la, *t, *la_rev, *t_rev;  // log_alpha_array / t_array (ref :105,:107) and their flipped copies (ref :166)
  double beta0, beta1, cos_s, cos_la0;
  int tables_f32;  // the double view of a schedule whose tables the reference holds in fp32 (see interp)

  DPM_HD F log_alpha(F tt) const {  // marginal_log_mean_coeff, ref :127-134
    if (discrete) return interp<F>(tt, t, la, total_N, sizeof(F) == 8 && tables_f32);
    if (cosine) {  // legacy :135-137, one rounding per tensor-scalar operation
      const F a = (((tt + (F)cos_s) / (F)(1. + cos_s)) * (F)M_PI) / (F)2;
      return f_log(f_cos(a)) - (F)cos_la0;
    }
    return (F)-0.25 * (tt * tt) * (F)(beta1 - beta0) - (F)0.5 * tt * (F)beta0;
  }
  DPM_HD F alpha(F tt) const { return f_exp(log_alpha(tt)); }                                  // ref :140
  DPM_HD F std_(F tt) const { return f_sqrt((F)1 - f_exp((F)2 * log_alpha(tt))); }             // ref :146
  DPM_HD F lambda(F tt) const {                                                                // ref :152-154
    F l = log_alpha(tt);
    return l - (F)0.5 * f_log((F)1 - f_exp((F)2 * l));
  }
  DPM_HD F inv_lambda(F lam) const {  // ref :156-167
    if (cosine) {  // legacy :171-175
      const F l = (F)-0.5 * f_logaddexp<F>((F)-2 * lam, (F)0);
      const F ac = f_acos(f_exp(l + (F)cos_la0));
      return (((ac * (F)2) * (F)(1. + cos_s)) / (F)M_PI) - (F)cos_s;
    }
    if (!discrete) {
      F tmp = (F)(2. * (beta1 - beta0)) * f_logaddexp<F>((F)-2 * lam, (F)0);
      F delta = (F)(beta0 * beta0) + tmp;
      return tmp / (f_sqrt(delta) + (F)beta0) / (F)(beta1 - beta0);
    }
    F l = (F)-0.5 * f_logaddexp<F>((F)0, (F)-2 * lam);
    return interp<F>(l, la_rev, t_rev, total_N, sizeof(F) == 8 && tables_f32);
  }
  // inverse_lambda of an fp32 TENSOR of lambdas (the logSNR grid: torch.linspace builds it in fp32, ref :470-472): the
  // reference's logaddexp then runs in fp32 (both of its operands are fp32 tensors, ref :165) and only the interpolation on
  // the tables is in the tables' type
  DPM_HD F inv_lambda_of_f32(float lam) const {
    if (sizeof(F) == 4 || !discrete) return inv_lambda((F)lam);
    const float l = -0.5f * f_logaddexp<float>(0.f, -2.f * lam);
    return interp<F>((F)l, la_rev, t_rev, total_N, sizeof(F) == 8 && tables_f32);
  }
};
typedef SchedViewT<float> SchedView;     // the fp32 view: the host planner's default and the device-side adaptive controller's
typedef SchedViewT<double> SchedView64;  // double-precision runs (host planner only)

// the float fields of dpm_stage in double + its integer fields: what the builders fill for a double-precision run
// (split into dpm_stage [integers, rounded floats] and dpm_stage_f64 [the doubles] by the planner)
struct Stage64 {
  int32_t index, form;
  uint32_t flags;
  int32_t model_type, guidance, outer_step, emits_state, x_src, xe_src, h1_slot, h2_slot, m_slot;
  double t_eval, t_input, t_out, alpha_e, sigma_e, cfg_scale, cg_scale, cx, c0, c1, c2, k[5], thr_ratio, thr_max, blend_alpha,
      blend_sigma;
  int32_t time_f64;  // the reference's time tensors at t_eval (bit 0) / t_out (bit 1) are doubles (see set_prologue)
};

// ---- coefficient builders (S = dpm_schedule on the host, SchedView on the device) ----------------------------
template <typename F>
struct MargT {
  F lam, la, sig;
};
template <class S>
DPM_HD inline MargT<typename S::F> marg(const S* s, typename S::F t) {
  return MargT<typename S::F>{s->lambda(t), s->log_alpha(t), s->std_(t)};
}

template <class ST>
DPM_HD inline void stage_init(ST* st) {
  *st = ST{};
  st->h1_slot = st->h2_slot = st->m_slot = -1;
  st->emits_state = 1;
  st->cfg_scale = 1;
  st->thr_ratio = (decltype(st->thr_ratio))0.995;
  st->thr_max = 1;
}

// time_f32 (double-precision runs only): the reference's time tensor at this evaluation is an fp32 tensor -- the grids
// torch.linspace builds are (ref :472-477), also in a double-precision run -- and get_model_input_time (ref :271-280) computes
// in ITS dtype; times that come out of inverse_lambda on double tables (singlestep inner nodes, the logSNR grid) are doubles.
template <class S, class ST>
DPM_HD inline void set_prologue(const S* s, typename S::F t_eval, int model_type, int guidance, double scale, ST* st,
                                bool time_f32 = false) {
  typedef typename S::F F;
  st->t_eval = t_eval;
  if (sizeof(F) == 8 && time_f32)
    st->t_input = s->discrete ? (F)(((float)t_eval - (float)(1. / s->total_N)) * 1000.f) : t_eval;
  else
    st->t_input = s->discrete ? (t_eval - (F)(1. / s->total_N)) * (F)1000 : t_eval;
  st->alpha_e = s->alpha(t_eval);
  st->sigma_e = s->std_(t_eval);
  st->model_type = model_type;
  st->guidance = guidance;
  st->cfg_scale = (F)scale;
  st->cg_scale = (F)scale * (F)st->sigma_e;  // ref :321
}

// dpm_solver_first_update (ref :547-592)
template <class S, class ST>
DPM_HD inline void coef_first(const S* s, bool pp, typename S::F ts, typename S::F tt, ST* st) {
  typedef typename S::F F;
  MargT<F> a = marg(s, ts), b = marg(s, tt);
  F h = b.lam - a.lam;
  st->form = DPM_FORM_LIN1;
  if (pp) {
    F phi_1 = f_expm1(-h);
    st->cx = b.sig / a.sig;
    st->c0 = f_exp(b.la) * phi_1;
  } else {
    F phi_1 = f_expm1(h);
    st->cx = f_exp(b.la - a.la);
    st->c0 = b.sig * phi_1;
  }
  st->t_out = tt;
}

// multistep_dpm_solver_second_update (ref :796-852)
template <class S, class ST>
DPM_HD inline void coef_ms2(const S* s, bool pp, int solver, typename S::F tp1, typename S::F tp0, typename S::F tt, ST* st) {
  typedef typename S::F F;
  F lam_p1 = s->lambda(tp1);
  MargT<F> p0 = marg(s, tp0), t = marg(s, tt);
  F a_t = f_exp(t.la);
  F h_0 = p0.lam - lam_p1;
  F h = t.lam - p0.lam;
  F r0 = h_0 / h;
  st->form = DPM_FORM_TWO;
  st->k[0] = (F)1 / r0;
  if (pp) {
    F phi_1 = f_expm1(-h);
    st->cx = t.sig / p0.sig;
    st->c0 = a_t * phi_1;
    st->c1 = solver == DPM_SOLVER_DPMSOLVER ? (F)0.5 * (a_t * phi_1) : -(a_t * (phi_1 / h + (F)1));
  } else {
    F phi_1 = f_expm1(h);
    st->cx = f_exp(t.la - p0.la);
    st->c0 = t.sig * phi_1;
    st->c1 = solver == DPM_SOLVER_DPMSOLVER ? (F)0.5 * (t.sig * phi_1) : t.sig * (phi_1 / h - (F)1);
  }
  st->t_out = tt;
}

// multistep_dpm_solver_third_update (ref :854-904); the reference ignores solver_type here
template <class S, class ST>
DPM_HD inline void coef_ms3(const S* s, bool pp, typename S::F tp2, typename S::F tp1, typename S::F tp0, typename S::F tt, ST* st) {
  typedef typename S::F F;
  F lam_p2 = s->lambda(tp2), lam_p1 = s->lambda(tp1);
  MargT<F> p0 = marg(s, tp0), t = marg(s, tt);
  F a_t = f_exp(t.la);
  F h_1 = lam_p1 - lam_p2;
  F h_0 = p0.lam - lam_p1;
  F h = t.lam - p0.lam;
  F r0 = h_0 / h, r1 = h_1 / h;
  st->form = DPM_FORM_MS3;
  st->k[0] = (F)1 / r0;
  st->k[1] = (F)1 / r1;
  st->k[2] = r0 / (r0 + r1);
  st->k[3] = (F)1 / (r0 + r1);
  if (pp) {
    F phi_1 = f_expm1(-h);
    F phi_2 = phi_1 / h + (F)1;
    F phi_3 = phi_2 / h - (F)0.5;
    st->cx = t.sig / p0.sig;
    st->c0 = a_t * phi_1;
    st->c1 = -(a_t * phi_2);
    st->c2 = a_t * phi_3;
  } else {
    F phi_1 = f_expm1(h);
    F phi_2 = phi_1 / h - (F)1;
    F phi_3 = phi_2 / h - (F)0.5;
    st->cx = f_exp(t.la - p0.la);
    st->c0 = t.sig * phi_1;
    st->c1 = t.sig * phi_2;
    st->c2 = t.sig * phi_3;
  }
  st->t_out = tt;
}

// r1/r2 of the singlestep solvers are Python floats (defaults / user floats: scalar-scalar arithmetic
// in double, one rounding when the product meets a tensor) or fp32 tensors (sample(): ref :1224-1227).
template <typename F>
struct RT {
  double d;
  bool tensor;
  DPM_HD F f() const { return (F)d; }
};
// (fp32 TENSORS r1 / r2: `0.5 / r1`, `r2 / r1`, `r2 - r1` are fp32 tensor operations in the reference whatever the dtype of
// the rest of the expression -- also in a double-precision call, where only their fp32 result is widened)
template <typename F>
DPM_HD inline F r_div(double num, RT<F> r) { return r.tensor ? (F)((float)num / (float)r.d) : (F)(num / r.d); }
template <typename F>
DPM_HD inline F r_ratio(RT<F> a, RT<F> b) { return (a.tensor) ? (F)((float)a.d / (float)b.d) : (F)(a.d / b.d); }
template <typename F>
DPM_HD inline F r_diff(RT<F> a, RT<F> b) { return (a.tensor) ? (F)((float)a.d - (float)b.d) : (F)(a.d - b.d); }

// singlestep_dpm_solver_second_update (ref :594-673): two stages
template <class S, class ST>
DPM_HD inline void coef_ss2(const S* s, bool pp, int solver, typename S::F ts, typename S::F tt, RT<typename S::F> r1, ST* A, ST* B,
              typename S::F* t_s1) {
  typedef typename S::F F;
  MargT<F> ms = marg(s, ts), mt = marg(s, tt);
  F h = mt.lam - ms.lam;
  F s1 = s->inv_lambda(ms.lam + r1.f() * h);
  MargT<F> m1 = marg(s, s1);
  F a_s1 = f_exp(m1.la), a_t = f_exp(mt.la);
  *t_s1 = s1;
  A->form = DPM_FORM_LIN1;
  B->form = DPM_FORM_TWO;
  B->flags |= DPM_F_BASE_HIST;
  B->k[0] = (F)1;
  if (pp) {
    F phi_11 = f_expm1(-r1.f() * h);
    F phi_1 = f_expm1(-h);
    A->cx = m1.sig / ms.sig;
    A->c0 = a_s1 * phi_11;
    B->cx = mt.sig / ms.sig;
    B->c0 = a_t * phi_1;
    B->c1 = solver == DPM_SOLVER_DPMSOLVER ? r_div(0.5, r1) * (a_t * phi_1)
                                           : -(r_div(1., r1) * (a_t * (phi_1 / h + (F)1)));
  } else {
    F phi_11 = f_expm1(r1.f() * h);
    F phi_1 = f_expm1(h);
    A->cx = f_exp(m1.la - ms.la);
    A->c0 = m1.sig * phi_11;
    B->cx = f_exp(mt.la - ms.la);
    B->c0 = mt.sig * phi_1;
    B->c1 = solver == DPM_SOLVER_DPMSOLVER ? r_div(0.5, r1) * (mt.sig * phi_1)
                                           : r_div(1., r1) * (mt.sig * (phi_1 / h - (F)1));
  }
  A->t_out = s1;
  B->t_out = tt;
}

// singlestep_dpm_solver_third_update (ref :675-794): three stages
template <class S, class ST>
DPM_HD inline void coef_ss3(const S* s, bool pp, int solver, typename S::F ts, typename S::F tt, RT<typename S::F> r1,
              RT<typename S::F> r2, ST* A, ST* B, ST* C, typename S::F* t_s1, typename S::F* t_s2) {
  typedef typename S::F F;
  MargT<F> ms = marg(s, ts), mt = marg(s, tt);
  F h = mt.lam - ms.lam;
  F s1 = s->inv_lambda(ms.lam + r1.f() * h);
  F s2 = s->inv_lambda(ms.lam + r2.f() * h);
  MargT<F> m1 = marg(s, s1), m2 = marg(s, s2);
  F a_s1 = f_exp(m1.la), a_s2 = f_exp(m2.la), a_t = f_exp(mt.la);
  *t_s1 = s1;
  *t_s2 = s2;
  A->form = DPM_FORM_LIN1;
  B->form = DPM_FORM_TWO;
  B->flags |= DPM_F_BASE_HIST;
  B->k[0] = (F)1;
  const bool taylor = solver == DPM_SOLVER_TAYLOR;
  C->form = taylor ? DPM_FORM_SS3T : DPM_FORM_TWO;
  if (!taylor) {
    C->flags |= DPM_F_BASE_HIST;
    C->k[0] = (F)1;
  } else {
    C->k[0] = r_div(1., r1);
    C->k[1] = r_div(1., r2);
    C->k[2] = r2.f();
    C->k[3] = r1.f();
    C->k[4] = r_diff(r2, r1);
  }
  F phi_1, phi_2, phi_3, phi_11, phi_12, phi_22;
  if (pp) {
    phi_11 = f_expm1(-r1.f() * h);
    phi_12 = f_expm1(-r2.f() * h);
    phi_1 = f_expm1(-h);
    phi_22 = f_expm1(-r2.f() * h) / (r2.f() * h) + (F)1;
    phi_2 = phi_1 / h + (F)1;
    phi_3 = phi_2 / h - (F)0.5;
    A->cx = m1.sig / ms.sig;
    A->c0 = a_s1 * phi_11;
    B->cx = m2.sig / ms.sig;
    B->c0 = a_s2 * phi_12;
    B->c1 = -(r_ratio(r2, r1) * (a_s2 * phi_22));
    C->cx = mt.sig / ms.sig;
    C->c0 = a_t * phi_1;
    if (!taylor) {
      C->c1 = -(r_div(1., r2) * (a_t * phi_2));
    } else {
      C->c1 = -(a_t * phi_2);
      C->c2 = a_t * phi_3;
    }
  } else {
    phi_11 = f_expm1(r1.f() * h);
    phi_12 = f_expm1(r2.f() * h);
    phi_1 = f_expm1(h);
    phi_22 = f_expm1(r2.f() * h) / (r2.f() * h) - (F)1;
    phi_2 = phi_1 / h - (F)1;
    phi_3 = phi_2 / h - (F)0.5;
    A->cx = f_exp(m1.la - ms.la);
    A->c0 = m1.sig * phi_11;
    B->cx = f_exp(m2.la - ms.la);
    B->c0 = m2.sig * phi_12;
    B->c1 = r_ratio(r2, r1) * (m2.sig * phi_22);
    C->cx = f_exp(mt.la - ms.la);
    C->c0 = mt.sig * phi_1;
    if (!taylor) {
      C->c1 = r_div(1., r2) * (mt.sig * phi_2);
    } else {
      C->c1 = mt.sig * phi_2;
      C->c2 = mt.sig * phi_3;
    }
  }
  A->t_out = s1;
  B->t_out = s2;
  C->t_out = tt;
}

// singlestep_dpm_solver_update (ref :906-930) as stage records: the body of dpm_coef_singlestep without its argument
// checks.  r_mode 0: r1 / r2 are Python floats (double arithmetic, one fp32 rounding), 1: fp32 tensors.
template <class S, class ST>
DPM_HD inline void singlestep_fill(const S* s, int algo, int solver_type, int order, typename S::F t_s, typename S::F t_t, double r1,
                                   double r2, int r_mode, ST* out) {
  typedef typename S::F F;
  const bool pp = algo == DPM_ALGO_DPMSOLVERPP;
  for (int i = 0; i < order; ++i) {
    stage_init(&out[i]);
    out[i].index = i;
    if (pp) out[i].flags |= DPM_F_TO_X0;
  }
  F te[3] = {t_s, (F)0, (F)0};
  RT<F> R1{r1, r_mode != 0}, R2{r2, r_mode != 0};
  if (order == 1) {
    coef_first(s, pp, t_s, t_t, &out[0]);
  } else if (order == 2) {
    coef_ss2(s, pp, solver_type, t_s, t_t, R1, &out[0], &out[1], &te[1]);
  } else {
    coef_ss3(s, pp, solver_type, t_s, t_t, R1, R2, &out[0], &out[1], &out[2], &te[1], &te[2]);
  }
  for (int i = 0; i < order; ++i) {
    set_prologue(s, te[i], DPM_MODEL_NOISE, DPM_GUIDE_NONE, 1., &out[i]);
    const bool last = i == order - 1;
    out[i].emits_state = last;
    out[i].x_src = DPM_SRC_STATE;
    out[i].xe_src = i == 0 ? DPM_SRC_STATE : DPM_SRC_TMP;
    if (i == 0 && order > 1) {
      out[i].flags |= DPM_F_STORE_M;
      out[i].m_slot = 0;
    }
    if (i >= 1) out[i].h1_slot = 0;
    if (order == 3 && solver_type == DPM_SOLVER_TAYLOR) {
      if (i == 1) {
        out[i].flags |= DPM_F_STORE_M;
        out[i].m_slot = 1;
      }
      if (i == 2) out[i].h2_slot = 1;
    }
  }
}
}  // namespace dpmc
