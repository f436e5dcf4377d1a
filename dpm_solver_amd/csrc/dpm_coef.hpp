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
// ---- correctly rounded fp32 elementary functions -------------------------------------------------------
DPM_HD inline float f_exp(float x) { return (float)exp((double)x); }
DPM_HD inline float f_log(float x) { return (float)log((double)x); }
DPM_HD inline float f_expm1(float x) { return (float)expm1((double)x); }
DPM_HD inline float f_log1p(float x) { return (float)log1p((double)x); }
DPM_HD inline float f_sqrt(float x) { return sqrtf(x); }
DPM_HD inline float f_cos(float x) { return (float)cos((double)x); }
DPM_HD inline float f_acos(float x) { return (float)acos((double)x); }
// torch.logaddexp
DPM_HD inline float f_logaddexp(float a, float b) {
  float m = a > b ? a : b;
  return m + f_log1p(f_exp(-fabsf(a - b)));
}

// interpolate_fn (ref :1253-1292): piecewise-linear through (xp, yp), xp ascending, outermost segments
// extended.  The reference locates the segment by sorting [x, xp]; a binary search finds the same one.
DPM_HD inline float interp32(float x, const float* xp, const float* yp, int K) {
  int lo = 0, hi = K;  // idx = #{xp < x} (std::lower_bound)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (xp[mid] < x) lo = mid + 1; else hi = mid;
  }
  const int idx = lo;
  int i0 = idx == 0 ? 0 : (idx == K ? K - 2 : idx - 1);
  int i1 = i0 + 1;
  return yp[i0] + (x - xp[i0]) * (yp[i1] - yp[i0]) / (xp[i1] - xp[i0]);
}


// ---- noise schedule (NoiseScheduleVP, ref :6-167) as a plain view: tables by pointer (host or device memory) ----
struct SchedView {
  int discrete, cosine, total_N;
  const float *la, *t, *la_rev, *t_rev;  // log_alpha_array / t_array (ref :105,:107) and their flipped copies (ref :166)
  double beta0, beta1, cos_s, cos_la0;

  DPM_HD float log_alpha(float tt) const {  // marginal_log_mean_coeff, ref :127-134
    if (discrete) return interp32(tt, t, la, total_N);
    if (cosine) {  // legacy :135-137, one fp32 rounding per tensor-scalar operation
      const float a = (((tt + (float)cos_s) / (float)(1. + cos_s)) * (float)M_PI) / 2.f;
      return f_log(f_cos(a)) - (float)cos_la0;
    }
    return -0.25f * (tt * tt) * (float)(beta1 - beta0) - 0.5f * tt * (float)beta0;
  }
  DPM_HD float alpha(float tt) const { return f_exp(log_alpha(tt)); }                                  // ref :140
  DPM_HD float std_(float tt) const { return f_sqrt(1.f - f_exp(2.f * log_alpha(tt))); }               // ref :146
  DPM_HD float lambda(float tt) const {                                                                // ref :152-154
    float l = log_alpha(tt);
    return l - 0.5f * f_log(1.f - f_exp(2.f * l));
  }
  DPM_HD float inv_lambda(float lam) const {  // ref :156-167
    if (cosine) {  // legacy :171-175
      const float l = -0.5f * f_logaddexp(-2.f * lam, 0.f);
      const float ac = f_acos(f_exp(l + (float)cos_la0));
      return (((ac * 2.f) * (float)(1. + cos_s)) / (float)M_PI) - (float)cos_s;
    }
    if (!discrete) {
      float tmp = (float)(2. * (beta1 - beta0)) * f_logaddexp(-2.f * lam, 0.f);
      float delta = (float)(beta0 * beta0) + tmp;
      return tmp / (f_sqrt(delta) + (float)beta0) / (float)(beta1 - beta0);
    }
    float l = -0.5f * f_logaddexp(0.f, -2.f * lam);
    return interp32(l, la_rev, t_rev, total_N);
  }
};

// ---- coefficient builders (S = dpm_schedule on the host, SchedView on the device) ----------------------------
struct Marg {
  float lam, la, sig;
};
template <class S>
DPM_HD inline Marg marg(const S* s, float t) { return Marg{s->lambda(t), s->log_alpha(t), s->std_(t)}; }

DPM_HD inline void stage_init(dpm_stage* st) {
  *st = dpm_stage{};
  st->h1_slot = st->h2_slot = st->m_slot = -1;
  st->emits_state = 1;
  st->cfg_scale = 1.f;
  st->thr_ratio = 0.995f;
  st->thr_max = 1.f;
}

template <class S>
DPM_HD inline void set_prologue(const S* s, float t_eval, int model_type, int guidance, double scale, dpm_stage* st) {
  st->t_eval = t_eval;
  // get_model_input_time (ref :271-280)
  st->t_input = s->discrete ? (t_eval - (float)(1. / s->total_N)) * 1000.f : t_eval;
  st->alpha_e = s->alpha(t_eval);
  st->sigma_e = s->std_(t_eval);
  st->model_type = model_type;
  st->guidance = guidance;
  st->cfg_scale = (float)scale;
  st->cg_scale = (float)scale * st->sigma_e;  // ref :321
}

// dpm_solver_first_update (ref :547-592)
template <class S>
DPM_HD inline void coef_first(const S* s, bool pp, float ts, float tt, dpm_stage* st) {
  Marg a = marg(s, ts), b = marg(s, tt);
  float h = b.lam - a.lam;
  st->form = DPM_FORM_LIN1;
  if (pp) {
    float phi_1 = f_expm1(-h);
    st->cx = b.sig / a.sig;
    st->c0 = f_exp(b.la) * phi_1;
  } else {
    float phi_1 = f_expm1(h);
    st->cx = f_exp(b.la - a.la);
    st->c0 = b.sig * phi_1;
  }
  st->t_out = tt;
}

// multistep_dpm_solver_second_update (ref :796-852)
template <class S>
DPM_HD inline void coef_ms2(const S* s, bool pp, int solver, float tp1, float tp0, float tt, dpm_stage* st) {
  float lam_p1 = s->lambda(tp1);
  Marg p0 = marg(s, tp0), t = marg(s, tt);
  float a_t = f_exp(t.la);
  float h_0 = p0.lam - lam_p1;
  float h = t.lam - p0.lam;
  float r0 = h_0 / h;
  st->form = DPM_FORM_TWO;
  st->k[0] = 1.f / r0;
  if (pp) {
    float phi_1 = f_expm1(-h);
    st->cx = t.sig / p0.sig;
    st->c0 = a_t * phi_1;
    st->c1 = solver == DPM_SOLVER_DPMSOLVER ? 0.5f * (a_t * phi_1) : -(a_t * (phi_1 / h + 1.f));
  } else {
    float phi_1 = f_expm1(h);
    st->cx = f_exp(t.la - p0.la);
    st->c0 = t.sig * phi_1;
    st->c1 = solver == DPM_SOLVER_DPMSOLVER ? 0.5f * (t.sig * phi_1) : t.sig * (phi_1 / h - 1.f);
  }
  st->t_out = tt;
}

// multistep_dpm_solver_third_update (ref :854-904); the reference ignores solver_type here
template <class S>
DPM_HD inline void coef_ms3(const S* s, bool pp, float tp2, float tp1, float tp0, float tt, dpm_stage* st) {
  float lam_p2 = s->lambda(tp2), lam_p1 = s->lambda(tp1);
  Marg p0 = marg(s, tp0), t = marg(s, tt);
  float a_t = f_exp(t.la);
  float h_1 = lam_p1 - lam_p2;
  float h_0 = p0.lam - lam_p1;
  float h = t.lam - p0.lam;
  float r0 = h_0 / h, r1 = h_1 / h;
  st->form = DPM_FORM_MS3;
  st->k[0] = 1.f / r0;
  st->k[1] = 1.f / r1;
  st->k[2] = r0 / (r0 + r1);
  st->k[3] = 1.f / (r0 + r1);
  if (pp) {
    float phi_1 = f_expm1(-h);
    float phi_2 = phi_1 / h + 1.f;
    float phi_3 = phi_2 / h - 0.5f;
    st->cx = t.sig / p0.sig;
    st->c0 = a_t * phi_1;
    st->c1 = -(a_t * phi_2);
    st->c2 = a_t * phi_3;
  } else {
    float phi_1 = f_expm1(h);
    float phi_2 = phi_1 / h - 1.f;
    float phi_3 = phi_2 / h - 0.5f;
    st->cx = f_exp(t.la - p0.la);
    st->c0 = t.sig * phi_1;
    st->c1 = t.sig * phi_2;
    st->c2 = t.sig * phi_3;
  }
  st->t_out = tt;
}

// r1/r2 of the singlestep solvers are Python floats (defaults / user floats: scalar-scalar arithmetic
// in double, one rounding when the product meets a tensor) or fp32 tensors (sample(): ref :1224-1227).
struct R {
  double d;
  bool tensor;
  DPM_HD float f() const { return (float)d; }
};
DPM_HD inline float r_div(double num, R r) { return r.tensor ? (float)num / r.f() : (float)(num / r.d); }
DPM_HD inline float r_ratio(R a, R b) { return (a.tensor) ? a.f() / b.f() : (float)(a.d / b.d); }
DPM_HD inline float r_diff(R a, R b) { return (a.tensor) ? a.f() - b.f() : (float)(a.d - b.d); }

// singlestep_dpm_solver_second_update (ref :594-673): two stages
template <class S>
DPM_HD inline void coef_ss2(const S* s, bool pp, int solver, float ts, float tt, R r1, dpm_stage* A, dpm_stage* B,
              float* t_s1) {
  Marg ms = marg(s, ts), mt = marg(s, tt);
  float h = mt.lam - ms.lam;
  float s1 = s->inv_lambda(ms.lam + r1.f() * h);
  Marg m1 = marg(s, s1);
  float a_s1 = f_exp(m1.la), a_t = f_exp(mt.la);
  *t_s1 = s1;
  A->form = DPM_FORM_LIN1;
  B->form = DPM_FORM_TWO;
  B->flags |= DPM_F_BASE_HIST;
  B->k[0] = 1.f;
  if (pp) {
    float phi_11 = f_expm1(-r1.f() * h);
    float phi_1 = f_expm1(-h);
    A->cx = m1.sig / ms.sig;
    A->c0 = a_s1 * phi_11;
    B->cx = mt.sig / ms.sig;
    B->c0 = a_t * phi_1;
    B->c1 = solver == DPM_SOLVER_DPMSOLVER ? r_div(0.5, r1) * (a_t * phi_1)
                                           : -(r_div(1., r1) * (a_t * (phi_1 / h + 1.f)));
  } else {
    float phi_11 = f_expm1(r1.f() * h);
    float phi_1 = f_expm1(h);
    A->cx = f_exp(m1.la - ms.la);
    A->c0 = m1.sig * phi_11;
    B->cx = f_exp(mt.la - ms.la);
    B->c0 = mt.sig * phi_1;
    B->c1 = solver == DPM_SOLVER_DPMSOLVER ? r_div(0.5, r1) * (mt.sig * phi_1)
                                           : r_div(1., r1) * (mt.sig * (phi_1 / h - 1.f));
  }
  A->t_out = s1;
  B->t_out = tt;
}

// singlestep_dpm_solver_third_update (ref :675-794): three stages
template <class S>
DPM_HD inline void coef_ss3(const S* s, bool pp, int solver, float ts, float tt, R r1, R r2, dpm_stage* A,
              dpm_stage* B, dpm_stage* C, float* t_s1, float* t_s2) {
  Marg ms = marg(s, ts), mt = marg(s, tt);
  float h = mt.lam - ms.lam;
  float s1 = s->inv_lambda(ms.lam + r1.f() * h);
  float s2 = s->inv_lambda(ms.lam + r2.f() * h);
  Marg m1 = marg(s, s1), m2 = marg(s, s2);
  float a_s1 = f_exp(m1.la), a_s2 = f_exp(m2.la), a_t = f_exp(mt.la);
  *t_s1 = s1;
  *t_s2 = s2;
  A->form = DPM_FORM_LIN1;
  B->form = DPM_FORM_TWO;
  B->flags |= DPM_F_BASE_HIST;
  B->k[0] = 1.f;
  const bool taylor = solver == DPM_SOLVER_TAYLOR;
  C->form = taylor ? DPM_FORM_SS3T : DPM_FORM_TWO;
  if (!taylor) {
    C->flags |= DPM_F_BASE_HIST;
    C->k[0] = 1.f;
  } else {
    C->k[0] = r_div(1., r1);
    C->k[1] = r_div(1., r2);
    C->k[2] = r2.f();
    C->k[3] = r1.f();
    C->k[4] = r_diff(r2, r1);
  }
  float phi_1, phi_2, phi_3, phi_11, phi_12, phi_22;
  if (pp) {
    phi_11 = f_expm1(-r1.f() * h);
    phi_12 = f_expm1(-r2.f() * h);
    phi_1 = f_expm1(-h);
    phi_22 = f_expm1(-r2.f() * h) / (r2.f() * h) + 1.f;
    phi_2 = phi_1 / h + 1.f;
    phi_3 = phi_2 / h - 0.5f;
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
    phi_22 = f_expm1(r2.f() * h) / (r2.f() * h) - 1.f;
    phi_2 = phi_1 / h - 1.f;
    phi_3 = phi_2 / h - 0.5f;
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
template <class S>
DPM_HD inline void singlestep_fill(const S* s, int algo, int solver_type, int order, float t_s, float t_t, double r1,
                                   double r2, int r_mode, dpm_stage* out) {
  const bool pp = algo == DPM_ALGO_DPMSOLVERPP;
  for (int i = 0; i < order; ++i) {
    stage_init(&out[i]);
    out[i].index = i;
    if (pp) out[i].flags |= DPM_F_TO_X0;
  }
  float te[3] = {t_s, 0.f, 0.f};
  R R1{r1, r_mode != 0}, R2{r2, r_mode != 0};
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
