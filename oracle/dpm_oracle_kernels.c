/*
 * dpm_oracle_kernels.c -- the data-parallel arithmetic of the sampling path restated in plain C.
 * TEST INFRASTRUCTURE ONLY (like oracle/dpm_oracle.py): only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load the library built from this file (oracle/_build/libdpm_oracle.so,
 * compiled by __graft_entry__.build_oracle() with gcc -O3 -ffp-contract=off -fopenmp).  The product path never does.
 *
 * What is here: every tensor expression of the reference's hot path (dpm_solver_pytorch.py; `ref :N` = its line N) on
 * fp32 arrays -- the noise -> data conversion, the classifier-free blend, dynamic thresholding, the first / second /
 * third order multistep updates in the reference's association, and the fused 2M stage a GPU launch performs.  The
 * SCALARS (alpha_t, sigma_t, the phi_k combinations) are arguments: they come from the numpy oracle's schedule code
 * (oracle/dpm_oracle.py: Schedule, Solver._marg), which is pinned to the reference's golden vectors.
 *
 * Pin: tests/test_oracle_c.py holds every function against the numpy oracle bit for bit on seeded inputs (IEEE + - * /
 * in fp32, no contraction: -ffp-contract=off), and a whole DPM-Solver++(2M) trajectory driven through these functions
 * against oracle.Solver.sample() and the committed golden fixtures (tests/golden/e2e.npz, produced by the unmodified
 * reference).
 *
 * Why it exists next to the numpy oracle: (1) an independent second restatement in the language the product's host side is
 * written in; (2) a CPU baseline that is a fair fight -- one fused pass per stage over all host cores (OpenMP), where
 * the numpy oracle, like the reference's ATen loop, makes ten passes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define DPMO_API __attribute__((visibility("default")))

DPMO_API int dpmo_version(void) { return 1; }

DPMO_API int dpmo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

DPMO_API void dpmo_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ref :330   noise = noise_uncond + guidance_scale * (noise - noise_uncond) */
DPMO_API void dpmo_cfg_blend(const float* nu, const float* nc, float scale, float* out, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) out[i] = nu[i] + scale * (nc[i] - nu[i]);
}

/* ref :439   x0 = (x - sigma_t * noise) / alpha_t */
DPMO_API void dpmo_eps_to_x0(const float* x, const float* eps, float alpha, float sigma, float* x0, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) x0[i] = (x[i] - sigma * eps[i]) / alpha;
}

/* ref :290-298  the other parameterisations -> noise, per-sample scalars broadcast by the caller (one time for all here) */
DPMO_API void dpmo_xstart_to_noise(const float* x, const float* out_, float alpha, float sigma, float* noise, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) noise[i] = (x[i] - alpha * out_[i]) / sigma;          /* ref :292 */
}
DPMO_API void dpmo_v_to_noise(const float* x, const float* out_, float alpha, float sigma, float* noise, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) noise[i] = alpha * out_[i] + sigma * x[i];            /* ref :295 */
}
DPMO_API void dpmo_score_to_noise(const float* out_, float sigma, float* noise, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) noise[i] = -sigma * out_[i];                          /* ref :298 */
}

/* ref :416-425  dynamic thresholding: s = quantile(|x0|, ratio) per sample (torch.quantile: linear interpolation at the
   fp32 rank ratio * (n - 1), ATen's lerp), s = max(s, max_val), x0 = clamp(x0, -s, s) / s */
static int cmp_f32(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}
DPMO_API int dpmo_dynamic_threshold(float* x0, int64_t batch, int64_t per, float ratio, float max_val, float* s_out) {
  int fail = 0;
#pragma omp parallel for schedule(dynamic)
  for (int64_t b = 0; b < batch; ++b) {
    float* row = x0 + b * per;
    float* a = (float*)malloc((size_t)per * sizeof(float));
    if (!a) {
      fail = 1;
      continue;
    }
    int has_nan = 0;
    for (int64_t i = 0; i < per; ++i) {
      a[i] = fabsf(row[i]);
      has_nan |= a[i] != a[i];
    }
    if (!has_nan) qsort(a, (size_t)per, sizeof(float), cmp_f32);
    const float rank = ratio * (float)(per - 1);
    const int64_t lo = (int64_t)floorf(rank), hi = (int64_t)ceilf(rank);
    const float w = rank - (float)lo;
    const float d = a[hi] - a[lo];
    float s = (w < 0.5f) ? fmaf(w, d, a[lo]) : fmaf(w - 1.0f, d, a[hi]);   /* ATen lerp: one fused multiply-add */
    if (has_nan) s = NAN;              /* torch.quantile: a row holding a NaN gives NaN */
    if (s == s && !(s > max_val)) s = max_val;   /* torch.maximum (a NaN stays) */
    if (s_out) s_out[b] = s;
    for (int64_t i = 0; i < per; ++i) {
      float v = row[i];
      v = v < -s ? -s : (v > s ? s : v);
      row[i] = v / s;
    }
    free(a);
  }
  return fail;
}

/* ref :573, :585   x_t = cx * x - c0 * model_s
     dpmsolver++: cx = sigma_t / sigma_s, c0 = alpha_t * expm1(-h);  dpmsolver: cx = exp(log_alpha_t - log_alpha_s), c0 = sigma_t * expm1(h) */
DPMO_API void dpmo_update_first(const float* x, const float* m, float cx, float c0, float* out, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) out[i] = cx * x[i] - c0 * m[i];
}

/* ref :827-851   D1_0 = k0 * (m0 - m1), k0 = 1 / r0;   x_t = cx * x - c0 * m0 - c1 * D1_0
     (c1 carries the sign of the reference's last term: 0.5 * (alpha_t * phi_1) for 'dpmsolver', -(alpha_t * (phi_1 / h + 1)) for
     'taylor' -- subtracting the negated coefficient is the reference's addition, bit for bit) */
DPMO_API void dpmo_update_ms2(const float* x, const float* m0, const float* m1, float cx, float c0, float k0, float c1, float* out,
                              int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float d = k0 * (m0[i] - m1[i]);
    out[i] = (cx * x[i] - c0 * m0[i]) - c1 * d;
  }
}

/* ref :879-903   D1_0 = k0 * (m0 - m1), D1_1 = k1 * (m1 - m2), D1 = D1_0 + k2 * (D1_0 - D1_1), D2 = k3 * (D1_0 - D1_1)
     dpmsolver++: x_t = cx * x - c0 * m0 + c1 * D1 - c2 * D2;   dpmsolver: x_t = cx * x - c0 * m0 - c1 * D1 - c2 * D2  (plus = 0) */
DPMO_API void dpmo_update_ms3(const float* x, const float* m0, const float* m1, const float* m2, float cx, float c0, float c1,
                              float c2, float k0, float k1, float k2, float k3, int plus, float* out, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float d10 = k0 * (m0[i] - m1[i]);
    const float d11 = k1 * (m1[i] - m2[i]);
    const float dd = d10 - d11;
    const float D1 = d10 + k2 * dd;
    const float D2 = k3 * dd;
    const float base = cx * x[i] - c0 * m0[i];
    out[i] = plus ? (base + c1 * D1) - c2 * D2 : (base - c1 * D1) - c2 * D2;
  }
}

/* The fused DPM-Solver++(2M) stage a GPU launch performs, as ONE pass: the model value from the fresh noise prediction
   (ref :439), the second-order multistep update (ref :827-831), both written.  first != 0: the first stage of a trajectory
   (ref :573), no cached value yet. */
DPMO_API void dpmo_stage_2m(const float* x, const float* eps, const float* m_prev, float alpha_e, float sigma_e, float cx, float c0,
                            float k0, float c1, int first, float* x_out, float* m_out, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float m = (x[i] - sigma_e * eps[i]) / alpha_e;
    const float base = cx * x[i] - c0 * m;
    x_out[i] = first ? base : base - c1 * (k0 * (m - m_prev[i]));
    if (m_out) m_out[i] = m;
  }
}
