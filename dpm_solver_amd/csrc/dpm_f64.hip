// dpm_f64.hip -- double-precision state (DPM_DTYPE_F64): the reference computes in whatever dtype torch's type promotion
// yields (ref :14, :105-107, :573-576), so `sample(x.double())` is a double-precision run -- every tensor operation in double,
// the scalars either doubles (NoiseScheduleVP(dtype=torch.float64): dpm_buffers.coef64) or fp32 values converted exactly (an
// fp32 schedule: the reference's coefficients are fp32 tensors there).  NOT a performance path: one run-time dispatched kernel,
// one element per lane, true IEEE divisions, the reference's association (compiled with -ffp-contract=off); dynamic
// thresholding by one workgroup per sample with an exact radix select on the 63-bit patterns of |x0| (torch.quantile's
// semantics in double: rank = q (n - 1) evaluated in double, ATen's lerp).
#include "dpm_device.hpp"

namespace {

struct KParams64 {
  double alpha_e, sigma_e, cfg_scale, cg_scale, cx, c0, c1, c2, k0, k1, k2, k3, k4, thr_ratio, thr_max, blend_alpha, blend_sigma;
  uint32_t flags;
  int32_t model_type, form, guidance;
};

KParams64 make_params64(const dpm_stage* st, const dpm_stage_f64* c) {
  KParams64 p;
  if (c) {
    p.alpha_e = c->alpha_e; p.sigma_e = c->sigma_e; p.cfg_scale = c->cfg_scale; p.cg_scale = c->cg_scale;
    p.cx = c->cx; p.c0 = c->c0; p.c1 = c->c1; p.c2 = c->c2;
    p.k0 = c->k[0]; p.k1 = c->k[1]; p.k2 = c->k[2]; p.k3 = c->k[3]; p.k4 = c->k[4];
    p.thr_ratio = c->thr_ratio; p.thr_max = c->thr_max; p.blend_alpha = c->blend_alpha; p.blend_sigma = c->blend_sigma;
  } else {  // fp32 scalars meeting double tensors: converted exactly (torch's type promotion)
    p.alpha_e = st->alpha_e; p.sigma_e = st->sigma_e; p.cfg_scale = st->cfg_scale; p.cg_scale = st->cg_scale;
    p.cx = st->cx; p.c0 = st->c0; p.c1 = st->c1; p.c2 = st->c2;
    p.k0 = st->k[0]; p.k1 = st->k[1]; p.k2 = st->k[2]; p.k3 = st->k[3]; p.k4 = st->k[4];
    p.thr_ratio = st->thr_ratio; p.thr_max = st->thr_max; p.blend_alpha = st->blend_alpha; p.blend_sigma = st->blend_sigma;
  }
  p.flags = st->flags;
  p.model_type = st->model_type;
  p.form = st->form;
  p.guidance = st->guidance;
  return p;
}

// raw network output -> noise prediction (noise_pred_fn, ref :288-298)
__device__ __forceinline__ double to_noise64(double o, double xe, const KParams64& p) {
  switch (p.model_type) {
    case DPM_MODEL_X_START: return (xe - p.alpha_e * o) / p.sigma_e;
    case DPM_MODEL_V: return p.alpha_e * o + p.sigma_e * xe;
    case DPM_MODEL_SCORE: return (-p.sigma_e) * o;
    default: return o;
  }
}
// everything up to (not including) thresholding: eps, or x0 when the stage converts (ref :322-330, :315-321, :439)
__device__ __forceinline__ double prologue64(double xe, double o0, double o1, double gg, const KParams64& p) {
  double eps;
  if (p.guidance == DPM_GUIDE_CFG) {
    const double nu = to_noise64(o1, xe, p), nc = to_noise64(o0, xe, p);
    eps = nu + p.cfg_scale * (nc - nu);
  } else if (p.guidance == DPM_GUIDE_CLASSIFIER) {
    eps = to_noise64(o0, xe, p) - p.cg_scale * gg;
  } else {
    eps = to_noise64(o0, xe, p);
  }
  if (p.flags & DPM_F_TO_X0) return (xe - p.sigma_e * eps) / p.alpha_e;
  return eps;
}
// the update forms, reference association (dpm_stage_kernel.hpp: combine)
__device__ __forceinline__ double combine64(double x, double mn, double h1, double h2, const KParams64& p) {
  switch (p.form) {
    case DPM_FORM_LIN1: return p.cx * x - p.c0 * mn;
    case DPM_FORM_TWO: {
      const double D = p.k0 * (mn - h1);
      const double P = (p.flags & DPM_F_BASE_HIST) ? h1 : mn;
      return (p.cx * x - p.c0 * P) - p.c1 * D;
    }
    case DPM_FORM_MS3: {
      const double D1_0 = p.k0 * (mn - h1), D1_1 = p.k1 * (h1 - h2), dd = D1_0 - D1_1;
      const double D1 = D1_0 + p.k2 * dd, D2 = p.k3 * dd;
      return ((p.cx * x - p.c0 * mn) - p.c1 * D1) - p.c2 * D2;
    }
    case DPM_FORM_SS3T: {
      const double D1_0 = p.k0 * (h2 - h1), D1_1 = p.k1 * (mn - h1);
      const double D1 = (p.k2 * D1_0 - p.k3 * D1_1) / p.k4, D2 = (2. * (D1_1 - D1_0)) / p.k4;
      return ((p.cx * x - p.c0 * h1) - p.c1 * D1) - p.c2 * D2;
    }
    default: return mn;
  }
}

struct Ptrs64 {
  const double *x, *xe, *e0, *e1, *g, *h1, *h2;
  double *xo, *mo, *xo2;
  const double *mask, *ba, *bb;
  int64_t mask_period, per_sample, eps_stride;
};

__device__ __forceinline__ double model_value64(const Ptrs64& q, const KParams64& p, int64_t i, int64_t ie) {
  const bool need_xe = (p.flags & DPM_F_TO_X0) || p.model_type == DPM_MODEL_X_START || p.model_type == DPM_MODEL_V;
  return prologue64(need_xe ? q.xe[i] : 0., q.e0[ie], p.guidance == DPM_GUIDE_CFG ? q.e1[ie] : 0.,
                    p.guidance == DPM_GUIDE_CLASSIFIER ? q.g[i] : 0., p);
}
__device__ __forceinline__ void finish64(const Ptrs64& q, const KParams64& p, int64_t i, double mn) {
  const int f = p.form;
  const double xv = f != DPM_FORM_DENOISE ? q.x[i] : 0.;
  const double h1 = (f == DPM_FORM_TWO || f == DPM_FORM_MS3 || f == DPM_FORM_SS3T) ? q.h1[i] : 0.;
  const double h2 = (f == DPM_FORM_MS3 || f == DPM_FORM_SS3T) ? q.h2[i] : 0.;
  double o = combine64(xv, mn, h1, h2, p);
  if (q.mask) {  // x * mask + (1 - mask) * (alpha * a + sigma * b): the DPM_F_BLEND epilogue
    const double m = q.mask[i % q.mask_period];
    const double r = q.bb ? p.blend_alpha * q.ba[i] + p.blend_sigma * q.bb[i] : q.ba[i];
    o = o * m + (1. - m) * r;
  }
  q.xo[i] = o;
  if (q.xo2) q.xo2[i] = o;
  if (p.flags & DPM_F_STORE_M) q.mo[i] = mn;
}

__global__ __launch_bounds__(256) void stage_kernel_f64(const Ptrs64 q, const KParams64 p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t ie = q.eps_stride ? (i / q.per_sample) * q.eps_stride + i % q.per_sample : i;
    finish64(q, p, i, model_value64(q, p, i, ie));
  }
}

// ---- dynamic thresholding in double (ref :416-425): one workgroup per sample.  The two order statistics torch.quantile
// interpolates between -- ascending ranks lo and lo + 1 of |x0| -- by a radix select over the 63-bit patterns (non-negative
// doubles order like their bit patterns): six passes of 11 / 11 / 11 / 11 / 11 / 8 bits, x0 recomputed from the inputs in
// every pass (same arithmetic, same bits), then clamp, divide, combine.
constexpr int T64 = 1024, NB64 = 2048;
__global__ __launch_bounds__(T64) void stage_thresh_kernel_f64(const Ptrs64 q, const KParams64 p, int64_t per_sample, int64_t lo,
                                                               int hi_differs, double w) {
  __shared__ uint32_t hist[NB64];
  __shared__ uint64_t sh_prefix, sh_min;
  __shared__ int64_t sh_rank, sh_cnt;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x, base = b * per_sample, ebase = b * (q.eps_stride ? q.eps_stride : per_sample);
  auto bits_at = [&](int64_t j) -> uint64_t {
    return (uint64_t)__double_as_longlong(model_value64(q, p, base + j, ebase + j)) & 0x7fffffffffffffffull;
  };
  uint64_t prefix = 0ull, known = 0ull;
  int64_t rank = lo;
  const int shifts[6] = {52, 41, 30, 19, 8, 0};
  const int widths[6] = {11, 11, 11, 11, 11, 8};
  for (int pass = 0; pass < 6; ++pass) {
    for (int j = tid; j < NB64; j += T64) hist[j] = 0u;
    __syncthreads();
    const int sh = shifts[pass];
    const uint64_t dm = (1ull << widths[pass]) - 1ull;
    for (int64_t j = tid; j < per_sample; j += T64) {
      const uint64_t u = bits_at(j);
      if ((u & known) == prefix) atomicAdd(&hist[(u >> sh) & dm], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int64_t r = rank;
      uint64_t bin = 0;
      for (uint64_t d = 0; d <= dm; ++d) {
        if (r < (int64_t)hist[d]) {
          bin = d;
          sh_cnt = hist[d];
          break;
        }
        r -= hist[d];
      }
      sh_prefix = prefix | (bin << sh);
      sh_rank = r;
    }
    __syncthreads();
    prefix = sh_prefix;
    known |= dm << sh;
    rank = sh_rank;
    __syncthreads();
  }
  const uint64_t a_bits = prefix;
  uint64_t b_bits = a_bits;
  if (hi_differs && rank + 1 >= sh_cnt) {  // the next order statistic is the smallest value above a (if any)
    if (tid == 0) sh_min = 0x7fffffffffffffffull;
    __syncthreads();
    uint64_t m = 0x7fffffffffffffffull;
    for (int64_t j = tid; j < per_sample; j += T64) {
      const uint64_t u = bits_at(j);
      if (u > a_bits && u < m) m = u;
    }
    atomicMin(reinterpret_cast<unsigned long long*>(&sh_min), (unsigned long long)m);
    __syncthreads();
    if (sh_min != 0x7fffffffffffffffull) b_bits = sh_min;
  }
  const double a = __longlong_as_double((long long)a_bits), bb = __longlong_as_double((long long)b_bits);
  const double diff = bb - a;
  const double qv = w < 0.5 ? a + w * diff : bb - diff * (1. - w);  // ATen lerp
  const double s = fmax(qv, p.thr_max);                               // ref :423
  for (int64_t j = tid; j < per_sample; j += T64) {
    const double x0 = model_value64(q, p, base + j, ebase + j);
    finish64(q, p, base + j, fmin(fmax(x0, -s), s) / s);              // ref :424
  }
}

__global__ __launch_bounds__(256) void add_noise_kernel_f64(const double* __restrict__ x, const double* __restrict__ noise,
                                                            double* __restrict__ out, int64_t n, double a, double sg) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a * x[i] + sg * noise[i];
}
__global__ __launch_bounds__(256) void blend_kernel_f64(const double* __restrict__ x, const double* __restrict__ mask,
                                                        const double* __restrict__ a, const double* __restrict__ b,
                                                        double* __restrict__ out, int64_t n, int64_t period, double alpha, double sigma) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double m = mask[i % period];
    const double r = b ? alpha * a[i] + sigma * b[i] : a[i];
    out[i] = x[i] * m + (1. - m) * r;
  }
}
}  // namespace

// entered from dpm_stage_launch (dpm_kernels.hip) for state_dtype == eps_dtype == DPM_DTYPE_F64; arguments checked there
int dpm_launch_f64(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop) {
  for (const void* ptr : {b->x, b->xe, b->e0, b->e1, b->g, b->h1, b->h2, (const void*)b->x_out, (const void*)b->m_out,
                          (const void*)b->x_out2, b->mask, b->blend_a, b->blend_b})
    if (!aligned(ptr, 8)) return dpm_set_error(DPM_ERR_ALIGN, "stage_launch: a double buffer is not 8-byte aligned");
  const KParams64 p = make_params64(st, b->coef64);
  Ptrs64 q;
  std::memset(&q, 0, sizeof q);
  q.x = static_cast<const double*>(b->x);
  q.xe = static_cast<const double*>(b->xe ? b->xe : b->x);
  q.e0 = static_cast<const double*>(b->e0);
  q.e1 = static_cast<const double*>(b->e1);
  q.g = static_cast<const double*>(b->g);
  q.h1 = static_cast<const double*>(b->h1);
  q.h2 = static_cast<const double*>(b->h2);
  q.xo = static_cast<double*>(b->x_out);
  q.mo = static_cast<double*>(b->m_out);
  q.xo2 = static_cast<double*>(b->x_out2);
  const bool blend = (st->flags & DPM_F_BLEND) != 0;
  q.mask = blend ? static_cast<const double*>(b->mask) : nullptr;
  q.ba = blend ? static_cast<const double*>(b->blend_a) : nullptr;
  q.bb = blend ? static_cast<const double*>(b->blend_b) : nullptr;
  q.mask_period = blend ? b->mask_period : 1;
  q.per_sample = b->n / b->batch;
  q.eps_stride = (b->eps_stride == q.per_sample) ? 0 : b->eps_stride;
  const LaunchCtx ctx{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)};
  if (st->flags & DPM_F_THRESH) {
    if (b->batch > 0x7fffffff) return dpm_set_error(DPM_ERR_UNSUPPORTED, "thresholding: batch out of range");
    // torch.quantile: rank = q * (n - 1) in the tensor's dtype (double here)
    const double rank = p.thr_ratio * (double)(q.per_sample - 1);
    const double lo = std::floor(rank), hi = std::ceil(rank);
    launch(stage_thresh_kernel_f64, dim3((unsigned)b->batch), dim3(T64), 0, ctx, q, p, q.per_sample, (int64_t)lo, hi != lo ? 1 : 0,
           rank - lo);
  } else {
    const DeviceInfo& di = device_info();
    int64_t blocks = (b->n + 255) / 256;
    const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * 16;
    if (blocks > cap) blocks = cap;
    launch(stage_kernel_f64, dim3((unsigned)blocks), dim3(256), 0, ctx, q, p, b->n);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "stage kernel launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

int dpm_add_noise_f64(double alpha, double sigma, const void* x, const void* noise, void* out, int64_t n, void* stream) {
  const DeviceInfo& di = device_info();
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(add_noise_kernel_f64, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const double*>(x), static_cast<const double*>(noise), static_cast<double*>(out), n, alpha, sigma);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "add_noise launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

int dpm_blend_f64(const void* x, const void* mask, const void* a, const void* b, double alpha, double sigma, void* out, int64_t n,
                  int64_t period, void* stream) {
  const DeviceInfo& di = device_info();
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(blend_kernel_f64, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const double*>(x), static_cast<const double*>(mask), static_cast<const double*>(a),
                     static_cast<const double*>(b), static_cast<double*>(out), n, period, alpha, sigma);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "blend launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}
