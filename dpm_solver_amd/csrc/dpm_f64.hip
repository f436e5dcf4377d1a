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

// the operands of one element, loaded before anything is computed or stored: U elements per lane in flight (the kernels below
// unroll by hand -- the pointers of Ptrs64 may alias as far as the compiler knows, so it would not hoist loads over stores)
struct In64 {
  double xe, o0, o1, gg;      // prologue
  double x, h1, h2, mk, ba, bb;  // combine + blend
};
__device__ __forceinline__ void load_prologue64(const Ptrs64& q, const KParams64& p, int64_t i, int64_t ie, In64& v) {
  const bool need_xe = (p.flags & DPM_F_TO_X0) || p.model_type == DPM_MODEL_X_START || p.model_type == DPM_MODEL_V;
  v.xe = need_xe ? q.xe[i] : 0.;
  v.o0 = q.e0[ie];
  v.o1 = p.guidance == DPM_GUIDE_CFG ? q.e1[ie] : 0.;
  v.gg = p.guidance == DPM_GUIDE_CLASSIFIER ? q.g[i] : 0.;
}
__device__ __forceinline__ void load_rest64(const Ptrs64& q, const KParams64& p, int64_t i, In64& v) {
  const int f = p.form;
  v.x = f != DPM_FORM_DENOISE ? q.x[i] : 0.;
  v.h1 = (f == DPM_FORM_TWO || f == DPM_FORM_MS3 || f == DPM_FORM_SS3T) ? q.h1[i] : 0.;
  v.h2 = (f == DPM_FORM_MS3 || f == DPM_FORM_SS3T) ? q.h2[i] : 0.;
  if (q.mask) {
    v.mk = q.mask[i % q.mask_period];
    v.ba = q.ba[i];
    v.bb = q.bb ? q.bb[i] : 0.;
  }
}
__device__ __forceinline__ double model_value64(const In64& v, const KParams64& p) { return prologue64(v.xe, v.o0, v.o1, v.gg, p); }
__device__ __forceinline__ void finish64(const Ptrs64& q, const KParams64& p, int64_t i, const In64& v, double mn) {
  double o = combine64(v.x, mn, v.h1, v.h2, p);
  if (q.mask) {  // x * mask + (1 - mask) * (alpha * a + sigma * b): the DPM_F_BLEND epilogue
    const double r = q.bb ? p.blend_alpha * v.ba + p.blend_sigma * v.bb : v.ba;
    o = o * v.mk + (1. - v.mk) * r;
  }
  q.xo[i] = o;
  if (q.xo2) q.xo2[i] = o;
  if (p.flags & DPM_F_STORE_M) q.mo[i] = mn;
}

constexpr int U64 = 2;  // elements per lane in flight: workgroup b owns the U64 consecutive 256-element rows from b * U64 on
__global__ __launch_bounds__(256) void stage_kernel_f64(const Ptrs64 q, const KParams64 p, int64_t n) {
  const int64_t i0 = (int64_t)blockIdx.x * (U64 * 256) + threadIdx.x;
  In64 v[U64];
#pragma unroll
  for (int u = 0; u < U64; ++u) {
    const int64_t i = i0 + u * 256 < n ? i0 + u * 256 : (i0 < n ? i0 : 0);  // clamped: the loads are unconditional
    const int64_t ie = q.eps_stride ? (i / q.per_sample) * q.eps_stride + i % q.per_sample : i;
    load_prologue64(q, p, i, ie, v[u]);
    load_rest64(q, p, i, v[u]);
  }
#pragma unroll
  for (int u = 0; u < U64; ++u) {
    const int64_t i = i0 + u * 256;
    if (i < n) finish64(q, p, i, v[u], model_value64(v[u], p));
  }
}

// ---- dynamic thresholding in double (ref :416-425): one workgroup per sample.  The two order statistics torch.quantile
// interpolates between -- ascending ranks lo and lo + 1 of |x0| -- by a radix select over the 63-bit patterns (non-negative
// doubles order like their bit patterns): histogram passes of 11 / 11 / 11 / 11 / 11 / 8 bits over the sample, x0 recomputed
// from the inputs in every pass (same arithmetic, same bits; a sample's inputs stay in L2 between the passes), until the
// selected bin holds few enough values to finish among them in LDS (normally after two passes: the binade, then 11 mantissa
// bits): one more pass collects the bin's members -- and the smallest value above the bin -- and the ranks are counted there.
// Then clamp, divide, combine.
constexpr int T64 = 1024, NB64 = 2048;
constexpr int CAP64 = NB64 / 2;  // the candidate list (64-bit patterns) reuses the histogram's LDS
constexpr int UT64 = 4;          // elements per lane in flight in the select passes (four operands each) ...
constexpr int UF64 = 2;          // ... and in the store phase (ten): 128 registers per lane at 1024 threads
__global__ __launch_bounds__(T64) void stage_thresh_kernel_f64(const Ptrs64 q, const KParams64 p, int64_t per_sample, int64_t lo,
                                                               int hi_differs, double w) {
  __shared__ __align__(8) uint32_t hist[NB64];
  __shared__ uint64_t sh_prefix, sh_min, sh_a, sh_b;
  __shared__ int64_t sh_rank, sh_cnt;
  __shared__ uint32_t sh_n;
  constexpr uint64_t TOP = 0x7fffffffffffffffull;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x, base = b * per_sample, ebase = b * (q.eps_stride ? q.eps_stride : per_sample);
  // the patterns of elements j, j + T64, .. (UT64 of them, loads issued together); beyond the sample: TOP + 1 (matches no prefix)
  bool nan_seen = false;  // torch.quantile: a sample holding a NaN anywhere gives NaN (every pass sees every element)
  auto bitsU = [&](int64_t j, uint64_t (&u)[UT64]) {
    In64 v[UT64];
#pragma unroll
    for (int r = 0; r < UT64; ++r) {
      const int64_t jj = j + (int64_t)r * T64 < per_sample ? j + (int64_t)r * T64 : j;
      load_prologue64(q, p, base + jj, ebase + jj, v[r]);
    }
#pragma unroll
    for (int r = 0; r < UT64; ++r)
      u[r] = j + (int64_t)r * T64 < per_sample ? ((uint64_t)__double_as_longlong(model_value64(v[r], p)) & TOP) : ~0ull;
#pragma unroll
    for (int r = 0; r < UT64; ++r) nan_seen |= u[r] != ~0ull && u[r] > 0x7ff0000000000000ull;
  };
  uint64_t prefix = 0ull, known = 0ull;
  int64_t rank = lo, cnt = per_sample;
  const int shifts[6] = {52, 41, 30, 19, 8, 0};
  const int widths[6] = {11, 11, 11, 11, 11, 8};
  for (int pass = 0; pass < 6 && cnt > CAP64; ++pass) {
    for (int j = tid; j < NB64; j += T64) hist[j] = 0u;
    __syncthreads();
    const int sh = shifts[pass];
    const uint64_t dm = (1ull << widths[pass]) - 1ull;
    for (int64_t j = tid; j < per_sample; j += (int64_t)UT64 * T64) {
      uint64_t u[UT64];
      bitsU(j, u);
#pragma unroll
      for (int r = 0; r < UT64; ++r)
        if (u[r] != ~0ull && (u[r] & known) == prefix) atomicAdd(&hist[(u[r] >> sh) & dm], 1u);
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
    cnt = sh_cnt;
    __syncthreads();
  }
  uint64_t a_bits, b_bits;
  if (known != TOP) {
    // few members left in the selected bin (all of the sample when it is that small): collect them and the smallest value
    // above the bin, finish by rank counting -- member v is the wanted one when #{smaller} <= rank < #{smaller} + #{equal}
    uint64_t* list = reinterpret_cast<uint64_t*>(hist);
    if (tid == 0) {
      sh_n = 0u;
      sh_min = TOP;
      sh_a = sh_b = 0ull;
    }
    __syncthreads();
    uint64_t m = TOP;
    for (int64_t j = tid; j < per_sample; j += (int64_t)UT64 * T64) {
      uint64_t u[UT64];
      bitsU(j, u);
#pragma unroll
      for (int r = 0; r < UT64; ++r) {
        if (u[r] == ~0ull) continue;
        if ((u[r] & known) == prefix)
          list[atomicAdd(&sh_n, 1u)] = u[r];
        else if ((u[r] & known) > prefix && u[r] < m)
          m = u[r];
      }
    }
    if (m != TOP) atomicMin(reinterpret_cast<unsigned long long*>(&sh_min), (unsigned long long)m);
    __syncthreads();
    const uint32_t nl = sh_n;  // == cnt
    if ((uint32_t)tid < nl) {
      const uint64_t v = list[tid];
      int64_t less = 0, eq = 0;
      for (uint32_t jj = 0; jj < nl; ++jj) {
        const uint64_t o = list[jj];
        less += o < v;
        eq += o == v;
      }
      if (less <= rank && rank < less + eq) sh_a = v;          // every thread holding this value writes the same bits
      if (less <= rank + 1 && rank + 1 < less + eq) sh_b = v;
    }
    __syncthreads();
    a_bits = sh_a;
    // the next order statistic: inside the bin when rank + 1 is, otherwise the smallest value above it (if there is none,
    // lo is the sample's last rank and torch reads the same element twice)
    b_bits = (rank + 1 < (int64_t)nl) ? sh_b : (sh_min != TOP ? sh_min : a_bits);
    if (!hi_differs) b_bits = a_bits;
  } else {
    // every bit settled by histograms (more than CAP64 copies of one value all the way down)
    a_bits = prefix;
    b_bits = a_bits;
    if (hi_differs && rank + 1 >= cnt) {  // the next order statistic is the smallest value above a (if any)
      if (tid == 0) sh_min = TOP;
      __syncthreads();
      uint64_t m = TOP;
      for (int64_t j = tid; j < per_sample; j += (int64_t)UT64 * T64) {
        uint64_t u[UT64];
        bitsU(j, u);
#pragma unroll
        for (int r = 0; r < UT64; ++r)
          if (u[r] != ~0ull && u[r] > a_bits && u[r] < m) m = u[r];
      }
      atomicMin(reinterpret_cast<unsigned long long*>(&sh_min), (unsigned long long)m);
      __syncthreads();
      if (sh_min != TOP) b_bits = sh_min;
    }
  }
  if (__syncthreads_or(nan_seen)) a_bits = b_bits = TOP;  // (a NaN pattern)
  const double a = __longlong_as_double((long long)a_bits), bb = __longlong_as_double((long long)b_bits);
  const double diff = bb - a;
  const double qv = w < 0.5 ? __builtin_fma(w, diff, a) : __builtin_fma(w - 1., diff, bb);  // ATen lerp: one fused multiply-add
  const double s = qv != qv ? qv : fmax(qv, p.thr_max);               // ref :423, torch.maximum keeps a NaN
  for (int64_t j = tid; j < per_sample; j += (int64_t)UF64 * T64) {
    In64 v[UF64];
#pragma unroll
    for (int r = 0; r < UF64; ++r) {
      const int64_t jj = j + (int64_t)r * T64 < per_sample ? j + (int64_t)r * T64 : j;
      load_prologue64(q, p, base + jj, ebase + jj, v[r]);
      load_rest64(q, p, base + jj, v[r]);
    }
#pragma unroll
    for (int r = 0; r < UF64; ++r) {
      const int64_t jj = j + (int64_t)r * T64;
      if (jj < per_sample) {
        const double mv = model_value64(v[r], p);
        finish64(q, p, base + jj, v[r], (mv != mv ? mv : fmin(fmax(mv, -s), s)) / s);  // ref :424, torch.clamp keeps a NaN
      }
    }
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
    const int64_t blocks = (b->n + U64 * 256 - 1) / (U64 * 256);  // one tile per workgroup, no grid-stride loop
    if (blocks > 0x7fffffff) return dpm_set_error(DPM_ERR_UNSUPPORTED, "stage_launch: double state of %lld elements", (long long)b->n);
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
