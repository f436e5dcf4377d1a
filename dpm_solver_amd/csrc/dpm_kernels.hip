// dpm_kernels.hip -- gfx950 (MI355X, CDNA4) device code of the DPM-Solver engine.
//
// C ABI entry points of the device side; the stage kernels live in dpm_device.hpp and are instantiated per dtype
// pair in dpm_stage_*.hip.
#include "dpm_device.hpp"
#include "dpm_coef.hpp"

namespace dpmk {
// one context per device ordinal (dpm_device.hpp): the only state of the library that outlives a call
DeviceContext& device_context(int dev) {
  static DeviceContext ctx[64];
  return ctx[(dev < 0 ? 0 : dev) % 64];
}
uint32_t* cluster_fault_word(int dev, bool create) {
  DeviceContext& c = device_context(dev);
  std::lock_guard<std::mutex> lk(c.mu);
  if (!c.fault && create) {
    void* p = nullptr;
    // portable: a host of several devices (one thread per device) maps the word into every device's address space
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) {
      c.fault = static_cast<uint32_t*>(p);
      *c.fault = 0u;
    }
  }
  return c.fault;
}
}  // namespace dpmk

// one translation unit per (state, eps) dtype pair
int dpm_launch_f32_f32(const dpm_stage*, const dpm_buffers*, void*, void*, void*, const dpm_stage*, const int32_t*);
int dpm_launch_f32_f16(const dpm_stage*, const dpm_buffers*, void*, void*, void*, const dpm_stage*, const int32_t*);
int dpm_launch_f32_bf16(const dpm_stage*, const dpm_buffers*, void*, void*, void*, const dpm_stage*, const int32_t*);
int dpm_launch_f16_f16(const dpm_stage*, const dpm_buffers*, void*, void*, void*, const dpm_stage*, const int32_t*);
int dpm_launch_bf16_bf16(const dpm_stage*, const dpm_buffers*, void*, void*, void*, const dpm_stage*, const int32_t*);
int dpm_launch_f64(const dpm_stage*, const dpm_buffers*, void*, void*, void*);                     // dpm_f64.hip
int dpm_add_noise_f64(double, double, const void*, const void*, void*, int64_t, void*);
int dpm_blend_f64(const void*, const void*, const void*, const void*, double, double, void*, int64_t, int64_t, void*);
int dpm_schedule_table_is_f64(const dpm_schedule* s);                                              // dpm_host.cpp
int dpm_launch_multi_f32_f32(const dpm_stage*, const dpm_buffers*, int, void*, void*, void*);
int dpm_launch_multi_f32_f16(const dpm_stage*, const dpm_buffers*, int, void*, void*, void*);
int dpm_launch_multi_f32_bf16(const dpm_stage*, const dpm_buffers*, int, void*, void*, void*);
int dpm_launch_multi_f16_f16(const dpm_stage*, const dpm_buffers*, int, void*, void*, void*);
int dpm_launch_multi_bf16_bf16(const dpm_stage*, const dpm_buffers*, int, void*, void*, void*);

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
int dpm_stage_launch_dyn(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop,
                         const dpm_stage* dyn, const int32_t* skip) {
  if (!st || !b) return dpm_set_error(DPM_ERR_ARG, "stage_launch: null pointer");
  if (b->n < 0 || b->batch < 1 || (b->n % b->batch) != 0)
    return dpm_set_error(DPM_ERR_ARG, "stage_launch: n=%lld is not a multiple of batch=%lld", (long long)b->n, (long long)b->batch);
  if (b->n == 0) return DPM_OK;  // empty batch: nothing to do (torch allows zero-sized tensors)
  const bool needs_x = st->form != DPM_FORM_DENOISE;
  const bool needs_h1 = st->form == DPM_FORM_TWO || st->form == DPM_FORM_MS3 || st->form == DPM_FORM_SS3T;
  const bool needs_h2 = st->form == DPM_FORM_MS3 || st->form == DPM_FORM_SS3T;
  const bool need_xe = (st->flags & DPM_F_TO_X0) || st->model_type == DPM_MODEL_X_START || st->model_type == DPM_MODEL_V;
  if (!b->e0 || !b->x_out) return dpm_set_error(DPM_ERR_ARG, "stage_launch: e0 / x_out must not be null");
  if ((needs_x || need_xe) && !b->x && !b->xe) return dpm_set_error(DPM_ERR_ARG, "stage_launch: x is null");
  if (needs_x && !b->x) return dpm_set_error(DPM_ERR_ARG, "stage_launch: x is null");
  if (needs_h1 && !b->h1) return dpm_set_error(DPM_ERR_ARG, "stage_launch: form %d needs h1", st->form);
  if (needs_h2 && !b->h2) return dpm_set_error(DPM_ERR_ARG, "stage_launch: form %d needs h2", st->form);
  if ((st->flags & DPM_F_STORE_M) && !b->m_out) return dpm_set_error(DPM_ERR_ARG, "stage_launch: STORE_M without m_out");
  if (st->guidance == DPM_GUIDE_CFG && !b->e1) return dpm_set_error(DPM_ERR_ARG, "stage_launch: CFG needs e1");
  if (st->guidance == DPM_GUIDE_CLASSIFIER && !b->g) return dpm_set_error(DPM_ERR_ARG, "stage_launch: classifier guidance needs g");
  if (st->flags & DPM_F_BLEND) {
    if (!b->mask || !b->blend_a || b->mask_period < 1)
      return dpm_set_error(DPM_ERR_ARG, "stage_launch: DPM_F_BLEND needs mask, blend_a and mask_period >= 1");
    if (b->n % b->mask_period != 0)
      return dpm_set_error(DPM_ERR_ARG, "stage_launch: n=%lld is not a multiple of mask_period=%lld", (long long)b->n,
                           (long long)b->mask_period);
    if (b->mask_period >= ((int64_t)1 << 31) && b->mask_period != b->n)
      return dpm_set_error(DPM_ERR_UNSUPPORTED, "stage_launch: a broadcast mask of 2^31 or more elements");
  }
  if (b->eps_stride != 0 && b->eps_stride < b->n / b->batch)
    return dpm_set_error(DPM_ERR_ARG, "stage_launch: eps_stride=%lld is smaller than a sample (%lld elements)",
                         (long long)b->eps_stride, (long long)(b->n / b->batch));
  dpm_buffers bb = *b;
  if (!bb.x) bb.x = bb.xe;  // DENOISE form: only the evaluation state exists
  const int sd = bb.state_dtype, ed = bb.eps_dtype;
  if (sd == DPM_DTYPE_F64 || ed == DPM_DTYPE_F64) {  // double-precision state: one run-time dispatched kernel (dpm_f64.hip)
    if (sd != ed) return dpm_set_error(DPM_ERR_UNSUPPORTED, "stage_launch: a double state needs double network outputs (state=%d eps=%d)", sd, ed);
    if (dyn) return dpm_set_error(DPM_ERR_UNSUPPORTED, "stage_launch: device-resident coefficients with a double state");
    return dpm_launch_f64(st, &bb, stream, ev_start, ev_stop);
  }
  // A LARGE launch takes the fused multi-request kernel's shape -- one workgroup per super-tile instead of a grid capped at 8
  // workgroups per CU walking the tiles in a loop, an XCD-contiguous tile mapping for 2-byte states, two tiles per workgroup
  // for 4-byte states -- as a "group" of one request (Tuning::big_tiles; the same arithmetic, the same bits).  Stages the fused
  // family does not build (thresholding, mask blend, classifier guidance, SS3T / DENOISE, unaligned or strided operands)
  // come back MULTI_NOT_BUILT and take the single-request path below.
  if (!dyn && !(st->flags & (DPM_F_THRESH | DPM_F_BLEND))) {
    const int big = tuning_for(bb.opts).big_tiles;
    if (big > 0 && (bb.n / EPT + 255) / 256 >= big) {
      int (*fn)(const dpm_stage*, const dpm_buffers*, int, void*, void*, void*) = nullptr;
      if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_F32) fn = dpm_launch_multi_f32_f32;
      else if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_F16) fn = dpm_launch_multi_f32_f16;
      else if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_BF16) fn = dpm_launch_multi_f32_bf16;
      else if (sd == DPM_DTYPE_F16 && ed == DPM_DTYPE_F16) fn = dpm_launch_multi_f16_f16;
      else if (sd == DPM_DTYPE_BF16 && ed == DPM_DTYPE_BF16) fn = dpm_launch_multi_bf16_bf16;
      if (fn) {
        const int rc = fn(st, &bb, 1, stream, ev_start, ev_stop);
        if (rc != MULTI_NOT_BUILT) return rc;
      }
    }
  }
  if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_F32) return dpm_launch_f32_f32(st, &bb, stream, ev_start, ev_stop, dyn, skip);
  if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_F16) return dpm_launch_f32_f16(st, &bb, stream, ev_start, ev_stop, dyn, skip);
  if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_BF16) return dpm_launch_f32_bf16(st, &bb, stream, ev_start, ev_stop, dyn, skip);
  if (sd == DPM_DTYPE_F16 && ed == DPM_DTYPE_F16) return dpm_launch_f16_f16(st, &bb, stream, ev_start, ev_stop, dyn, skip);
  if (sd == DPM_DTYPE_BF16 && ed == DPM_DTYPE_BF16) return dpm_launch_bf16_bf16(st, &bb, stream, ev_start, ev_stop, dyn, skip);
  return dpm_set_error(DPM_ERR_UNSUPPORTED, "stage_launch: unsupported dtype pair state=%d eps=%d", sd, ed);
}

int dpm_stage_launch_ev(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop) {
  return dpm_stage_launch_dyn(st, b, stream, ev_start, ev_stop, nullptr, nullptr);
}

extern "C" int dpm_stage_launch(const dpm_stage* st, const dpm_buffers* b, void* stream) {
  return dpm_stage_launch_ev(st, b, stream, nullptr, nullptr);
}

// One stage of n_req requests.  ev_start / ev_stop (optional) are arrays of n_req events: a fused launch of the
// requests [r0, r1) is bracketed by ev_start[r0] / ev_stop[r0] and fused_first[r] = r0 for its members (a request
// launched on its own has fused_first[r] = r), so the caller can spread the duration.
int dpm_stage_launch_multi_ev(const dpm_stage* st, const dpm_buffers* bs, int n_req, void* stream, void** ev_start,
                              void** ev_stop, int* fused_first) {
  if (!st || !bs || n_req < 1) return dpm_set_error(DPM_ERR_ARG, "stage_launch_multi: bad arguments");
  int done = 0;  // requests already advanced by fused launches
  bool same = n_req > 1 && tuning_for(bs[0].opts).multi_fuse != 0;
  for (int r = 1; r < n_req && same; ++r)
    same = bs[r].n == bs[0].n && bs[r].batch == bs[0].batch && bs[r].state_dtype == bs[0].state_dtype &&
           bs[r].eps_dtype == bs[0].eps_dtype;
  if (same && bs[0].n > 0 && bs[0].batch > 0 && bs[0].n % bs[0].batch == 0) {
    // the argument checks of the single launch, once per request
    const bool needs_h1 = st->form == DPM_FORM_TWO || st->form == DPM_FORM_MS3;
    const bool needs_h2 = st->form == DPM_FORM_MS3;
    for (int r = 0; r < n_req && same; ++r) {
      const dpm_buffers& b = bs[r];
      same = b.e0 && b.x_out && (b.x || b.xe) && (!needs_h1 || b.h1) && (!needs_h2 || b.h2) &&
             (!(st->flags & DPM_F_STORE_M) || b.m_out) && (st->guidance != DPM_GUIDE_CFG || b.e1);
    }
    const int sd = bs[0].state_dtype, ed = bs[0].eps_dtype;
    int (*fn)(const dpm_stage*, const dpm_buffers*, int, void*, void*, void*) = nullptr;
    if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_F32) fn = dpm_launch_multi_f32_f32;
    else if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_F16) fn = dpm_launch_multi_f32_f16;
    else if (sd == DPM_DTYPE_F32 && ed == DPM_DTYPE_BF16) fn = dpm_launch_multi_f32_bf16;
    else if (sd == DPM_DTYPE_F16 && ed == DPM_DTYPE_F16) fn = dpm_launch_multi_f16_f16;
    else if (sd == DPM_DTYPE_BF16 && ed == DPM_DTYPE_BF16) fn = dpm_launch_multi_bf16_bf16;
    if (same && fn) {
      int r0 = 0;
      for (; r0 < n_req; r0 += MULTI_MAX) {
        const int cnt = std::min(MULTI_MAX, n_req - r0);
        const int rc = fn(st, bs + r0, cnt, stream, ev_start ? ev_start[r0] : nullptr, ev_stop ? ev_stop[r0] : nullptr);
        if (rc == MULTI_NOT_BUILT) break;  // no fused variant for this stage, or a buffer of this group is unaligned
        if (rc) return rc;
        if (fused_first)
          for (int r = r0; r < r0 + cnt; ++r) fused_first[r] = r0;
        done = r0 + cnt;
      }
    }
  }
  for (int r = done; r < n_req; ++r) {
    const int rc = dpm_stage_launch_ev(st, &bs[r], stream, ev_start ? ev_start[r] : nullptr, ev_stop ? ev_stop[r] : nullptr);
    if (rc) return rc;
    if (fused_first) fused_first[r] = r;
  }
  return DPM_OK;
}

extern "C" int dpm_stage_launch_multi(const dpm_stage* st, const dpm_buffers* bs, int n_req, void* stream) {
  return dpm_stage_launch_multi_ev(st, bs, n_req, stream, nullptr, nullptr, nullptr);
}

int dpm_timing_begin(int n, void*** starts, void*** stops) {
  void** a = new (std::nothrow) void*[2 * (size_t)n]();
  if (!a) return dpm_set_error(DPM_ERR_NOMEM, "out of memory");
  for (int i = 0; i < 2 * n; ++i) {
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(static_cast<hipEvent_t>(a[j]));
      delete[] a;
      return dpm_set_error((int)rc, "hipEventCreate: %s", hipGetErrorString(rc));
    }
    a[i] = e;
  }
  *starts = a;
  *stops = a + n;
  return DPM_OK;
}

// `recorded` (optional, n flags): event pairs that were never handed to a launch (members of a fused group other than its
// first) are skipped, their ms stays untouched
int dpm_timing_end(int n, void** starts, void** stops, void* stream, float* ms, const unsigned char* recorded) {
  int ret = DPM_OK;
  hipError_t rc = hipStreamSynchronize(static_cast<hipStream_t>(stream));
  if (rc != hipSuccess) ret = dpm_set_error((int)rc, "hipStreamSynchronize: %s", hipGetErrorString(rc));
  for (int i = 0; i < n; ++i) {
    if (ms && ret == DPM_OK && (!recorded || recorded[i])) {
      rc = hipEventElapsedTime(&ms[i], static_cast<hipEvent_t>(starts[i]), static_cast<hipEvent_t>(stops[i]));
      if (rc != hipSuccess) ret = dpm_set_error((int)rc, "hipEventElapsedTime(stage %d): %s", i, hipGetErrorString(rc));
    }
  }
  for (int i = 0; i < 2 * n; ++i) (void)hipEventDestroy(static_cast<hipEvent_t>(starts[i]));
  delete[] starts;
  return ret;
}

extern "C" size_t dpm_threshold_workspace_bytes(int64_t batch, int64_t per_sample) {
  if (batch < 1 || per_sample < 1) return 0;
  const DeviceInfo& di = device_info();
  // 0 when one workgroup per sample is the plan (the sample lives in that workgroup's LDS); else the zeroed
  // per-sample histograms and barrier counters the clusters merge through
  return (size_t)thr_ws_bytes(batch, per_sample, di.n_cu > 0 ? di.n_cu : 256);
}

extern "C" int dpm_add_noise_launch_f64(const dpm_schedule* s, const double* t_host, int nt, const void* x, const void* noise,
                                        void* out, int64_t n, void* stream) {
  if (!s || !t_host || !x || !noise || !out || nt < 1 || n < 0) return dpm_set_error(DPM_ERR_ARG, "add_noise: bad arguments");
  if (n == 0) return DPM_OK;
  for (int j = 0; j < nt; ++j) {  // the schedule in double at the double time (fp32 tables are promoted exactly, ref :127-134)
    double a64 = 0., s64 = 0.;
    dpm_schedule_eval_f64(s, DPM_EVAL_ALPHA, &t_host[j], 1, &a64);
    dpm_schedule_eval_f64(s, DPM_EVAL_STD, &t_host[j], 1, &s64);
    const int rc = dpm_add_noise_f64(a64, s64, x, static_cast<const double*>(noise) + (int64_t)j * n,
                                     static_cast<double*>(out) + (int64_t)j * n, n, stream);
    if (rc) return rc;
  }
  return DPM_OK;
}

extern "C" int dpm_add_noise_launch(const dpm_schedule* s, const float* t_host, int nt, const void* x, const void* noise,
                                    void* out, int64_t n, int dtype, void* stream) {
  if (!s || !t_host || !x || !noise || !out || nt < 1 || n < 0) return dpm_set_error(DPM_ERR_ARG, "add_noise: bad arguments");
  if (n == 0) return DPM_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == DPM_DTYPE_F64) {  // double state: alpha / sigma in double on a dtype=float64 schedule, else fp32 values promoted
    for (int j = 0; j < nt; ++j) {
      double a64 = 0., s64 = 0.;
      if (dpm_schedule_table_is_f64(s)) {
        const double tj = (double)t_host[j];
        dpm_schedule_eval_f64(s, DPM_EVAL_ALPHA, &tj, 1, &a64);
        dpm_schedule_eval_f64(s, DPM_EVAL_STD, &tj, 1, &s64);
      } else {
        float a = 0.f, sg = 0.f;
        dpm_schedule_eval(s, DPM_EVAL_ALPHA, &t_host[j], 1, &a);
        dpm_schedule_eval(s, DPM_EVAL_STD, &t_host[j], 1, &sg);
        a64 = a;
        s64 = sg;
      }
      const int rc = dpm_add_noise_f64(a64, s64, x, static_cast<const double*>(noise) + (int64_t)j * n,
                                       static_cast<double*>(out) + (int64_t)j * n, n, stream);
      if (rc) return rc;
    }
    return DPM_OK;
  }
  const DeviceInfo& di = device_info();
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * 16;
  if (blocks > cap) blocks = cap;
  const bool v2 = n % EPT == 0;
  for (int j = 0; j < nt; ++j) {
    float a = 0.f, sg = 0.f;
    dpm_schedule_eval(s, DPM_EVAL_ALPHA, &t_host[j], 1, &a);
    dpm_schedule_eval(s, DPM_EVAL_STD, &t_host[j], 1, &sg);
    const int64_t off = (int64_t)j * n;
#define DPM_AN(T)                                                                                                       \
  do {                                                                                                                  \
    const T* xp = (const T*)x;                                                                                          \
    const T* np_ = (const T*)noise + off;                                                                               \
    T* op = (T*)out + off;                                                                                              \
    const size_t al = sizeof(T) * EPT;                                                                                  \
    if (v2 && aligned(xp, al) && aligned(np_, al) && aligned(op, al)) {                                                 \
      int64_t bl = (n / EPT + 255) / 256;                                                                               \
      if (bl > cap) bl = cap;                                                                                           \
      hipLaunchKernelGGL((add_noise_kernel<T, true>), dim3((unsigned)bl), dim3(256), 0, st, xp, np_, op, n, a, sg);     \
    } else {                                                                                                            \
      hipLaunchKernelGGL((add_noise_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, xp, np_, op, n, a, sg); \
    }                                                                                                                   \
  } while (0)
    switch (dtype) {
      case DPM_DTYPE_F32: DPM_AN(float); break;
      case DPM_DTYPE_F16: DPM_AN(__half); break;
      case DPM_DTYPE_BF16: DPM_AN(bf16_t); break;
      default: return dpm_set_error(DPM_ERR_UNSUPPORTED, "add_noise: unsupported dtype %d", dtype);
    }
#undef DPM_AN
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "add_noise launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

extern "C" int dpm_blend_launch(const void* x, const void* mask, const void* a, const void* b, float alpha, float sigma,
                                void* out, int64_t n, int64_t mask_period, int dtype, void* stream) {
  if (!x || !mask || !a || !out || n < 0 || mask_period < 1) return dpm_set_error(DPM_ERR_ARG, "blend: bad arguments");
  if (n == 0) return DPM_OK;
  if (dtype == DPM_DTYPE_F64) return dpm_blend_f64(x, mask, a, b, (double)alpha, (double)sigma, out, n, mask_period, stream);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const DeviceInfo& di = device_info();
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * 16;
  if (blocks > cap) blocks = cap;
  KExt ext;
  std::memset(&ext, 0, sizeof ext);
  ext.mask_period = mask_period;
  ext.blend_alpha = alpha;
  ext.blend_sigma = sigma;
  switch (dtype) {
    case DPM_DTYPE_F32:
      hipLaunchKernelGGL(blend_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (const float*)mask,
                         (const float*)a, (const float*)b, (float*)out, n, ext);
      break;
    case DPM_DTYPE_F16:
      hipLaunchKernelGGL(blend_kernel<__half>, dim3((unsigned)blocks), dim3(256), 0, st, (const __half*)x,
                         (const __half*)mask, (const __half*)a, (const __half*)b, (__half*)out, n, ext);
      break;
    case DPM_DTYPE_BF16:
      hipLaunchKernelGGL(blend_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x,
                         (const bf16_t*)mask, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n, ext);
      break;
    default: return dpm_set_error(DPM_ERR_UNSUPPORTED, "blend: unsupported dtype %d", dtype);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "blend launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

extern "C" int dpm_adaptive_error_launch(const void* x_lower, const void* x_higher, const void* x_prev, float atol,
                                         float rtol, float* e_out, int64_t batch, int64_t per_sample, int dtype,
                                         void* stream) {
  if (!x_lower || !x_higher || !x_prev || !e_out || batch < 1 || per_sample < 1)
    return dpm_set_error(DPM_ERR_ARG, "adaptive_error: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t me = hipMemsetAsync(e_out + batch, 0, sizeof(float), st);  // the batch-maximum slot
  if (me != hipSuccess) return dpm_set_error((int)me, "hipMemsetAsync: %s", hipGetErrorString(me));
  switch (dtype) {
    case DPM_DTYPE_F32:
      hipLaunchKernelGGL(adaptive_error_kernel<float>, dim3((unsigned)batch), dim3(1024), 0, st, (const float*)x_lower,
                         (const float*)x_higher, (const float*)x_prev, atol, rtol, e_out, per_sample);
      break;
    case DPM_DTYPE_F16:
      hipLaunchKernelGGL(adaptive_error_kernel<__half>, dim3((unsigned)batch), dim3(1024), 0, st, (const __half*)x_lower,
                         (const __half*)x_higher, (const __half*)x_prev, atol, rtol, e_out, per_sample);
      break;
    case DPM_DTYPE_BF16:
      hipLaunchKernelGGL(adaptive_error_kernel<bf16_t>, dim3((unsigned)batch), dim3(1024), 0, st, (const bf16_t*)x_lower,
                         (const bf16_t*)x_higher, (const bf16_t*)x_prev, atol, rtol, e_out, per_sample);
      break;
    default: return dpm_set_error(DPM_ERR_UNSUPPORTED, "adaptive_error: unsupported dtype %d", dtype);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "adaptive_error launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

// ------------------------------------------------------------------------------------------------
// adaptive solver, controller on the device (include/dpm_hip.h: dpm_adaptive_*; ref :956-1010)
// ------------------------------------------------------------------------------------------------
dpmc::SchedView dpm_schedule_view(const dpm_schedule* s);  // dpm_host.cpp
int dpm_stage_launch_dyn(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop,
                         const dpm_stage* dyn, const int32_t* skip);

namespace {
struct AdaptiveDev {  // device-resident controller state
  float s, lambda_s, lambda_0, h, t_0, t_err, theta, t;
  int32_t order, nfe, done, iters, accept, accepted, pad0, pad1;
  dpm_stage st[5];  // 0, 1: the lower-order update's stages; 2, 3, 4: the higher-order update's
};
struct AdaptiveStatic {
  int32_t algo, solver, model_type, guidance;
  double scale;
};

// the stage records of one iteration s -> t (order 2: DPM-Solver-12, order 3: DPM-Solver-23, ref :976-993), in three
// independent pieces so that the device can run them on different wavefronts: the lower-order update's stages, the
// higher-order update's, and the prologue scalars (alpha, sigma, model time, guidance) of one stage
template <class S>
DPM_HD void adaptive_plan_lower(const S* sv, const AdaptiveStatic& c, int order, float s, float t, dpm_stage* st) {
  if (order == 2) {
    dpmc::singlestep_fill(sv, c.algo, c.solver, 1, s, t, 0., 0., 0, &st[0]);
    st[1] = st[0];
  } else {
    dpmc::singlestep_fill(sv, c.algo, c.solver, 2, s, t, 1. / 3., 0., 0, &st[0]);
  }
}
template <class S>
DPM_HD void adaptive_plan_higher(const S* sv, const AdaptiveStatic& c, int order, float s, float t, dpm_stage* st) {
  if (order == 2) {
    dpmc::singlestep_fill(sv, c.algo, c.solver, 2, s, t, 0.5, 0., 0, &st[2]);
    st[4] = st[3];
  } else {
    dpmc::singlestep_fill(sv, c.algo, c.solver, 3, s, t, 1. / 3., 2. / 3., 0, &st[2]);
  }
}
template <class S>
DPM_HD void adaptive_plan_prologue(const S* sv, const AdaptiveStatic& c, dpm_stage* st) {
  dpmc::set_prologue(sv, st->t_eval, c.model_type, c.guidance, c.scale, st);
}
template <class S>
DPM_HD void adaptive_plan(const S* sv, const AdaptiveStatic& c, int order, float s, float t, dpm_stage* st) {
  adaptive_plan_lower(sv, c, order, s, t, st);
  adaptive_plan_higher(sv, c, order, s, t, st);
  for (int i = 0; i < 5; ++i) adaptive_plan_prologue(sv, c, &st[i]);
}

__global__ void adaptive_reset_kernel(AdaptiveDev* S, float s, float lambda_s, float lambda_0, float h, float t_0,
                                      float t_err, float theta, int order, int32_t* status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  S->s = s;
  S->lambda_s = lambda_s;
  S->lambda_0 = lambda_0;
  S->h = h;
  S->t_0 = t_0;
  S->t_err = t_err;
  S->theta = theta;
  S->t = s;
  S->order = order;
  S->nfe = 0;
  S->iters = 0;
  S->accept = 0;
  S->accepted = 0;
  S->done = !(fabsf(s - t_0) > t_err);  // the loop condition of ref :995 before the first iteration
  for (int i = 0; i < 4; ++i) __hip_atomic_store(&status[i], i == 0 ? S->done : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// decision on the previous iteration (one thread), plan of the next one -- double-precision exp / log / expm1 chains,
// spread over the block's four wavefronts: lower-order stages on one, higher-order stages on another, then the five
// prologues round-robin --, then the time vectors of the network calls (all threads)
__global__ __launch_bounds__(256) void adaptive_begin_kernel(AdaptiveDev* S, dpmc::SchedView sv, AdaptiveStatic c,
                                                             float* e_dev, float* tvec, int64_t tv_len, int32_t* status) {
  __shared__ float te[3], ti[3];
  __shared__ int live;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) {
    if (S->iters > 0 && !S->done) {
      const float E = *e_dev;
      const int acc = E <= 1.f;  // ref :1002
      if (acc) {
        S->s = S->t;
        S->lambda_s = sv.lambda(S->s);
      }
      // ref :1007: h = min(theta * h * E^(-1/order), lambda_0 - lambda_s); the power in double like the reference's float()
      const float pw = (float)pow((double)E, -1.0 / (double)S->order);
      const float hn = (S->theta * S->h) * pw;
      const float room = S->lambda_0 - S->lambda_s;
      S->h = room < hn ? room : hn;
      S->nfe += S->order;
      S->accept = acc;
      S->accepted += acc;
      if (!(fabsf(S->s - S->t_0) > S->t_err)) S->done = 1;  // ref :995
    } else {
      S->accept = 0;
    }
    *e_dev = 0.f;
    live = !S->done;
    if (live) S->t = sv.inv_lambda(S->lambda_s + S->h);  // ref :996
  }
  __syncthreads();
  if (live && lane == 0) {
    if (wave == 0) adaptive_plan_lower(&sv, c, S->order, S->s, S->t, S->st);
    if (wave == 1) adaptive_plan_higher(&sv, c, S->order, S->s, S->t, S->st);
  }
  __syncthreads();
  if (live && lane == 0) {
    adaptive_plan_prologue(&sv, c, &S->st[wave]);
    if (wave == 0) adaptive_plan_prologue(&sv, c, &S->st[4]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (live)
      for (int j = 0; j < 3; ++j) {
        te[j] = S->st[2 + j].t_eval;
        ti[j] = S->st[2 + j].t_input;
      }
    // the verdict as of THIS begin, by its index: a host that looks at begin #j (after waiting for it) reads the same
    // value on every rank of a sharded run, however far its device has run ahead
    __hip_atomic_store(&status[8 + (S->iters & 31)], S->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    S->iters += 1;
    __hip_atomic_store(&status[1], S->nfe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&status[2], S->iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&status[3], S->accepted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&status[0], S->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  if (!live) return;
  for (int j = 0; j < 3; ++j)
    for (int64_t i = threadIdx.x; i < tv_len; i += blockDim.x) {
      tvec[(int64_t)(2 * j) * tv_len + i] = te[j];
      tvec[(int64_t)(2 * j + 1) * tv_len + i] = ti[j];
    }
}

// an accepted step becomes the state: x <- x_higher, x_prev <- x_lower (ref :1003-1005)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void adaptive_commit_kernel(const AdaptiveDev* S, T* __restrict__ x, T* __restrict__ xp,
                                                              const T* __restrict__ xl, const T* __restrict__ xh, int64_t n) {
  if (!S->accept) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC) {
    const int64_t nv = n * (int64_t)sizeof(T) / 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
      reinterpret_cast<u32x4*>(x)[i] = ld16<true>(reinterpret_cast<const u32x4*>(xh) + i);
      reinterpret_cast<u32x4*>(xp)[i] = ld16<true>(reinterpret_cast<const u32x4*>(xl) + i);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      x[i] = xh[i];
      xp[i] = xl[i];
    }
  }
}

// error norm with G workgroups per sample: partial sums of squares in double, the last workgroup of a sample adds
// them in a fixed order (deterministic), E_b = sqrt(mean) and the batch maximum by atomicMax on the bit pattern
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void adaptive_error_kernel2(const T* __restrict__ xl, const T* __restrict__ xh,
                                                              const T* __restrict__ xp, float atol, float rtol,
                                                              int64_t per_sample, int G, double* __restrict__ partial,
                                                              uint32_t* __restrict__ counters, float* __restrict__ e_max,
                                                              const int32_t* skip) {
  if (skip && *skip) return;
  __shared__ double part[4];
  __shared__ int last;
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int64_t chunk = ((per_sample + G - 1) / G + 7) / 8 * 8;
  const int64_t lo = (int64_t)g * chunk, hi = lo + chunk < per_sample ? lo + chunk : per_sample;
  const int64_t base = (int64_t)b * per_sample;
  double acc = 0.;
  auto term = [&](float l, float h, float pv) {
    const float delta = fmaxf(atol, rtol * fmaxf(fabsf(l), fabsf(pv)));
    const float v = (h - l) / delta;
    acc += (double)(v * v);
  };
  if (VEC) {
    for (int64_t i = lo + (int64_t)threadIdx.x * EPT; i < hi; i += (int64_t)blockDim.x * EPT) {
      float l[EPT], h[EPT], pv[EPT];
      load_pack<true>(xl, (base + i) / EPT, l);
      load_pack<true>(xh, (base + i) / EPT, h);
      load_pack<true>(xp, (base + i) / EPT, pv);
#pragma unroll
      for (int j = 0; j < EPT; ++j) term(l[j], h[j], pv[j]);
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
      term(to_f32(xl[base + i]), to_f32(xh[base + i]), to_f32(xp[base + i]));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
    __threadfence();
    last = atomicAdd(&counters[b], 1u) == (uint32_t)(G - 1);
    if (last) {
      __threadfence();
      double t = 0.;
      for (int q = 0; q < G; ++q) t += __hip_atomic_load(&partial[(int64_t)b * G + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float e = sqrtf((float)(t / (double)per_sample));
      atomicMax(reinterpret_cast<unsigned int*>(e_max), __float_as_uint(e));  // E >= 0: bit patterns order like values
      counters[b] = 0u;  // ready for the next launch
    }
  }
}
}  // namespace

struct dpm_adaptive {
  dpm_adaptive_desc d;
  AdaptiveStatic c;
  dpmc::SchedView sv;        // table pointers into `tables` (device)
  float* tables = nullptr;   // 4 x K floats
  AdaptiveDev* dev = nullptr;
  int32_t* status = nullptr; // host-mapped: done, nfe, iterations, accepted
  double* partial = nullptr; // error norm scratch, grown on demand
  uint32_t* counters = nullptr;
  int64_t scratch_blocks = 0, scratch_batch = 0;
  dpm_stage tmpl[5];
  float s0, lambda_s0, lambda_0;
};

extern "C" void dpm_adaptive_destroy(dpm_adaptive* a) {
  if (!a) return;
  if (a->tables) (void)hipFree(a->tables);
  if (a->dev) (void)hipFree(a->dev);
  if (a->status) (void)hipHostFree(a->status);
  if (a->partial) (void)hipFree(a->partial);
  if (a->counters) (void)hipFree(a->counters);
  delete a;
}

extern "C" int dpm_adaptive_create(const dpm_schedule* s, const dpm_adaptive_desc* d, dpm_adaptive** out) {
  if (!s || !d || !out) return dpm_set_error(DPM_ERR_ARG, "adaptive_create: null pointer");
  if (d->order != 2 && d->order != 3)
    return dpm_set_error(DPM_ERR_ARG, "For adaptive step size solver, order must be 2 or 3, got %d", d->order);
  if (d->algorithm_type < 0 || d->algorithm_type > 1 || d->solver_type < 0 || d->solver_type > 1 || d->model_type < 0 ||
      d->model_type > 3 || d->guidance < 0 || d->guidance > 2)
    return dpm_set_error(DPM_ERR_ARG, "adaptive_create: enumeration out of range");
  dpm_adaptive* a = new (std::nothrow) dpm_adaptive;
  if (!a) return dpm_set_error(DPM_ERR_NOMEM, "out of memory");
  a->d = *d;
  a->c = AdaptiveStatic{d->algorithm_type, d->solver_type, d->model_type, d->guidance, d->guidance_scale};
  const dpmc::SchedView hv = dpm_schedule_view(s);
  a->sv = hv;
  hipError_t e = hipSuccess;
  if (hv.discrete) {
    const size_t K = (size_t)hv.total_N;
    e = hipMalloc(&a->tables, 4 * K * sizeof(float));
    const float* src[4] = {hv.la, hv.t, hv.la_rev, hv.t_rev};
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
      e = hipMemcpy(a->tables + i * K, src[i], K * sizeof(float), hipMemcpyHostToDevice);
    a->sv.la = a->tables;
    a->sv.t = a->tables + K;
    a->sv.la_rev = a->tables + 2 * K;
    a->sv.t_rev = a->tables + 3 * K;
  } else {
    a->sv.la = a->sv.t = a->sv.la_rev = a->sv.t_rev = nullptr;
  }
  if (e == hipSuccess) e = hipMalloc(&a->dev, sizeof(AdaptiveDev));
  if (e == hipSuccess) e = hipMemset(a->dev, 0, sizeof(AdaptiveDev));
  void* st = nullptr;
  if (e == hipSuccess) e = hipHostMalloc(&st, 256, hipHostMallocMapped);
  if (e != hipSuccess) {
    dpm_adaptive_destroy(a);
    return dpm_set_error((int)e, "adaptive_create: %s", hipGetErrorString(e));
  }
  a->status = static_cast<int32_t*>(st);
  std::memset(a->status, 0, 256);
  // host copy of the plan for the static fields (form, flags, slots) and the initial scalars
  a->s0 = (float)d->t_start;
  a->lambda_s0 = hv.lambda(a->s0);
  a->lambda_0 = hv.lambda((float)d->t_end);
  const float t1 = hv.inv_lambda(a->lambda_s0 + (float)d->h_init);
  adaptive_plan(&hv, a->c, d->order, a->s0, t1, a->tmpl);
  *out = a;
  return DPM_OK;
}

extern "C" int dpm_adaptive_stage_template(const dpm_adaptive* a, int which, dpm_stage* out) {
  if (!a || !out || which < 0 || which > 4) return dpm_set_error(DPM_ERR_ARG, "adaptive_stage_template: bad arguments");
  *out = a->tmpl[which];
  return DPM_OK;
}

extern "C" int dpm_adaptive_reset(dpm_adaptive* a, void* stream) {
  if (!a) return dpm_set_error(DPM_ERR_ARG, "adaptive_reset: null handle");
  hipLaunchKernelGGL(adaptive_reset_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a->dev, a->s0,
                     a->lambda_s0, a->lambda_0, (float)a->d.h_init, (float)a->d.t_end, (float)a->d.t_err, (float)a->d.theta,
                     a->d.order, a->status);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "adaptive_reset: %s", hipGetErrorString(e));
  return DPM_OK;
}

extern "C" int dpm_adaptive_begin(dpm_adaptive* a, void* x, void* x_prev, const void* x_lower, const void* x_higher, int64_t n,
                                  int dtype, float* e_dev, float* t_vectors, int64_t tv_len, void* stream) {
  // n == 0: an empty shard of a batch-sharded run -- it still runs the controller (every rank must take the same
  // decisions and issue the same collectives) but has no state to commit, and its tensors have no storage
  if (!a || !e_dev || !t_vectors || n < 0 || tv_len < 1 || (n > 0 && (!x || !x_prev || !x_lower || !x_higher)))
    return dpm_set_error(DPM_ERR_ARG, "adaptive_begin: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adaptive_begin_kernel, dim3(1), dim3(256), 0, st, a->dev, a->sv, a->c, e_dev, t_vectors, tv_len, a->status);
  if (n > 0) {
    const DeviceInfo& di = device_info();
    const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * 8;
#define DPM_COMMIT(T)                                                                                                     \
  do {                                                                                                                    \
    const bool v = (n * (int64_t)sizeof(T)) % 16 == 0 && aligned(x, 16) && aligned(x_prev, 16) && aligned(x_lower, 16) && \
                   aligned(x_higher, 16);                                                                                 \
    int64_t bl = ((v ? n * (int64_t)sizeof(T) / 16 : n) + 255) / 256;                                                     \
    if (bl > cap) bl = cap;                                                                                               \
    if (v)                                                                                                                \
      hipLaunchKernelGGL((adaptive_commit_kernel<T, true>), dim3((unsigned)bl), dim3(256), 0, st, a->dev, (T*)x,          \
                         (T*)x_prev, (const T*)x_lower, (const T*)x_higher, n);                                           \
    else                                                                                                                  \
      hipLaunchKernelGGL((adaptive_commit_kernel<T, false>), dim3((unsigned)bl), dim3(256), 0, st, a->dev, (T*)x,         \
                         (T*)x_prev, (const T*)x_lower, (const T*)x_higher, n);                                           \
  } while (0)
    switch (dtype) {
      case DPM_DTYPE_F32: DPM_COMMIT(float); break;
      case DPM_DTYPE_F16: DPM_COMMIT(__half); break;
      case DPM_DTYPE_BF16: DPM_COMMIT(bf16_t); break;
      default: return dpm_set_error(DPM_ERR_UNSUPPORTED, "adaptive_begin: unsupported dtype %d", dtype);
    }
#undef DPM_COMMIT
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "adaptive_begin: %s", hipGetErrorString(e));
  return DPM_OK;
}

extern "C" int dpm_adaptive_stage_launch(dpm_adaptive* a, int which, const dpm_stage* st, const dpm_buffers* b, void* stream) {
  if (!a || !st || !b || which < 0 || which > 4) return dpm_set_error(DPM_ERR_ARG, "adaptive_stage_launch: bad arguments");
  return dpm_stage_launch_dyn(st, b, stream, nullptr, nullptr, &a->dev->st[which], &a->dev->done);
}

extern "C" int dpm_adaptive_error(dpm_adaptive* a, const void* x_lower, const void* x_higher, const void* x_prev,
                                  int64_t batch, int64_t per_sample, int dtype, float* e_dev, void* stream) {
  if (!a || !e_dev || batch < 0) return dpm_set_error(DPM_ERR_ARG, "adaptive_error: bad arguments");
  if (batch == 0) return DPM_OK;  // an empty shard contributes nothing to the maximum (begin left *e_dev = 0)
  if (!x_lower || !x_higher || !x_prev || per_sample < 1) return dpm_set_error(DPM_ERR_ARG, "adaptive_error: bad arguments");
  const DeviceInfo& di = device_info();
  const int n_cu = di.n_cu > 0 ? di.n_cu : 256;
  // enough workgroups to fill the chip twice, at least 8192 elements each
  int64_t G = std::max<int64_t>(1, std::min<int64_t>((2 * (int64_t)n_cu + batch - 1) / batch, per_sample / 8192));
  if (G > 64) G = 64;
  const int64_t blocks = batch * G;
  if (blocks > a->scratch_blocks || batch > a->scratch_batch) {  // grow the scratch (outside any capture: create-time sizes)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs);
    if (cs != hipStreamCaptureStatusNone)
      return dpm_set_error(DPM_ERR_UNSUPPORTED, "adaptive_error: run one iteration outside the capture first (scratch allocation)");
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (a->partial) (void)hipFree(a->partial);
    if (a->counters) (void)hipFree(a->counters);
    a->partial = nullptr;
    a->counters = nullptr;
    hipError_t e = hipMalloc(&a->partial, (size_t)blocks * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&a->counters, (size_t)batch * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(a->counters, 0, (size_t)batch * sizeof(uint32_t));
    if (e != hipSuccess) return dpm_set_error((int)e, "adaptive_error: %s", hipGetErrorString(e));
    a->scratch_blocks = blocks;
    a->scratch_batch = batch;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
#define DPM_ERRK(T)                                                                                                     \
  do {                                                                                                                  \
    const size_t al = sizeof(T) * EPT;                                                                                  \
    const bool v = per_sample % EPT == 0 && aligned(x_lower, al) && aligned(x_higher, al) && aligned(x_prev, al);       \
    if (v)                                                                                                              \
      hipLaunchKernelGGL((adaptive_error_kernel2<T, true>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x_lower, \
                         (const T*)x_higher, (const T*)x_prev, (float)a->d.atol, (float)a->d.rtol, per_sample, (int)G,   \
                         a->partial, a->counters, e_dev, &a->dev->done);                                                \
    else                                                                                                                \
      hipLaunchKernelGGL((adaptive_error_kernel2<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x_lower, \
                         (const T*)x_higher, (const T*)x_prev, (float)a->d.atol, (float)a->d.rtol, per_sample, (int)G,   \
                         a->partial, a->counters, e_dev, &a->dev->done);                                                \
  } while (0)
  switch (dtype) {
    case DPM_DTYPE_F32: DPM_ERRK(float); break;
    case DPM_DTYPE_F16: DPM_ERRK(__half); break;
    case DPM_DTYPE_BF16: DPM_ERRK(bf16_t); break;
    default: return dpm_set_error(DPM_ERR_UNSUPPORTED, "adaptive_error: unsupported dtype %d", dtype);
  }
#undef DPM_ERRK
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "adaptive_error: %s", hipGetErrorString(e));
  return DPM_OK;
}

extern "C" int dpm_adaptive_done_at(const dpm_adaptive* a, int begin_index) {
  if (!a || begin_index < 0) return -1;
  const volatile int32_t* s = a->status;
  return s[8 + (begin_index & 31)];
}

extern "C" int dpm_adaptive_poll(const dpm_adaptive* a, int* done, int* nfe, int* iterations, int* accepted) {
  if (!a) return dpm_set_error(DPM_ERR_ARG, "adaptive_poll: null handle");
  const volatile int32_t* s = a->status;
  if (done) *done = s[0];
  if (nfe) *nfe = s[1];
  if (iterations) *iterations = s[2];
  if (accepted) *accepted = s[3];
  return DPM_OK;
}

// ------------------------------------------------------------------------------------------------
// hipGraph capture of a trajectory (dpm_plan_run under stream capture)
// ------------------------------------------------------------------------------------------------
struct dpm_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int result = -1;
  int nodes = 0;
};

extern "C" int dpm_graph_create(const dpm_plan* p, const dpm_run_buffers* rb, dpm_model_cb model, void* user, void* stream,
                                dpm_graph** out) {
  if (!p || !rb || !out) return dpm_set_error(DPM_ERR_ARG, "graph_create: null pointer");
  if (!stream) return dpm_set_error(DPM_ERR_ARG, "graph_create: capture needs a non-null stream");
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)device_info();  // query device properties before the capture starts
  dpm_graph* g = new (std::nothrow) dpm_graph;
  if (!g) return dpm_set_error(DPM_ERR_NOMEM, "out of memory");
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) {
    delete g;
    return dpm_set_error((int)e, "hipStreamBeginCapture: %s", hipGetErrorString(e));
  }
  const int rc = dpm_plan_run(p, rb, model, user, stream, &g->result);
  e = hipStreamEndCapture(st, &g->graph);  // always end the capture, also after a failed launch
  if (rc) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return rc;
  }
  if (e != hipSuccess) {
    delete g;
    return dpm_set_error((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
  }
  size_t nn = 0;
  if (hipGraphGetNodes(g->graph, nullptr, &nn) == hipSuccess) g->nodes = (int)nn;
  e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g->graph);
    delete g;
    return dpm_set_error((int)e, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  *out = g;
  return DPM_OK;
}

extern "C" int dpm_graph_launch(dpm_graph* g, void* stream) {
  if (!g || !g->exec) return dpm_set_error(DPM_ERR_ARG, "graph_launch: null graph");
  hipError_t e = hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return dpm_set_error((int)e, "hipGraphLaunch: %s", hipGetErrorString(e));
  return DPM_OK;
}

extern "C" int dpm_graph_result(const dpm_graph* g) { return g ? g->result : -1; }
extern "C" int dpm_graph_num_nodes(const dpm_graph* g) { return g ? g->nodes : 0; }

extern "C" void dpm_graph_destroy(dpm_graph* g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
}

extern "C" int dpm_cluster_timeout_poll(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  uint32_t* w = cluster_fault_word(dev, false);
  if (!w || !*w) return 0;
  *w = 0u;
  return 1;
}

extern "C" int dpm_device_info(int* n_cu, int* lds_bytes, char* arch, int arch_len) {
  const DeviceInfo& di = device_info();
  if (!di.ok) return dpm_set_error((int)hipErrorNoDevice, "no HIP device available");
  if (n_cu) *n_cu = di.n_cu;
  if (lds_bytes) *lds_bytes = di.lds;
  if (arch && arch_len > 0) {
    std::strncpy(arch, di.arch, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return DPM_OK;
}
