// dpm_stage_f32_f16.hip -- stage kernels for state dtype float, network-output dtype __half (see dpm_device.hpp):
// the TWO and SS3T update forms, the fused multi-request launchers and the pair's catch-all kernels; dpm_stage_f32_f16_b.hip holds the other forms
#define DPM_CATCHALL_HOME
#include "dpm_device.hpp"

template const void* dpm_catchall_thresh<float, __half>();
template const void* dpm_catchall_scalar<float, __half, false>();
template const void* dpm_catchall_scalar<float, __half, true>();

int dpm_launch_f32_f16_b(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop, const dpm_stage* dyn,
          const int32_t* skip, const dpm_buffers* multi, int n_multi);

int dpm_launch_f32_f16(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop, const dpm_stage* dyn,
          const int32_t* skip) {
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop), dyn, skip};
  const int rc = launch_form<float, __half, FORMS_A>(st, b, s);
  return rc == FORM_ELSEWHERE ? dpm_launch_f32_f16_b(st, b, stream, ev_start, ev_stop, dyn, skip, nullptr, 0) : rc;
}

int dpm_launch_multi_f32_f16(const dpm_stage* st, const dpm_buffers* bs, int n_req, void* stream, void* ev_start, void* ev_stop) {
  if ((st->flags & DPM_F_THRESH) && !(st->flags & DPM_F_BLEND)) {  // one thresholding launch over all requests' samples
    const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop),
                      nullptr, nullptr, bs, n_req};
    const int rc = launch_form<float, __half, FORMS_A>(st, bs, s);
    return rc == FORM_ELSEWHERE ? dpm_launch_f32_f16_b(st, bs, stream, ev_start, ev_stop, nullptr, nullptr, bs, n_req) : rc;
  }
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)};
  return launch_multi_typed<float, __half>(st, bs, n_req, s);
}
