// dpm_stage_f32_f32.hip -- stage kernels for state dtype float, network-output dtype float (see dpm_device.hpp):
// the TWO and SS3T update forms, the fused multi-request launchers and the pair's catch-all kernels; dpm_stage_f32_f32_b.hip holds the other forms
#define DPM_CATCHALL_HOME
#include "dpm_device.hpp"

template const void* dpm_catchall_thresh<float, float>();
template const void* dpm_catchall_scalar<float, float, false>();
template const void* dpm_catchall_scalar<float, float, true>();

int dpm_launch_f32_f32_b(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop, const dpm_stage* dyn,
          const int32_t* skip, const dpm_buffers* multi, int n_multi);

int dpm_launch_f32_f32(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop, const dpm_stage* dyn,
          const int32_t* skip) {
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop), dyn, skip};
  const int rc = launch_form<float, float, FORMS_A>(st, b, s);
  return rc == FORM_ELSEWHERE ? dpm_launch_f32_f32_b(st, b, stream, ev_start, ev_stop, dyn, skip, nullptr, 0) : rc;
}

int dpm_launch_multi_f32_f32(const dpm_stage* st, const dpm_buffers* bs, int n_req, void* stream, void* ev_start, void* ev_stop) {
  if ((st->flags & DPM_F_THRESH) && !(st->flags & DPM_F_BLEND)) {  // one thresholding launch over all requests' samples
    const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop),
                      nullptr, nullptr, bs, n_req};
    const int rc = launch_form<float, float, FORMS_A>(st, bs, s);
    return rc == FORM_ELSEWHERE ? dpm_launch_f32_f32_b(st, bs, stream, ev_start, ev_stop, nullptr, nullptr, bs, n_req) : rc;
  }
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)};
  return launch_multi_typed<float, float>(st, bs, n_req, s);
}
