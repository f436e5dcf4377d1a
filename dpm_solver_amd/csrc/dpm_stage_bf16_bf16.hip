// dpm_stage_bf16_bf16.hip -- stage kernels for state dtype bf16_t, network-output dtype bf16_t (see dpm_device.hpp)
#include "dpm_device.hpp"

int dpm_launch_bf16_bf16(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop, const dpm_stage* dyn,
          const int32_t* skip) {
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop), dyn, skip};
  return launch_form<bf16_t, bf16_t>(st, b, s);
}

int dpm_launch_multi_bf16_bf16(const dpm_stage* st, const dpm_buffers* bs, int n_req, void* stream, void* ev_start, void* ev_stop) {
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)};
  return launch_multi_typed<bf16_t, bf16_t>(st, bs, n_req, s);
}
