// dpm_stage_f16_f16.hip -- stage kernels for state dtype __half, network-output dtype __half (see dpm_device.hpp)
#include "dpm_device.hpp"

int dpm_launch_f16_f16(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop, const dpm_stage* dyn,
          const int32_t* skip) {
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop), dyn, skip};
  return launch_form<__half, __half>(st, b, s);
}

int dpm_launch_multi_f16_f16(const dpm_stage* st, const dpm_buffers* bs, int n_req, void* stream, void* ev_start, void* ev_stop) {
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)};
  return launch_multi_typed<__half, __half>(st, bs, n_req, s);
}
