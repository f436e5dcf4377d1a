// dpm_stage_f32_bf16_b.hip -- stage kernels for state dtype float, network-output dtype bf16_t (see dpm_device.hpp):
// the LIN1, MS3 and DENOISE update forms (a translation unit of its own for the build time; entered from
// dpm_stage_f32_bf16.hip)
#include "dpm_device.hpp"

int dpm_launch_f32_bf16_b(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop, const dpm_stage* dyn,
          const int32_t* skip, const dpm_buffers* multi, int n_multi) {
  const LaunchCtx s{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop), dyn, skip,
                    multi, n_multi};
  return launch_form<float, bf16_t, FORMS_B>(st, b, s);
}
