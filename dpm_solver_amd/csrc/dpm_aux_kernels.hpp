// dpm_aux_kernels.hpp -- add_noise, the stand-alone mask blend and the adaptive solver's error norm (part of
// dpm_device.hpp; include that)
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// add_noise (ref :1012-1030):  out = alpha*x + sigma*noise
// ------------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void add_noise_kernel(const T* __restrict__ x, const T* __restrict__ nz,
                                                        T* __restrict__ out, int64_t n, float alpha, float sigma) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC) {  // n % 8 == 0, pointers 16/32-byte aligned: one 8-element group per lane and iteration
    for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < n / EPT; gi += stride) {
      float a[EPT], b[EPT], o[EPT];
      load_pack<false>(x, gi, a);
      load_pack<true>(nz, gi, b);
#pragma unroll
      for (int j = 0; j < EPT; ++j) o[j] = alpha * a[j] + sigma * b[j];
      store_pack<false>(out, gi, o);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      out[i] = from_f32<T>(alpha * to_f32(x[i]) + sigma * to_f32(nz[i]));
  }
}

// ------------------------------------------------------------------------------------------------
// stand-alone mask blend (the epilogue of KExt as its own launch: callable use of the corrector, and the
// correction of x_T before the first multistep update, ref :1180)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void blend_kernel(const T* __restrict__ x, const T* __restrict__ mask,
                                                    const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                    int64_t n, KExt ext) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = from_f32<T>(blend_ref(to_f32(x[i]), to_f32(mask[i % ext.mask_period]), to_f32(a[i]), b ? to_f32(b[i]) : 0.f,
                                   b != nullptr, ext));
}

// ------------------------------------------------------------------------------------------------
// adaptive solver error norm (ref :999-1001): one workgroup per sample
//   delta = max(atol, rtol*max(|x_lower|, |x_prev|));  E_b = sqrt(mean(((x_higher - x_lower)/delta)^2))
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void adaptive_error_kernel(const T* __restrict__ xl, const T* __restrict__ xh,
                                                              const T* __restrict__ xp, float atol, float rtol,
                                                              float* __restrict__ e_out, int64_t per_sample) {
  __shared__ double part[16];
  const int64_t base = (int64_t)blockIdx.x * per_sample;
  double acc = 0.;
  for (int64_t i = threadIdx.x; i < per_sample; i += blockDim.x) {
    const float l = to_f32(xl[base + i]), h = to_f32(xh[base + i]), pv = to_f32(xp[base + i]);
    const float delta = fmaxf(atol, rtol * fmaxf(fabsf(l), fabsf(pv)));
    const float v = (h - l) / delta;
    acc += (double)(v * v);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
    const float e = sqrtf((float)(t / (double)per_sample));
    e_out[blockIdx.x] = e;
    // batch maximum (ref :1001) in the extra slot: E >= 0, so the bit patterns order like the values
    atomicMax(reinterpret_cast<unsigned int*>(e_out + gridDim.x), __float_as_uint(e));
  }
}

}  // namespace
