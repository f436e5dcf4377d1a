// dpm_device.hpp -- gfx950 (MI355X, CDNA4) device code of the DPM-Solver engine: element types, the fused stage
// kernels (streaming + dynamic thresholding) and their launch plumbing.  Included by one translation unit per
// (state dtype, eps dtype) pair (dpm_stage_*.hip) so the instantiation matrix compiles in parallel, and by
// dpm_kernels.hip (C ABI entry points, add_noise, adaptive error norm, calibration).
//
// One fused, HBM-streaming kernel per solver stage:
//
//   raw network output(s) --CFG blend / classifier term--> --x_start|v|score -> eps--> --eps -> x0-->
//   --dynamic thresholding--> mn   ;   x_out = exponential-integrator combination of x, mn, h1, h2
//
// Memory-bound (~0.5 flop/B): no MFMA.  What matters is (i) 16-byte coalesced accesses -- each lane moves
// 8 consecutive elements per tensor per iteration, a wavefront 512 contiguous elements, (ii) all loads of
// an iteration issued before the first use, (iii) >= 2048 workgroups of 256 threads so every CU holds 8
// waves per SIMD, (iv) the per-stage scalars arrive as kernel arguments, i.e. in SGPRs via the scalar
// cache, so the vector pipeline only ever sees the five streams.  The arithmetic keeps the reference's
// association and is compiled with -ffp-contract=off: given equal coefficients the result is bit-identical
// to the reference's chain of ATen kernels (no fused multiply-adds there either).
//
// `ref :NNN` = line in the reference's dpm_solver_pytorch.py.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <algorithm>
#include <cstring>
#include <cmath>
#ifdef DPM_THR_TIMING
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif
#include <mutex>
#include <type_traits>
#include <new>

#include "dpm_hip.h"

int dpm_set_error(int code, const char* fmt, ...);  // dpm_host.cpp

namespace dpmk {
struct Tuning {
  int unroll = 0;           // tiles per workgroup iteration: 1, 2;  0 = default
  int nontemporal = -1;     // nt mask; -1 = per-dtype default
  int blocks_per_cu = 8;    // grid cap = CUs x this (8 x 256 threads = every SIMD holds 8 waves)
  int assume_resident = 0;  // treat every launch as dpm_buffers.inputs_resident (benchmarking a frozen loop from Python)
  int multi_fuse = 1;       // dpm_stage_launch_multi: 1 = one fused launch per group of requests, 0 = one launch per request
  int cluster_in_graph = 0; // 1: thresholding keeps workgroup clusters under stream capture also for samples that fit one workgroup
  int cluster_one_hop = 1;  // 0: clusters always take the general route (merged histograms, several barriers)
  int multi_xcd_remap = -1; // fused launch maps workgroup b to tile (b % 8) * span + b / 8 (one contiguous eighth of the
                            // tile space per XCD): 1 on, 0 off, -1 per dtype (on for 2-byte states: +1.4 %; fp32: -3 %)
  int multi_blocks_per_cu = 0;  // grid cap of the fused launch in workgroups per CU; 0 = one super-tile per workgroup
};
extern Tuning g_tuning;     // defined in dpm_kernels.hip
// device-wide chain of clustered thresholding launches (see launch_typed): one instance per device for the library
struct ClusterChain {
  std::mutex mu;
  hipEvent_t ev = nullptr;
  bool recorded = false;
};
ClusterChain& cluster_chain(int dev);  // defined in dpm_kernels.hip
// host-mapped word a clustered kernel raises when one of its waits timed out; nullptr until created (create = false
// never allocates: stream capture)
uint32_t* cluster_fault_word(bool create);
// bfloat16 storage (a named type with external linkage: it is a template argument of functions shared between
// translation units, dpm_catchall_* below)
struct bf16_t {
  uint16_t v;
};
}  // namespace dpmk
using dpmk::Tuning;
using dpmk::g_tuning;
using dpmk::ClusterChain;
using dpmk::cluster_chain;
using dpmk::cluster_fault_word;

// The catch-all kernels of a dtype pair (run-time form / guidance: the one-element-per-lane stage kernel and the general
// thresholding kernel) are compiled in ONE of the pair's two translation units (dpm_stage_<pair>.hip defines
// DPM_CATCHALL_HOME); the sibling unit launches them through their host-side handles.
template <typename TS, typename TE>
const void* dpm_catchall_thresh();
template <typename TS, typename TE, bool DYN>
const void* dpm_catchall_scalar();

namespace {

// ------------------------------------------------------------------------------------------------
// element types
// ------------------------------------------------------------------------------------------------
using dpmk::bf16_t;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__half v) { return __half2float(v); }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float((uint32_t)v.v << 16); }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return bf16_t{(uint16_t)((u >> 16) | 0x40)};  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                                   // round to nearest even
  return bf16_t{(uint16_t)(u >> 16)};
}

constexpr int EPT = 8;  // elements per lane per access group: 2 x 16 B (fp32) or 1 x 16 B (fp16/bf16)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
// Output stores are WRITE-THROUGH (`sc0 sc1`): the line goes to the memory side and is dropped from the XCD's L2 instead
// of lingering there dirty.  Nothing reads a stage's outputs from this L2 again (the next kernel starts with its L2
// invalidated, and may run the element on another XCD), so keeping them only pollutes the cache during the kernel and
// leaves a write-back for the kernel boundary: [256,4,64,64] fp16 2M stage 7.29 -> 6.39 us per launch (70.5 -> 80.4 % of
// HBM peak), fp32 13.22 -> 13.02 (profiles/r01_store_policy.md).  Written as inline assembly: a `volatile` store
// compiles to the same instruction but the compiler follows each one with `s_waitcnt vmcnt(0)`, which serialises the
// stores (fp16 6.65 us, HBM-cold 8.9 instead of 8.4).  The compiler does not know about these stores: the two wait
// states a 16-byte store needs before its data registers may be rewritten (gfx940+) are in the string, and its own
// `vmcnt` bookkeeping stays correct because loads return in order among themselves -- an unknown older or younger
// store can only make one of its waits longer, never shorter.  -DDPM_STORE_WRITE_THROUGH=0 restores plain /
// non-temporal stores (the NT flag) for comparison.
#ifndef DPM_STORE_WRITE_THROUGH
#define DPM_STORE_WRITE_THROUGH 1
#endif
// WT = false: a store instruction that leaves gaps (32-byte lane stride): the halves of a line have to meet in L2 first
template <bool NT, bool WT = true>
__device__ __forceinline__ void st16(u32x4* p, u32x4 v) {
  if (DPM_STORE_WRITE_THROUGH && WT)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}
template <bool NT, typename V2>
__device__ __forceinline__ void st8(V2* p, V2 v) {
  if (DPM_STORE_WRITE_THROUGH)
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}


// 8 consecutive elements of group `group` -> fp32.  Always global_load_dwordx4 (x2 for fp32).
template <bool NT>
__device__ __forceinline__ void load_pack(const float* __restrict__ p, int64_t group, float (&out)[EPT]) {
  const u32x4* q = reinterpret_cast<const u32x4*>(p) + group * 2;
  const u32x4 a = ld16<NT>(q), b = ld16<NT>(q + 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out[j] = __uint_as_float(a[j]);
    out[4 + j] = __uint_as_float(b[j]);
  }
}
template <bool NT>
__device__ __forceinline__ void load_pack(const __half* __restrict__ p, int64_t group, float (&out)[EPT]) {
  const u32x4 a = ld16<NT>(reinterpret_cast<const u32x4*>(p) + group);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out[2 * j] = __half2float(__ushort_as_half((unsigned short)(a[j] & 0xffffu)));
    out[2 * j + 1] = __half2float(__ushort_as_half((unsigned short)(a[j] >> 16)));
  }
}
template <bool NT>
__device__ __forceinline__ void load_pack(const bf16_t* __restrict__ p, int64_t group, float (&out)[EPT]) {
  const u32x4 a = ld16<NT>(reinterpret_cast<const u32x4*>(p) + group);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out[2 * j] = __uint_as_float(a[j] << 16);
    out[2 * j + 1] = __uint_as_float(a[j] & 0xffff0000u);
  }
}

template <bool NT>
__device__ __forceinline__ void store_pack(float* __restrict__ p, int64_t group, const float (&in)[EPT]) {
  u32x4 a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = __float_as_uint(in[j]);
    b[j] = __float_as_uint(in[4 + j]);
  }
  u32x4* q = reinterpret_cast<u32x4*>(p) + group * 2;
  // 32 consecutive bytes per lane, i.e. two instructions that each fill every other 16 bytes (the layout of the
  // extended kernel when its inputs are strided or masked): written through, every half line would travel on its own --
  // guided-diffusion's strided 6-channel stage 48.8 -> 77 us.  Cached stores let L2 merge them.
  st16<NT, false>(q, a);
  st16<NT, false>(q + 1, b);
}
// two fp32 -> one dword of two fp16, round to nearest even (one v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t pack_half2(float lo, float hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 v = __builtin_convertvector(f2{lo, hi}, h2);
  return __builtin_bit_cast(uint32_t, v);
}
// two fp32 -> one dword of two bf16, round to nearest even (gfx950: one v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf162(float lo, float hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 v = __builtin_convertvector(f2{lo, hi}, b2);
  return __builtin_bit_cast(uint32_t, v);
}
template <bool NT>
__device__ __forceinline__ void store_pack(__half* __restrict__ p, int64_t group, const float (&in)[EPT]) {
  u32x4 a;
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = pack_half2(in[2 * j], in[2 * j + 1]);
  st16<NT>(reinterpret_cast<u32x4*>(p) + group, a);
}
template <bool NT>
__device__ __forceinline__ void store_pack(bf16_t* __restrict__ p, int64_t group, const float (&in)[EPT]) {
  u32x4 a;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    a[j] = pack_bf162(in[2 * j], in[2 * j + 1]);
  st16<NT>(reinterpret_cast<u32x4*>(p) + group, a);
}

// Tile-level access for the streaming kernel.  A tile is the 2048 elements of one workgroup iteration (256 lanes x 8).
// `split` (4-byte state, tile complete): lane t takes elements [4t, 4t+4) and [1024+4t, 1024+4t+4) of the tile, so
// each of the two global_load_dwordx4 of a wavefront covers 1 KiB of consecutive addresses; otherwise lane t takes the
// 8 consecutive elements [8t, 8t+8) (one 16-byte access for 2-byte types, two adjacent ones for fp32).  The op is
// elementwise, so any mapping that is the same for every tensor of the launch is correct.  In the split case gi is
// (first group of the tile) + lane; the tile may start at any group of the tensor (strided network outputs).
template <bool NT, typename T>
__device__ __forceinline__ void load_tile(const T* __restrict__ p, int64_t gi, bool split, float (&out)[EPT]) {
  if constexpr (sizeof(T) == 4) {
    if (split) {
      const u32x4* q = reinterpret_cast<const u32x4*>(p) + (2 * gi - (int64_t)threadIdx.x);
      const u32x4 a = ld16<NT>(q), b = ld16<NT>(q + 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        out[j] = __uint_as_float(a[j]);
        out[4 + j] = __uint_as_float(b[j]);
      }
      return;
    }
  } else {
    if (split) {  // 2-byte network output next to a 4-byte state: the same elements as two 8-byte accesses
      typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t* q = reinterpret_cast<const u32x2_t*>(p) + (2 * gi - (int64_t)threadIdx.x);
      const u32x2_t a = NT ? __builtin_nontemporal_load(q) : *q;
      const u32x2_t b = NT ? __builtin_nontemporal_load(q + 256) : *(q + 256);
      const uint32_t w[4] = {a[0], a[1], b[0], b[1]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (std::is_same<T, __half>::value) {
          out[2 * j] = __half2float(__ushort_as_half((unsigned short)(w[j] & 0xffffu)));
          out[2 * j + 1] = __half2float(__ushort_as_half((unsigned short)(w[j] >> 16)));
        } else {
          out[2 * j] = __uint_as_float(w[j] << 16);
          out[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
      }
      return;
    }
  }
  load_pack<NT>(p, gi, out);
}
template <bool NT, typename T>
__device__ __forceinline__ void store_tile(T* __restrict__ p, int64_t gi, bool split, const float (&in)[EPT]) {
  if constexpr (sizeof(T) == 4) {
    if (split) {
      u32x4 a, b;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = __float_as_uint(in[j]);
        b[j] = __float_as_uint(in[4 + j]);
      }
      u32x4* q = reinterpret_cast<u32x4*>(p) + (2 * gi - (int64_t)threadIdx.x);
      st16<NT>(q, a);
      st16<NT>(q + 256, b);
      return;
    }
  }
  store_pack<NT>(p, gi, in);
}

// ------------------------------------------------------------------------------------------------
// per-stage scalars (kernel argument => SGPRs)
// ------------------------------------------------------------------------------------------------
struct KParams {
  // Field order matters to the optimiser, not to the hardware: with alpha_e, sigma_e and cfg_scale adjacent the SLP
  // vectoriser loads them as two OVERLAPPING <2 x float> in the mode-dispatching kernels, SROA then cannot split the
  // argument copy and the backend parks those 12 bytes in LDS (3 KB per workgroup and an LDS round trip per use).
  // Integers in between keep every float a scalar kernarg load.
  float alpha_e;
  uint32_t flags;
  float sigma_e;
  int32_t model_type;
  float cfg_scale;
  int32_t form;      // DPM_FORM_* / DPM_GUIDE_*: read by the run-time dispatched kernels (FORM_RT / GUIDE_RT)
  float cg_scale;
  int32_t guidance;
  float cx, c0, c1, c2;
  float k0, k1, k2, k3, k4;
  float inv_alpha;   // RN(1 / alpha_e), used by the specialised prologue (see div_by_alpha)
  float inv_sigma;   // RN(1 / sigma_e)
  uint32_t fastdiv;  // bit 0 / 1: alpha_e / sigma_e pass div_invariant_ok (general prologue)
};

// x / alpha_e for a wave-uniform divisor whose correctly rounded reciprocal r = RN(1/alpha) is known: q = RN(x*r),
// then one exact-residual correction q' = RN(q + RN(x - q*alpha) * r) (both fused: the residual is exact).  This is
// the correctly rounded quotient -- bit-identical to IEEE division, which the reference uses -- for every finite
// x whose quotient is a normal number, provided alpha's significand is not all ones (Markstein's theorem; the launch
// falls back to the generic prologue with a true division when that guard fails).  3 VALU ops instead of ~12.
// The arithmetic below is written once for V = float and V = f32x2 (two adjacent elements): the streaming kernel works
// on adjacent pairs so that the packed fp32 instructions (v_pk_mul/add/fma_f32) take their operands from the register
// pairs the loads and conversions produce, and v_cvt_pk_f16_f32 packs the two halves of one output dword directly.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f32x2 vfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <typename V>
__device__ __forceinline__ V div_by_alpha(V x, const KParams& p) {
  const V q = x * p.inv_alpha;
  const V e = vfma(-q, (V)(p.alpha_e), x);
  return vfma(e, (V)(p.inv_alpha), q);
}
// the same for the general prologue: divisor d with reciprocal r when the host-side guard passed (`fast`, wave-uniform),
// a true division otherwise -- identical bits either way
template <typename V>
__device__ __forceinline__ V div_uniform(V x, float d, float r, bool fast) {
  if (fast) {
    const V q = x * r;
    const V e = vfma(-q, (V)(d), x);
    return vfma(e, (V)(r), q);
  }
  return x / d;
}

// the guard of div_by_alpha / div_uniform on the device (the host's twin is div_invariant_ok below)
__device__ __forceinline__ bool div_invariant_ok_dev(float d) {
  const uint32_t u = __float_as_uint(d), ex = (u >> 23) & 0xffu;
  return ex > 32u && ex < 222u && (u & 0x7fffffu) != 0x7fffffu;
}
// coefficients computed on the device (LaunchCtx::dyn, kernels instantiated with DYN = true): every float of the stage
// record comes from device memory; flags, form, model type and guidance kind stay the host's.  Built unconditionally from
// loads: a conditional overwrite of the kernel argument keeps part of it addressable and the backend parks it in LDS.
__device__ __forceinline__ KParams params_from_dyn(const KParams& p, const dpm_stage* d) {
  KParams q;
  q.alpha_e = d->alpha_e;
  q.sigma_e = d->sigma_e;
  q.cfg_scale = p.cfg_scale;
  q.cg_scale = d->cg_scale;
  q.cx = d->cx;
  q.c0 = d->c0;
  q.c1 = d->c1;
  q.c2 = d->c2;
  q.k0 = d->k[0];
  q.k1 = d->k[1];
  q.k2 = d->k[2];
  q.k3 = d->k[3];
  q.k4 = d->k[4];
  q.flags = p.flags;
  q.model_type = p.model_type;
  q.inv_alpha = 1.0f / q.alpha_e;
  q.form = p.form;
  q.guidance = p.guidance;
  q.inv_sigma = 1.0f / q.sigma_e;
  q.fastdiv = (div_invariant_ok_dev(q.alpha_e) ? 1u : 0u) | (div_invariant_ok_dev(q.sigma_e) ? 2u : 0u);
  return q;
}

// Compile-time knowledge about the prologue: a prologue mode PM = model_type * 2 + (eps -> x0 ? 1 : 0) fixes the network's
// parameterisation and the conversion at compile time (branch-free inner loop, divisions by the wave-uniform alpha /
// sigma as multiplications by their reciprocal + one exact-residual correction); PM_RT reads everything from the stage
// record at run time (true divisions when a divisor fails the guard).  Kernels are instantiated for the two modes of a
// noise-prediction network (SPEC_NOISE_EPS, SPEC_NOISE_X0: the common case) and as SPEC_GENERIC, which picks the mode
// once per tile iteration (wave-uniform switch) and runs the same straight-line code for x_start / v / score networks.
enum { PM_RT = -1, SPEC_NOISE_EPS = DPM_MODEL_NOISE * 2, SPEC_NOISE_X0 = DPM_MODEL_NOISE * 2 + 1, SPEC_GENERIC = 100 };

template <int SPEC>
__device__ __forceinline__ bool spec_need_xe(const KParams& p) {
  if (SPEC >= 0 && SPEC != SPEC_GENERIC)
    return (SPEC & 1) || (SPEC >> 1) == DPM_MODEL_X_START || (SPEC >> 1) == DPM_MODEL_V;
  return (p.flags & DPM_F_TO_X0) || p.model_type == DPM_MODEL_X_START || p.model_type == DPM_MODEL_V;
}
// the mode SPEC_GENERIC dispatches to: PM_RT when a divisor this stage needs fails the division-by-invariant guard
__device__ __forceinline__ int generic_mode(const KParams& p) {
  const bool tox0 = (p.flags & DPM_F_TO_X0) != 0;
  const bool ok = (!tox0 || (p.fastdiv & 1u)) && (p.model_type != DPM_MODEL_X_START || (p.fastdiv & 2u));
  return ok ? p.model_type * 2 + (tox0 ? 1 : 0) : PM_RT;
}

// raw network output -> noise prediction (noise_pred_fn, ref :288-298)
template <int PM, typename V>
__device__ __forceinline__ V to_noise(V o, V xe, const KParams& p) {
  const int model = PM >= 0 ? (PM >> 1) : p.model_type;
  switch (model) {
    case DPM_MODEL_X_START:
      return div_uniform(xe - p.alpha_e * o, p.sigma_e, p.inv_sigma, PM >= 0 || (p.fastdiv & 2u) != 0u);
    case DPM_MODEL_V: return p.alpha_e * o + p.sigma_e * xe;
    case DPM_MODEL_SCORE: return (-p.sigma_e) * o;
    default: return o;
  }
}

// everything up to (not including) thresholding: returns eps, or x0 when the stage converts (DPM_F_TO_X0)
// FORM_RT / GUIDE_RT as template arguments: the form / guidance kind is read from the stage record at run time
// (wave-uniform branches).  The catch-all kernels -- the one-element-per-lane fallback and the general thresholding
// kernel -- are instantiated once per dtype pair this way instead of once per (form, guidance, xe) combination.
constexpr int FORM_RT = -1, GUIDE_RT = -1;
template <int GUIDE>
__device__ __forceinline__ bool guide_is(int what, const KParams& p) {
  return GUIDE == GUIDE_RT ? p.guidance == what : GUIDE == what;
}

template <int GUIDE, int SPEC = PM_RT, typename V = float>
__device__ __forceinline__ V prologue(V xe, V o0, V o1, V gg, const KParams& p) {
  static_assert(SPEC != SPEC_GENERIC, "SPEC_GENERIC dispatches to a mode (stage_tiles); the prologue takes the mode");
  V eps;
  if (guide_is<GUIDE>(DPM_GUIDE_CFG, p)) {  // ref :326-330: uncond + scale * (cond - uncond)
    V nu = to_noise<SPEC>(o1, xe, p), nc = to_noise<SPEC>(o0, xe, p);
    eps = nu + p.cfg_scale * (nc - nu);
  } else if (guide_is<GUIDE>(DPM_GUIDE_CLASSIFIER, p)) {  // ref :321
    eps = to_noise<SPEC>(o0, xe, p) - p.cg_scale * gg;
  } else {
    eps = to_noise<SPEC>(o0, xe, p);
  }
  if (SPEC >= 0) return (SPEC & 1) ? div_by_alpha(xe - p.sigma_e * eps, p) : eps;  // ref :439, division by invariant
  if (p.flags & DPM_F_TO_X0) return div_uniform(xe - p.sigma_e * eps, p.alpha_e, p.inv_alpha, (p.fastdiv & 1u) != 0u);  // ref :439
  return eps;
}

// the exponential-integrator combination, reference association
template <int FORM, typename V>
__device__ __forceinline__ V combine(V x, V mn, V h1, V h2, const KParams& p) {
  if (FORM == DPM_FORM_LIN1) {
    return p.cx * x - p.c0 * mn;  // ref :573-576, :585-588
  } else if (FORM == DPM_FORM_TWO) {
    V D = p.k0 * (mn - h1);
    V P = (p.flags & DPM_F_BASE_HIST) ? h1 : mn;
    return (p.cx * x - p.c0 * P) - p.c1 * D;  // ref :827-851 (multistep), :636-669, :728-778 (singlestep)
  } else if (FORM == DPM_FORM_MS3) {
    V D1_0 = p.k0 * (mn - h1);  // ref :880-883
    V D1_1 = p.k1 * (h1 - h2);
    V dd = D1_0 - D1_1;
    V D1 = D1_0 + p.k2 * dd;
    V D2 = p.k3 * dd;
    return ((p.cx * x - p.c0 * mn) - p.c1 * D1) - p.c2 * D2;  // ref :888-903
  } else if (FORM == DPM_FORM_SS3T) {
    V D1_0 = p.k0 * (h2 - h1);  // h1 = model_s, h2 = model_s1, mn = model_s2; ref :741-750, :780-789
    V D1_1 = p.k1 * (mn - h1);
    V D1 = (p.k2 * D1_0 - p.k3 * D1_1) / p.k4;
    V D2 = (2.f * (D1_1 - D1_0)) / p.k4;
    return ((p.cx * x - p.c0 * h1) - p.c1 * D1) - p.c2 * D2;
  } else {
    return mn;  // DPM_FORM_DENOISE, ref :541-545
  }
}

template <int FORM, typename V>
__device__ __forceinline__ V combine_any(V x, V mn, V h1, V h2, const KParams& p) {
  if constexpr (FORM != FORM_RT) {
    return combine<FORM>(x, mn, h1, h2, p);
  } else {
    switch (p.form) {
      case DPM_FORM_LIN1: return combine<DPM_FORM_LIN1>(x, mn, h1, h2, p);
      case DPM_FORM_TWO: return combine<DPM_FORM_TWO>(x, mn, h1, h2, p);
      case DPM_FORM_MS3: return combine<DPM_FORM_MS3>(x, mn, h1, h2, p);
      case DPM_FORM_SS3T: return combine<DPM_FORM_SS3T>(x, mn, h1, h2, p);
      default: return combine<DPM_FORM_DENOISE>(x, mn, h1, h2, p);
    }
  }
}
template <int FORM>
__device__ __forceinline__ bool form_needs_x(const KParams& p) {
  return FORM == FORM_RT ? p.form != DPM_FORM_DENOISE : FORM != DPM_FORM_DENOISE;
}
template <int FORM>
__device__ __forceinline__ bool form_needs_h1(const KParams& p) {
  const int f = FORM == FORM_RT ? p.form : FORM;
  return f == DPM_FORM_TWO || f == DPM_FORM_MS3 || f == DPM_FORM_SS3T;
}
template <int FORM>
__device__ __forceinline__ bool form_needs_h2(const KParams& p) {
  const int f = FORM == FORM_RT ? p.form : FORM;
  return f == DPM_FORM_MS3 || f == DPM_FORM_SS3T;
}

template <int FORM>
struct FormTraits {
  static constexpr bool needs_x = FORM != DPM_FORM_DENOISE;
  static constexpr bool needs_h1 = FORM == DPM_FORM_TWO || FORM == DPM_FORM_MS3 || FORM == DPM_FORM_SS3T;
  static constexpr bool needs_h2 = FORM == DPM_FORM_MS3 || FORM == DPM_FORM_SS3T;
};

// ------------------------------------------------------------------------------------------------
// extensions around the update: all optional, all wave-uniform
//   * eps_stride: the network output is a channel slice of a wider tensor (learned-variance models return
//     [B,2C,H,W] and the solver uses out[:, :C], runners/diffusion.py:596-603): sample b of e0/e1 starts at
//     b*eps_stride instead of b*per_sample, so no .contiguous() copy is needed;
//   * xo2: second copy of x_out -- the other half of the [2B,...] network input of classifier-free guidance
//     (replaces torch.cat([x] * 2), ref :326);
//   * mask / ba / bb: the mask blend the DiffEdit / inpainting callers run as correcting_xt_fn after every update
//     (scripts/diffedit_inpaint.ipynb cell 6):  x <- x*mask + (1 - mask)*(blend_alpha*ba + blend_sigma*bb)
//     (bb null: x <- x*mask + (1 - mask)*ba), mask indexed modulo mask_period (broadcast [H,W] / [C,H,W] masks).
// ------------------------------------------------------------------------------------------------
struct KExt {
  void* xo2;
  const void* mask;
  const void* ba;
  const void* bb;
  int64_t mask_period;  // elements
  int64_t per_sample;   // elements of one sample (eps_stride != 0 only)
  int64_t eps_stride;   // elements between samples of e0 / e1; 0 = contiguous
  float blend_alpha, blend_sigma;
};

// reference association of the blend: x * mask + (1 - mask) * (alpha * a + sigma * b), one rounding per operation
__device__ __forceinline__ float blend_ref(float v, float m, float a, float b, bool has_b, const KExt& e) {
  const float r = has_b ? e.blend_alpha * a + e.blend_sigma * b : a;
  return v * m + (1.f - m) * r;
}

// ------------------------------------------------------------------------------------------------
// the streaming stage kernel
// ------------------------------------------------------------------------------------------------
// model values of the U tiles of one workgroup iteration for prologue mode PM (the loaded registers arrive by reference:
// a plain forced-inline function, so that they stay registers).  The empty asm statement is a side effect: without one a
// switch over these calls is if-converted into computing every mode and selecting.
template <int GUIDE, bool XE, int PM, int U, bool NEEDS_X>
__device__ __forceinline__ void tile_models(const float (&vx)[U][EPT], const float (&vxe)[U][EPT], const float (&v0)[U][EPT],
                                            const float (&v1)[U][EPT], const float (&vg)[U][EPT], const bool need_xe,
                                            const KParams& p, f32x2 (&mnv)[U][EPT / 2]) {
  asm volatile("");  // no clobbers: a "memory" clobber would force the kernel arguments behind `p` into memory
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int q = 0; q < EPT; q += 2) {  // adjacent pairs: see f32x2
      const f32x2 z = {0.f, 0.f};
      const f32x2 x2 = NEEDS_X || (!XE && need_xe) ? f32x2{vx[u][q], vx[u][q + 1]} : z;
      const f32x2 xe2 = XE ? f32x2{vxe[u][q], vxe[u][q + 1]} : x2;
      mnv[u][q / 2] = prologue<GUIDE, PM>(xe2, f32x2{v0[u][q], v0[u][q + 1]},
                                          GUIDE == DPM_GUIDE_CFG ? f32x2{v1[u][q], v1[u][q + 1]} : z,
                                          GUIDE == DPM_GUIDE_CLASSIFIER ? f32x2{vg[u][q], vg[u][q + 1]} : z, p);
    }
}

// EXT = the launch uses one of the KExt extensions (duplicate store, strided network output, mask blend): the same
// tiling, with the extra index arithmetic and streams compiled in.  EXT launches have no ragged tail (the scalar kernel
// takes those) and use the split layout only when no per-sample / per-period index is involved.
// One workgroup iteration: the U tiles that start at tile t0 of ONE tensor set (shared by the single-request kernel and
// the fused multi-request kernel below).  Everything outside the two unrolled loops is wave-uniform scalar work.
template <typename TS, typename TE, int FORM, int GUIDE, bool XE, int SPEC, int U, int NT, bool EXT>
__device__ __forceinline__ void stage_tiles(const TS* __restrict__ x, const TS* __restrict__ xe,
                                            const TE* __restrict__ e0, const TE* __restrict__ e1,
                                            const TE* __restrict__ g, const TS* __restrict__ h1,
                                            const TS* __restrict__ h2, TS* __restrict__ xo, TS* __restrict__ mo,
                                            const int64_t ngroups, const int64_t t0, const KParams& p, const KExt& ext) {
  using FT = FormTraits<FORM>;
  constexpr bool SPLIT = sizeof(TS) == 4;  // see load_tile
  const bool need_xe = spec_need_xe<SPEC>(p);
  const bool store_m = p.flags & DPM_F_STORE_M;
  const TS* mask = EXT ? static_cast<const TS*>(ext.mask) : nullptr;
  const TS* ba = EXT ? static_cast<const TS*>(ext.ba) : nullptr;
  const TS* bb = EXT ? static_cast<const TS*>(ext.bb) : nullptr;
  TS* xo2 = EXT ? static_cast<TS*>(ext.xo2) : nullptr;
  const int64_t gps = EXT ? ext.per_sample / EPT : 1, sgroups = EXT ? ext.eps_stride / EPT : 0;
  const int64_t mgroups = EXT ? ext.mask_period / EPT : 1;
  const bool small = ngroups < (int64_t)0x7fffffff;  // 32-bit index arithmetic is enough (n < 2^34 elements)
  // the split layout needs every tensor of the launch indexed by whole tiles: a mask whose period is a multiple of the
  // 2048-element tile ([64,64] and larger masks)
  // (a strided network output: samples made of whole tiles, so that a tile's groups are consecutive in e0 / e1 too)
  const bool can_split = SPLIT && (!EXT || ((!ext.eps_stride || gps % 256 == 0) &&
                                            (!mask || ext.mask_period % (256 * EPT) == 0)));
  float vx[U][EPT], vxe[U][EPT], v0[U][EPT], v1[U][EPT], vg[U][EPT], vh1[U][EPT], vh2[U][EPT];
  float vm[EXT ? U : 1][EPT], va[EXT ? U : 1][EPT], vb[EXT ? U : 1][EPT];
  // Lanes past the end of the last tile load a clamped (valid) group and only skip the store: loads and arithmetic
  // stay in straight-line code, so the loaded registers are consumed where they land (no copies at a join).
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t gr = (t0 + u) * 256 + threadIdx.x;
    const int64_t gi = gr < ngroups ? gr : ngroups - 1;
    const bool split = can_split && (t0 + u) * 256 + 256 <= ngroups;
    int64_t ge = gi;  // group index into the network outputs
    if (EXT && ext.eps_stride) {
      if (small) {
        const uint32_t q = (uint32_t)gi / (uint32_t)gps;
        ge = (int64_t)q * sgroups + ((uint32_t)gi - q * (uint32_t)gps);
      } else {
        ge = (gi / gps) * sgroups + gi % gps;
      }
    }
    if (FT::needs_x || (!XE && need_xe)) load_tile<(NT & 1) != 0>(x, gi, split, vx[u]);
    if (XE && need_xe) load_tile<(NT & 1) != 0>(xe, gi, split, vxe[u]);
    load_tile<(NT & 1) != 0>(e0, ge, split, v0[u]);
    if (GUIDE == DPM_GUIDE_CFG) load_tile<(NT & 1) != 0>(e1, ge, split, v1[u]);
    if (GUIDE == DPM_GUIDE_CLASSIFIER) load_tile<(NT & 1) != 0>(g, gi, split, vg[u]);
    if (FT::needs_h1) load_tile<(NT & 1) != 0>(h1, gi, split, vh1[u]);
    if (FT::needs_h2) load_tile<(NT & 1) != 0>(h2, gi, split, vh2[u]);
    if (EXT && mask) {
      if (split) {  // mask, known image and noise in the state's split layout: whole 1 KiB runs per access
        const int64_t mtiles = mgroups / 256;
        const int64_t mt = small ? (int64_t)((uint32_t)(t0 + u) % (uint32_t)mtiles) : (t0 + u) % mtiles;
        load_tile<false>(mask, mt * 256 + threadIdx.x, true, vm[u]);
        load_tile<(NT & 1) != 0>(ba, gi, true, va[u]);
        if (bb) load_tile<(NT & 1) != 0>(bb, gi, true, vb[u]);
      } else {
        const int64_t gm = small ? (int64_t)((uint32_t)gi % (uint32_t)mgroups) : gi % mgroups;
        load_pack<false>(mask, gm, vm[u]);
        load_pack<false>(ba, gi, va[u]);
        if (bb) load_pack<(NT & 1) != 0>(bb, gi, vb[u]);
      }
    }
  }
  // the model values of all U tiles; SPEC_GENERIC picks the prologue mode here, once per workgroup iteration
  f32x2 mnv[U][EPT / 2];
#define DPM_MODELS(PM_) tile_models<GUIDE, XE, PM_, U, FT::needs_x>(vx, vxe, v0, v1, vg, need_xe, p, mnv)
  if constexpr (SPEC == SPEC_GENERIC) {
    switch (generic_mode(p)) {
      case 0: DPM_MODELS(0); break;
      case 1: DPM_MODELS(1); break;
      case 2: DPM_MODELS(2); break;
      case 3: DPM_MODELS(3); break;
      case 4: DPM_MODELS(4); break;
      case 5: DPM_MODELS(5); break;
      case 6: DPM_MODELS(6); break;
      case 7: DPM_MODELS(7); break;
      default: DPM_MODELS(PM_RT); break;
    }
  } else {
    DPM_MODELS(SPEC);
  }
#undef DPM_MODELS
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t gi = (t0 + u) * 256 + threadIdx.x;
    const bool split = can_split && (t0 + u) * 256 + 256 <= ngroups;
    float ox[EPT], om[EPT];
#pragma unroll
    for (int q = 0; q < EPT; q += 2) {
      const f32x2 z = {0.f, 0.f};
      const f32x2 mn = mnv[u][q / 2];
      const f32x2 o = combine<FORM>(FT::needs_x ? f32x2{vx[u][q], vx[u][q + 1]} : z, mn,
                                    FT::needs_h1 ? f32x2{vh1[u][q], vh1[u][q + 1]} : z,
                                    FT::needs_h2 ? f32x2{vh2[u][q], vh2[u][q + 1]} : z, p);
      om[q] = mn.x;
      om[q + 1] = mn.y;
      ox[q] = o.x;
      ox[q + 1] = o.y;
    }
    if (EXT && mask) {
#pragma unroll
      for (int j = 0; j < EPT; ++j)
        ox[j] = blend_ref(to_f32(from_f32<TS>(ox[j])), vm[EXT ? u : 0][j], va[EXT ? u : 0][j],
                          bb ? vb[EXT ? u : 0][j] : 0.f, bb != nullptr, ext);
    }
    if (gi < ngroups) {
      store_tile<(NT & 2) != 0>(xo, gi, split, ox);
      if (EXT && xo2) store_tile<(NT & 2) != 0>(xo2, gi, split, ox);
      if (store_m) store_tile<(NT & 4) != 0>(mo, gi, split, om);
    }
  }
}

template <typename TS, typename TE, int FORM, int GUIDE, bool XE, int SPEC, int U, int NT, bool EXT, bool DYN = false>
__global__ __launch_bounds__(256) void stage_kernel(const TS* __restrict__ x, const TS* __restrict__ xe,
                                                    const TE* __restrict__ e0, const TE* __restrict__ e1,
                                                    const TE* __restrict__ g, const TS* __restrict__ h1,
                                                    const TS* __restrict__ h2, TS* __restrict__ xo,
                                                    TS* __restrict__ mo, int64_t n, const KParams p_arg, KExt ext,
                                                    const dpm_stage* dyn, const int32_t* skip) {
  using FT = FormTraits<FORM>;
  if constexpr (DYN) {  // device-resident coefficients: see LaunchCtx
    if (*skip) return;
  }
  KParams p_dyn;  // (a copy of the argument, even a const one, would leave part of it in memory -> LDS)
  if constexpr (DYN) p_dyn = params_from_dyn(p_arg, dyn);
  const KParams& p = DYN ? p_dyn : p_arg;
  const int64_t ngroups = n / EPT;
  // a tile = 256 consecutive groups (one per lane of the workgroup); a workgroup iteration covers U tiles and
  // issues the loads of all of them before the first use
  const int64_t ntiles = (ngroups + 255) / 256;
  for (int64_t t0 = (int64_t)blockIdx.x * U; t0 < ntiles; t0 += (int64_t)gridDim.x * U)
    stage_tiles<TS, TE, FORM, GUIDE, XE, SPEC, U, NT, EXT>(x, xe, e0, e1, g, h1, h2, xo, mo, ngroups, t0, p, ext);
  if constexpr (!EXT) {
    // ragged tail (n % 8 elements): first lanes of block 0, scalar
    const bool need_xe = spec_need_xe<SPEC>(p);
    const bool store_m = p.flags & DPM_F_STORE_M;
    const int64_t tail0 = ngroups * EPT;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < n) {
      const int64_t i = tail0 + threadIdx.x;
      const float xv = (FT::needs_x || (!XE && need_xe)) ? to_f32(x[i]) : 0.f;
      const float xev = XE ? (need_xe ? to_f32(xe[i]) : 0.f) : xv;
      const float mn = prologue<GUIDE, SPEC == SPEC_GENERIC ? (int)PM_RT : SPEC>(
          xev, to_f32(e0[i]), GUIDE == DPM_GUIDE_CFG ? to_f32(e1[i]) : 0.f, GUIDE == DPM_GUIDE_CLASSIFIER ? to_f32(g[i]) : 0.f, p);
      xo[i] = from_f32<TS>(combine<FORM>(xv, mn, FT::needs_h1 ? to_f32(h1[i]) : 0.f, FT::needs_h2 ? to_f32(h2[i]) : 0.f, p));
      if (store_m) mo[i] = from_f32<TS>(mn);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fused multi-request stage: ONE launch advances up to DPM_MULTI_MAX independent sampling requests that are at the
// same stage of the same plan (same scalars, element count and dtypes; their own buffers).  A server holding R requests
// in flight -- or one caller sampling R batches side by side -- pays the launch's ramp-up and drain (2-3 us of a
// 42 MB launch's 8.5 us when its inputs come from HBM) once per R x 42 MB instead of once per 42 MB.  The pointer table
// is a kernel argument (kernarg segment -> scalar loads with a wave-uniform index); the virtual tile index runs over
// request-major super-tiles (U tiles of one request), so consecutive workgroups stream consecutive addresses.
// ------------------------------------------------------------------------------------------------
constexpr int MULTI_MAX = DPM_MULTI_MAX;
struct MultiTab {
  const void* x[MULTI_MAX];
  const void* e0[MULTI_MAX];
  const void* e1[MULTI_MAX];
  const void* h1[MULTI_MAX];
  const void* h2[MULTI_MAX];
  void* xo[MULTI_MAX];
  void* mo[MULTI_MAX];
};

template <typename TS, typename TE, int FORM, int GUIDE, int SPEC, int U, int NT>
__global__ __launch_bounds__(256) void stage_kernel_multi(const MultiTab tab, int64_t n, uint32_t nreq, uint32_t spr,
                                                          KParams p, uint32_t xcd_span) {
  const int64_t ngroups = n / EPT;
  KExt ext = {};
  const uint32_t total = nreq * spr;  // spr = super-tiles (U tiles) per request
  for (uint32_t v0 = blockIdx.x; v0 < (xcd_span ? 8u * xcd_span : total); v0 += gridDim.x) {
    // xcd_span != 0 (tuning): workgroup b runs on XCD b % 8 -- give every XCD one contiguous eighth of the tile space
    const uint32_t v = xcd_span ? (v0 & 7u) * xcd_span + (v0 >> 3) : v0;
    if (v >= total) continue;
    const uint32_t r = v / spr;
    const int64_t t0 = (int64_t)(v - r * spr) * U;
    stage_tiles<TS, TE, FORM, GUIDE, false, SPEC, U, NT, false>(
        static_cast<const TS*>(tab.x[r]), nullptr, static_cast<const TE*>(tab.e0[r]), static_cast<const TE*>(tab.e1[r]),
        nullptr, static_cast<const TS*>(tab.h1[r]), static_cast<const TS*>(tab.h2[r]), static_cast<TS*>(tab.xo[r]),
        static_cast<TS*>(tab.mo[r]), ngroups, t0, p, ext);
  }
}

// same arithmetic, one element per lane: used when a pointer is not 16/32-byte aligned (views with offsets), for ragged
// extended launches and for (form, xe) combinations the streaming family does not instantiate.  ONE kernel per dtype
// pair: form and guidance are read from the stage record (wave-uniform branches), xe always points at the state the
// network saw (= x when there is no separate one).
template <typename TS, typename TE, bool DYN = false>
__global__ __launch_bounds__(256) void stage_kernel_scalar(const TS* __restrict__ x, const TS* __restrict__ xe,
                                                           const TE* __restrict__ e0, const TE* __restrict__ e1,
                                                           const TE* __restrict__ g, const TS* __restrict__ h1,
                                                           const TS* __restrict__ h2, TS* __restrict__ xo,
                                                           TS* __restrict__ mo, int64_t n, const KParams p_arg, KExt ext,
                                                           const dpm_stage* dyn, const int32_t* skip) {
  if constexpr (DYN) {
    if (*skip) return;
  }
  KParams p_dyn;  // (a copy of the argument, even a const one, would leave part of it in memory -> LDS)
  if constexpr (DYN) p_dyn = params_from_dyn(p_arg, dyn);
  const KParams& p = DYN ? p_dyn : p_arg;
  const bool need_xe = (p.flags & DPM_F_TO_X0) || p.model_type == DPM_MODEL_X_START || p.model_type == DPM_MODEL_V;
  const bool store_m = p.flags & DPM_F_STORE_M;
  const bool nx = form_needs_x<FORM_RT>(p), nh1 = form_needs_h1<FORM_RT>(p), nh2 = form_needs_h2<FORM_RT>(p);
  const bool cfg = p.guidance == DPM_GUIDE_CFG, clsg = p.guidance == DPM_GUIDE_CLASSIFIER;
  const TS* mask = static_cast<const TS*>(ext.mask);
  const TS* ba = static_cast<const TS*>(ext.ba);
  const TS* bb = static_cast<const TS*>(ext.bb);
  TS* xo2 = static_cast<TS*>(ext.xo2);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t ie = ext.eps_stride ? (i / ext.per_sample) * ext.eps_stride + i % ext.per_sample : i;
    const float xv = nx ? to_f32(x[i]) : 0.f;
    const float xev = need_xe ? to_f32(xe[i]) : 0.f;
    const float mn = prologue<GUIDE_RT>(xev, to_f32(e0[ie]), cfg ? to_f32(e1[ie]) : 0.f, clsg ? to_f32(g[i]) : 0.f, p);
    float o = combine_any<FORM_RT>(xv, mn, nh1 ? to_f32(h1[i]) : 0.f, nh2 ? to_f32(h2[i]) : 0.f, p);
    if (mask) {
      o = to_f32(from_f32<TS>(o));  // the reference blends the stored state
      o = blend_ref(o, to_f32(mask[i % ext.mask_period]), to_f32(ba[i]), bb ? to_f32(bb[i]) : 0.f, bb != nullptr, ext);
    }
    const TS ov = from_f32<TS>(o);
    xo[i] = ov;
    if (xo2) xo2[i] = ov;
    if (store_m) mo[i] = from_f32<TS>(mn);
  }
}

// ------------------------------------------------------------------------------------------------
// dynamic thresholding (ref :416-425)
//
//   s = quantile(|x0|, ratio) over the sample;  s = max(s, max_val);  x0 <- clamp(x0, -s, s) / s;
//   then the same combine / epilogue as the streaming kernel.
//
// A *cluster* of k workgroups owns one sample at a time (k = 1 when a sample fits one workgroup's LDS and the batch
// alone fills the chip; k > 1 spreads small batches and large samples -- 3x256x256 pixels -- over many CUs).  Each
// workgroup computes x0 for its chunk of the sample ONCE into LDS, so HBM sees every stream exactly once (5N for the
// 2M stage).  The quantile needs two exact order statistics of |x0| (non-negative floats order like their bit
// patterns).  Two routes to a short candidate list that provably holds them:
//   * top-K front end (ratio close to 1: K = n - rank is a small part of the thread count): every thread keeps the
//     largest |x0| it produced; the K-th largest element of the sample is at least the K-th largest of those maxima, so
//     a histogram of ONE value per thread bounds the top digit, and the elements at or above it are the candidates;
//   * otherwise the level-0 histogram (top 11 bits) of all elements, built by LDS atomics during the load phase; the
//     candidates are the elements of the selected bin, the smallest value of the higher bins rides along.
// The candidates are compacted (count in registers, wavefront scan, one LDS atomic per wavefront), exchanged through the
// workspace when k > 1, and -- when there are at most T of them, the usual case -- finished by rank counting: every
// thread counts the candidates smaller than its own one.  Longer lists (plateaus, K > T/4) run the remaining levels of
// an 11/11/9-bit radix select and a min-above search.  Bins are located by a workgroup-wide prefix sum (16 bytes of
// histogram per thread, DPP wavefront scan).  The fractional rank is the reference's fp32 `ratio*(n-1)` and the
// interpolation is ATen's lerp.
//
// Cluster barriers are single-use counters in a zeroed workspace (agent-scope atomics); the launch keeps the grid
// within the number of co-resident workgroups, so waiting workgroups can always be joined by their peers.
// ------------------------------------------------------------------------------------------------
constexpr int THR_THREADS = 512;
constexpr int THR_NB = 2048;                 // bins per radix level
// workspace words per sample (k > 1): 3 level histograms, the histogram of the per-thread maxima and the candidate list
// of the top-K front end, counters (a 256-byte multiple)
constexpr int THR_WS_WORDS = 5 * THR_NB + 64;
constexpr int THR_WS_MAXH = 3 * THR_NB;
constexpr int THR_WS_LIST = 4 * THR_NB;
constexpr int THR_WS_CNT = 5 * THR_NB;  // [0..3] barriers of the radix levels / min-above, [4..5] barriers of the top-K front
                                        // end, [8] min-above complement, [9], [10] list cursors
constexpr int THR_CHUNK_MAX = 12288;         // elements of a sample one workgroup keeps in LDS (48 KiB)
constexpr int THR_CAP = 4096;                // candidates (elements sharing the selected top digit) kept compacted
constexpr int THR_GCAP = THR_NB;             // cluster-wide candidates exchanged through the level-1 histogram's words
// single-exchange route of a cluster (cluster_select_once): every workgroup publishes the elements of its chunk that
// can still be among the sample's K largest into its own slot of the workspace -- header + values, every word tagged
constexpr int THR_ROWS = 2;  // tile rows a thread keeps in flight in the streaming phases (3 and 6 measured: no faster)
// fine digits of the single-exchange route: |x0| bits >> THR_FSHIFT (8 exponent + 9 mantissa bits: 0.2 % wide bins),
// THR_NB of them below a maximum (a factor 54).  Measured against 1.5 % bins (shift 17) on [64,3,256,256]: 55.7 -> 53.1 us
// per stage -- the union's values crowd into ~40 of the coarse bins and their LDS atomics serialise.
constexpr int THR_FSHIFT = 14;
constexpr int THR_KMAX = 256;                // largest cluster the single-exchange route serves
constexpr int THR_MISC = 32 + 2 * THR_KMAX;  // scalar LDS words of the thresholding kernel (see stage_thresh_kernel)
constexpr int THR_SLOT_CAP = 256;            // values one workgroup may publish
constexpr int THR_SLOT_HDR = 8;              // [0] tag | count (or overflow), [1] tag | bound, [2] tag | chunk maximum
constexpr int THR_SLOTW = THR_SLOT_CAP + THR_SLOT_HDR;
constexpr uint32_t THR_TAG = 0x80000000u;    // |x0| bit patterns have bit 31 clear: a tagged word is never 0
constexpr uint32_t THR_OVERFLOW = 0x40000000u;
constexpr int THR_WS_DONE = THR_WS_CNT + 12; // workgroups of the cluster that are through with the workspace
constexpr uint32_t THR_SPIN_LIMIT = 1u << 22; // polls (about a microsecond each) before a wait gives up: seconds

struct ThrParams {
  int64_t per_sample;
  int32_t lo, hi;  // floor / ceil of the fp32 rank (ascending order)
  float w;         // fractional part
  float max_val;
  int32_t chunk;   // elements per workgroup of a cluster (multiple of 4 when the vector path is on)
  int32_t k;       // workgroups per cluster
  int32_t groups;  // clusters in the grid
  int32_t batch;
  int32_t vec;     // 1: 4-element vector accesses are legal for every tensor of this launch
  int32_t topk;    // > 0: K = per_sample - lo is small enough for the top-K front end of the select
  int32_t mrank;   // top-K: ascending rank of the K-th largest per-thread maximum among the contributing threads
  int32_t fastdiv; // 1: noise-prediction network + eps -> x0 with a divisor that passes div_invariant_ok (see div_by_alpha)
  int32_t quota;   // > 0: single-exchange cluster route; values beyond this rank of the per-thread maxima are not published
  int32_t kbig;    // K = per_sample - lo (the wanted element is the K-th largest of the sample)
  int32_t slot_pub; // entries of a slot that are always written (values, then the bare tag)
  int32_t slot_cap; // values per workgroup slot: a power of two <= THR_SLOT_CAP with k * slot_cap <= THR_CAP
  int32_t slot_shift; // log2(slot_cap)
  int32_t debug_reject; // testing: run the single-exchange select but always take the general route afterwards
  int64_t ws_stride; // words per sample in ws
  uint32_t* ws;    // k > 1: batch x ws_stride words, all zero between launches (the kernel cleans up after itself)
  uint32_t* fault; // host-mapped word: set when a cluster wait timed out (the launch's results are then garbage)
#ifdef DPM_THR_TIMING
  uint64_t* tdbg;  // 16 timestamps per workgroup (tools/thr_timeline.py)
#endif
};

// inclusive prefix sum over the 64 lanes of a wavefront: DPP row shifts inside the rows of 16 lanes, then the two row
// broadcasts (no LDS traffic, six VALU instructions).  Needs all 64 lanes active.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// 4 consecutive elements (one 16-byte / 8-byte access); NT = streaming (non-temporal) access for data that is dead
// after this kernel
template <bool NT = false>
__device__ __forceinline__ void load4(const float* __restrict__ p, int64_t i, float (&o)[4]) {
  const u32x4 a = ld16<NT>(reinterpret_cast<const u32x4*>(p + i));
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = __uint_as_float(a[j]);
}
template <bool NT = false>
__device__ __forceinline__ void load4(const __half* __restrict__ p, int64_t i, float (&o)[4]) {
  const u32x2* q = reinterpret_cast<const u32x2*>(p + i);
  const u32x2 a = NT ? __builtin_nontemporal_load(q) : *q;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    o[2 * j] = __half2float(__ushort_as_half((unsigned short)(a[j] & 0xffffu)));
    o[2 * j + 1] = __half2float(__ushort_as_half((unsigned short)(a[j] >> 16)));
  }
}
template <bool NT = false>
__device__ __forceinline__ void load4(const bf16_t* __restrict__ p, int64_t i, float (&o)[4]) {
  const u32x2* q = reinterpret_cast<const u32x2*>(p + i);
  const u32x2 a = NT ? __builtin_nontemporal_load(q) : *q;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    o[2 * j] = __uint_as_float(a[j] << 16);
    o[2 * j + 1] = __uint_as_float(a[j] & 0xffff0000u);
  }
}
template <bool NT = false>
__device__ __forceinline__ void store4(float* __restrict__ p, int64_t i, const float (&v)[4]) {
  u32x4 a;
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = __float_as_uint(v[j]);
  st16<NT>(reinterpret_cast<u32x4*>(p + i), a);
}
template <bool NT = false>
__device__ __forceinline__ void store4(__half* __restrict__ p, int64_t i, const float (&v)[4]) {
  u32x2 a;
#pragma unroll
  for (int j = 0; j < 2; ++j)
    a[j] = pack_half2(v[2 * j], v[2 * j + 1]);
  st8<NT>(reinterpret_cast<u32x2*>(p + i), a);
}
template <bool NT = false>
__device__ __forceinline__ void store4(bf16_t* __restrict__ p, int64_t i, const float (&v)[4]) {
  u32x2 a;
#pragma unroll
  for (int j = 0; j < 2; ++j) a[j] = pack_bf162(v[2 * j], v[2 * j + 1]);
  st8<NT>(reinterpret_cast<u32x2*>(p + i), a);
}

// A wait on another workgroup gives up after THR_SPIN_LIMIT polls (seconds): the peers of a cluster are co-resident by
// construction, so this only happens when something else keeps them off the chip that long (two clustered graphs
// replayed concurrently on different streams) or on a true deadlock.  It never traps: the waiter raises the library's
// host-mapped fault word, stops waiting for the rest of the launch (its results are garbage) and the kernel terminates;
// the next clustered launch returns DPM_ERR_FAULT.
__device__ __forceinline__ void raise_fault(uint32_t* fault) {
  if (fault) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// all workgroups of a cluster meet here; `cnt` is a zero-initialised single-use counter.  Everything the cluster
// shares travels as agent-scope atomics and sc1 loads, so no cache write-back / invalidate is needed: drain this
// wave's atomics, arrive with a relaxed atomic, poll with relaxed sc1 loads (MI355X_MICROARCH.md, barrier-counter).
// `dead` (LDS word): a previous wait of this workgroup timed out -- do not wait again.
__device__ __forceinline__ void cluster_barrier(uint32_t* cnt, uint32_t k, uint32_t* dead, uint32_t* fault) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (!*dead && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > THR_SPIN_LIMIT) {
        *dead = 1u;
        raise_fault(fault);
      }
    }
  }
  __syncthreads();
}

// keep m1 >= m2 >= m3 >= m4, the four largest values seen so far (duplicates are separate entries): inserting u into a
// sorted list replaces every entry by the median of itself, its larger neighbour and u -- one v_med3_u32 each
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;  // the backend folds this shape into v_med3_u32
  const uint32_t t = hi < c ? hi : c;
  return lo > t ? lo : t;
}
__device__ __forceinline__ void top4_insert(uint32_t u, uint32_t& m1, uint32_t& m2, uint32_t& m3, uint32_t& m4) {
  m4 = med3_u32(m3, m4, u);
  m3 = med3_u32(m2, m3, u);
  m2 = med3_u32(m1, m2, u);
  m1 = u > m1 ? u : m1;
}

// Every thread owns 4 consecutive bins of the workgroup's LDS histogram (THR_NB = 4 T): one conflict-free 16-byte
// read, a wavefront scan, the wavefront totals through LDS.  The thread whose bins hold the ascending `rank`
// publishes misc[0] = bin, misc[1] = rank inside that bin, misc[2] = the bin's count.  The histogram is left ZEROED.
template <int T>
__device__ __forceinline__ void locate_bin(uint32_t* hist, uint32_t* misc, uint32_t rank, int tid) {
  static_assert(THR_NB == 4 * T, "one 16-byte histogram slice per thread");
  u32x4* h4 = reinterpret_cast<u32x4*>(hist);
  const u32x4 v = h4[tid];
  h4[tid] = u32x4{0u, 0u, 0u, 0u};
  const uint32_t tot = (v[0] + v[1]) + (v[2] + v[3]);
  const uint32_t incl = wave_incl_scan(tot);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wavefront-uniform: scalar compares below
  if ((tid & 63) == 63) misc[16 + wave] = incl;
  __syncthreads();
  static_assert(T / 64 == 8, "two 16-byte reads of the wavefront totals");
  const u32x4 w0 = *reinterpret_cast<const u32x4*>(misc + 16), w1 = *reinterpret_cast<const u32x4*>(misc + 20);
  uint32_t before = 0u;
#pragma unroll
  for (int w = 0; w < 4; ++w) before += (w < wave ? w0[w] : 0u) + (w + 4 < wave ? w1[w] : 0u);
  const uint32_t excl = before + incl - tot;
  if (rank >= excl && rank - excl < tot) {
    uint32_t r = rank - excl, cbin = v[0];
    int j = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (j == q - 1 && r >= cbin) {
        r -= cbin;
        cbin = v[q];
        j = q;
      }
    misc[0] = (uint32_t)(tid * 4 + j);
    misc[1] = r;
    misc[2] = cbin;
  }
  __syncthreads();
}

// Append the elements of sx0[0..n) whose top digit d satisfies (GE ? d >= bin : d == bin) to cand[] (capacity THR_CAP;
// misc[4] counts all of them): count in registers, wavefront scan, ONE LDS atomic per wavefront for the base slot,
// write -- not one atomic round trip per 64 elements.  Returns this lane's minimum of the elements above digit `bin`.
// The digit of a value is (u >> shift) - dbase, clamped at 0 (shift = 20, dbase = 0: the top 11 bits).
template <int T, bool GE>
__device__ __forceinline__ uint32_t compact_candidates(const float* sx0, int n, uint32_t bin, uint32_t* misc,
                                                       uint32_t* cand, int tid, int shift = 20, uint32_t dbase = 0u) {
  constexpr int NIT = THR_CHUNK_MAX / (T * 4);
  constexpr uint32_t ABS = 0x7fffffffu;
  u32x4 q[NIT];
  uint32_t cnt = 0u, hi = ABS;
  const int last = n > 0 ? ((n - 1) & ~3) : 0;  // rows beyond the end re-read the last group: all LDS reads issue at once
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid * 4 + it * T * 4;
    q[it] = *reinterpret_cast<const u32x4*>(sx0 + (i < n ? i : last));
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid * 4 + it * T * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t u = q[it][j] & ABS;
      const uint32_t dr = u >> shift, d = dr > dbase ? dr - dbase : 0u;
      const bool in = i + j < n;
      cnt += (in && (GE ? d >= bin : d == bin)) ? 1u : 0u;
      if (!GE && in && d > bin && u < hi) hi = u;
    }
  }
  const uint32_t incl = wave_incl_scan(cnt);
  uint32_t slot = 0u;
  if ((tid & 63) == 63 && incl) slot = atomicAdd(&misc[4], incl);
  uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)slot, 63) + incl - cnt;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid * 4 + it * T * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t u = q[it][j] & ABS;
      const uint32_t dr = u >> shift, d = dr > dbase ? dr - dbase : 0u;
      if (i + j < n && (GE ? d >= bin : d == bin)) {
        if (off < (uint32_t)THR_CAP) cand[off] = u;
        ++off;
      }
    }
  }
  return hi;
}

// maximum over the 64 lanes of a wavefront, valid in lane 63 (the DPP ladder of wave_incl_scan with max; 0 is the identity)
__device__ __forceinline__ uint32_t wave_max_to_lane63(uint32_t v) {
#define DPM_DPP_MAX(ctrl, rmask, bc)                                                              \
  {                                                                                               \
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, bc);    \
    v = o > v ? o : v;                                                                            \
  }
  DPM_DPP_MAX(0x111, 0xf, true)
  DPM_DPP_MAX(0x112, 0xf, true)
  DPM_DPP_MAX(0x114, 0xf, true)
  DPM_DPP_MAX(0x118, 0xf, true)
  DPM_DPP_MAX(0x142, 0xa, false)
  DPM_DPP_MAX(0x143, 0xc, false)
#undef DPM_DPP_MAX
  return v;
}

// nc <= T candidates in cand[]: every thread counts the candidates smaller than its own one; the element of ascending
// rank r is the largest candidate with at most r smaller ones.  misc[6] <- rank-th, misc[7] <- (rank+1)-th (or the
// largest candidate when there is none).  One pass -- instead of three histogram levels + a min search.
// rank_count expects misc[6] = misc[7] = 0 and 32 sentinels (0xffffffff: never smaller than anything, the list becomes a
// multiple of 32) behind the list, both visible to the workgroup (a barrier behind the writes); one barrier at its end.
template <int T>
__device__ __forceinline__ void rank_count(const uint32_t* cand, uint32_t nc, uint32_t rank, uint32_t* misc, int tid) {
  if ((uint32_t)(tid & ~63) < nc) {  // wavefronts beyond the list have nothing to do
    const uint32_t my = (uint32_t)tid < nc ? cand[tid] : 0xffffffffu;
    uint32_t lt = 0u;
    for (uint32_t j = 0; j < nc; j += 32) {  // broadcast reads, eight in flight
      u32x4 q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = *reinterpret_cast<const u32x4*>(cand + j + 4 * e);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        lt += (q[e][0] < my ? 1u : 0u) + (q[e][1] < my ? 1u : 0u) + (q[e][2] < my ? 1u : 0u) + (q[e][3] < my ? 1u : 0u);
    }
    const uint32_t ma = wave_max_to_lane63(((uint32_t)tid < nc && lt <= rank) ? my : 0u);
    const uint32_t mb = wave_max_to_lane63(((uint32_t)tid < nc && lt <= rank + 1u) ? my : 0u);
    if ((tid & 63) == 63) {
      if (ma) atomicMax(&misc[6], ma);
      if (mb) atomicMax(&misc[7], mb);
    }
  }
  __syncthreads();
}

template <int T>
__device__ __forceinline__ void rank_select(uint32_t* cand, uint32_t nc, uint32_t rank, uint32_t* misc, int tid) {
  if (tid == 0) {
    misc[6] = 0u;
    misc[7] = 0u;
  }
  if (tid < 32) cand[nc + tid] = 0xffffffffu;
  __syncthreads();
  rank_count<T>(cand, nc, rank, misc, tid);
}

// workgroup-wide exclusive prefix sum of one value per thread (wavefront scan + the wavefront totals through misc[16..]);
// misc[24] <- grand total.  Two barriers.
template <int T>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* misc, int tid) {
  static_assert(T / 64 == 8, "eight wavefronts");
  const uint32_t incl = wave_incl_scan(v);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __syncthreads();  // misc[16..24] may still be read by a previous user
  if ((tid & 63) == 63) misc[16 + wave] = incl;
  __syncthreads();
  uint32_t before = 0u, total = 0u;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const uint32_t t = misc[16 + w];
    before += w < wave ? t : 0u;
    total += t;
  }
  if (tid == 0) misc[24] = total;
  return before + incl - v;
}

// nc candidates in cand[] (any number up to THR_CAP): 11/11/9-bit radix select of the element of ascending rank `rank`
// and of its successor.  hist must be all zero on entry and is left all zero.  a <- element, b <- next order statistic
// (or a when there is none).
template <int T>
__device__ __forceinline__ void list_select(const uint32_t* cand, uint32_t nc, uint32_t rank, uint32_t* hist, uint32_t* misc,
                                            int tid, uint32_t& a, uint32_t& b) {
  uint32_t prefix = 0u, known = 0u, cnt_sel = 0u;
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = pass == 0 ? 20 : pass == 1 ? 9 : 0;
    const uint32_t dmask = pass == 2 ? 0x1ffu : 0x7ffu;
    for (uint32_t i = tid; i < nc; i += T) {
      const uint32_t u = cand[i];
      if ((u & known) == prefix) atomicAdd(&hist[(u >> shift) & dmask], 1u);
    }
    __syncthreads();
    locate_bin<T>(hist, misc, rank, tid);
    prefix |= misc[0] << shift;
    known |= dmask << shift;
    rank = misc[1];
    cnt_sel = misc[2];
  }
  a = prefix;
  b = prefix;
  if (rank + 1u >= cnt_sel) {  // the successor is the smallest candidate above a (if any)
    // misc[12], not misc[3]: that one carries the general route's minimum above the selected digit across this call
    if (tid == 0) misc[12] = 0x7fffffffu;
    __syncthreads();
    uint32_t m = 0x7fffffffu;
    for (uint32_t i = tid; i < nc; i += T) {
      const uint32_t u = cand[i];
      if (u > prefix && u < m) m = u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t o = __shfl_xor(m, d, 64);
      m = o < m ? o : m;
    }
    if ((tid & 63) == 0) atomicMin(&misc[12], m);
    __syncthreads();
    if (misc[12] != 0x7fffffffu) b = misc[12];
  }
}

// The same for a list whose values spread over many fine digits (the union of a cluster's candidates: the upper tail of
// the sample): ONE histogram level over 14-bit digits relative to `umax` (any value >= the list's maximum; 1.5 % wide
// bins), then rank counting among the handful of members of the selected bin -- ~1.4 us instead of 3 us of rank counting
// over the whole list (340 entries) or three histogram levels.  Falls back to list_select when the bin is crowded
// (plateaus).  Entry: hist all zero, misc[13] = 0x7fffffff, misc[14] = 0, all visible (a barrier behind the writes).
// Exit: returns true when hist[0 .. T + 32) may hold leftovers (the bin's members), false when hist is all zero.
template <int T>
__device__ __forceinline__ bool union_select(const uint32_t* cand, uint32_t nc, uint32_t rank, uint32_t umax, uint32_t* hist,
                                             uint32_t* misc, int tid, uint32_t& a, uint32_t& b) {
  constexpr int PER = THR_CAP / T;
  const int lane = tid & 63;
  const uint32_t top = umax >> THR_FSHIFT;
  const uint32_t dbase = top > (uint32_t)(THR_NB - 1) ? top - (uint32_t)(THR_NB - 1) : 0u;
  uint32_t v[PER], d[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t i = (uint32_t)tid + (uint32_t)j * T;
    v[j] = i < nc ? cand[i] : 0u;
    const uint32_t dr = v[j] >> THR_FSHIFT;
    d[j] = dr > dbase ? dr - dbase : 0u;
    if (i < nc) atomicAdd(&hist[d[j]], 1u);
  }
  __syncthreads();
  locate_bin<T>(hist, misc, rank, tid);
  const uint32_t bin = misc[0], r_in = misc[1], cnt_bin = misc[2];
  if (cnt_bin > (uint32_t)T) {  // crowded bin: the general list select (hist is zero again)
    list_select<T>(cand, nc, rank, hist, misc, tid, a, b);
    return false;
  }
  // members of the bin -> hist[0..cnt_bin) (the zeroed histogram doubles as the buffer) + rank_count's sentinels and
  // zeroed result words; minimum of the higher bins
  uint32_t above = 0x7fffffffu;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if ((uint32_t)tid + (uint32_t)j * T < nc) {
      if (d[j] == bin) hist[atomicAdd(&misc[14], 1u)] = v[j];
      if (d[j] > bin && v[j] < above) above = v[j];
    }
  }
  if (tid < 32) hist[cnt_bin + tid] = 0xffffffffu;
  if (tid == 32) {
    misc[6] = 0u;
    misc[7] = 0u;
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) {
    const uint32_t o = __shfl_xor(above, dd, 64);
    above = o < above ? o : above;
  }
  if (lane == 0 && above != 0x7fffffffu) atomicMin(&misc[13], above);
  __syncthreads();
  rank_count<T>(hist, cnt_bin, r_in, misc, tid);
  a = misc[6];
  b = r_in + 1u < cnt_bin ? misc[7] : (misc[13] != 0x7fffffffu ? misc[13] : a);
  return true;
}

// Single-exchange select of a cluster (k workgroups own one sample).  The wanted order statistics are the K-th and
// (K-1)-th largest |x0| of the sample, K = per_sample - lo.  Every workgroup publishes ALL elements of its chunk at or
// above a bound of its own choosing -- the bound of the `quota`-th largest of its per-thread maxima, so about `quota`
// values, where quota = the chunk's expected share K/k of the top K plus six standard deviations -- into its slot of the
// workspace, reads the other slots, and finishes on the union U by itself (rank counting or a radix select in LDS; all
// workgroups hold identical data).  The result is exact whenever the K-th largest of U is not below any workgroup's
// bound M_c = the smallest value it would have published: every unpublished element is then smaller than K elements of
// U, so top-K(U) = top-K(sample).  Otherwise (a slot overflowed, U too small, K-th(U) < max M_c: samples whose large
// values cluster in one chunk) every workgroup reaches the same verdict from the same data and the cluster takes the
// general route with merged histograms -- no extra exchange for the decision.
// One hop: tagged words (bit 31, never set in |x0|) written with sc1 stores into zeroed slots, readers poll the words
// they need -- no drain -> arrive -> poll -> read-back barrier.  Digits here are 14 bits (8 exponent + 6 mantissa bits)
// relative to the chunk's maximum: 1.5 % wide bins instead of 12.5 %, so a bound admits ~10 % more than `quota`, not 2x.
// Returns true with a (K-th largest) and b ((K-1)-th largest, = a when K = 1); false = not solved, LDS state
// (hist zero, misc[4] = 0) ready for the general route.
template <int T>
__device__ __forceinline__ bool cluster_select_once(const float* sx0, int n, bool vec, uint32_t m1, uint32_t m2,
                                                    uint32_t m3, uint32_t m4, bool has, uint32_t* hist, uint32_t* misc,
                                                    uint32_t* cand, uint32_t* slots, const ThrParams& tp, uint32_t k, int c,
                                                    int tid, uint32_t& a_out, uint32_t& b_out, bool stamp) {
#ifdef DPM_THR_TIMING
#define DPM_R1STAMP(j) \
  if (tid == 0 && stamp) tp.tdbg[(int64_t)blockIdx.x * 16 + (j)] = wall_clock64();
#else
#define DPM_R1STAMP(j)
  (void)stamp;
#endif
  const int lane = tid & 63;
  const uint32_t K = (uint32_t)tp.kbig;
  const uint32_t cap = (uint32_t)tp.slot_cap;
  uint32_t* sc = misc + 32;  // [2 k], k <= THR_KMAX: counts and list offsets of the k slots
  // 1. the chunk's maximum (misc[8]: the kernel reduces it on the way out of phase 1) -> digit base
  const uint32_t cmax = misc[8];
  const uint32_t top = cmax >> THR_FSHIFT;
  const uint32_t dbase = top > (uint32_t)(THR_NB - 1) ? top - (uint32_t)(THR_NB - 1) : 0u;
  auto digit = [&](uint32_t u) {
    const uint32_t d = u >> THR_FSHIFT;
    return d > dbase ? d - dbase : 0u;
  };
  // 2. histogram of one value per thread, bound = digit of the quota-th largest maximum
  if (has) atomicAdd(&hist[digit(m1)], 1u);
  __syncthreads();
  const int P = vec ? (n + 3) / 4 : n;  // threads that produced at least one element
  const uint32_t Pl = (uint32_t)(P < T ? P : T);
  locate_bin<T>(hist, misc, Pl > (uint32_t)tp.quota ? Pl - (uint32_t)tp.quota : 0u, tid);
  const uint32_t bin_lo = Pl ? misc[0] : 0u;
  DPM_R1STAMP(8)
  // 3. this chunk's candidates: a thread's are among its four largest values unless even the fourth qualifies
  {
    const int mine = vec ? (has ? 4 * ((n - tid * 4 + T * 4 - 1) / (T * 4)) : 0) : (has ? (n - tid + T - 1) / T : 0);
    const bool c1 = mine > 0 && digit(m1) >= bin_lo, c2 = mine > 1 && digit(m2) >= bin_lo;
    const bool c3 = mine > 2 && digit(m3) >= bin_lo, c4 = mine > 3 && digit(m4) >= bin_lo;
    if (__ballot(c4 && mine > 4)) {
      (void)compact_candidates<T, true>(sx0, n, bin_lo, misc, cand, tid, THR_FSHIFT, dbase);
    } else {
      const uint32_t cnt = (c1 ? 1u : 0u) + (c2 ? 1u : 0u) + (c3 ? 1u : 0u) + (c4 ? 1u : 0u);
      const uint32_t incl = wave_incl_scan(cnt);
      uint32_t slot = 0u;
      if (lane == 63 && incl) slot = atomicAdd(&misc[4], incl);
      uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)slot, 63) + incl - cnt;
      if (c1 && off < (uint32_t)THR_CAP) cand[off] = m1;
      off += c1 ? 1u : 0u;
      if (c2 && off < (uint32_t)THR_CAP) cand[off] = m2;
      off += c2 ? 1u : 0u;
      if (c3 && off < (uint32_t)THR_CAP) cand[off] = m3;
      off += c3 ? 1u : 0u;
      if (c4 && off < (uint32_t)THR_CAP) cand[off] = m4;
    }
  }
  __syncthreads();
  // 4. publish: values, then the header (count, bound, chunk maximum); every word carries the tag
  const uint32_t ncl = misc[4];
  const bool over = ncl > cap;
  uint32_t* mine_slot = slots + (size_t)c * THR_SLOTW;
  // the first `pub` entries of a slot are always written -- the tag alone beyond the count -- so that readers can wait
  // for them without knowing the count (step 5)
  const uint32_t pub = (uint32_t)tp.slot_pub;
  {
    const uint32_t nv = over ? 0u : ncl, nw = nv > pub ? nv : pub;
    for (uint32_t i = tid; i < nw; i += T)
      __hip_atomic_store(&mine_slot[THR_SLOT_HDR + i], (i < nv ? cand[i] : 0u) | THR_TAG, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) {
    const uint32_t bound = bin_lo ? (bin_lo + dbase) << THR_FSHIFT : 0u;  // smallest |x0| with that digit; digit 0 = everything
    __hip_atomic_store(&mine_slot[2], cmax | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mine_slot[1], bound | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mine_slot[0], (over ? THR_OVERFLOW : ncl) | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  DPM_R1STAMP(9)
  // 5. the other workgroups' slots.  Word p of the slot area (slot p >> shift, entry p & (W - 1)) belongs to thread
  // p mod T whatever the counts turn out to be, and the first `pub` entries of every slot get written whatever the count:
  // headers and values are polled TOGETHER, every round's loads issued back to back -- one round trip after the last
  // peer has published, not one for the headers and another for the values.  Entries beyond `pub` (a chunk with more
  // candidates than expected) are fetched in step 6.
  constexpr int PER = THR_CAP / T;
  const int shift = tp.slot_shift;
  const uint32_t W = 1u << shift, words = k << shift;  // <= THR_CAP
  uint32_t w[PER];
  {
    const bool own = (uint32_t)tid < k;
    const uint32_t* sl = slots + (size_t)(own ? tid : 0) * THR_SLOTW;
    uint32_t h0 = THR_TAG, h1 = THR_TAG, h2 = THR_TAG;
    bool act[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const uint32_t q = (uint32_t)tid + (uint32_t)j * T;
      act[j] = q < words && (q & (W - 1u)) < pub;
      w[j] = act[j] ? 0u : THR_TAG;
    }
    if (own) h0 = h1 = h2 = 0u;
    uint32_t spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const uint32_t q = (uint32_t)tid + (uint32_t)j * T;
        if (!(w[j] & THR_TAG))
          w[j] = __hip_atomic_load(slots + (size_t)(q >> shift) * THR_SLOTW + THR_SLOT_HDR + (q & (W - 1u)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!(h0 & THR_TAG)) h0 = __hip_atomic_load(&sl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(h1 & THR_TAG)) h1 = __hip_atomic_load(&sl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(h2 & THR_TAG)) h2 = __hip_atomic_load(&sl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t all = h0 & h1 & h2;
#pragma unroll
      for (int j = 0; j < PER; ++j) all &= w[j];
      if ((all & THR_TAG) || misc[30]) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > THR_SPIN_LIMIT) {
        misc[30] = 1u;
        raise_fault(tp.fault);
      }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (!act[j]) w[j] = 0u;  // not fetched yet (step 6 does if the slot's count reaches that far)
    DPM_R1STAMP(10)
    const bool bad = own && (!(h0 & THR_TAG) || !(h1 & THR_TAG) || !(h2 & THR_TAG) || (h0 & THR_OVERFLOW));
    const uint32_t cnt_t = (own && !bad) ? (h0 & 0xffffu) : 0u;
    const uint32_t bnd_t = own ? (h1 & ~THR_TAG) : 0u, max_t = own ? (h2 & ~THR_TAG) : 0u;
    if (k <= 64u) {  // the usual cluster sizes: wavefront 0 holds every header -- no barrier until the results are out
      if (tid < 64) {
        const uint32_t incl = wave_incl_scan(cnt_t);
        if (own) {
          sc[tid] = cnt_t;
          sc[k + tid] = incl - cnt_t;
        }
        const uint32_t wb = wave_max_to_lane63(bnd_t), wx = wave_max_to_lane63(max_t);
        const bool anybad = __ballot(bad) != 0;
        if (lane == 63) {
          misc[24] = incl;  // entries of the union
          misc[9] = wb;     // largest bound
          misc[12] = wx;    // maximum of the sample (digit base of union_select)
          misc[10] = anybad ? 1u : 0u;
        }
      }
    } else {  // misc[9], [10], [12] start at zero (sample start)
      const uint32_t off_t = block_excl_scan<T>(cnt_t, misc, tid);  // misc[24] <- total
      if (own) {
        sc[tid] = cnt_t;
        sc[k + tid] = off_t;
      }
      const uint32_t wb = wave_max_to_lane63(bnd_t), wx = wave_max_to_lane63(max_t);
      if (lane == 63 && wb) atomicMax(&misc[9], wb);
      if (lane == 63 && wx) atomicMax(&misc[12], wx);
      if (__ballot(bad) && lane == 0) misc[10] = 1u;
    }
    if (tid == 0) {
      misc[13] = 0x7fffffffu;  // union_select: smallest value above the selected bin
      misc[14] = 0u;           //               members of the selected bin appended so far
    }
  }
  __syncthreads();
  const uint32_t total = misc[24], bound_max = misc[9], umax = misc[12];
  const bool ok = !misc[10] && total >= K && total <= (uint32_t)THR_CAP;
  // 6. the union -> cand[]: entry i of slot s goes to off[s] + i
  if (ok) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const uint32_t q = (uint32_t)tid + (uint32_t)j * T;
      const uint32_t sl = q >> shift, i = q & (W - 1u);
      if (q < words && i < sc[sl]) {
        const uint32_t* src = slots + (size_t)sl * THR_SLOTW + THR_SLOT_HDR + i;
        uint32_t spins = 0;
        while (!(w[j] & THR_TAG) && !misc[30]) {
          __builtin_amdgcn_s_sleep(1);
          w[j] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > THR_SPIN_LIMIT) {
            misc[30] = 1u;
            raise_fault(tp.fault);
          }
        }
        cand[sc[k + sl] + i] = w[j] & ~THR_TAG;
      }
    }
  }
  __syncthreads();
  DPM_R1STAMP(11)
  // 7. K-th and (K-1)-th largest of the union
  bool valid = ok && !misc[30];
  bool leftovers = false;  // hist[0 .. T + 32) holds the selected bin's members
  if (valid) {
    uint32_t a, b;
    const uint32_t rank = total - K;  // ascending
    if (total <= 64u) {
      rank_select<T>(cand, total, rank, misc, tid);
      a = misc[6];
      b = rank + 1u < total ? misc[7] : a;
    } else {
      leftovers = union_select<T>(cand, total, rank, umax, hist, misc, tid, a, b);
    }
    a_out = a;
    b_out = b;
    DPM_R1STAMP(12)
#ifdef DPM_THR_DEBUG
    if (tid == 0 && c < 2) printf("[r1] c=%d ncl=%u total=%u K=%u rank=%u bound_max=%08x a=%08x b=%08x\n", c, ncl, total, K, rank, bound_max, a, b);
#endif
    // an unpublished element of some chunk could be among the K largest when the K-th of the union is below a bound
    valid = a >= bound_max && !tp.debug_reject;
  }
  if (!valid) {  // the general route expects its LDS state: hist all zero, no candidates
    if (leftovers) {
      hist[tid] = 0u;
      hist[tid + T] = 0u;
    }
    if (tid == 0) misc[4] = 0u;
    __syncthreads();
  }
  return valid;
}

// HOT != 0: the usual configuration fixed at compile time -- 16-byte accesses legal, noise-prediction network with the
// division by the invariant alpha, no mask blend; HOT = 1 with the top-K front end, HOT = 2 with the full level-0
// histogram -- so that the load and store loops are straight-line code without the wave-uniform branches of the general
// prologue and their operands (the kernel is as sensitive to its instruction count as to HBM).
// Everything else runs the same source with HOT = 0.
template <typename TS, typename TE, int FORM, int GUIDE, bool XE, int T, int HOT>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) void stage_thresh_kernel(
    const TS* __restrict__ x, const TS* __restrict__ xe, const TE* __restrict__ e0, const TE* __restrict__ e1,
    const TE* __restrict__ g, const TS* __restrict__ h1, const TS* __restrict__ h2, TS* __restrict__ xo,
    TS* __restrict__ mo, KParams p, ThrParams tp, KExt ext) {
  // FORM / GUIDE may be FORM_RT / GUIDE_RT (HOT = 0: the general kernel, one per dtype pair): read from p then
  const bool nx = form_needs_x<FORM>(p), nh1 = form_needs_h1<FORM>(p), nh2 = form_needs_h2<FORM>(p);
  const bool g_cfg = guide_is<GUIDE>(DPM_GUIDE_CFG, p), g_cls = guide_is<GUIDE>(DPM_GUIDE_CLASSIFIER, p);
  constexpr int BPT = THR_NB / T;  // histogram bins per thread when all threads touch the histogram
  constexpr uint32_t ABS = 0x7fffffffu;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  float* sx0 = reinterpret_cast<float*>(lds_raw);                    // [chunk]
  uint32_t* hist = reinterpret_cast<uint32_t*>(sx0 + tp.chunk);      // [THR_NB]
  uint32_t* misc = hist + THR_NB;                                    // [THR_MISC]: [0..2] locate_bin result, [3] min-above, [4] candidate
                                                                     // count, [5] list cursor, [16..23] wavefront totals
                                                                     // [32..] cluster_select_once: slot counts, offsets
  uint32_t* cand = misc + THR_MISC;                                  // [THR_CAP + 32] candidates (+ sentinels)
  const bool store_m = p.flags & DPM_F_STORE_M;
  const bool vec = HOT != 0 || tp.vec != 0;
  const uint32_t k = (uint32_t)tp.k;
  const bool route1 = k > 1 && tp.quota > 0;                              // single-exchange cluster select
  const bool track = HOT == 1 || (HOT == 0 && (tp.topk > 0 || route1));  // phase 1 keeps every thread's four largest |x0|
  const bool topk = track && tp.topk > 0;                                 // top-K front end of the general route
  const bool fastdiv = HOT != 0 || tp.fastdiv != 0;
  const int64_t eps_stride = ext.eps_stride;
  const int grp = k == 1 ? (int)blockIdx.x : (int)(blockIdx.x / k);
  const int c = k == 1 ? 0 : (int)(blockIdx.x % k);
  const TS* mask = HOT != 0 ? nullptr : static_cast<const TS*>(ext.mask);
  const TS* ba = HOT != 0 ? nullptr : static_cast<const TS*>(ext.ba);
  const TS* bb = HOT != 0 ? nullptr : static_cast<const TS*>(ext.bb);
  TS* xo2 = static_cast<TS*>(ext.xo2);
  if (threadIdx.x == 0) misc[30] = 0u;  // set when a wait on a peer workgroup timed out (see raise_fault)
  for (int s_idx = grp; s_idx < tp.batch; s_idx += tp.groups) {
    // The thread index is re-materialised per sample: otherwise the compiler hoists every per-thread predicate of the
    // body (dozens of 64-bit lane masks) out of this loop, runs out of SGPRs and pays v_readlane pairs all over the select.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int64_t base = (int64_t)s_idx * tp.per_sample + (int64_t)c * tp.chunk;
    const int64_t ebase = (int64_t)s_idx * (eps_stride ? eps_stride : tp.per_sample) + (int64_t)c * tp.chunk;
    const int64_t left = tp.per_sample - (int64_t)c * tp.chunk;
    const int n = left <= 0 ? 0 : (left < tp.chunk ? (int)left : tp.chunk);
    uint32_t* ws = k == 1 ? nullptr : tp.ws + (int64_t)s_idx * tp.ws_stride;
    // mask index of element base + i without a 64-bit division per element (launch: period < 2^31 or period == n)
    const bool mfull = ext.mask_period >= ((int64_t)1 << 31);
    const uint32_t mbase = (mask && !mfull) ? (uint32_t)(base % ext.mask_period) : 0u;
    const uint32_t mper = (uint32_t)ext.mask_period;
    // -DDPM_THR_TIMING (tools/thr_timeline.py): wall-clock stamps of the first sample a workgroup processes -- 0 start,
    // 1 x0 in LDS, 4 maxima histogram, 5 bin located, 6 candidates compacted / exchanged, 7 rank counting, 2 threshold
    // known, 3 end.  DESIGN.md section 5 quotes them.
#ifdef DPM_THR_TIMING
#define DPM_TSTAMP(j) \
  if (tid == 0 && s_idx == grp) tp.tdbg[(int64_t)blockIdx.x * 16 + (j)] = wall_clock64();
#else
#define DPM_TSTAMP(j)
#endif
    DPM_TSTAMP(0)

    // phase 1: x0 of this workgroup's chunk -> LDS.  On the way: the largest |x0| of every thread (top-K front end),
    // or the level-0 histogram (top 11 bits of |x0|) -- its LDS atomics overlap the global loads
#pragma unroll
    for (int j = 0; j < BPT; ++j) hist[j * T + tid] = 0u;
    if (tid == 0) {
      misc[4] = 0u;   // candidate counter
      misc[3] = ABS;  // smallest value above the selected top digit (cluster exchange)
      misc[8] = 0u;   // cluster_select_once: chunk maximum, largest bound, bad-slot flag, maximum of the sample
      misc[9] = 0u;
      misc[10] = 0u;
      misc[12] = 0u;
    }
    __syncthreads();
    uint32_t m1 = 0u, m2 = 0u, m3 = 0u, m4 = 0u;  // top-K: the four largest |x0| bit patterns this thread produced
    if (vec) {
      // THR_ROWS tile rows per iteration, the loads of all of them issued before the first use: the phase is bound by
      // the bytes one workgroup keeps in flight (two workgroups per CU, and while one of them selects only one streams)
      for (int i0 = tid * 4; i0 < n; i0 += THR_ROWS * T * 4) {
        float vx[THR_ROWS][4], v0[THR_ROWS][4], v1[THR_ROWS][4], vg[THR_ROWS][4];
#pragma unroll
        for (int r = 0; r < THR_ROWS; ++r) {
          const int ir = i0 + r * T * 4 < n ? i0 + r * T * 4 : i0;  // clamped: loads are unconditional
          load4(XE ? xe : x, base + ir, vx[r]);
          load4<true>(e0, ebase + ir, v0[r]);                     // the network outputs are dead after this kernel
          if (g_cfg) load4<true>(e1, ebase + ir, v1[r]);
          if (g_cls) load4<true>(g, base + ir, vg[r]);
        }
#pragma unroll
        for (int r = 0; r < THR_ROWS; ++r) {
          const int i = i0 + r * T * 4;
          if (r == 0 || i < n) {
            float o[4];
            if (fastdiv) {  // uniform: the common parameterisation, division by the invariant alpha (3 VALU ops for ~12)
#pragma unroll
              for (int j = 0; j < 4; j += 2) {  // adjacent pairs: packed fp32 instructions
                const f32x2 z = {0.f, 0.f};
                const f32x2 rr = prologue<GUIDE, SPEC_NOISE_X0, f32x2>(
                    f32x2{vx[r][j], vx[r][j + 1]}, f32x2{v0[r][j], v0[r][j + 1]},
                    g_cfg ? f32x2{v1[r][j], v1[r][j + 1]} : z, g_cls ? f32x2{vg[r][j], vg[r][j + 1]} : z, p);
                o[j] = rr[0];
                o[j + 1] = rr[1];
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                o[j] = prologue<GUIDE>(vx[r][j], v0[r][j], g_cfg ? v1[r][j] : 0.f, g_cls ? vg[r][j] : 0.f, p);
            }
            {  // LDS, not global memory: a plain 16-byte store (store4 writes through to global memory)
              u32x4 a;
#pragma unroll
              for (int j = 0; j < 4; ++j) a[j] = __float_as_uint(o[j]);
              *reinterpret_cast<u32x4*>(sx0 + i) = a;
            }
            if (track) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                top4_insert(__float_as_uint(o[j]) & ABS, m1, m2, m3, m4);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) atomicAdd(&hist[(__float_as_uint(o[j]) & ABS) >> 20], 1u);
            }
          }
        }
      }
    } else {
#pragma unroll 4
      for (int i = tid; i < n; i += T) {
        const float xev = to_f32(XE ? xe[base + i] : x[base + i]);
        const float o = prologue<GUIDE>(xev, to_f32(e0[ebase + i]), g_cfg ? to_f32(e1[ebase + i]) : 0.f,
                                        g_cls ? to_f32(g[base + i]) : 0.f, p);
        sx0[i] = o;
        const uint32_t u = __float_as_uint(o) & ABS;
        if (track)
          top4_insert(u, m1, m2, m3, m4);
        else
          atomicAdd(&hist[u >> 20], 1u);
      }
    }
    const bool has = (vec ? tid * 4 : tid) < n;  // this thread produced at least one element (launch: ThrParams.mrank)
    if (route1) {  // cluster_select_once starts from the chunk's maximum: reduced here, behind the barrier phase 1 ends with
      const uint32_t wm = wave_max_to_lane63(has ? m1 : 0u);
      if (lane == 63 && wm) atomicMax(&misc[8], wm);
    }
    __syncthreads();
    DPM_TSTAMP(1)

    // phase 2: the lo-th smallest |x0| of the whole sample.
    uint32_t prefix = 0u, known = 0u, rank = (uint32_t)tp.lo, cnt_sel = 0u;
    uint32_t hi = ABS, nc = 0u;
    bool use_cand = false, local_only = k == 1;  // local_only: no further cluster-wide step is needed
    bool hist_ready = !track;                    // the level-0 histogram of the whole chunk exists
    bool fast = false;                           // the candidates are few: finish by rank counting
    // clusters first try to settle the sample with ONE exchange (cluster_select_once); the general route below is the
    // fallback for samples whose large values sit in one chunk, and the only route when the quantile is not near 1
    uint32_t a1 = 0u, b1 = 0u;
    const bool solved = route1 && cluster_select_once<T>(sx0, n, vec, m1, m2, m3, m4, has, hist, misc, cand,
                                                         ws + THR_WS_WORDS, tp, k, c, tid, a1, b1, s_idx == grp);
    const bool general = !solved;                // the general route runs (for clusters: it dirties the merged histograms)

    if (general && topk) {
      // Top-K front end (the usual case: ratio close to 1, K = n - lo elements at or above the wanted one, K much smaller
      // than the number of threads).  The K-th largest element of the sample is at least the K-th largest of the
      // per-thread maxima (those are K distinct elements), so every element that can still matter has a top digit >=
      // the digit of that maximum: a histogram of ONE value per thread instead of one LDS atomic per element on a
      // few hot bins (|x0| of one sample sits in a handful of exponents), then the usual compaction.
      if (has) atomicAdd(&hist[m1 >> 20], 1u);
      __syncthreads();
      DPM_TSTAMP(4)
      if (k > 1) {
        uint32_t* gh = ws + THR_WS_MAXH;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
          const uint32_t v = hist[j * T + tid];
          if (v) __hip_atomic_fetch_add(&gh[j * T + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cluster_barrier(ws + THR_WS_CNT + 4, k, misc + 30, tp.fault);
#pragma unroll
        for (int j = 0; j < BPT; ++j)
          hist[j * T + tid] = __hip_atomic_load(&gh[j * T + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
      }
      locate_bin<T>(hist, misc, (uint32_t)tp.mrank, tid);
      const uint32_t bin_lo = misc[0];
      DPM_TSTAMP(5)
      {
        // The candidates of a thread are among its four largest values unless even the fourth reaches the digit (and
        // the thread has more elements): wavefronts where that happens anywhere sweep their LDS rows instead.
        const int mine = vec ? (has ? 4 * ((n - tid * 4 + T * 4 - 1) / (T * 4)) : 0) : (has ? (n - tid + T - 1) / T : 0);
        const bool c1 = mine > 0 && (m1 >> 20) >= bin_lo, c2 = mine > 1 && (m2 >> 20) >= bin_lo;
        const bool c3 = mine > 2 && (m3 >> 20) >= bin_lo, c4 = mine > 3 && (m4 >> 20) >= bin_lo;
        if (__ballot(c4 && mine > 4)) {
          (void)compact_candidates<T, true>(sx0, n, bin_lo, misc, cand, tid);
        } else {
          const uint32_t cnt = (c1 ? 1u : 0u) + (c2 ? 1u : 0u) + (c3 ? 1u : 0u) + (c4 ? 1u : 0u);
          const uint32_t incl = wave_incl_scan(cnt);
          uint32_t slot = 0u;
          if (lane == 63 && incl) slot = atomicAdd(&misc[4], incl);
          uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)slot, 63) + incl - cnt;
          if (c1 && off < (uint32_t)THR_CAP) cand[off] = m1;
          off += c1 ? 1u : 0u;
          if (c2 && off < (uint32_t)THR_CAP) cand[off] = m2;
          off += c2 ? 1u : 0u;
          if (c3 && off < (uint32_t)THR_CAP) cand[off] = m3;
          off += c3 ? 1u : 0u;
          if (c4 && off < (uint32_t)THR_CAP) cand[off] = m4;
        }
      }
      __syncthreads();
      nc = misc[4];
      bool ok = nc <= (uint32_t)THR_CAP;
      if (k > 1) {  // one list for the cluster; every workgroup then finishes on identical data by itself
        uint32_t* gl = ws + THR_WS_LIST;
        uint32_t* gcnt = ws + THR_WS_CNT + 10;
        if (tid == 0) misc[5] = __hip_atomic_fetch_add(gcnt, nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t slot0 = misc[5];
        if (ok && slot0 <= (uint32_t)THR_GCAP && nc <= (uint32_t)THR_GCAP - slot0)
          for (uint32_t i = tid; i < nc; i += T)
            __hip_atomic_store(&gl[slot0 + i], cand[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cluster_barrier(ws + THR_WS_CNT + 5, k, misc + 30, tp.fault);
        const uint32_t total = __hip_atomic_load(gcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = total <= (uint32_t)THR_GCAP;
        if (ok) {
          nc = total;
          for (uint32_t i = tid; i < nc; i += T)
            cand[i] = __hip_atomic_load(&gl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
      }
      DPM_TSTAMP(6)
      if (ok && nc >= (uint32_t)tp.topk) {
        use_cand = true;
        local_only = true;
        rank = nc - (uint32_t)tp.topk;  // ascending rank of the wanted element inside the list
        fast = nc <= (uint32_t)T;
      } else {  // plateaus: too many elements share the digit -- start over with the full histograms
        nc = 0u;
        if (tid == 0) misc[4] = 0u;
      }
      hist_ready = false;
    }

    // 11 + 11 + 9-bit radix select over the candidates, or over the whole chunk.  In the latter case the elements
    // that share the selected top digit -- the only ones levels 1, 2 and the min-above search can still care about --
    // are compacted into `cand` after level 0 (wave-aggregated append); everything above that digit only matters
    // through its minimum, kept per lane in `hi`.
#pragma unroll 1
    for (int pass = 0; pass < 3 && !fast && general; ++pass) {
      const int shift = pass == 0 ? 20 : pass == 1 ? 9 : 0;
      const uint32_t dmask = pass == 2 ? 0x1ffu : 0x7ffu;
      if (pass > 0 || !hist_ready) {  // the histogram is all zero here (sample start / locate_bin)
        if (use_cand) {
          for (uint32_t i = tid; i < nc; i += T) {
            const uint32_t u = cand[i];
            if ((u & known) == prefix) atomicAdd(&hist[(u >> shift) & dmask], 1u);
          }
        } else {
          for (int i = tid * 4; i < n; i += T * 4) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(sx0 + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t u = q[j] & ABS;
              if (i + j < n && (u & known) == prefix) atomicAdd(&hist[(u >> shift) & dmask], 1u);
            }
          }
        }
        __syncthreads();
      }
      if (!local_only) {  // merge into the sample's histogram of this level, wait for the peers, read the sum back
        uint32_t* gh = ws + pass * THR_NB;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
          const uint32_t v = hist[j * T + tid];
          if (v) __hip_atomic_fetch_add(&gh[j * T + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cluster_barrier(ws + THR_WS_CNT + pass, k, misc + 30, tp.fault);
#pragma unroll
        for (int j = 0; j < BPT; ++j)
          hist[j * T + tid] = __hip_atomic_load(&gh[j * T + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
      }
      locate_bin<T>(hist, misc, rank, tid);
      prefix |= misc[0] << shift;
      known |= dmask << shift;
      rank = misc[1];
      cnt_sel = misc[2];
#ifdef DPM_THR_DEBUG
      if (tid == 0 && c < 2) printf("[gen] c=%d pass=%d prefix=%08x rank=%u cnt_sel=%u nc=%u use_cand=%d local_only=%d\n", c, pass, prefix, rank, cnt_sel, nc, (int)use_cand, (int)local_only);
#endif
      if (pass == 0 && !use_cand) {  // compact this chunk's candidates, remember the smallest value of the higher digits
        const uint32_t bin0 = prefix >> 20;
        hi = compact_candidates<T, false>(sx0, n, bin0, misc, cand, tid);
        __syncthreads();
        nc = misc[4];
        use_cand = nc <= (uint32_t)THR_CAP;
        if (k > 1 && cnt_sel <= (uint32_t)THR_GCAP) {
          // The whole cluster's candidates fit one list: exchange them (and the minimum of the higher digits) once.
          // Every workgroup then finishes levels 1, 2 and the min-above search on identical data by itself -- two
          // cluster barriers per sample instead of four.
          uint32_t* gl = ws + THR_NB;               // the level-1 histogram's words double as the list
          uint32_t* ghi = ws + THR_WS_CNT + 8;      // complement of the smallest value above the selected digit
          uint32_t* gcnt = ws + THR_WS_CNT + 9;     // list slots handed out so far
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(hi, d, 64);
            hi = o < hi ? o : hi;
          }
          if (lane == 0) atomicMin(&misc[3], hi);
          __syncthreads();
          if (tid == 0) {
            misc[5] = __hip_atomic_fetch_add(gcnt, nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(ghi, ABS - misc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          __syncthreads();
          const uint32_t slot0 = misc[5];
          for (uint32_t i = tid; i < nc; i += T)
            __hip_atomic_store(&gl[slot0 + i], cand[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          cluster_barrier(ws + THR_WS_CNT + 1, k, misc + 30, tp.fault);
          nc = cnt_sel;
          for (uint32_t i = tid; i < nc; i += T)
            cand[i] = __hip_atomic_load(&gl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hi = ABS - __hip_atomic_load(ghi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          use_cand = true;
          local_only = true;
          __syncthreads();
        }
        fast = use_cand && local_only && nc <= (uint32_t)T;
      }
    }
    uint32_t a_bits = prefix;
    float a, b;
    if (solved) {
      a = __uint_as_float(a1);
      b = tp.hi != tp.lo ? __uint_as_float(b1) : a;
    } else if (fast) {
      // the candidates hold the wanted element at ascending position `rank`, and -- unless it is their largest -- the next
      // order statistic too; otherwise that one is the smallest value of the higher digits
      rank_select<T>(cand, nc, rank, misc, tid);
      DPM_TSTAMP(7)
      a_bits = misc[6];
      a = __uint_as_float(a_bits);
      b = a;
      if (tp.hi != tp.lo) {
        if (rank + 1u < nc) {
          b = __uint_as_float(misc[7]);
        } else {
          if (k == 1) {  // `hi` is still per lane
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
              const uint32_t o = __shfl_xor(hi, d, 64);
              hi = o < hi ? o : hi;
            }
            if (lane == 0) atomicMin(&misc[3], hi);
            __syncthreads();
            hi = misc[3];
          }
          b = __uint_as_float(hi);
        }
      }
    } else {
      a = __uint_as_float(a_bits);
      b = a;
      if (tp.hi != tp.lo && rank + 1u >= cnt_sel) {
        // the next order statistic is the smallest value above a: wavefront min, one atomic per wave
        if (tid == 0) misc[3] = ABS;
        __syncthreads();
        uint32_t m = hi;
        if (use_cand) {
          for (uint32_t i = tid; i < nc; i += T) {
            const uint32_t u = cand[i];
            if (u > a_bits && u < m) m = u;
          }
        } else {
          for (int i = tid * 4; i < n; i += T * 4) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(sx0 + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t u = q[j] & ABS;
              if (i + j < n && u > a_bits && u < m) m = u;
            }
          }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          const uint32_t o = __shfl_xor(m, d, 64);
          m = o < m ? o : m;
        }
        if (lane == 0) atomicMin(&misc[3], m);
        __syncthreads();
        if (!local_only) {  // workspace words start at zero: keep the minimum as a maximum of the complement
          uint32_t* gm = ws + THR_WS_CNT + 8;
          if (tid == 0) __hip_atomic_fetch_max(gm, ABS - misc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          cluster_barrier(ws + THR_WS_CNT + 3, k, misc + 30, tp.fault);
          b = __uint_as_float(ABS - __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        } else {
          b = __uint_as_float(misc[3]);
        }
      }
    }
    DPM_TSTAMP(2)
    // This workgroup is through with the sample's workspace.  The last of the cluster to say so puts every word it and
    // its peers dirtied back to zero (end of the sample loop): the workspace is all zero between launches, so no launch
    // has to clear it first.  The returning atomic is in flight during phase 3.
    uint32_t done_old = 0u;
    if (k > 1 && tid == 0) done_old = __hip_atomic_fetch_add(ws + THR_WS_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // torch.quantile 'linear' = ATen lerp(a, b, w)
    const float diff = b - a;
    const float q = tp.w < 0.5f ? a + tp.w * diff : b - diff * (1.f - tp.w);
    const float s = fmaxf(q, tp.max_val);  // ref :423
    // x0 / s for every element of the sample: the same division by an invariant (the guard of div_by_alpha, evaluated
    // here because s is born on the device); ref :424 divides
    const uint32_t s_bits = __float_as_uint(s), s_ex = (s_bits >> 23) & 0xffu;
    const bool s_fast = s_ex > 32u && s_ex < 222u && (s_bits & 0x7fffffu) != 0x7fffffu;
    const float inv_s = 1.f / s;

    // phase 3: clamp, scale, combine, epilogue, store
    if (vec) {
      // THR_ROWS tile rows per iteration, the loads of all issued before the first use (explicit: the write-through
      // stores are assembly the loop unroller will not duplicate)
      for (int i0 = tid * 4; i0 < n; i0 += THR_ROWS * T * 4) {
        float vx[THR_ROWS][4], vh1[THR_ROWS][4], vh2[THR_ROWS][4];
#pragma unroll
        for (int r = 0; r < THR_ROWS; ++r) {
          const int64_t gi = base + (i0 + r * T * 4 < n ? i0 + r * T * 4 : i0);  // clamped: loads are unconditional
          if (nx) load4<true>(x, gi, vx[r]);                 // last use of x and of the cached model values
          if (nh1) load4<true>(h1, gi, vh1[r]);
          if (nh2) load4<true>(h2, gi, vh2[r]);
        }
#pragma unroll
        for (int r = 0; r < THR_ROWS; ++r) {
          const int i = i0 + r * T * 4;
          if (r == 0 || i < n) {  // per lane: a later row may end before this lane
            const int64_t gi = base + i;
            float vm[4], va[4], vb[4], o[4], om[4];
            if (mask) {
              load4(mask, mfull ? gi : (int64_t)((mbase + (uint32_t)i) % mper), vm);
              load4(ba, gi, va);
              if (bb) load4(bb, gi, vb);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) om[j] = fminf(fmaxf(sx0[i + j], -s), s);  // ref :424
            if (s_fast) {
#pragma unroll
              for (int j = 0; j < 4; j += 2) {
                const f32x2 c2 = {om[j], om[j + 1]};
                const f32x2 qd = c2 * inv_s;
                const f32x2 q2 = vfma(vfma(-qd, (f32x2)(s), c2), (f32x2)(inv_s), qd);
                om[j] = q2[0];
                om[j + 1] = q2[1];
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) om[j] = om[j] / s;
            }
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              const f32x2 z = {0.f, 0.f};
              const f32x2 c2 = combine_any<FORM, f32x2>(nx ? f32x2{vx[r][j], vx[r][j + 1]} : z, f32x2{om[j], om[j + 1]},
                                                    nh1 ? f32x2{vh1[r][j], vh1[r][j + 1]} : z,
                                                    nh2 ? f32x2{vh2[r][j], vh2[r][j + 1]} : z, p);
              o[j] = c2[0];
              o[j + 1] = c2[1];
            }
            if (mask) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                o[j] = blend_ref(to_f32(from_f32<TS>(o[j])), vm[j], va[j], bb ? vb[j] : 0.f, bb != nullptr, ext);
            }
            store4(xo, gi, o);
            if (xo2) store4(xo2, gi, o);
            if (store_m) store4<true>(mo, gi, om);                  // read again only after the next network call
          }
        }
      }
    } else {
      for (int i = tid; i < n; i += T) {
        const int64_t gi = base + i;
        const float mn = fminf(fmaxf(sx0[i], -s), s) / s;  // ref :424
        const float xv = nx ? to_f32(x[gi]) : 0.f;
        float o = combine_any<FORM>(xv, mn, nh1 ? to_f32(h1[gi]) : 0.f, nh2 ? to_f32(h2[gi]) : 0.f, p);
        if (mask)
          o = blend_ref(to_f32(from_f32<TS>(o)), to_f32(mask[mfull ? gi : (int64_t)((mbase + (uint32_t)i) % mper)]),
                        to_f32(ba[gi]), bb ? to_f32(bb[gi]) : 0.f, bb != nullptr, ext);
        const TS ov = from_f32<TS>(o);
        xo[gi] = ov;
        if (xo2) xo2[gi] = ov;
        if (store_m) mo[gi] = from_f32<TS>(mn);
      }
    }
    if (k > 1) {
      if (tid == 0) misc[11] = done_old == k - 1u ? 1u : 0u;
      __syncthreads();
      if (misc[11]) {
        uint32_t* slots = ws + THR_WS_WORDS;
        for (uint32_t i = tid; i < k * (uint32_t)THR_SLOTW; i += T) slots[i] = 0u;
        if (general || !route1) {
          for (uint32_t i = tid; i < (uint32_t)THR_WS_WORDS; i += T) ws[i] = 0u;
        } else if (tid == 0) {
          ws[THR_WS_DONE] = 0u;
        }
      }
    }
    __syncthreads();  // the next sample of this cluster reuses the LDS
#ifdef DPM_THR_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stamp means: stores issued AND drained
#endif
    DPM_TSTAMP(3)
  }
}

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

// ------------------------------------------------------------------------------------------------
// launch plumbing
// ------------------------------------------------------------------------------------------------
struct DeviceInfo {
  int n_cu = 0;
  int lds = 0;
  char arch[64] = {0};
  bool ok = false;
};

inline const DeviceInfo& device_info() {
  static thread_local int cached_dev = -1;
  static thread_local DeviceInfo info;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return info;
  if (dev != cached_dev || !info.ok) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      info.n_cu = prop.multiProcessorCount;
      info.lds = (int)prop.maxSharedMemoryPerMultiProcessor;
      std::strncpy(info.arch, prop.gcnArchName, sizeof(info.arch) - 1);
      info.ok = true;
      cached_dev = dev;
    }
  }
  return info;
}

inline bool aligned(const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// the division-by-invariant of the specialised prologue is exact unless alpha's significand is all ones (or alpha is
// not a normal number): then the generic prologue, with a true division, runs instead
inline bool div_invariant_ok(float alpha) {
  uint32_t u;
  std::memcpy(&u, &alpha, 4);
  const uint32_t ex = (u >> 23) & 0xffu;
  return ex != 0u && ex != 0xffu && (u & 0x7fffffu) != 0x7fffffu && ex > 32u && ex < 222u;
}

inline KParams make_params(const dpm_stage* st) {
  KParams p;
  p.alpha_e = st->alpha_e;
  p.inv_alpha = 1.0f / st->alpha_e;
  p.sigma_e = st->sigma_e;
  p.cfg_scale = st->cfg_scale;
  p.cg_scale = st->cg_scale;
  p.cx = st->cx;
  p.c0 = st->c0;
  p.c1 = st->c1;
  p.c2 = st->c2;
  p.k0 = st->k[0];
  p.k1 = st->k[1];
  p.k2 = st->k[2];
  p.k3 = st->k[3];
  p.k4 = st->k[4];
  p.flags = st->flags;
  p.model_type = st->model_type;
  p.form = st->form;
  p.guidance = st->guidance;
  p.inv_sigma = 1.0f / st->sigma_e;
  p.fastdiv = (div_invariant_ok(st->alpha_e) ? 1u : 0u) | (div_invariant_ok(st->sigma_e) ? 2u : 0u);
  return p;
}

// cluster shape of the thresholding kernel: k workgroups per sample, `chunk` elements each.  Depends only on the
// batch, the sample size and the CU count, so dpm_threshold_workspace_bytes() and the launch agree.
struct ThrPlan {
  int64_t k, chunk;
};
inline ThrPlan thr_plan(int64_t batch, int64_t per_sample, int n_cu) {
  const int64_t kmin = (per_sample + THR_CHUNK_MAX - 1) / THR_CHUNK_MAX;        // what LDS allows
  const int64_t kfill = (2 * (int64_t)n_cu) / (batch < 1 ? 1 : batch);          // spread a small batch over the chip
  const int64_t kmax = std::max<int64_t>(1, per_sample / 2048);                 // but keep >= 2 elements per lane
  int64_t k = std::max(kmin, std::min(std::min(kfill, kmax), (int64_t)n_cu));
  if (k < 1) k = 1;
  int64_t chunk = (per_sample + k - 1) / k;
  chunk = (chunk + 3) / 4 * 4;
  return ThrPlan{k, chunk};
}
// words per sample: the merged histograms / lists / counters of the general route + one slot per workgroup of the cluster
inline int64_t thr_ws_stride(int64_t k) { return (int64_t)THR_WS_WORDS + k * (int64_t)THR_SLOTW; }
inline int64_t thr_ws_bytes(int64_t batch, int64_t per_sample, int n_cu) {
  const ThrPlan pl = thr_plan(batch, per_sample, n_cu);
  return pl.k > 1 ? batch * thr_ws_stride(pl.k) * 4 : 0;
}

// launch-shape defaults (measured on MI355X: profiles/r01_tuning.md, r01_tuning_v3.txt, and r01_tuning_v4.txt with the
// write-through stores) and the run-time tuning hooks.  nt mask: bit 0 = nt loads; bits 1, 2 = nt x_out / m_out store,
// which only matter in a -DDPM_STORE_WRITE_THROUGH=0 build.  Two situations, two optima:
//   * a network ran since the inputs were written (every real sampling loop): the streams come from HBM and streaming
//     (nt) loads win -- [256,4,64,64] HBM-cold: fp16 8.4 vs 9.3 us, fp32 15.3 vs 16.3-16.5 us against the default cache
//     policy.  This is the default (DefNT = 5).
//   * the previous launch wrote the inputs (dpm_buffers.inputs_resident: frozen-model loops such as dpm_plan_run
//     without a model callback): they sit in the Infinity Cache and the default policy wins, with two tiles per
//     workgroup iteration when there is work for it -- fp16 5.5 vs 7.4-7.6 us, fp32 12.35 vs 12.9 us.  Variants exist for
//     the 2M / first-order kernels (HotCombo).
constexpr int DEF_U = 1;
template <typename TS>
struct DefNT {
  static constexpr int value = 5;
};

template <int FORM, int GUIDE, bool XE>
struct HotCombo {
  static constexpr bool value = (FORM == DPM_FORM_TWO || FORM == DPM_FORM_LIN1) && GUIDE == DPM_GUIDE_NONE && !XE;
};

struct LaunchCtx {
  hipStream_t stream;
  hipEvent_t start, stop;  // both null: plain launch; else hipExtLaunchKernelGGL brackets the kernel itself
  // device-resident coefficients (the adaptive solver's on-device controller, dpm_kernels.hip): the float fields of
  // the stage record are read from `dyn` (device memory) by the kernel instead of from its arguments, and the launch
  // is a no-op when *skip != 0.  Honoured by the general-prologue kernels only.
  const dpm_stage* dyn = nullptr;
  const int32_t* skip = nullptr;
};

template <typename K, typename... Args>
void launch(K kern, dim3 grid, dim3 block, size_t lds, const LaunchCtx& c, Args... args) {
  if (c.start || c.stop)
    hipExtLaunchKernelGGL(kern, grid, block, lds, c.stream, c.start, c.stop, 0, args...);
  else
    hipLaunchKernelGGL(kern, grid, block, lds, c.stream, args...);
}

template <typename TS, typename TE, int FORM, int GUIDE, bool XE>
int launch_typed(const dpm_stage* st, const dpm_buffers* b, const LaunchCtx& stream) {
  const KParams p = make_params(st);
  const TS* x = static_cast<const TS*>(b->x);
  const TS* xe = static_cast<const TS*>(b->xe);
  const TE* e0 = static_cast<const TE*>(b->e0);
  const TE* e1 = static_cast<const TE*>(b->e1);
  const TE* g = static_cast<const TE*>(b->g);
  const TS* h1 = static_cast<const TS*>(b->h1);
  const TS* h2 = static_cast<const TS*>(b->h2);
  TS* xo = static_cast<TS*>(b->x_out);
  TS* mo = static_cast<TS*>(b->m_out);
  const DeviceInfo& di = device_info();
  const int n_cu = di.n_cu > 0 ? di.n_cu : 256;
  KExt ext;
  std::memset(&ext, 0, sizeof ext);
  const bool blend = (st->flags & DPM_F_BLEND) != 0;
  ext.xo2 = b->x_out2;
  ext.mask = blend ? b->mask : nullptr;
  ext.ba = blend ? b->blend_a : nullptr;
  ext.bb = blend ? b->blend_b : nullptr;
  ext.mask_period = blend ? b->mask_period : 0;
  ext.per_sample = b->n / b->batch;
  ext.eps_stride = (b->eps_stride == ext.per_sample) ? 0 : b->eps_stride;
  ext.blend_alpha = st->blend_alpha;
  ext.blend_sigma = st->blend_sigma;
  const bool use_ext = ext.xo2 || ext.mask || ext.eps_stride;

  if (st->flags & DPM_F_THRESH) {
    if (stream.dyn) return dpm_set_error(DPM_ERR_UNSUPPORTED, "dynamic thresholding with device-resident coefficients");
    const int64_t per_sample = b->n / b->batch;
    if (b->batch > 0x7fffffff || per_sample > ((int64_t)1 << 40))
      return dpm_set_error(DPM_ERR_UNSUPPORTED, "thresholding: batch / sample size out of range");
    ThrPlan pl = thr_plan(b->batch, per_sample, n_cu);
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream.stream, &cap_status);
    const bool capturing = cap_status != hipStreamCaptureStatusNone;
    // Clusters wait for each other inside the kernel, which is only safe while no OTHER clustered launch can hold part of
    // the chip at the same time.  Eager launches of this process are chained device-wide (below); a captured graph is
    // replayed outside that chain, possibly next to another graph on another stream.  Under capture a sample that fits
    // one workgroup's LDS therefore takes the cluster-free shape (one workgroup per sample) unless the caller opts in
    // (DPM_TUNE_CLUSTER_IN_GRAPH); larger samples have no such shape and keep their clusters, with bounded waits.
    if (capturing && pl.k > 1 && per_sample <= THR_CHUNK_MAX && !g_tuning.cluster_in_graph) {
      pl.k = 1;
      pl.chunk = (per_sample + 3) / 4 * 4;
    }
    ThrParams tp;
    std::memset(&tp, 0, sizeof tp);
    tp.per_sample = per_sample;
    // torch.quantile: rank = q * (n - 1) evaluated in fp32 (q is an fp32 tensor)
    const float rank = st->thr_ratio * (float)(per_sample - 1);
    tp.lo = (int32_t)floorf(rank);
    tp.hi = (int32_t)ceilf(rank);
    tp.w = rank - (float)tp.lo;
    tp.max_val = st->thr_max;
    tp.chunk = (int32_t)pl.chunk;
    tp.k = (int32_t)pl.k;
    tp.batch = (int32_t)b->batch;
    tp.fastdiv = st->model_type == DPM_MODEL_NOISE && (st->flags & DPM_F_TO_X0) && div_invariant_ok(st->alpha_e);
    const size_t a4s = sizeof(TS) * 4, a4e = sizeof(TE) * 4;
    tp.vec = per_sample % 4 == 0 && ext.eps_stride % 4 == 0 && ext.mask_period % 4 == 0 && aligned(x, a4s) &&
             aligned(xe, a4s) && aligned(h1, a4s) && aligned(h2, a4s) && aligned(xo, a4s) && aligned(mo, a4s) &&
             aligned(ext.xo2, a4s) && aligned(ext.mask, a4s) && aligned(ext.ba, a4s) && aligned(ext.bb, a4s) &&
             aligned(e0, a4e) && aligned(e1, a4e) && aligned(g, a4e);
    {
      // top-K front end: a = the K-th largest element.  It needs at most one wanted element per contributing thread and
      // pays when the K-th largest per-thread maximum sits in the sparse upper tail (K a small part of the threads) and
      // the candidates (a small multiple of K) fit the rank-counting finish (<= THR_THREADS of them).
      const int64_t K = per_sample - (int64_t)tp.lo;
      int64_t P = 0;
      for (int64_t c = 0; c < pl.k; ++c) {
        const int64_t n_c = std::max<int64_t>(0, std::min<int64_t>(pl.chunk, per_sample - c * pl.chunk));
        P += std::min<int64_t>(THR_THREADS, tp.vec ? (n_c + 3) / 4 : n_c);
      }
      if (K >= 1 && K <= P / 4 && K <= THR_THREADS / 4) {  // beyond: the candidates outgrow the rank-counting finish
        tp.topk = (int32_t)K;
        tp.mrank = (int32_t)(P - K);
      }
      // single-exchange cluster route (cluster_select_once): a chunk's share of the K largest is ~ K/k; publishing the
      // ~quota = K/k + 6 sigma + 8 largest values of every chunk makes the one-hop answer exact except for samples whose
      // large values sit in one chunk (those fall back inside the kernel).  Needs room in the slots for the 14-bit digit's
      // granularity (x1.5) and a union that fits the LDS list.
      if (pl.k > 1 && pl.k <= THR_KMAX && K >= 1 && K < ((int64_t)1 << 30)) {
        const double mu = (double)K / (double)pl.k;
        const int64_t quota = (int64_t)std::ceil(mu + 6.0 * std::sqrt(mu) + 8.0);
        // slot size: the smallest power of two >= 64 with room for the quota and the digit granularity (fewer words to
        // fetch per slot); at most THR_SLOT_CAP and THR_CAP / k
        int slot_shift = 6;
        while (((int64_t)1 << slot_shift) < quota * 3 / 2 && slot_shift < 8) ++slot_shift;
        while (slot_shift > 0 && ((int64_t)1 << slot_shift) > std::min<int64_t>(THR_SLOT_CAP, THR_CAP / pl.k)) --slot_shift;
        const int64_t slot_cap = (int64_t)1 << slot_shift;
        if (quota * 3 / 2 <= slot_cap && g_tuning.cluster_one_hop) {
          tp.quota = (int32_t)quota;
          tp.kbig = (int32_t)K;
          tp.slot_cap = (int32_t)slot_cap;
          tp.slot_pub = (int32_t)std::min<int64_t>(slot_cap, quota + quota / 4 + 4);
          tp.slot_shift = slot_shift;
          tp.debug_reject = g_tuning.cluster_one_hop == 2;
        }
      }
    }
#ifdef DPM_THR_TIMING
    // debug build only: the DPM_THR_TIMING_LAUNCH-th thresholding launch of the process (default 40) is synchronised
    // and its stamps are written to $DPM_THR_TIMING_FILE, one line of 16 values per workgroup
    static uint64_t* t_dev = nullptr;
    static int t_launches = 0;
    if (!t_dev) (void)hipMalloc(&t_dev, 4096 * 16 * sizeof(uint64_t));
    tp.tdbg = t_dev;
    auto t_dump = [&](int64_t wgs) {
      const char* path = getenv("DPM_THR_TIMING_FILE");
      const char* at = getenv("DPM_THR_TIMING_LAUNCH");
      if (!path || ++t_launches != (at ? atoi(at) : 40) || wgs > 4096) return;
      (void)hipStreamSynchronize(stream.stream);
      std::vector<uint64_t> h((size_t)wgs * 16);
      (void)hipMemcpy(h.data(), t_dev, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
      if (FILE* f = fopen(path, "w")) {
        for (int64_t i = 0; i < wgs; ++i) {
          for (int j = 0; j < 16; ++j) fprintf(f, "%llu ", (unsigned long long)h[(size_t)i * 16 + j]);
          fprintf(f, "\n");
        }
        fclose(f);
      }
    };
#else
    auto t_dump = [](int64_t) {};
#endif
    const size_t lds_bytes = (size_t)pl.chunk * 4 + THR_NB * 4 + THR_MISC * 4 + (THR_CAP + 32) * 4;
    // the compile-time specialisation exists for the forms / guidance kinds samplers combine with thresholding
    constexpr bool HOT_BUILT = (FORM == DPM_FORM_LIN1 || FORM == DPM_FORM_TWO || FORM == DPM_FORM_MS3) &&
                               (GUIDE == DPM_GUIDE_NONE || GUIDE == DPM_GUIDE_CFG) && !XE;
    const bool hot = HOT_BUILT && tp.vec && tp.fastdiv && !ext.mask;
    // the general kernel reads form / guidance from the stage record and always takes the evaluation state through xe
    using ThrKernel = decltype(&stage_thresh_kernel<TS, TE, FORM_RT, GUIDE_RT, true, THR_THREADS, 0>);
    auto kern = reinterpret_cast<ThrKernel>(const_cast<void*>(dpm_catchall_thresh<TS, TE>()));
    if constexpr (HOT_BUILT) {
      if (hot)
        kern = (tp.topk > 0 || tp.quota > 0) ? stage_thresh_kernel<TS, TE, FORM, GUIDE, XE, THR_THREADS, 1>
                                            : stage_thresh_kernel<TS, TE, FORM, GUIDE, XE, THR_THREADS, 2>;
    }
    if (!xe) xe = x;
    int64_t grid = b->batch;
    tp.groups = (int32_t)b->batch;
    if (pl.k > 1) {
      // clusters synchronise through spin barriers: every workgroup of the grid must be resident at once
      static thread_local int occ_dev = -1, occ = 0;
      static thread_local size_t occ_lds = 0;
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (dev != occ_dev || lds_bytes != occ_lds) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), THR_THREADS,
                                                                    lds_bytes);
        if (e != hipSuccess) return dpm_set_error((int)e, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
        occ_dev = dev;
        occ_lds = lds_bytes;
        occ = nb;
      }
      const int64_t cap = (int64_t)n_cu * (occ < 1 ? 1 : (occ > 2 ? 2 : occ));
      if (pl.k > cap)
        return dpm_set_error(DPM_ERR_UNSUPPORTED, "dynamic thresholding: a sample of %lld elements needs %lld co-resident "
                             "workgroups, the device holds %lld", (long long)per_sample, (long long)pl.k, (long long)cap);
      if (!b->workspace)
        return dpm_set_error(DPM_ERR_ARG,
                             "dynamic thresholding of %lld samples x %lld elements needs a workspace of "
                             "dpm_threshold_workspace_bytes() = %lld bytes",
                             (long long)b->batch, (long long)per_sample, (long long)thr_ws_bytes(b->batch, per_sample, n_cu));
      const int64_t groups = std::min<int64_t>(b->batch, cap / pl.k);
      tp.groups = (int32_t)groups;
      tp.ws = static_cast<uint32_t*>(b->workspace);
      tp.ws_stride = thr_ws_stride(pl.k);
      grid = groups * pl.k;
      // No clearing of the workspace here: the caller hands it over zero-filled once, the kernel leaves it zero-filled
      // (dpm_threshold_workspace_bytes).  A wait that timed out in an earlier clustered launch is reported now.
      uint32_t* fault = cluster_fault_word(!capturing);
      if (fault && *fault) {
        *fault = 0u;
        return dpm_set_error(DPM_ERR_FAULT, "a clustered dynamic-thresholding launch gave up waiting for a peer workgroup "
                             "(another clustered launch held the GPU concurrently?); its results are invalid and its "
                             "workspace must be zero-filled again");
      }
      tp.fault = fault;
      // Two clustered launches on different streams could each hold part of the CUs with spinning workgroups and
      // starve the other's missing peers.  Within this process they are therefore chained device-wide: wait for the
      // previous clustered launch (whatever its stream), record after this one.  (Not under stream capture, where an
      // event recorded outside the capture cannot be waited on; see above.)
      if (!capturing) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        ClusterChain& ch = cluster_chain(dev);
        std::lock_guard<std::mutex> lk(ch.mu);
        if (!ch.ev && hipEventCreateWithFlags(&ch.ev, hipEventDisableTiming) != hipSuccess) ch.ev = nullptr;
        if (ch.ev && ch.recorded) (void)hipStreamWaitEvent(stream.stream, ch.ev, 0);
        launch(kern, dim3((unsigned)grid), dim3(THR_THREADS), lds_bytes, stream, x, xe, e0, e1, g, h1, h2, xo, mo, p, tp, ext);
        if (ch.ev && hipEventRecord(ch.ev, stream.stream) == hipSuccess) ch.recorded = true;
        t_dump(grid);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return dpm_set_error((int)e, "stage kernel launch failed: %s", hipGetErrorString(e));
        return DPM_OK;
      }
    }
    launch(kern, dim3((unsigned)grid), dim3(THR_THREADS), lds_bytes, stream, x, xe, e0, e1, g, h1, h2, xo, mo, p, tp, ext);
    t_dump(grid);
  } else {
    const size_t as = sizeof(TS) * EPT, ae = sizeof(TE) * EPT;
    bool vec = aligned(x, as) && aligned(xe, as) && aligned(h1, as) && aligned(h2, as) && aligned(xo, as) &&
               aligned(mo, as) && aligned(e0, ae) && aligned(e1, ae) && aligned(g, ae);
    if (use_ext)  // the extended vector kernel has no ragged tail and indexes whole 8-element groups
      vec = vec && aligned(ext.xo2, as) && aligned(ext.mask, as) && aligned(ext.ba, as) && aligned(ext.bb, as) &&
            b->n % EPT == 0 && ext.mask_period % EPT == 0 &&
            (!ext.eps_stride || (ext.per_sample % EPT == 0 && ext.eps_stride % EPT == 0));
    // what the streaming family instantiates (binary size: one kernel per combination and dtype pair):
    //   * a separate evaluation state (xe != x) only occurs in the singlestep mid / final stages: forms TWO and SS3T;
    //   * the compile-time prologues (noise-prediction network) for the forms samplers spend their time in -- LIN1, TWO,
    //     MS3; SS3T and DENOISE run the general prologue (true division: the same bits);
    //   everything else goes through the one-element-per-lane kernel.
    constexpr bool COMBO_BUILT = !XE || FORM == DPM_FORM_TWO || FORM == DPM_FORM_SS3T;
    constexpr bool SPEC_BUILT = FORM == DPM_FORM_LIN1 || FORM == DPM_FORM_TWO || FORM == DPM_FORM_MS3;
    // device-resident coefficients (adaptive solver): DYN kernels exist for the forms it launches -- first-order,
    // second-order and the singlestep-3 'taylor' combination -- without the KExt extensions; anything else takes the
    // one-element-per-lane kernel
    constexpr bool DYN_BUILT = COMBO_BUILT && (FORM == DPM_FORM_LIN1 || FORM == DPM_FORM_TWO || FORM == DPM_FORM_SS3T);
    const bool dyn_vec = stream.dyn && DYN_BUILT && !use_ext;
    if (!vec || !COMBO_BUILT || (stream.dyn && !dyn_vec)) {
      int64_t blocks = (b->n + 255) / 256;
      const int64_t cap = (int64_t)n_cu * 16;
      if (blocks > cap) blocks = cap;
      using ScalarKernel = decltype(&stage_kernel_scalar<TS, TE, false>);  // the DYN = true variant has the same signature
      const void* k = stream.dyn ? dpm_catchall_scalar<TS, TE, true>() : dpm_catchall_scalar<TS, TE, false>();
      launch(reinterpret_cast<ScalarKernel>(const_cast<void*>(k)), dim3((unsigned)blocks), dim3(256), 0, stream, x, xe ? xe : x,
             e0, e1, g, h1, h2, xo, mo, b->n, p, ext, stream.dyn, stream.skip);
    } else if constexpr (COMBO_BUILT) {
      const bool noise = SPEC_BUILT && !stream.dyn && st->model_type == DPM_MODEL_NOISE &&
                         (!(st->flags & DPM_F_TO_X0) || div_invariant_ok(st->alpha_e));
      const int spec = !noise ? SPEC_GENERIC : ((st->flags & DPM_F_TO_X0) ? SPEC_NOISE_X0 : SPEC_NOISE_EPS);
      const int64_t ntiles = ((b->n / EPT) + 255) / 256;
      const Tuning tn = g_tuning;
      const bool big = ntiles >= 4 * (int64_t)n_cu;  // two tiles per iteration only when there is work for it
      auto grid_for = [&](int u) {
        int64_t blocks = (ntiles + u - 1) / u;
        const int64_t cap = (int64_t)n_cu * tn.blocks_per_cu;
        if (blocks > cap) blocks = cap;
        return dim3((unsigned)(blocks < 1 ? 1 : blocks));
      };
#define DPM_LAUNCH(SPEC_, U_, NT_, EXT_)                                                                             \
  launch(stage_kernel<TS, TE, FORM, GUIDE, XE, SPEC_, U_, NT_, EXT_>, grid_for(U_), dim3(256), 0, stream, x, xe, e0, e1, \
         g, h1, h2, xo, mo, b->n, p, ext, stream.dyn, stream.skip)
      if (dyn_vec) {
        if constexpr (DYN_BUILT)
          launch(stage_kernel<TS, TE, FORM, GUIDE, XE, SPEC_GENERIC, 1, DefNT<TS>::value, false, true>, grid_for(1), dim3(256),
                 0, stream, x, xe, e0, e1, g, h1, h2, xo, mo, b->n, p, ext, stream.dyn, stream.skip);
      } else if (use_ext) {
        // (tiles per iteration, nt mask) of the inputs-from-HBM table below; x_out stays cacheable (it is the next
        // network input), so bit 1 is never set
        constexpr int EU = (sizeof(TS) == 4 && sizeof(TE) == 2) ? 2 : 1;
        constexpr int ENT = sizeof(TS) == 2 ? 1 : (sizeof(TE) == 4 ? 5 : 1);
        const bool two = EU == 2 && big && SPEC_BUILT;
        if (spec == SPEC_GENERIC) {
          DPM_LAUNCH(SPEC_GENERIC, 1, ENT, true);
        } else if constexpr (SPEC_BUILT) {
          if (spec == SPEC_NOISE_X0) {
            if (two) DPM_LAUNCH(SPEC_NOISE_X0, EU, ENT, true); else DPM_LAUNCH(SPEC_NOISE_X0, 1, ENT, true);
          } else {
            if (two) DPM_LAUNCH(SPEC_NOISE_EPS, EU, ENT, true); else DPM_LAUNCH(SPEC_NOISE_EPS, 1, ENT, true);
          }
        }
      } else if (spec == SPEC_GENERIC) {
        DPM_LAUNCH(SPEC_GENERIC, 1, DefNT<TS>::value, false);
      } else if constexpr (SPEC_BUILT) {
        if (spec == SPEC_NOISE_EPS) {
          DPM_LAUNCH(SPEC_NOISE_EPS, DEF_U, DefNT<TS>::value, false);
        } else if constexpr (HotCombo<FORM, GUIDE, XE>::value) {
          // the north-star kernels (2M / 1st-order update, no guidance): (tiles per iteration, nt mask) by situation and
          // dtypes, from profiles/r01_tuning_v3.txt / r01_tuning_v4.txt:
          //   inputs cache-resident: default policy, two tiles per iteration when there is work for it
          //   inputs from HBM:       2-byte state (1, nt loads); fp32 + fp32 (1 | 2, nt loads + nt m store);
          //                          fp32 state + 2-byte network output (1 | 2, nt loads)  [SD under autocast]
          const bool resident = b->inputs_resident != 0 || tn.assume_resident != 0;
          constexpr int CNT = sizeof(TS) == 2 ? 1 : (sizeof(TE) == 4 ? 5 : 1);  // nt mask of the HBM situation
#ifdef DPM_TUNING_VARIANTS  // tools/tune.py single: every (tiles per iteration, nt mask)
          if (tn.unroll > 0 && tn.nontemporal >= 0) {
            switch (tn.unroll * 8 + (tn.nontemporal & 7)) {
              case 8 + 0: DPM_LAUNCH(SPEC_NOISE_X0, 1, 0, false); break;
              case 8 + 1: DPM_LAUNCH(SPEC_NOISE_X0, 1, 1, false); break;
              case 8 + 5: DPM_LAUNCH(SPEC_NOISE_X0, 1, 5, false); break;
              case 16 + 0: DPM_LAUNCH(SPEC_NOISE_X0, 2, 0, false); break;
              case 16 + 1: DPM_LAUNCH(SPEC_NOISE_X0, 2, 1, false); break;
              case 16 + 5: DPM_LAUNCH(SPEC_NOISE_X0, 2, 5, false); break;
              default: DPM_LAUNCH(SPEC_NOISE_X0, DEF_U, DefNT<TS>::value, false); break;
            }
          } else
#endif
          if (resident) {
            if (big) DPM_LAUNCH(SPEC_NOISE_X0, 2, 0, false); else DPM_LAUNCH(SPEC_NOISE_X0, 1, 0, false);
          } else if (sizeof(TS) == 2 || !big) {
            DPM_LAUNCH(SPEC_NOISE_X0, 1, CNT, false);
          } else {
            DPM_LAUNCH(SPEC_NOISE_X0, 2, CNT, false);
          }
        } else {
          DPM_LAUNCH(SPEC_NOISE_X0, DEF_U, DefNT<TS>::value, false);
        }
      }
#undef DPM_LAUNCH
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "stage kernel launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

// ---- fused multi-request launch (stage_kernel_multi).  Returns DPM_ERR_UNSUPPORTED *without* setting an error text
// when this (form, guidance, prologue) has no fused variant: the caller then launches the requests one by one.
constexpr int MULTI_NOT_BUILT = -1000;
// Launch shape of the fused kernel (profiles/r02_tune_multi.txt, 32 x [256,4,64,64], kernel-only per request-stage):
// the inputs of a fused launch always come from HBM (R x 42 MB of other requests' traffic passed since they were
// written) -> streaming loads; one super-tile per workgroup -- a grid of all R x tiles workgroups, no grid-stride loop:
// fp16 7.95 us with the grid capped at 8 workgroups per CU, 7.7 / 7.5 at 16 / 32 per CU, 6.93 uncapped (0.757 of the
// HBM peak; fp32 15.4 -> 13.9, fp32 state + fp16 output 13.3 -> 12.6); two tiles per workgroup for 4-byte states.
template <typename TS, typename TE>
struct MultiShape {
  static constexpr int U = (sizeof(TS) == 4) ? 2 : 1;
  static constexpr int NT = 1;
};

template <typename TS, typename TE, int FORM, int GUIDE, int SPEC>
int launch_multi_spec(const dpm_stage* st, const dpm_buffers* bs, int n_req, const LaunchCtx& c) {
  const DeviceInfo& di = device_info();
  const int n_cu = di.n_cu > 0 ? di.n_cu : 256;
  const Tuning tn = g_tuning;
  MultiTab tab;
  std::memset(&tab, 0, sizeof tab);
  for (int r = 0; r < n_req; ++r) {
    tab.x[r] = bs[r].x ? bs[r].x : bs[r].xe;
    tab.e0[r] = bs[r].e0;
    tab.e1[r] = bs[r].e1;
    tab.h1[r] = bs[r].h1;
    tab.h2[r] = bs[r].h2;
    tab.xo[r] = bs[r].x_out;
    tab.mo[r] = bs[r].m_out;
  }
  const KParams p = make_params(st);
  const int64_t n = bs[0].n;
  const int64_t ntiles = ((n / EPT) + 255) / 256;
  auto go = [&](auto kern, int u) {
    const int64_t spr = (ntiles + u - 1) / u;
    int64_t blocks = spr * n_req;
    const bool remap = tn.multi_xcd_remap < 0 ? sizeof(TS) == 2 : tn.multi_xcd_remap != 0;
    const uint32_t span = remap ? (uint32_t)((blocks + 7) / 8) : 0u;
    if (span) blocks = (int64_t)span * 8;
    if (tn.multi_blocks_per_cu > 0) {  // tuning hook: cap the grid, workgroups loop over the super-tiles
      const int64_t cap = (int64_t)n_cu * tn.multi_blocks_per_cu;
      if (blocks > cap) blocks = cap;
    }
    launch(kern, dim3((unsigned)blocks), dim3(256), 0, c, tab, n, (uint32_t)n_req, (uint32_t)spr, p, span);
  };
  constexpr int DU = MultiShape<TS, TE>::U, DN = MultiShape<TS, TE>::NT;
#ifdef DPM_TUNING_VARIANTS  // tools/tune.py multi: every (tiles per iteration, nt mask) of the 2M kernel
  if constexpr (FORM == DPM_FORM_TWO && GUIDE == DPM_GUIDE_NONE && SPEC == SPEC_NOISE_X0) {
    if (tn.unroll > 0 && tn.nontemporal >= 0) {
      switch (tn.unroll * 8 + (tn.nontemporal & 7)) {
        case 8 + 0: go(stage_kernel_multi<TS, TE, FORM, GUIDE, SPEC, 1, 0>, 1); break;
        case 8 + 1: go(stage_kernel_multi<TS, TE, FORM, GUIDE, SPEC, 1, 1>, 1); break;
        case 8 + 5: go(stage_kernel_multi<TS, TE, FORM, GUIDE, SPEC, 1, 5>, 1); break;
        case 16 + 0: go(stage_kernel_multi<TS, TE, FORM, GUIDE, SPEC, 2, 0>, 2); break;
        case 16 + 1: go(stage_kernel_multi<TS, TE, FORM, GUIDE, SPEC, 2, 1>, 2); break;
        case 16 + 5: go(stage_kernel_multi<TS, TE, FORM, GUIDE, SPEC, 2, 5>, 2); break;
        default: go(stage_kernel_multi<TS, TE, FORM, GUIDE, SPEC, DU, DN>, DU); break;
      }
      hipError_t e2 = hipGetLastError();
      if (e2 != hipSuccess) return dpm_set_error((int)e2, "fused stage kernel launch failed: %s", hipGetErrorString(e2));
      return DPM_OK;
    }
  }
#endif
  go(stage_kernel_multi<TS, TE, FORM, GUIDE, SPEC, DU, DN>, DU);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "fused stage kernel launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

// every request of the group: same stage, n, batch, dtypes (checked by the caller); here: is there a fused variant, and
// do all buffers allow 16-byte accesses?
template <typename TS, typename TE>
int launch_multi_typed(const dpm_stage* st, const dpm_buffers* bs, int n_req, const LaunchCtx& c) {
  if (st->flags & (DPM_F_THRESH | DPM_F_BLEND)) return MULTI_NOT_BUILT;
  if (st->guidance == DPM_GUIDE_CLASSIFIER) return MULTI_NOT_BUILT;
  if (st->form != DPM_FORM_LIN1 && st->form != DPM_FORM_TWO && st->form != DPM_FORM_MS3) return MULTI_NOT_BUILT;
  const size_t as = sizeof(TS) * EPT, ae = sizeof(TE) * EPT;
  if (bs[0].n % EPT != 0) return MULTI_NOT_BUILT;
  for (int r = 0; r < n_req; ++r) {
    const dpm_buffers& b = bs[r];
    if (b.x_out2 || (b.eps_stride && b.eps_stride != b.n / b.batch)) return MULTI_NOT_BUILT;
    if (b.xe && b.x && b.xe != b.x) return MULTI_NOT_BUILT;
    if (!(aligned(b.x, as) && aligned(b.xe, as) && aligned(b.h1, as) && aligned(b.h2, as) && aligned(b.x_out, as) &&
          aligned(b.m_out, as) && aligned(b.e0, ae) && aligned(b.e1, ae)))
      return MULTI_NOT_BUILT;
  }
  const bool x0 = (st->flags & DPM_F_TO_X0) != 0;
  const bool cfg = st->guidance == DPM_GUIDE_CFG;
  // x_start / v / score networks (and an alpha the division-by-invariant guard rejects) take the general prologue
  const bool generic = st->model_type != DPM_MODEL_NOISE || (x0 && !div_invariant_ok(st->alpha_e));
#define DPM_MULTI(FORM_)                                                                                        \
  (generic ? (cfg ? launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_CFG, SPEC_GENERIC>(st, bs, n_req, c)             \
                  : launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_NONE, SPEC_GENERIC>(st, bs, n_req, c))           \
   : cfg   ? (x0 ? launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_CFG, SPEC_NOISE_X0>(st, bs, n_req, c)             \
                 : launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_CFG, SPEC_NOISE_EPS>(st, bs, n_req, c))           \
           : (x0 ? launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_NONE, SPEC_NOISE_X0>(st, bs, n_req, c)            \
                 : launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_NONE, SPEC_NOISE_EPS>(st, bs, n_req, c)))
  switch (st->form) {
    case DPM_FORM_LIN1: return DPM_MULTI(DPM_FORM_LIN1);
    case DPM_FORM_TWO: return DPM_MULTI(DPM_FORM_TWO);
    default: return DPM_MULTI(DPM_FORM_MS3);
  }
#undef DPM_MULTI
}

template <typename TS, typename TE, int FORM, int GUIDE>
int launch_xe(const dpm_stage* st, const dpm_buffers* b, const LaunchCtx& s) {
  return (b->xe != nullptr && b->xe != b->x) ? launch_typed<TS, TE, FORM, GUIDE, true>(st, b, s)
                                            : launch_typed<TS, TE, FORM, GUIDE, false>(st, b, s);
}

template <typename TS, typename TE, int FORM>
int launch_guide(const dpm_stage* st, const dpm_buffers* b, const LaunchCtx& s) {
  switch (st->guidance) {
    case DPM_GUIDE_NONE: return launch_xe<TS, TE, FORM, DPM_GUIDE_NONE>(st, b, s);
    case DPM_GUIDE_CFG: return launch_xe<TS, TE, FORM, DPM_GUIDE_CFG>(st, b, s);
    case DPM_GUIDE_CLASSIFIER: return launch_xe<TS, TE, FORM, DPM_GUIDE_CLASSIFIER>(st, b, s);
  }
  return dpm_set_error(DPM_ERR_ARG, "unknown guidance %d", st->guidance);
}

// The single-request launchers of one dtype pair are spread over two translation units (compile time: the build is the
// slowest unit).  FORMS = the update forms this unit instantiates (bit f = form f); a stage of another form returns
// FORM_ELSEWHERE and the caller (dpm_stage_<pair>.hip) passes it on to the sibling unit.
constexpr int FORM_ELSEWHERE = -1001;
constexpr unsigned FORMS_A = (1u << DPM_FORM_TWO) | (1u << DPM_FORM_SS3T);  // + the fused multi-request launchers
constexpr unsigned FORMS_B = (1u << DPM_FORM_LIN1) | (1u << DPM_FORM_MS3) | (1u << DPM_FORM_DENOISE);
template <typename TS, typename TE, unsigned FORMS>
int launch_form(const dpm_stage* st, const dpm_buffers* b, const LaunchCtx& s) {
  switch (st->form) {
#define DPM_FORM_CASE(F)                                            \
  case F:                                                           \
    if constexpr ((FORMS >> F) & 1u) return launch_guide<TS, TE, F>(st, b, s); \
    return FORM_ELSEWHERE;
    DPM_FORM_CASE(DPM_FORM_LIN1)
    DPM_FORM_CASE(DPM_FORM_TWO)
    DPM_FORM_CASE(DPM_FORM_MS3)
    DPM_FORM_CASE(DPM_FORM_SS3T)
    DPM_FORM_CASE(DPM_FORM_DENOISE)
#undef DPM_FORM_CASE
  }
  return dpm_set_error(DPM_ERR_ARG, "unknown update form %d", st->form);
}

}  // namespace

#ifdef DPM_CATCHALL_HOME
template <typename TS, typename TE>
const void* dpm_catchall_thresh() {
  return reinterpret_cast<const void*>(&stage_thresh_kernel<TS, TE, FORM_RT, GUIDE_RT, true, THR_THREADS, 0>);
}
template <typename TS, typename TE, bool DYN>
const void* dpm_catchall_scalar() {
  return reinterpret_cast<const void*>(&stage_kernel_scalar<TS, TE, DYN>);
}
#endif
