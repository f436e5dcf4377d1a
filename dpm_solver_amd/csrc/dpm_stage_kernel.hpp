// dpm_stage_kernel.hpp -- the streaming stage kernels: per-stage scalars (KParams), prologues, update forms, the KExt
// extensions, stage_kernel / stage_kernel_multi / stage_kernel_scalar (part of dpm_device.hpp; include that)
#pragma once

namespace {

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
// x whose quotient is a normal number and whose residual does not underflow (|x| above ~2^-100 = 8e-31), provided
// alpha's significand is not all ones (Markstein's theorem; the launch falls back to the generic prologue with a
// true division when that guard fails).  3 VALU ops instead of ~12.  Below that magnitude the result may be one ulp
// off, and an infinite x gives NaN where the division gives inf (tools/fuzz_gpu_kernel.py --extreme measures both).
// The arithmetic below is written once for V = float and V = f32x2 (two adjacent elements): the streaming kernel works
// on adjacent pairs so that the packed fp32 instructions (v_pk_mul/add/fma_f32) take their operands from the register
// pairs the loads and conversions produce, and v_cvt_pk_f16_f32 packs the two halves of one output dword directly.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f32x2 vfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// a value rounded to the network output's storage type TE and back (identity for fp32): where the reference's arithmetic
// runs on the NETWORK's tensors -- the classifier-free blend of a noise-prediction network's two outputs, ref :326-330 --
// a half-precision network makes every operation of that expression a half operation (computed in fp32, rounded once each)
template <typename TE>
__device__ __forceinline__ float round_as(float v) {
  return to_f32(from_f32<TE>(v));
}
template <typename TE>
__device__ __forceinline__ f32x2 round_as(f32x2 v) {
  return f32x2{round_as<TE>(v[0]), round_as<TE>(v[1])};
}
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
// (Since round 5 the launchers instantiate SPEC_NOISE_X0 and SPEC_GENERIC only: the eps form measured no faster with a
// prologue of its own, profiles/r05_kernel_budget.md.)
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

// TE = storage type of the network's outputs.  It matters in ONE place: a noise-prediction network under classifier-free
// guidance.  The reference blends the two halves of the network's output as they come -- `noise_uncond + guidance_scale *
// (noise - noise_uncond)` on the network's own tensors (ref :326-330) -- so with an fp16 / bf16 network (Stable Diffusion
// under autocast) the subtraction, the product with the Python-float scale and the sum are three half-precision operations,
// each computed in fp32 and rounded to the half type (torch's opmath), before the first fp32 schedule tensor promotes
// anything.  Reproduced here operation by operation; x_start / v / score networks are converted with fp32 schedule tensors
// first (ref :290-298), so their blend -- like everything after it -- is fp32.
template <int GUIDE, int SPEC = PM_RT, typename V = float, typename TE = float>
__device__ __forceinline__ V prologue(V xe, V o0, V o1, V gg, const KParams& p) {
  static_assert(SPEC != SPEC_GENERIC, "SPEC_GENERIC dispatches to a mode (stage_tiles); the prologue takes the mode");
  V eps;
  if (guide_is<GUIDE>(DPM_GUIDE_CFG, p)) {  // ref :326-330: uncond + scale * (cond - uncond)
    V nu = to_noise<SPEC>(o1, xe, p), nc = to_noise<SPEC>(o0, xe, p);
    if (sizeof(TE) == 2 && (SPEC >= 0 ? (SPEC >> 1) == DPM_MODEL_NOISE : p.model_type == DPM_MODEL_NOISE))
      eps = round_as<TE>(nu + round_as<TE>(p.cfg_scale * round_as<TE>(nc - nu)));
    else
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

// Are the solver's MODEL VALUES half-precision tensors in the reference?  Only in the noise-prediction form (algorithm_type
// 'dpmsolver': the model value IS the network's noise, ref :444-451) of a noise network whose output nothing fp32 has touched
// -- unguided, or classifier-free (a half blend, see prologue) -- with a half-precision network.  Then every DIFFERENCE OF TWO
// MODEL VALUES in the update formulas (ref :636-669, :728-789, :827-851, :880-903) is a half operation, rounded to the
// network's dtype before the (1,)-shaped fp32 coefficient promotes the product; reproduced in combine().  In the
// data-prediction form (dpmsolver++) the model value is x0 = (x - sigma eps) / alpha, fp32 from the start.
template <typename TE>
__device__ __forceinline__ bool model_values_are_half(const KParams& p) {
  return sizeof(TE) == 2 && !(p.flags & DPM_F_TO_X0) && p.model_type == DPM_MODEL_NOISE && p.guidance != DPM_GUIDE_CLASSIFIER;
}
// hm (wave-uniform): round the difference of two model values through TE
template <typename TE, typename V>
__device__ __forceinline__ V mdiff(V a, V b, bool hm) {
  if (sizeof(TE) == 2 && hm) return round_as<TE>(a - b);
  return a - b;
}

// the exponential-integrator combination, reference association
template <int FORM, typename V, typename TE = float>
__device__ __forceinline__ V combine(V x, V mn, V h1, V h2, const KParams& p, bool hm = false) {
  if (FORM == DPM_FORM_LIN1) {
    return p.cx * x - p.c0 * mn;  // ref :573-576, :585-588
  } else if (FORM == DPM_FORM_TWO) {
    V D = p.k0 * mdiff<TE>(mn, h1, hm);
    V P = (p.flags & DPM_F_BASE_HIST) ? h1 : mn;
    return (p.cx * x - p.c0 * P) - p.c1 * D;  // ref :827-851 (multistep), :636-669, :728-778 (singlestep)
  } else if (FORM == DPM_FORM_MS3) {
    V D1_0 = p.k0 * mdiff<TE>(mn, h1, hm);  // ref :880-883
    V D1_1 = p.k1 * mdiff<TE>(h1, h2, hm);
    V dd = D1_0 - D1_1;
    V D1 = D1_0 + p.k2 * dd;
    V D2 = p.k3 * dd;
    return ((p.cx * x - p.c0 * mn) - p.c1 * D1) - p.c2 * D2;  // ref :888-903
  } else if (FORM == DPM_FORM_SS3T) {
    // h1 = model_s, h2 = model_s1, mn = model_s2; ref :741-750, :780-789.  (With half model values the reference's D1 / D2
    // chain is half arithmetic too -- r1, r2 arrive as 0-dim tensors, which do not promote -- but HOW torch rounds a 0-dim
    // operand differs between its CPU and GPU kernels and between the left and the right operand; only the two tensor-tensor
    // differences are reproduced, the rest of this one formula stays fp32: INTEGRATION.md, behavioural notes.)
    V D1_0 = p.k0 * mdiff<TE>(h2, h1, hm);
    V D1_1 = p.k1 * mdiff<TE>(mn, h1, hm);
    V D1 = (p.k2 * D1_0 - p.k3 * D1_1) / p.k4;
    V D2 = (2.f * (D1_1 - D1_0)) / p.k4;
    return ((p.cx * x - p.c0 * h1) - p.c1 * D1) - p.c2 * D2;
  } else {
    return mn;  // DPM_FORM_DENOISE, ref :541-545
  }
}

template <int FORM, typename V, typename TE = float>
__device__ __forceinline__ V combine_any(V x, V mn, V h1, V h2, const KParams& p, bool hm = false) {
  if constexpr (FORM != FORM_RT) {
    return combine<FORM, V, TE>(x, mn, h1, h2, p, hm);
  } else {
    switch (p.form) {
      case DPM_FORM_LIN1: return combine<DPM_FORM_LIN1, V, TE>(x, mn, h1, h2, p, hm);
      case DPM_FORM_TWO: return combine<DPM_FORM_TWO, V, TE>(x, mn, h1, h2, p, hm);
      case DPM_FORM_MS3: return combine<DPM_FORM_MS3, V, TE>(x, mn, h1, h2, p, hm);
      case DPM_FORM_SS3T: return combine<DPM_FORM_SS3T, V, TE>(x, mn, h1, h2, p, hm);
      default: return combine<DPM_FORM_DENOISE, V, TE>(x, mn, h1, h2, p, hm);
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
template <int GUIDE, bool XE, int PM, int U, bool NEEDS_X, typename TE>
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
      mnv[u][q / 2] = prologue<GUIDE, PM, f32x2, TE>(xe2, f32x2{v0[u][q], v0[u][q + 1]},
                                          GUIDE == DPM_GUIDE_CFG ? f32x2{v1[u][q], v1[u][q + 1]} : z,
                                          GUIDE == DPM_GUIDE_CLASSIFIER ? f32x2{vg[u][q], vg[u][q + 1]} : z, p);
    }
}

// EXT = the launch uses one of the KExt extensions (duplicate store, strided network output, mask blend): the same
// tiling, with the extra index arithmetic and streams compiled in.  EXT launches have no ragged tail (the scalar kernel
// takes those) and use the split layout only when no per-sample / per-period index is involved.
// One workgroup iteration: the U tiles that start at tile t0 of ONE tensor set (shared by the single-request kernel and
// the fused multi-request kernel below).  Everything outside the two unrolled loops is wave-uniform scalar work.
// DMA = the read streams travel by LDS-DMA (glds16: global memory -> this wavefront's LDS rows -> ds_read_b128) instead of
// through global_load_dwordx4 into registers: the lone-launch variant of the north-star kernels (2-byte state and network
// output, unguided noise-prediction network, dpmsolver++: x, eps and -- second order -- the cached model value).  Same
// elements per lane, same arithmetic, same bits; `dma_rows` = the workgroup's dynamic LDS (3 KiB per wavefront).
template <typename TS, typename TE, int FORM, int GUIDE, bool XE, int SPEC, int U, int NT, bool EXT, bool DMA = false>
__device__ __forceinline__ void stage_tiles(const TS* __restrict__ x, const TS* __restrict__ xe,
                                            const TE* __restrict__ e0, const TE* __restrict__ e1,
                                            const TE* __restrict__ g, const TS* __restrict__ h1,
                                            const TS* __restrict__ h2, TS* __restrict__ xo, TS* __restrict__ mo,
                                            const int64_t ngroups, const int64_t t0, const KParams& p, const KExt& ext,
                                            u32x4* dma_rows = nullptr) {
  using FT = FormTraits<FORM>;
  static_assert(!DMA || (sizeof(TS) == 2 && sizeof(TE) == 2 && !EXT && !XE && GUIDE == DPM_GUIDE_NONE && SPEC == SPEC_NOISE_X0 &&
                         U == 1 && (FORM == DPM_FORM_LIN1 || FORM == DPM_FORM_TWO)),
                "the LDS-DMA variant exists for the lone-launch north-star kernels only");
  constexpr bool SPLIT = sizeof(TS) == 4;  // see load_tile
  const bool need_xe = spec_need_xe<SPEC>(p);
  const bool store_m = p.flags & DPM_F_STORE_M;
  // (the compile-time data-prediction prologue never has half model values; everything else asks the stage record)
  const bool hm = (SPEC >= 0 && SPEC != SPEC_GENERIC && (SPEC & 1)) ? false : model_values_are_half<TE>(p);
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
  if constexpr (DMA) {
    // three (two: first order) 16-byte LDS-DMA loads per lane, one wait, three ds_read_b128 of the lane's own bytes
    const int64_t gr = t0 * 256 + tile_lane();
    const int64_t gi = gr < ngroups ? gr : ngroups - 1;
    const uint32_t lane = threadIdx.x & 63u;
    glds16<(NT & 1) != 0>(reinterpret_cast<const u32x4*>(x) + gi, dma_rows);
    glds16<(NT & 1) != 0>(reinterpret_cast<const u32x4*>(e0) + gi, dma_rows + 64);
    if (FT::needs_h1) glds16<(NT & 1) != 0>(reinterpret_cast<const u32x4*>(h1) + gi, dma_rows + 128);
    lds_dma_wait();
    unpack8(dma_rows[lane], x, vx[0]);
    unpack8(dma_rows[64 + lane], e0, v0[0]);
    if (FT::needs_h1) unpack8(dma_rows[128 + lane], h1, vh1[0]);
  }
#pragma unroll
  for (int u = 0; u < (DMA ? 0 : U); ++u) {
    const int64_t gr = (t0 + u) * 256 + tile_lane();
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
        load_tile<false>(mask, mt * 256 + tile_lane(), true, vm[u]);
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
#define DPM_MODELS(PM_) tile_models<GUIDE, XE, PM_, U, FT::needs_x, TE>(vx, vxe, v0, v1, vg, need_xe, p, mnv)
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
    const int64_t gi = (t0 + u) * 256 + tile_lane();
    const bool split = can_split && (t0 + u) * 256 + 256 <= ngroups;
    float ox[EPT], om[EPT];
#pragma unroll
    for (int q = 0; q < EPT; q += 2) {
      const f32x2 z = {0.f, 0.f};
      const f32x2 mn = mnv[u][q / 2];
      const f32x2 o = combine<FORM, f32x2, TE>(FT::needs_x ? f32x2{vx[u][q], vx[u][q + 1]} : z, mn,
                                    FT::needs_h1 ? f32x2{vh1[u][q], vh1[u][q + 1]} : z,
                                    FT::needs_h2 ? f32x2{vh2[u][q], vh2[u][q + 1]} : z, p, hm);
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

// Launched with 256 or 512 threads: each 256-lane group of a workgroup walks tiles of its own, so the two shapes run the same
// lanes over the same tiles and differ only in how many workgroups the dispatcher has to place -- 1024 instead of 2048 for
// a [256,4,64,64] request.  Inside a network loop (rocprofv3 rows, same box, alternating runs, profiles/r04_block_threads.md)
// the lone 2M launch takes 8.3-8.6 us instead of 8.6-8.7 (fp16), 14.2-14.3 instead of 14.5-14.7 (fp32), 13.1 instead of 13.6
// (fp32 state, fp16 network); 1024 threads were measured too: no better for these, 1.6 % worse for the last, and a 128-VGPR
// budget the extended kernels do not fit.
constexpr int STAGE_MAX_THREADS = 512;
template <typename TS, typename TE, int FORM, int GUIDE, bool XE, int SPEC, int U, int NT, bool EXT, bool DYN = false>
__global__ __launch_bounds__(STAGE_MAX_THREADS) void stage_kernel(const TS* __restrict__ x, const TS* __restrict__ xe,
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
  const uint32_t per = blockDim.x >> 8;                                                     // 256-lane groups per workgroup
  const uint32_t sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));   // this wavefront's (wave-uniform)
  for (int64_t t0 = ((int64_t)blockIdx.x * per + sub) * U; t0 < ntiles; t0 += (int64_t)gridDim.x * per * U)
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
      const float mn = prologue<GUIDE, SPEC == SPEC_GENERIC ? (int)PM_RT : SPEC, float, TE>(
          xev, to_f32(e0[i]), GUIDE == DPM_GUIDE_CFG ? to_f32(e1[i]) : 0.f, GUIDE == DPM_GUIDE_CLASSIFIER ? to_f32(g[i]) : 0.f, p);
      xo[i] = from_f32<TS>(combine<FORM, float, TE>(xv, mn, FT::needs_h1 ? to_f32(h1[i]) : 0.f, FT::needs_h2 ? to_f32(h2[i]) : 0.f, p,
                                                    model_values_are_half<TE>(p)));
      if (store_m) mo[i] = from_f32<TS>(mn);
    }
  }
}

// LAB build (launch_stream): the lone-launch variant of the north-star kernels with its read streams on the LDS-DMA path
// (stage_tiles<..., DMA = true>) -- measured equal to the register path (profiles/r05_lone_floor.md), not in the product.
// One tile per 256-lane group and iteration like stage_kernel; dynamic LDS = 3 KiB per wavefront.  The rows of a wavefront
// are reused by its next tile: the ds_reads of the previous tile are complete before the stores it waited on are issued
// (their results are the stores' operands), so the next tile's LDS-DMA cannot overtake them.
template <typename TS, typename TE, int FORM, int NT>
__global__ __launch_bounds__(STAGE_MAX_THREADS) void stage_kernel_dma(const TS* __restrict__ x, const TE* __restrict__ e0,
                                                                      const TS* __restrict__ h1, TS* __restrict__ xo,
                                                                      TS* __restrict__ mo, int64_t n, const KParams p) {
  extern __shared__ __align__(16) unsigned char dma_lds[];
  const int64_t ngroups = n / EPT;
  const int64_t ntiles = (ngroups + 255) / 256;
  const uint32_t per = blockDim.x >> 8;
  const uint32_t sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  u32x4* rows = reinterpret_cast<u32x4*>(dma_lds) + (size_t)wave * 192;
  const KExt ext = {};
  for (int64_t t0 = (int64_t)blockIdx.x * per + sub; t0 < ntiles; t0 += (int64_t)gridDim.x * per)
    stage_tiles<TS, TE, FORM, DPM_GUIDE_NONE, false, SPEC_NOISE_X0, 1, NT, false, true>(x, nullptr, e0, nullptr, nullptr, h1, nullptr,
                                                                                      xo, mo, ngroups, t0, p, ext, rows);
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
  void* xo2[MULTI_MAX];  // classifier-free guidance: the second half of the [2B, ...] network input (or null)
};

template <typename TS, typename TE, int FORM, int GUIDE, int SPEC, int U, int NT>
__global__ __launch_bounds__(STAGE_MAX_THREADS) void stage_kernel_multi(const MultiTab tab, int64_t n, uint32_t nreq, uint32_t spr,
                                                          KParams p, uint32_t xcd_span) {
  const int64_t ngroups = n / EPT;
  KExt ext = {};
  const uint32_t total = nreq * spr;  // spr = super-tiles (U tiles) per request
  // 256 or 512 threads per workgroup: every 256-lane group takes super-tiles of its own (see stage_kernel)
  const uint32_t per = blockDim.x >> 8;
  const uint32_t sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const uint32_t wgs = xcd_span ? 8u * ((xcd_span + per - 1u) / per) : (total + per - 1u) / per;
  for (uint32_t b = blockIdx.x; b < wgs; b += gridDim.x) {
    // xcd_span != 0 (tuning): workgroup b runs on XCD b % 8 -- give every XCD one contiguous eighth of the tile space
    const uint32_t in_xcd = (b >> 3) * per + sub;
    if (xcd_span && in_xcd >= xcd_span) continue;
    const uint32_t v = xcd_span ? (b & 7u) * xcd_span + in_xcd : b * per + sub;
    if (v >= total) continue;
    const uint32_t r = v / spr;
    const int64_t t0 = (int64_t)(v - r * spr) * U;
    // the guided variants carry the duplicate store of the CFG network input (KExt::xo2) -- the EXT flavour of the tile body
    constexpr bool DUP = GUIDE == DPM_GUIDE_CFG;
    if constexpr (DUP) ext.xo2 = tab.xo2[r];
    stage_tiles<TS, TE, FORM, GUIDE, false, SPEC, U, NT, DUP>(
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
  const bool hm = model_values_are_half<TE>(p);
  const TS* mask = static_cast<const TS*>(ext.mask);
  const TS* ba = static_cast<const TS*>(ext.ba);
  const TS* bb = static_cast<const TS*>(ext.bb);
  TS* xo2 = static_cast<TS*>(ext.xo2);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t ie = ext.eps_stride ? (i / ext.per_sample) * ext.eps_stride + i % ext.per_sample : i;
    const float xv = nx ? to_f32(x[i]) : 0.f;
    const float xev = need_xe ? to_f32(xe[i]) : 0.f;
    const float mn = prologue<GUIDE_RT, PM_RT, float, TE>(xev, to_f32(e0[ie]), cfg ? to_f32(e1[ie]) : 0.f, clsg ? to_f32(g[i]) : 0.f, p);
    float o = combine_any<FORM_RT, float, TE>(xv, mn, nh1 ? to_f32(h1[i]) : 0.f, nh2 ? to_f32(h2[i]) : 0.f, p, hm);
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

}  // namespace
