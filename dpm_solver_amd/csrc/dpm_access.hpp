// dpm_access.hpp -- element types and global-memory access of the stage kernels: 16-byte packs, write-through stores,
// the split-tile layout (part of dpm_device.hpp; include that)
#pragma once

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
__device__ __forceinline__ __half from_f32<__half>(float v) {
  // What is converted is the ROUNDED fp32 value (torch's opmath: the operation in fp32, then one conversion).  Without the
  // empty asm the compiler folds a preceding fp32 multiply into v_fma_mixlo_f16, which rounds the exact product ONCE to
  // fp16: one fp16 ulp off wherever the fp32 rounding lands on an fp16 tie -- guidance_scale 7.3 times an 11-bit
  // difference does in ~1 % of elements (the classifier-free blend of an fp16 network in the one-element-per-lane and
  // thresholding kernels; found by tools/fuzz_gpu_kernel.py in round 6, build() checks the listing for the instruction).
  asm("" : "+v"(v));
  return __float2half_rn(v);
}
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


// LDS-DMA: 16 bytes per lane straight from global memory into LDS at (wave-uniform base, in M0) + lane * 16, no VGPR in
// between (global_load_lds_dwordx4; aux 2 = nt).  `lds_row` = this WAVEFRONT's 1 KiB row: lane l's bytes land at row + 16 l,
// so a lane reads back exactly what it would have loaded.  The issuing wavefront waits with `s_waitcnt vmcnt(0)` (lds_dma_wait)
// before its ds_read -- rows are private to a wavefront, no barrier.  Measured on the lone [256,4,64,64] launch inside a network
// loop (tools/floor.py, profiles/r05_lone_floor.md): the no-arithmetic kernel of the 2M stage's five streams takes 9.20 us by
// events with its three read streams on this path against 9.62 through registers.
template <bool NT>
__device__ __forceinline__ void glds16(const u32x4* gsrc, u32x4* lds_row) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_row, 16, 0, NT ? 2 : 0);
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the 8 elements of one 16-byte pack of a 2-byte type -> fp32
__device__ __forceinline__ void unpack8(const u32x4& a, const __half*, float (&out)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out[2 * j] = __half2float(__ushort_as_half((unsigned short)(a[j] & 0xffffu)));
    out[2 * j + 1] = __half2float(__ushort_as_half((unsigned short)(a[j] >> 16)));
  }
}
__device__ __forceinline__ void unpack8(const u32x4& a, const bf16_t*, float (&out)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out[2 * j] = __uint_as_float(a[j] << 16);
    out[2 * j + 1] = __uint_as_float(a[j] & 0xffff0000u);
  }
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
  unpack8(ld16<NT>(reinterpret_cast<const u32x4*>(p) + group), p, out);
}
template <bool NT>
__device__ __forceinline__ void load_pack(const bf16_t* __restrict__ p, int64_t group, float (&out)[EPT]) {
  unpack8(ld16<NT>(reinterpret_cast<const u32x4*>(p) + group), p, out);
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

// A tile belongs to 256 consecutive lanes.  The streaming kernel may be launched with 256 or 512 threads: every
// 256-lane group of a workgroup then takes tiles of its own (stage_kernel) -- fewer, larger workgroups for the dispatcher
// to place, the same lanes doing the same work.  (In a kernel bounded to 256 threads the mask folds away.)
__device__ __forceinline__ uint32_t tile_lane() { return threadIdx.x & 255u; }

// Tile-level access for the streaming kernel.  A tile is the 2048 elements of one 256-lane group's iteration (256 lanes x 8).
// `split` (4-byte state, tile complete): lane t takes elements [4t, 4t+4) and [1024+4t, 1024+4t+4) of the tile, so
// each of the two global_load_dwordx4 of a wavefront covers 1 KiB of consecutive addresses; otherwise lane t takes the
// 8 consecutive elements [8t, 8t+8) (one 16-byte access for 2-byte types, two adjacent ones for fp32).  The op is
// elementwise, so any mapping that is the same for every tensor of the launch is correct.  In the split case gi is
// (first group of the tile) + lane; the tile may start at any group of the tensor (strided network outputs).
template <bool NT, typename T>
__device__ __forceinline__ void load_tile(const T* __restrict__ p, int64_t gi, bool split, float (&out)[EPT]) {
  if constexpr (sizeof(T) == 4) {
    if (split) {
      const u32x4* q = reinterpret_cast<const u32x4*>(p) + (2 * gi - (int64_t)tile_lane());
      const u32x4 a = ld16<NT>(q), b = ld16<NT>(q + 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        out[j] = __uint_as_float(a[j]);
        out[4 + j] = __uint_as_float(b[j]);
      }
      return;
    }
  } else {
    if (split) {
      // 2-byte network output next to a 4-byte state.  The state's split layout gives lane t of the tile the elements
      // [4t, 4t+4) and [1024+4t, +4): as two 8-byte accesses per lane (round 2) the network-output streams ran at the
      // 8-byte rate of the load path (MI355X_MICROARCH.md: 0.54-0.70x the 16-byte rate).  Now ONE 16-byte access per lane:
      // wavefront w needs two 512-byte runs of the tile, elements [256w, +256) and [1024+256w, +256); its lanes 0..31
      // fetch the first run and lanes 32..63 the second, 8 consecutive elements each, and four ds_permute_b32 (the LDS
      // crossbar, no LDS memory) hand every lane its two 4-element groups: source lane s < 32 owns what the even lane 2s
      // (dwords 0, 1) and the odd lane 2s+1 (dwords 2, 3) need of the first run, lane 32+s the same of the second run; every
      // permute moves one dword to each of the 64 lanes (first-run dwords to the even lanes while second-run dwords go to
      // the odd ones, then the other way round).  Requires all 64 lanes active: load_tile runs in straight-line code.
      const int l = (int)(threadIdx.x & 63u), w = (int)(tile_lane() >> 6);
      const bool lo = l < 32;
      const u32x4* q = reinterpret_cast<const u32x4*>(p) + ((gi - (int64_t)tile_lane()) + 32 * w + (lo ? l : l + 96));
      const u32x4 v = ld16<NT>(q);
      const int s2 = (lo ? l : l - 32) * 2;
      const int to_a = (lo ? s2 : s2 + 1) * 4, to_b = (lo ? s2 + 1 : s2) * 4;  // byte address = destination lane * 4
      const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_permute(to_a, (int)(lo ? v[0] : v[2]));
      const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_permute(to_a, (int)(lo ? v[1] : v[3]));
      const uint32_t r3 = (uint32_t)__builtin_amdgcn_ds_permute(to_b, (int)(lo ? v[2] : v[0]));
      const uint32_t r4 = (uint32_t)__builtin_amdgcn_ds_permute(to_b, (int)(lo ? v[3] : v[1]));
      const bool even = (l & 1) == 0;
      const uint32_t w4[4] = {even ? r1 : r3, even ? r2 : r4, even ? r3 : r1, even ? r4 : r2};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (std::is_same<T, __half>::value) {
          out[2 * j] = __half2float(__ushort_as_half((unsigned short)(w4[j] & 0xffffu)));
          out[2 * j + 1] = __half2float(__ushort_as_half((unsigned short)(w4[j] >> 16)));
        } else {
          out[2 * j] = __uint_as_float(w4[j] << 16);
          out[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u);
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
      u32x4* q = reinterpret_cast<u32x4*>(p) + (2 * gi - (int64_t)tile_lane());
      st16<NT>(q, a);
      st16<NT>(q + 256, b);
      return;
    }
  }
  store_pack<NT>(p, gi, in);
}

}  // namespace
