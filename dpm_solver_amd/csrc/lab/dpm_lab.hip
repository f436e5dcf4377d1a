// dpm_lab.hip -- the LAB build's own entry points (include/dpm_lab.h): process-global tuning knobs, event-bracketed
// launches, the no-arithmetic calibration / floor kernels, the side-stream helpers.  Compiled into libdpm_lab.so only
// (__graft_entry__.build_lab: every source of the product library with -DDPM_LAB=1, plus csrc/lab/); the product library
// contains none of this.
#include "dpm_device.hpp"
#include "dpm_lab.h"

#include <vector>

#if !DPM_LAB
#error "csrc/lab/ is compiled with -DDPM_LAB=1 only"
#endif

namespace dpmk {
Tuning g_lab_tuning;
}
using dpmk::g_lab_tuning;

// launch hooks of the product sources (dpm_kernels.hip)
int dpm_stage_launch_ev(const dpm_stage* st, const dpm_buffers* b, void* stream, void* ev_start, void* ev_stop);
int dpm_timing_begin(int n, void*** starts, void*** stops);
int dpm_timing_end(int n, void** starts, void** stops, void* stream, float* ms, const unsigned char* recorded);

extern "C" int dpm_lab_build(void) { return 1; }
extern "C" const void* dpm_lab_device_context(int dev) { return &device_context(dev); }

extern "C" int dpm_stage_launch_timed(const dpm_stage* st, const dpm_buffers* b, void* stream, float* ms) {
  if (!ms) return dpm_set_error(DPM_ERR_ARG, "null pointer");
  void **starts = nullptr, **stops = nullptr;
  int rc = dpm_timing_begin(1, &starts, &stops);
  if (rc) return rc;
  rc = dpm_stage_launch_ev(st, b, stream, starts[0], stops[0]);
  int rc2 = dpm_timing_end(1, starts, stops, stream, rc ? nullptr : ms, nullptr);
  return rc ? rc : rc2;
}

// ---- event-bracketed launches without a synchronisation per launch (include/dpm_lab.h: dpm_trace_*)
struct dpm_trace {
  int cap = 0;
  void** starts = nullptr;  // 2 * cap events: starts, then stops (dpm_timing_begin's layout)
  void** stops = nullptr;
  std::vector<unsigned char> used;
};

extern "C" int dpm_trace_create(int capacity, dpm_trace** out) {
  if (!out || capacity < 1 || capacity > (1 << 20)) return dpm_set_error(DPM_ERR_ARG, "trace_create: bad arguments");
  dpm_trace* t = new (std::nothrow) dpm_trace;
  if (!t) return dpm_set_error(DPM_ERR_NOMEM, "out of memory");
  int rc = dpm_timing_begin(capacity, &t->starts, &t->stops);
  if (rc) {
    delete t;
    return rc;
  }
  t->cap = capacity;
  t->used.assign((size_t)capacity, 0);
  *out = t;
  return DPM_OK;
}

extern "C" int dpm_stage_launch_traced(const dpm_stage* st, const dpm_buffers* b, void* stream, dpm_trace* t, int slot) {
  if (!t || slot < 0 || slot >= t->cap) return dpm_set_error(DPM_ERR_ARG, "stage_launch_traced: slot %d outside the trace", slot);
  const int rc = dpm_stage_launch_ev(st, b, stream, t->starts[slot], t->stops[slot]);
  if (!rc) t->used[(size_t)slot] = 1;
  return rc;
}

extern "C" int dpm_trace_read(dpm_trace* t, void* stream, float* ms, int n) {
  if (!t || !ms || n < 0) return dpm_set_error(DPM_ERR_ARG, "trace_read: bad arguments");
  hipError_t rc = hipStreamSynchronize(static_cast<hipStream_t>(stream));
  if (rc != hipSuccess) return dpm_set_error((int)rc, "hipStreamSynchronize: %s", hipGetErrorString(rc));
  for (int i = 0; i < n; ++i) {
    ms[i] = -1.f;
    if (i < t->cap && t->used[(size_t)i]) {
      rc = hipEventElapsedTime(&ms[i], static_cast<hipEvent_t>(t->starts[i]), static_cast<hipEvent_t>(t->stops[i]));
      if (rc != hipSuccess) return dpm_set_error((int)rc, "hipEventElapsedTime(slot %d): %s", i, hipGetErrorString(rc));
      t->used[(size_t)i] = 0;
    }
  }
  return DPM_OK;
}

extern "C" void dpm_trace_destroy(dpm_trace* t) {
  if (!t) return;
  for (int i = 0; i < 2 * t->cap; ++i) (void)hipEventDestroy(static_cast<hipEvent_t>(t->starts[i]));
  delete[] t->starts;
  delete t;
}

// ---- prefetch: read buffers and drop the data (dpm_prefetch_launch)
namespace {
constexpr int PREFETCH_MAX = 8;
struct PrefetchTab {
  const u32x4* p[PREFETCH_MAX];
  int64_t nvec[PREFETCH_MAX];
};
template <bool NT>
__global__ __launch_bounds__(256) void prefetch_kernel(const PrefetchTab tab, int n_buf) {
  for (int r = 0; r < n_buf; ++r) {
    const u32x4* p = tab.p[r];
    const int64_t nv = tab.nvec[r];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
      const u32x4 v = ld16<NT>(p + i);
      asm volatile("" ::"v"(v));  // keeps the load; the data is not wanted
    }
  }
}
}  // namespace

extern "C" int dpm_prefetch_launch(const void* const* bufs, const int64_t* bytes, int n_buf, int policy, void* stream) {
  if (!bufs || !bytes || n_buf < 0 || n_buf > PREFETCH_MAX) return dpm_set_error(DPM_ERR_ARG, "prefetch: bad arguments (<= %d buffers)", PREFETCH_MAX);
  PrefetchTab tab;
  std::memset(&tab, 0, sizeof tab);
  int64_t total = 0;
  int k = 0;
  for (int i = 0; i < n_buf; ++i) {
    if (!bufs[i] || bytes[i] < 16) continue;
    if (!aligned(bufs[i], 16)) return dpm_set_error(DPM_ERR_ALIGN, "prefetch: buffer %d is not 16-byte aligned", i);
    tab.p[k] = static_cast<const u32x4*>(bufs[i]);
    tab.nvec[k] = bytes[i] / 16;
    total += tab.nvec[k];
    ++k;
  }
  if (!k) return DPM_OK;
  const DeviceInfo& di = device_info();
  const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * 4;
  int64_t blocks = (total / k + 255) / 256;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (policy == 1)
    hipLaunchKernelGGL(prefetch_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, tab, k);
  else
    hipLaunchKernelGGL(prefetch_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, tab, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "prefetch launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

// ------------------------------------------------------------------------------------------------
// calibration kernels: what the memory system sustains for this access pattern and size, with no arithmetic.
// kind 0: copy (1 read + 1 write stream); kind 1: 3 read + 2 write streams (the 2M stage's pattern); kind 2: 4 read + 1 write
// streams (the same bytes: what a 2M stage would move that re-derives the previous model value from the previous state and
// network output instead of storing it -- `e` is read).
// ------------------------------------------------------------------------------------------------
namespace {
template <int BLOCK, int KIND, int NT>
__global__ __launch_bounds__(BLOCK) void calib_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                      const u32x4* __restrict__ c, u32x4* __restrict__ d,
                                                      u32x4* __restrict__ e, int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < nvec; i += stride) {
    if (KIND == 0) {
      st16<(NT & 2) != 0>(d + i, ld16<(NT & 1) != 0>(a + i));
    } else if (KIND == 2) {
      const u32x4 va = ld16<(NT & 1) != 0>(a + i), vb = ld16<(NT & 1) != 0>(b + i), vc = ld16<(NT & 1) != 0>(c + i);
      const u32x4 ve = ld16<(NT & 1) != 0>(e + i);
      st16<(NT & 2) != 0>(d + i, (va ^ vb) ^ (vc ^ ve));
    } else {
      const u32x4 va = ld16<(NT & 1) != 0>(a + i), vb = ld16<(NT & 1) != 0>(b + i), vc = ld16<(NT & 1) != 0>(c + i);
      st16<(NT & 2) != 0>(d + i, va ^ vb);
      st16<(NT & 4) != 0>(e + i, vb ^ vc);
    }
  }
}

template <int BLOCK, int KIND>
void calib_nt(int nt, dim3 grid, const LaunchCtx& c, const u32x4* a, const u32x4* b, const u32x4* cc, u32x4* d, u32x4* e,
              int64_t nvec) {
  switch (nt) {
    case 1: launch(calib_kernel<BLOCK, KIND, 1>, grid, dim3(BLOCK), 0, c, a, b, cc, d, e, nvec); break;
    case 5: launch(calib_kernel<BLOCK, KIND, 5>, grid, dim3(BLOCK), 0, c, a, b, cc, d, e, nvec); break;
    case 7: launch(calib_kernel<BLOCK, KIND, 7>, grid, dim3(BLOCK), 0, c, a, b, cc, d, e, nvec); break;
    default: launch(calib_kernel<BLOCK, KIND, 0>, grid, dim3(BLOCK), 0, c, a, b, cc, d, e, nvec); break;
  }
}
}  // namespace

extern "C" int dpm_calib_launch(int kind, int block, int blocks_per_cu, int nt, const void* a, const void* b, const void* c,
                                void* d, void* e, int64_t nbytes, void* stream, float* ms) {
  if (!a || !d || nbytes < 16 || (kind >= 1 && (!b || !c || !e))) return dpm_set_error(DPM_ERR_ARG, "calib: bad arguments");
  const int64_t nvec = nbytes / 16;
  const DeviceInfo& di = device_info();
  int64_t blocks = (nvec + block - 1) / block;
  const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * blocks_per_cu;
  if (blocks > cap) blocks = cap;
  void **starts = nullptr, **stops = nullptr;
  if (ms) {
    int rc = dpm_timing_begin(1, &starts, &stops);
    if (rc) return rc;
  }
  const LaunchCtx ctx{static_cast<hipStream_t>(stream), ms ? static_cast<hipEvent_t>(starts[0]) : nullptr,
                      ms ? static_cast<hipEvent_t>(stops[0]) : nullptr};
  const u32x4 *pa = (const u32x4*)a, *pb = (const u32x4*)b, *pc = (const u32x4*)c;
  u32x4 *pd = (u32x4*)d, *pe = (u32x4*)e;
  const dim3 grid((unsigned)blocks);
  int rc = DPM_OK;
  if (kind == 0 && block == 256) calib_nt<256, 0>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else if (kind == 0 && block == 512) calib_nt<512, 0>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else if (kind == 0 && block == 1024) calib_nt<1024, 0>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else if (kind == 1 && block == 256) calib_nt<256, 1>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else if (kind == 1 && block == 512) calib_nt<512, 1>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else if (kind == 1 && block == 1024) calib_nt<1024, 1>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else if (kind == 2 && block == 256) calib_nt<256, 2>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else if (kind == 2 && block == 512) calib_nt<512, 2>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else if (kind == 2 && block == 1024) calib_nt<1024, 2>(nt, grid, ctx, pa, pb, pc, pd, pe, nvec);
  else rc = dpm_set_error(DPM_ERR_ARG, "calib: kind %d / block %d not built", kind, block);
  if (ms) {
    int rc2 = dpm_timing_end(1, starts, stops, stream, rc ? nullptr : ms, nullptr);
    if (!rc) rc = rc2;
  }
  return rc;
}

extern "C" int dpm_tuning_set(int knob, int value) {
  switch (knob) {
    case DPM_TUNE_UNROLL:
      if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
        return dpm_set_error(DPM_ERR_ARG, "unroll must be 0 (default), 1, 2, 4 or 8 (4 and 8: tuning builds only)");
      g_lab_tuning.unroll = value;
      return DPM_OK;
    case DPM_TUNE_NONTEMPORAL: g_lab_tuning.nontemporal = value < 0 ? -1 : (value & 7); return DPM_OK;
    case DPM_TUNE_BLOCKS_PER_CU:
      if (value < 1 || value > 64) return dpm_set_error(DPM_ERR_ARG, "blocks_per_cu must be in 1..64");
      g_lab_tuning.blocks_per_cu = value;
      return DPM_OK;
    case DPM_TUNE_ASSUME_RESIDENT: g_lab_tuning.assume_resident = value != 0; return DPM_OK;
    case DPM_TUNE_MULTI_FUSE: g_lab_tuning.multi_fuse = value != 0; return DPM_OK;
    case DPM_TUNE_CLUSTER_IN_GRAPH: g_lab_tuning.cluster_in_graph = value != 0; return DPM_OK;
    case DPM_TUNE_MULTI_XCD_REMAP: g_lab_tuning.multi_xcd_remap = value < 0 ? -1 : (value != 0); return DPM_OK;
    case DPM_TUNE_CLUSTER_ONE_HOP: g_lab_tuning.cluster_one_hop = value < 0 ? 0 : (value > 2 ? 2 : value); return DPM_OK;
    case DPM_TUNE_THR_PREDICT: g_lab_tuning.thr_predict = value != 0; return DPM_OK;
    case DPM_TUNE_THR_SPIN_LIMIT:
      if (value < 0) return dpm_set_error(DPM_ERR_ARG, "thr_spin_limit must be >= 0");
      g_lab_tuning.thr_spin_limit = value;
      return DPM_OK;
    case DPM_TUNE_THR_DEBUG_FAULT:
      if (value < 0 || value > 3) return dpm_set_error(DPM_ERR_ARG, "thr_debug_fault must be 0 .. 3");
      g_lab_tuning.thr_debug_fault = value;
      return DPM_OK;
    case DPM_TUNE_THR_ELECT: g_lab_tuning.thr_elect = value < 0 ? -1 : (value != 0); return DPM_OK;
    case DPM_TUNE_FORCE_GENERIC: g_lab_tuning.force_generic = value != 0; return DPM_OK;
    case DPM_TUNE_THR_STAGGER: g_lab_tuning.thr_stagger = value < 0 ? 0 : value; return DPM_OK;
    case DPM_TUNE_LDS_DMA: g_lab_tuning.lds_dma = value < 0 ? -1 : (value != 0); return DPM_OK;
    case DPM_TUNE_BLOCK_THREADS:
      if (value != 0 && value != 256 && value != 512)
        return dpm_set_error(DPM_ERR_ARG, "block_threads must be 0 (by size), 256 or 512");
      g_lab_tuning.block_threads = value;
      return DPM_OK;
    case DPM_TUNE_MULTI_BLOCKS_PER_CU:
      if (value < 0 || value > 4096) return dpm_set_error(DPM_ERR_ARG, "multi_blocks_per_cu must be in 0..4096");
      g_lab_tuning.multi_blocks_per_cu = value;
      return DPM_OK;
    case DPM_TUNE_BIG_TILES: g_lab_tuning.big_tiles = value < 0 ? DPM_BIG_TILES_DEFAULT : value; return DPM_OK;
  }
  return dpm_set_error(DPM_ERR_ARG, "unknown tuning knob %d", knob);
}

extern "C" int dpm_tuning_get(int knob) {
  switch (knob) {
    case DPM_TUNE_UNROLL: return g_lab_tuning.unroll;
    case DPM_TUNE_NONTEMPORAL: return g_lab_tuning.nontemporal;
    case DPM_TUNE_BLOCKS_PER_CU: return g_lab_tuning.blocks_per_cu;
    case DPM_TUNE_ASSUME_RESIDENT: return g_lab_tuning.assume_resident;
    case DPM_TUNE_MULTI_FUSE: return g_lab_tuning.multi_fuse;
    case DPM_TUNE_CLUSTER_IN_GRAPH: return g_lab_tuning.cluster_in_graph;
    case DPM_TUNE_MULTI_XCD_REMAP: return g_lab_tuning.multi_xcd_remap;
    case DPM_TUNE_CLUSTER_ONE_HOP: return g_lab_tuning.cluster_one_hop;
    case DPM_TUNE_MULTI_BLOCKS_PER_CU: return g_lab_tuning.multi_blocks_per_cu;
    case DPM_TUNE_BIG_TILES: return g_lab_tuning.big_tiles;
    case DPM_TUNE_THR_PREDICT: return g_lab_tuning.thr_predict;
    case DPM_TUNE_THR_SPIN_LIMIT: return g_lab_tuning.thr_spin_limit;
    case DPM_TUNE_THR_DEBUG_FAULT: return g_lab_tuning.thr_debug_fault;
    case DPM_TUNE_BLOCK_THREADS: return g_lab_tuning.block_threads;
    case DPM_TUNE_THR_ELECT: return g_lab_tuning.thr_elect;
    case DPM_TUNE_FORCE_GENERIC: return g_lab_tuning.force_generic;
    case DPM_TUNE_THR_STAGGER: return g_lab_tuning.thr_stagger;
    case DPM_TUNE_LDS_DMA: return g_lab_tuning.lds_dma;
  }
  return -1;
}


// ------------------------------------------------------------------------------------------------
// The floor of the lone 2M launch (include/dpm_lab.h: dpm_floor_launch): three read streams, two write streams, no
// arithmetic, over load path x rows in flight x workgroup shape x wave priority x store policy.
// ------------------------------------------------------------------------------------------------
namespace {
// LDS-DMA: 16 bytes per lane straight into LDS at (wave-uniform base in M0) + lane * 16; aux 2 = nt
template <bool NT>
__device__ __forceinline__ void glds16(const u32x4* gsrc, uint32_t* lds_row) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_row, 16, 0, NT ? 2 : 0);
}
template <int STORE>
__device__ __forceinline__ void floor_store(u32x4* p, u32x4 v) {
  if (STORE == 0)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if (STORE == 2)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// ROWS 16-byte accesses per lane and stream in flight; a 256-lane group covers ROWS x 4 KiB of every stream per step, rows
// 4 KiB apart (every wavefront instruction covers 1 KiB of consecutive addresses)
template <int PATH, int ROWS, bool NT, int STORE>
__global__ __launch_bounds__(1024) void floor_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                                     const u32x4* __restrict__ c, u32x4* __restrict__ d,
                                                     u32x4* __restrict__ e, int64_t nvec, int prio) {
  extern __shared__ __align__(16) unsigned char floor_lds[];
  switch (prio) {  // s_setprio takes an immediate
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }
  const uint32_t per = blockDim.x >> 8;
  const uint32_t sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const uint32_t lane256 = threadIdx.x & 255u;
  const int64_t nsteps = (nvec + 256 * ROWS - 1) / (256 * ROWS);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // LDS rows of this wavefront: [stream][row][64 lanes x 16 B]
  u32x4* lrow = reinterpret_cast<u32x4*>(floor_lds) + (size_t)wave * 3 * ROWS * 64;
  const uint32_t lane = threadIdx.x & 63u;
  for (int64_t s = (int64_t)blockIdx.x * per + sub; s < nsteps; s += (int64_t)gridDim.x * per) {
    int64_t idx[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int64_t i = (s * ROWS + r) * 256 + lane256;
      idx[r] = i < nvec ? i : nvec - 1;  // clamped: loads are unconditional, stores guarded
    }
    u32x4 va[ROWS], vb[ROWS], vc[ROWS];
    if constexpr (PATH == 0) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        va[r] = ld16<NT>(a + idx[r]);
        vb[r] = ld16<NT>(b + idx[r]);
        vc[r] = ld16<NT>(c + idx[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        glds16<NT>(a + idx[r], reinterpret_cast<uint32_t*>(lrow + (0 * ROWS + r) * 64));
        glds16<NT>(b + idx[r], reinterpret_cast<uint32_t*>(lrow + (1 * ROWS + r) * 64));
        glds16<NT>(c + idx[r], reinterpret_cast<uint32_t*>(lrow + (2 * ROWS + r) * 64));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's own rows have landed (no other reads them)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        va[r] = lrow[(0 * ROWS + r) * 64 + lane];
        vb[r] = lrow[(1 * ROWS + r) * 64 + lane];
        vc[r] = lrow[(2 * ROWS + r) * 64 + lane];
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int64_t i = (s * ROWS + r) * 256 + lane256;
      if (i < nvec) {
        floor_store<STORE>(d + i, va[r] ^ vb[r]);
        floor_store<STORE>(e + i, vb[r] ^ vc[r]);
      }
    }
    if constexpr (PATH == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are reused by the next step
  }
}

template <int PATH, int ROWS, bool NT>
void floor_go(int store, dim3 grid, dim3 block, size_t lds, const LaunchCtx& c, const u32x4* a, const u32x4* b, const u32x4* cc,
              u32x4* d, u32x4* e, int64_t nvec, int prio) {
  switch (store) {
    case 1: launch(floor_kernel<PATH, ROWS, NT, 1>, grid, block, lds, c, a, b, cc, d, e, nvec, prio); break;
    case 2: launch(floor_kernel<PATH, ROWS, NT, 2>, grid, block, lds, c, a, b, cc, d, e, nvec, prio); break;
    default: launch(floor_kernel<PATH, ROWS, NT, 0>, grid, block, lds, c, a, b, cc, d, e, nvec, prio); break;
  }
}
template <int PATH, int ROWS>
void floor_nt(int nt, int store, dim3 grid, dim3 block, size_t lds, const LaunchCtx& c, const u32x4* a, const u32x4* b,
              const u32x4* cc, u32x4* d, u32x4* e, int64_t nvec, int prio) {
  if (nt) floor_go<PATH, ROWS, true>(store, grid, block, lds, c, a, b, cc, d, e, nvec, prio);
  else floor_go<PATH, ROWS, false>(store, grid, block, lds, c, a, b, cc, d, e, nvec, prio);
}
template <int PATH>
int floor_rows(int rows, int nt, int store, dim3 grid, dim3 block, const LaunchCtx& c, const u32x4* a, const u32x4* b,
               const u32x4* cc, u32x4* d, u32x4* e, int64_t nvec, int prio) {
  const size_t lds = PATH == 1 ? (size_t)(block.x / 64) * 3 * rows * 1024 : 0;
  switch (rows) {
    case 1: floor_nt<PATH, 1>(nt, store, grid, block, lds, c, a, b, cc, d, e, nvec, prio); return DPM_OK;
    case 2: floor_nt<PATH, 2>(nt, store, grid, block, lds, c, a, b, cc, d, e, nvec, prio); return DPM_OK;
    case 4: floor_nt<PATH, 4>(nt, store, grid, block, lds, c, a, b, cc, d, e, nvec, prio); return DPM_OK;
  }
  return dpm_set_error(DPM_ERR_ARG, "floor: rows must be 1, 2 or 4");
}
}  // namespace

static int floor_launch_ev(const dpm_floor_desc* f, const void* a, const void* b, const void* c, void* d, void* e,
                           int64_t nbytes, void* stream, void* ev_start, void* ev_stop) {
  if (!f || !a || !b || !c || !d || !e || nbytes < 16) return dpm_set_error(DPM_ERR_ARG, "floor: bad arguments");
  if (f->block != 256 && f->block != 512 && f->block != 1024) return dpm_set_error(DPM_ERR_ARG, "floor: block must be 256, 512 or 1024");
  if (f->prio < 0 || f->prio > 3 || f->store < 0 || f->store > 2 || f->load_path < 0 || f->load_path > 1)
    return dpm_set_error(DPM_ERR_ARG, "floor: prio 0..3, store 0..2, load_path 0..1");
  for (const void* p : {a, b, c, (const void*)d, (const void*)e})
    if (!aligned(p, 16)) return dpm_set_error(DPM_ERR_ALIGN, "floor: buffers must be 16-byte aligned");
  const int64_t nvec = nbytes / 16;
  const int rows = f->rows > 0 ? f->rows : 1;
  if (f->load_path == 1 && (size_t)(f->block / 64) * 3 * rows * 1024 > 65536)
    return dpm_set_error(DPM_ERR_ARG, "floor: LDS-DMA rows of a %d-thread workgroup at %d rows exceed 64 KiB", f->block, rows);
  const int64_t per = f->block / 256;
  const int64_t nsteps = (nvec + 256 * rows - 1) / (256 * rows);
  int64_t blocks = (nsteps + per - 1) / per;
  const DeviceInfo& di = device_info();
  if (f->blocks_per_cu > 0) {
    const int64_t cap = (int64_t)(di.n_cu > 0 ? di.n_cu : 256) * f->blocks_per_cu;
    if (blocks > cap) blocks = cap;
  }
  const LaunchCtx ctx{static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)};
  const u32x4 *pa = (const u32x4*)a, *pb = (const u32x4*)b, *pc = (const u32x4*)c;
  u32x4 *pd = (u32x4*)d, *pe = (u32x4*)e;
  const dim3 grid((unsigned)blocks), block((unsigned)f->block);
  int rc = f->load_path == 1 ? floor_rows<1>(rows, f->nt, f->store, grid, block, ctx, pa, pb, pc, pd, pe, nvec, f->prio)
                             : floor_rows<0>(rows, f->nt, f->store, grid, block, ctx, pa, pb, pc, pd, pe, nvec, f->prio);
  if (!rc) {
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) rc = dpm_set_error((int)he, "floor launch failed: %s", hipGetErrorString(he));
  }
  return rc;
}

extern "C" int dpm_floor_launch(const dpm_floor_desc* f, const void* a, const void* b, const void* c, void* d, void* e,
                                int64_t nbytes, void* stream, float* ms) {
  if (!ms) return floor_launch_ev(f, a, b, c, d, e, nbytes, stream, nullptr, nullptr);
  void **starts = nullptr, **stops = nullptr;
  int rc = dpm_timing_begin(1, &starts, &stops);
  if (rc) return rc;
  rc = floor_launch_ev(f, a, b, c, d, e, nbytes, stream, starts[0], stops[0]);
  int rc2 = dpm_timing_end(1, starts, stops, stream, rc ? nullptr : ms, nullptr);
  return rc ? rc : rc2;
}

// the floor kernel in a trace slot (no synchronisation): what tools/floor.py puts into the stage kernel's place inside a loop
extern "C" int dpm_floor_launch_traced(const dpm_floor_desc* f, const void* a, const void* b, const void* c, void* d, void* e,
                                       int64_t nbytes, void* stream, dpm_trace* t, int slot) {
  if (!t || slot < 0 || slot >= t->cap) return dpm_set_error(DPM_ERR_ARG, "floor_launch_traced: slot %d outside the trace", slot);
  const int rc = floor_launch_ev(f, a, b, c, d, e, nbytes, stream, t->starts[slot], t->stops[slot]);
  if (!rc) t->used[(size_t)slot] = 1;
  return rc;
}

// ---- page touch: one 4-byte load per `stride` bytes of each buffer (dpm_pagetouch_launch)
namespace {
struct TouchTab {
  const unsigned char* p[8];
  int64_t n[8];  // touches per buffer
};
__global__ __launch_bounds__(256) void pagetouch_kernel(const TouchTab tab, int n_buf, int64_t stride) {
  for (int r = 0; r < n_buf; ++r) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tab.n[r]; i += (int64_t)gridDim.x * 256) {
      const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(tab.p[r] + i * stride));
      asm volatile("" ::"v"(v));
    }
  }
}
}  // namespace

extern "C" int dpm_pagetouch_launch(const void* const* bufs, const int64_t* bytes, int n_buf, int64_t stride, void* stream) {
  if (!bufs || !bytes || n_buf < 0 || n_buf > 8 || stride < 4 || stride % 4) return dpm_set_error(DPM_ERR_ARG, "pagetouch: bad arguments");
  TouchTab tab;
  std::memset(&tab, 0, sizeof tab);
  int k = 0;
  int64_t most = 0;
  for (int i = 0; i < n_buf; ++i) {
    if (!bufs[i] || bytes[i] < 4) continue;
    if (!aligned(bufs[i], 4)) return dpm_set_error(DPM_ERR_ALIGN, "pagetouch: buffer %d is not 4-byte aligned", i);
    tab.p[k] = static_cast<const unsigned char*>(bufs[i]);
    tab.n[k] = (bytes[i] - 4) / stride + 1;
    most = std::max(most, tab.n[k]);
    ++k;
  }
  if (!k) return DPM_OK;
  int64_t blocks = (most + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(pagetouch_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tab, k, stride);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "pagetouch launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}
