// dpm_device.hpp -- gfx950 (MI355X, CDNA4) device code of the DPM-Solver engine: element types, the fused stage
// kernels (streaming + dynamic thresholding) and their launch plumbing, in five parts included at the end of this
// file (dpm_access.hpp, dpm_stage_kernel.hpp, dpm_thresh_{common,select,kernel}.hpp, dpm_aux_kernels.hpp, dpm_launch.hpp).  Included by
// two translation units per (state dtype, eps dtype) pair (dpm_stage_*.hip) so the instantiation matrix compiles in
// parallel, and by dpm_kernels.hip (C ABI entry points, add_noise, adaptive error norm, calibration).
//
// One fused, HBM-streaming kernel per solver stage:
//
//   raw network output(s) --CFG blend / classifier term--> --x_start|v|score -> eps--> --eps -> x0-->
//   --dynamic thresholding--> mn   ;   x_out = exponential-integrator combination of x, mn, h1, h2
//
// Memory-bound (~0.5 flop/B): no MFMA.  What matters is (i) 16-byte coalesced accesses -- each lane moves
// 8 consecutive elements per tensor per iteration, a wavefront 512 contiguous elements, (ii) all loads of
// an iteration issued before the first use, (iii) >= 2048 groups of 256 lanes (in workgroups of one or two) so every CU
// holds 8 waves per SIMD, (iv) the per-stage scalars arrive as kernel arguments, i.e. in SGPRs via the scalar
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

#ifndef DPM_LAB
#define DPM_LAB 0   // 1: the lab build (include/dpm_lab.h): process-global tuning knobs, fault injection, experiments
#endif

// Round 6 (tools/big_single.py, profiles/r06_big_single.md): plain sample() on ONE [8192,4,64,64] fp16 tensor ran its 1.3 GB
// stages at 0.68-0.72 of the HBM peak through the capped, looping single-request shape and at 0.77-0.79 through the fused
// kernel's uncapped one.  Kernel-only with inputs from HBM, single shape -> fused shape: 65536 tiles 206.7 -> 185.6 us (fp16),
// 449.9 -> 411.1 (fp32), 408.7 -> 370.7 (fp32 state, fp16 network); 32768 tiles 100.0 -> 94.3 / 209.2 -> 191.1 / 180.4 -> 171.0;
// 16384 tiles 51.8 -> 50.0 / 97.2 -> 94.7 / 88.2 -> 86.7; at 8192 tiles and below the two are within the +-3 % of the
// measurement (and the single shape is ahead at 2048: fp32 + fp16 14.2 against 15.1 us).
#ifndef DPM_BIG_TILES_DEFAULT
#define DPM_BIG_TILES_DEFAULT 16384
#endif
namespace dpmk {
// Launch-shape parameters.  In the PRODUCT build this is a constant: every launch starts from the defaults below (the
// measured best) and takes what the caller may choose per call from dpm_launch_opts -- no process-global mutable state.
// The LAB build keeps one mutable instance behind dpm_tuning_set / dpm_tuning_get for the tuning tools and the tests.
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
  int thr_predict = 1;      // clustered thresholding: predict the select bound from the previous stages (thr_hint)
  int thr_spin_limit = 1 << 12;  // clustered thresholding: polls before a wait on a peer gives up (THR_SPIN_LIMIT)
  int block_threads = 0;    // streaming kernel: threads per workgroup (256 / 512); 0 = by size (launch_stream)
  int big_tiles = DPM_BIG_TILES_DEFAULT;  // a single launch of at least this many 2048-element tiles takes the fused kernel's
                            // launch shape (one workgroup per super-tile, no grid-stride loop, XCD-contiguous tiles for 2-byte
                            // states): dpm_stage_launch hands it to the multi-request launcher as a group of one; 0 = never
  int lds_dma = -1;         // lone-launch north-star kernels: read streams by LDS-DMA (1) / through registers (0); -1 = default
  int force_generic = 0;    // LAB: take the run-time-prologue kernels (SPEC_GENERIC / HOT 3) where a compile-time one exists (A/B)
  int thr_elect = -1;       // clustered thresholding: one elected reducer per sample (1) / every workgroup reads every slot (0)
  int thr_stagger = 0;      // LAB: start offset between thresholding clusters (ThrParams.stagger)
  int thr_debug_fault = 0;  // LAB ONLY (fault injection; compiled out of the product kernels): 1 = every cluster wait gives
                            // up at its first unsuccessful poll, 2 / 3 = workgroup 1 of every cluster takes no part from the
                            // start, with / without marking the sample
};
#if DPM_LAB
extern Tuning g_lab_tuning;  // defined in csrc/lab/dpm_lab.hip
inline Tuning base_tuning() { return g_lab_tuning; }
#else
inline Tuning base_tuning() { return Tuning{}; }
#endif
// the parameters of ONE call: the defaults (lab: the knobs) + the caller's dpm_launch_opts
inline Tuning tuning_for(const dpm_launch_opts* o) {
  Tuning t = base_tuning();
  if (o) {
    if (o->cluster_in_graph) t.cluster_in_graph = 1;
    if (o->no_fuse) t.multi_fuse = 0;
    if (o->thr_spin_limit > 0) t.thr_spin_limit = o->thr_spin_limit;
  }
  return t;
}
// Per-device context of the library -- the only state that outlives a call: the chain of clustered thresholding launches
// of the device (see launch_thresh) and the host-mapped word a clustered kernel raises when one of its waits timed out
// (and was recovered from: diagnostics only).  One instance per device ordinal, created on first use.
struct DeviceContext {
  std::mutex mu;
  hipEvent_t ev = nullptr;      // recorded behind the device's last eager clustered launch
  bool recorded = false;
  uint32_t* fault = nullptr;    // host-mapped (portable) diagnostics word; nullptr until created
};
DeviceContext& device_context(int dev);                   // defined in dpm_kernels.hip
uint32_t* cluster_fault_word(int dev, bool create);       // create = false never allocates (stream capture)
// bfloat16 storage (a named type with external linkage: it is a template argument of functions shared between
// translation units, dpm_catchall_* below)
struct bf16_t {
  uint16_t v;
};
}  // namespace dpmk
using dpmk::Tuning;
using dpmk::tuning_for;
using dpmk::DeviceContext;
using dpmk::device_context;
using dpmk::cluster_fault_word;

// The catch-all kernels of a dtype pair (run-time form / guidance: the one-element-per-lane stage kernel and the general
// thresholding kernel) are compiled in ONE of the pair's two translation units (dpm_stage_<pair>.hip defines
// DPM_CATCHALL_HOME); the sibling unit launches them through their host-side handles.
template <typename TS, typename TE>
const void* dpm_catchall_thresh();
template <typename TS, typename TE, bool DYN>
const void* dpm_catchall_scalar();

#include "dpm_access.hpp"
#include "dpm_stage_kernel.hpp"
#include "dpm_thresh_common.hpp"
#include "dpm_thresh_select.hpp"
#include "dpm_thresh_kernel.hpp"
#include "dpm_aux_kernels.hpp"
#include "dpm_launch.hpp"
