/*
 * dpm_lab.h -- measurement, tuning and experiment entry points of the LAB build (libdpm_lab.so).
 *
 * The lab library is the product library's sources compiled with -DDPM_LAB=1 plus csrc/lab/: it exports everything
 * include/dpm_hip.h declares AND what is declared here.  tools/ and the fault-injection tests load it (through
 * DPM_SOLVER_AMD_LIB); nothing a sampler calls lives here, and the product library (libdpm_hip.so) exports none of it:
 *   - process-global launch-shape knobs (dpm_tuning_*) incl. fault injection into the clustered thresholding kernels;
 *   - event-bracketed launches (dpm_stage_launch_timed, dpm_trace_*);
 *   - no-arithmetic memory-system kernels: dpm_calib_launch (stream patterns) and dpm_floor_launch (the floor of the
 *     lone 3-read + 2-write launch over load path, bytes in flight, workgroup shape, wave priority);
 *   - rejected experiments kept for their records: dpm_prefetch_launch / dpm_pagetouch_launch (profiles/r03_in_loop.md,
 *     r05_lone_floor.md), dpm_resident_* (profiles/r04_resident.md).
 */
#ifndef DPM_LAB_H
#define DPM_LAB_H

#include "dpm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1: this library is the lab build */
DPM_API int dpm_lab_build(void);

/* ---- launch-shape knobs, PROCESS-GLOBAL and unsynchronised (lab only: a tool sets them around its own launches) ----- */
enum {
  DPM_TUNE_UNROLL = 0, DPM_TUNE_NONTEMPORAL = 1, DPM_TUNE_BLOCKS_PER_CU = 2, DPM_TUNE_ASSUME_RESIDENT = 3,
  DPM_TUNE_MULTI_FUSE = 4,          /* 1 (default): dpm_stage_launch_multi fuses; 0: one launch per request          */
  DPM_TUNE_MULTI_BLOCKS_PER_CU = 5, /* grid cap of the fused launch, workgroups per CU; 0 (default) = no cap      */
  DPM_TUNE_CLUSTER_IN_GRAPH = 6,    /* as dpm_launch_opts.cluster_in_graph                                          */
  DPM_TUNE_CLUSTER_ONE_HOP = 7,     /* 0: clusters skip the single-exchange select (testing the general route); 2: run it
                                       and reject its answer                                                         */
  DPM_TUNE_MULTI_XCD_REMAP = 8,     /* fused launch gives every XCD one contiguous eighth of the tiles: 1 on, 0 off,
                                       -1 (default) on for 2-byte states only (measured +1.4 % fp16, -3 % fp32)    */
  DPM_TUNE_THR_PREDICT = 9,         /* 1 (default): clustered thresholding launches predict the select bound from
                                       dpm_buffers.thr_hint; 0: the hint is still maintained but never used          */
  DPM_TUNE_THR_SPIN_LIMIT = 10,     /* as dpm_launch_opts.thr_spin_limit; default 4096                                */
  DPM_TUNE_THR_DEBUG_FAULT = 11,    /* fault injection.  1: every cluster wait gives up at its first unsuccessful poll;
                                       2 / 3: workgroup 1 of every cluster takes no part in its cluster from the start,
                                       with / without marking its samples (the peers see the mark / run out of polls).
                                       Results must not change.  0 (default): off                                     */
  DPM_TUNE_BLOCK_THREADS = 12,      /* streaming stage kernel: threads per workgroup, 256 / 512; 0 (default): by size  */
  DPM_TUNE_FORCE_GENERIC = 13,      /* 1: launch the run-time-prologue kernels (SPEC_GENERIC / thresholding HOT 3) also where a
                                       compile-time specialisation exists: the A/B behind the kernel-count budget        */
  DPM_TUNE_LDS_DMA = 15,            /* lone 2-byte 2M / first-order launch: read streams by LDS-DMA (1) or through registers (0);
                                       -1 (default): through registers, like the product (measured equal: r05_lone_floor.md) */
  DPM_TUNE_THR_STAGGER = 16,        /* clustered thresholding: cluster g starts (g % groups) * ticks x 0.1 us late; value = groups << 16
                                       | ticks (groups 0 = 2); 0 (default): off (profiles/r05_thresholding.md)         */
  DPM_TUNE_BIG_TILES = 17,          /* a single stage launch of at least this many 2048-element tiles is launched with the fused
                                       kernel's shape (as a group of one request); 0: never; default: the library's            */
  DPM_TUNE_THR_ELECT = 14           /* clustered thresholding: 1 = one elected reducer per sample selects on the union and
                                       publishes the result (k slot reads per sample), 0 = every workgroup reads every slot
                                       (k^2); -1 (default): the library's choice                                        */
};
DPM_API int dpm_tuning_set(int knob, int value);
DPM_API int dpm_tuning_get(int knob);

/* ---- event-bracketed launches: hipExtLaunchKernelGGL start / stop events around the kernel itself ---------------- */
DPM_API int dpm_stage_launch_timed(const dpm_stage* st, const dpm_buffers* b, void* stream, float* ms);
/* kernel durations INSIDE a real loop, without synchronising between launches: a trace owns `capacity` start/stop event
   pairs; dpm_stage_launch_traced is dpm_stage_launch with the pair of `slot` bracketing the kernel itself,
   dpm_trace_read synchronises the stream once and fills ms[0..n) (-1 for slots never used). */
typedef struct dpm_trace dpm_trace;
DPM_API int dpm_trace_create(int capacity, dpm_trace** out);
DPM_API int dpm_stage_launch_traced(const dpm_stage* st, const dpm_buffers* b, void* stream, dpm_trace* t, int slot);
DPM_API int dpm_trace_read(dpm_trace* t, void* stream, float* ms, int n);
DPM_API void dpm_trace_destroy(dpm_trace* t);

/* ---- memory-system calibration with no arithmetic ------------------------------------------------------------------
   kind 0: copy; kind 1: 3 read + 2 write streams, the 2M stage's pattern; kind 2: 4 read + 1 write streams, `e` read;
   nbytes per stream; block in {256,512,1024}; nt mask: bit 0 nt loads, bits 1 / 2 nt store of d / e (plain builds only);
   ms (optional) = kernel time by events. */
DPM_API int dpm_calib_launch(int kind, int block, int blocks_per_cu, int nt, const void* a, const void* b, const void* c,
                     void* d, void* e, int64_t nbytes, void* stream, float* ms);

/* The FLOOR of the lone 2M launch: 3 read + 2 write streams of nbytes each (a, b, c -> d = a ^ b, e = b ^ c), no
   arithmetic, swept over what bounds a 42 MB launch that starts cold behind a network's last kernel:
     load_path  0: global_load_dwordx4 into registers (the stage kernel's path)
                1: LDS-DMA -- global_load_lds_dwordx4 into the wavefront's own LDS rows, ds_read_b128 back
                   (MI355X_MICROARCH.md: an all-LDS-DMA prologue burst runs ~12-13 B/cyc/CU against ~10-11 for mixed loads)
     rows       16-byte loads per lane and stream issued before the first use: 1, 2, 4 (bytes in flight per CU =
                rows x 3 streams x 16 B x resident lanes)
     block      threads per workgroup: 256, 512, 1024; every 256-lane group walks tiles of its own (as the stage kernel)
     blocks_per_cu  grid cap in workgroups per CU (0 = one tile-row set per 256-lane group, no loop)
     nt         1: streaming (non-temporal) loads
     prio       s_setprio value the wavefronts start with (0 .. 3): priority against the tail of the previous kernel
     store      0: write-through stores (sc0 sc1, the product's), 1: plain stores, 2: non-temporal stores
   ms (optional) = kernel time by events.  Results are checked by tools/floor.py (d == a ^ b, e == b ^ c). */
typedef struct dpm_floor_desc {
  int32_t load_path, rows, block, blocks_per_cu, nt, prio, store, reserved;
} dpm_floor_desc;
DPM_API int dpm_floor_launch(const dpm_floor_desc* f, const void* a, const void* b, const void* c, void* d, void* e, int64_t nbytes,
                     void* stream, float* ms);
/* the same launch bracketed by the event pair of trace slot `slot` (no synchronisation: inside a loop) */
DPM_API int dpm_floor_launch_traced(const dpm_floor_desc* f, const void* a, const void* b, const void* c, void* d, void* e,
                            int64_t nbytes, void* stream, dpm_trace* t, int slot);

/* ---- side-stream helpers measured against the lone launch's ramp-up (neither is used by any loop of the library) ----
   dpm_prefetch_launch: read `n_buf` device buffers (bytes[i] each, 16-byte aligned) and discard the data (policy 0:
   default loads, 1: streaming loads): rejected in round 3 (profiles/r03_in_loop.md).
   dpm_pagetouch_launch: ONE 4-byte load per `stride` bytes (4096 = one per page) of each buffer -- kilobytes, not
   megabytes: warms the TLB and the first-byte path without moving the data (round 5, profiles/r05_lone_floor.md). */
DPM_API int dpm_prefetch_launch(const void* const* bufs, const int64_t* bytes, int n_buf, int policy, void* stream);
DPM_API int dpm_pagetouch_launch(const void* const* bufs, const int64_t* bytes, int n_buf, int64_t stride, void* stream);

/* ---- EXPERIMENT (rejected, profiles/r04_resident.md): a resident stage kernel woken by a stream-ordered write --------
   One launch per trajectory on a side stream keeps `workgroups` workgroups on the chip (capped at what is co-resident);
   per stage the host enqueues dpm_resident_signal behind the network's last kernel.  Covers the unguided 20-step
   DPM-Solver++(2M) trajectory: noise-prediction network, forms LIN1 / TWO, equal fp16 or fp32 dtypes, n a multiple of
   2048.  Waits are bounded (spin limit + abort word). */
DPM_API int dpm_resident_create(const dpm_stage* stages, const dpm_buffers* bufs, int n_stages, int workgroups, int sleep, void** out);
DPM_API int dpm_resident_start(void* handle, const void* x_first, void* x_last_out, void* side_stream);
DPM_API int dpm_resident_signal(void* handle, int stage, const void* eps, void* stream);
DPM_API void dpm_resident_destroy(void* handle);

/* ---- per-device context probe (tests): the address of the library's context of device `dev` (the chain of clustered
   launches, the diagnostics word), so that a host-only test can check devices do not share one */
DPM_API const void* dpm_lab_device_context(int dev);

#ifdef __cplusplus
}
#endif
#endif /* DPM_LAB_H */
