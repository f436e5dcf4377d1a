// dpm_thresh_common.hpp -- dynamic thresholding (ref :416-425), part 1 of 3: the constants of the LDS / workspace layout,
// ThrParams / ThrTab, the 4-element accessors and the primitives of the cluster protocol (bounded waits, give-up, barrier).
// Part of dpm_device.hpp (include that); dpm_thresh_select.hpp and dpm_thresh_kernel.hpp build on it.
#pragma once

namespace {

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
// wavefronts per SIMD the run-time dispatched catch-all thresholding kernel (HOT = 0) is compiled for: 4 = two workgroups per
// CU at a 128-register budget (a handful of spills to scratch), 2 = no register limit, one workgroup per CU
#ifndef DPM_THR_CATCHALL_WAVES
#define DPM_THR_CATCHALL_WAVES 4
#endif
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
constexpr int THR_SLOTW = THR_SLOT_CAP + THR_SLOT_HDR;  // (128-byte aligned slots, 288 words, were measured: no difference)
constexpr uint32_t THR_TAG = 0x80000000u;    // |x0| bit patterns have bit 31 clear: a tagged word is never 0
constexpr uint32_t THR_OVERFLOW = 0x40000000u;
constexpr int THR_WS_DONE = THR_WS_CNT + 12; // workgroups of the cluster that are through with the workspace
constexpr int THR_HINT_W = DPM_THR_HINT_WORDS;
// the predicted bound sits this far below the extrapolated order statistic: with the statistic within a few percent of
// its extrapolation the union stays ~1.3 K entries (K = the wanted rank from the top) and holds the K-th largest
constexpr float THR_HINT_MARGIN = 0.94f;
constexpr int THR_WS_POISON = THR_WS_CNT + 16; // a workgroup of the cluster is out of the protocol on this sample (see give_up)
// LAB build, elected reducer (ThrParams.elect): the verdict workgroup 0 of a cluster publishes for its peers -- per slot area
// (searched: + 0, predicted: + 4) [0] tag | a, [1] tag | b, [2] tag | valid
constexpr int THR_WS_RESULT = THR_WS_CNT + 24;
// polls (a microsecond or two each: a dependent sc1 load + s_sleep) before a wait on a peer gives up -- milliseconds.
// Giving up is safe (the workgroup then computes the sample's order statistics alone, solo_select), so the limit only
// trades a stall against redundant work when the peers are off the chip (ThrParams.spin_limit, DPM_TUNE_THR_SPIN_LIMIT)
constexpr uint32_t THR_SPIN_LIMIT = 1u << 12;

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
  int32_t bpr;      // > 0: the launch fuses several requests of bpr samples each (ThrTab); sample s belongs to request s / bpr
  int32_t slot_pub; // entries of a slot that are always written (values, then the bare tag)
  int32_t slot_cap; // values per workgroup slot: a power of two <= THR_SLOT_CAP with k * slot_cap <= THR_CAP
  int32_t slot_shift; // log2(slot_cap)
  int32_t debug_reject; // testing: run the single-exchange select but always take the general route afterwards
  int64_t ws_stride; // words per sample in ws
  uint32_t* ws;    // k > 1: batch x ws_stride words, all zero between launches (the kernel cleans up after itself)
  uint32_t* fault; // host-mapped word: set when a cluster wait timed out and was recovered from (diagnostics only:
                   // dpm_cluster_timeout_poll; the launch's results are correct either way)
  uint32_t spin_limit;  // polls before a wait on a peer gives up (THR_SPIN_LIMIT)
#if DPM_LAB
  int32_t debug_fault;  // LAB build only (DPM_TUNE_THR_DEBUG_FAULT): 2 / 3 = workgroup 1 of every cluster takes no part in its cluster
  int32_t elect;        // LAB build only (DPM_TUNE_THR_ELECT): workgroup 0 of a cluster reads the k slots, selects on the union
                        // and publishes the verdict; its peers make one wait and read three words -- k slot reads per sample
                        // instead of k^2 (profiles/r05_thresholding.md)
  int32_t stagger;      // LAB build only (DPM_TUNE_THR_STAGGER): cluster g starts (g % groups) * ticks of 0.1 us late -- low 16
                        // bits = ticks, high bits = groups (0 = 2): clusters that walk several large samples stay out of
                        // phase, so that one group streams while another selects (profiles/r05_thresholding.md)
#endif
  float* hint;     // dpm_buffers.thr_hint (THR_HINT_W floats per sample) or null: the selected order statistic of the
                   // previous two stages -> predicted select bound of this one (cluster_select_once, `pbound`)
  int32_t hint_reset; // this is the first stage of a trajectory: the stored values are stale, overwrite without reading
  int32_t hint_predict; // 0: maintain the hint but do not use it (DPM_TUNE_THR_PREDICT)
#ifdef DPM_THR_TIMING  // (lab build only)
  uint64_t* tdbg;  // 16 timestamps per workgroup (tools/thr_timeline.py)
#endif
};

// pointer table of a fused multi-request thresholding launch (a kernel argument, like MultiTab): request r's tensors
// and its workspace
struct ThrTab {
  const void* x[MULTI_MAX];
  const void* e0[MULTI_MAX];
  const void* e1[MULTI_MAX];
  const void* h1[MULTI_MAX];
  const void* h2[MULTI_MAX];
  void* xo[MULTI_MAX];
  void* mo[MULTI_MAX];
  uint32_t* ws[MULTI_MAX];
};

// torch.quantile gives NaN for a sample that holds a NaN ANYWHERE (ref :420), whatever the rank asked for.  Phase 1 stores
// every NaN as the one pattern 0x7fffffff (canon_nan): the largest |x0| key there is and the only one with the top digit
// 0x7ff, so "the sample holds a NaN" is "bin 2047 of a level-0 histogram (of the elements, or of the per-thread maxima) is
// not empty" (locate_bin with a nan_tag) or "the largest chunk maximum is that pattern" (cluster_select_once) -- words every
// route already merges across the cluster.  The verdict is misc[THR_NANW] == the sample's tag (sample index + 1: never
// reset, a stale tag of an earlier sample does not match).
constexpr uint32_t THR_NAN_KEY = 0x7fffffffu;
constexpr int THR_NANW = 15;
__device__ __forceinline__ float canon_nan(float v) { return v != v ? __uint_as_float(THR_NAN_KEY) : v; }

// torch.clamp(v, -s, s) (ref :424): a NaN element stays NaN (fminf / fmaxf return their other operand), a NaN bound makes
// the division that follows NaN anyway.  Non-finite values: INTEGRATION.md, behavioural notes.
__device__ __forceinline__ float clamp_ref(float v, float s) {
  const float c = fminf(fmaxf(v, -s), s);
  return v != v ? v : c;
}

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

// A wait on another workgroup gives up after ThrParams.spin_limit polls (milliseconds).  Within one process the peers of a
// cluster are co-resident by construction (grid capped at the occupancy, clustered launches chained), so this happens
// when something else keeps them off the chip: another PROCESS running clusters on the same GPU, two clustered graphs
// replayed concurrently, a kernel of another stream holding the CUs.  Giving up is harmless and local.  The workgroup
// LEAVES the cluster protocol for the rest of the launch: it writes nothing more into any sample's workspace, arrives at
// no further barrier, and computes the order statistics of its samples ALONE from global memory (solo_select) -- the
// same bits as without the timeout.  What it contributed before (always complete: every contribution precedes the
// wait it belongs to) stays valid for the peers; peers that wait for something it will no longer deliver give up in
// turn -- at once when they see the sample's THR_WS_POISON mark, which only shortens their wait, after their own polls
// otherwise.  Nothing is reported to the caller except the host-mapped diagnostic word (dpm_cluster_timeout_poll).
// The reference cannot fail here (ref :416-425); neither can this.
// (Round 4 first let such a workgroup run on through the protocol with whatever it had read and discard its result:
// the forced-fault sweeps found a neighbouring chunk changed once in a few thousand launches and workspace words left
// dirty, profiles/r04_thresholding.md.)
__device__ __forceinline__ void raise_fault(uint32_t* fault) {
  if (fault) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// `dead` (LDS word): this workgroup gave up a wait -- it is out of the protocol for the rest of the launch
__device__ __forceinline__ void give_up(uint32_t* dead, uint32_t* poison, uint32_t* fault) {
  *dead = 1u;
  raise_fault(fault);
  __hip_atomic_store(poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the peers need not poll to the end
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (landed long before this workgroup reports itself done)
}
// the `spins`-th unsuccessful poll of a wait: time to give up?  The mark of a peer that did is looked at every 8th poll, by
// the workgroup's first wavefront only (`looks`): the others learn it through the LDS word give_up sets, and a wait in which
// only they are left runs to its own limit (every wavefront looking cost 3 % more read traffic at cfg5's size).
__device__ __forceinline__ bool wait_is_over(uint32_t spins, const uint32_t* poison, const ThrParams& tp, bool looks = true) {
  if (spins > tp.spin_limit) return true;
  return looks && (spins & 7u) == 0u && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// all workgroups of a cluster meet here; `cnt` is a zero-initialised single-use counter.  Everything the cluster
// shares travels as agent-scope atomics and sc1 loads, so no cache write-back / invalidate is needed: drain this
// wave's atomics, arrive with a relaxed atomic, poll with relaxed sc1 loads (MI355X_MICROARCH.md, barrier-counter).
// Returns false when the wait was given up (`dead`, an LDS word, is set): the caller leaves the cluster protocol -- it
// contributes nothing and arrives nowhere from then on.  Hence the invariant the peers rely on: a counter that reaches
// k was reached by k workgroups that were each still in the protocol, with every contribution of theirs drained.
__device__ __forceinline__ bool cluster_barrier(uint32_t* cnt, uint32_t k, uint32_t* dead, uint32_t* ws, const ThrParams& tp) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
      __builtin_amdgcn_s_sleep(2);
      if (wait_is_over(++spins, ws + THR_WS_POISON, tp)) {
        give_up(dead, ws + THR_WS_POISON, tp.fault);
        break;
      }
    }
  }
  __syncthreads();
  return *dead == 0u;
}

}  // namespace
