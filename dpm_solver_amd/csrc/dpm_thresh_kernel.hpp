// dpm_thresh_kernel.hpp -- dynamic thresholding (ref :416-425): exact order statistics in LDS, workgroup clusters,
// stage_thresh_kernel (part of dpm_device.hpp; include that)
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
// `part` (optional: T zeroed words, visible like the sentinels): short lists -- the union of a small cluster is ~120
// entries, two wavefronts' worth of candidates -- are counted by ALL eight wavefronts: the 64-candidate groups are
// replicated over the wavefronts and every replica counts against its own part of the list (the counting is the longest
// single step of a small cluster's select: nc compare-and-add pairs per candidate), the partial counts meet in `part`
// through LDS atomics; one more barrier.
template <int T>
__device__ __forceinline__ void rank_count(const uint32_t* cand, uint32_t nc, uint32_t rank, uint32_t* misc, int tid,
                                           uint32_t* part = nullptr) {
  static_assert(T / 64 == 8, "eight wavefronts");
  const uint32_t groups = (nc + 63u) >> 6;  // wavefronts' worth of candidates
  const uint32_t gp = groups <= 1u ? 1u : (groups <= 2u ? 2u : (groups <= 4u ? 4u : 8u));
  const uint32_t P = part ? 8u / gp : 1u;   // replicas of every group = parts of the list
  uint32_t lt = 0u;
  if (P > 1u) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t g = wave & (gp - 1u), pr = wave / gp;
    const uint32_t ci = g * 64u + (uint32_t)(tid & 63);
    if (g * 64u < nc) {
      const uint32_t my = ci < nc ? cand[ci] : 0xffffffffu;
      const uint32_t nchunk = (nc + 31u) >> 5, per = (nchunk + P - 1u) / P;
      const uint32_t j1 = (pr + 1u) * per < nchunk ? (pr + 1u) * per : nchunk;
      for (uint32_t j = pr * per * 32u; j < j1 * 32u; j += 32u) {
        u32x4 q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = *reinterpret_cast<const u32x4*>(cand + j + 4 * e);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          lt += (q[e][0] < my ? 1u : 0u) + (q[e][1] < my ? 1u : 0u) + (q[e][2] < my ? 1u : 0u) + (q[e][3] < my ? 1u : 0u);
      }
      if (ci < nc && lt) atomicAdd(&part[ci], lt);
    }
    __syncthreads();
  }
  if ((uint32_t)(tid & ~63) < nc) {  // wavefronts beyond the list have nothing to do
    const uint32_t my = (uint32_t)tid < nc ? cand[tid] : 0xffffffffu;
    if (P > 1u) {
      lt = (uint32_t)tid < nc ? part[tid] : 0u;
    } else {
      for (uint32_t j = 0; j < nc; j += 32) {  // broadcast reads, eight in flight
        u32x4 q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = *reinterpret_cast<const u32x4*>(cand + j + 4 * e);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          lt += (q[e][0] < my ? 1u : 0u) + (q[e][1] < my ? 1u : 0u) + (q[e][2] < my ? 1u : 0u) + (q[e][3] < my ? 1u : 0u);
      }
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

// scratch of rank_count's partial counts inside cand[]: behind the longest list rank_select is called with (T entries + 32
// sentinels)
constexpr int THR_PART_OFF = 2048;

template <int T>
__device__ __forceinline__ void rank_select(uint32_t* cand, uint32_t nc, uint32_t rank, uint32_t* misc, int tid) {
  if (tid == 0) {
    misc[6] = 0u;
    misc[7] = 0u;
  }
  if (tid < 32) cand[nc + tid] = 0xffffffffu;
  const bool split = nc <= (uint32_t)(T / 2);  // (workgroup-uniform) lists that leave wavefronts without candidates
  if (split) cand[THR_PART_OFF + tid] = 0u;
  __syncthreads();
  rank_count<T>(cand, nc, rank, misc, tid, split ? cand + THR_PART_OFF : nullptr);
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
                                                    int tid, uint32_t& a_out, uint32_t& b_out, bool stamp,
                                                    uint32_t* poison, const uint32_t pbound = 0u) {
  // pbound != 0 (bit pattern of a positive float): the bound is PREDICTED from the previous stages' thresholds (same value
  // in every workgroup of the cluster) instead of searched in the histogram of the per-thread maxima: no histogram, no
  // locate_bin, and a union of ~1.3 K entries instead of k * quota.  Every element >= pbound of every chunk is published,
  // so the answer is exact whenever the union holds at least K entries; fewer (the prediction was too high), a slot
  // overflow or a union beyond the list capacity (too low) fail the attempt exactly like the searched bound does.
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
  // 2. histogram of one value per thread, bound = digit of the quota-th largest maximum (or the predicted bound)
  uint32_t bin_lo = 0u;
  if (!pbound) {
    if (has) atomicAdd(&hist[digit(m1)], 1u);
    __syncthreads();
    const int P = vec ? (n + 3) / 4 : n;  // threads that produced at least one element
    const uint32_t Pl = (uint32_t)(P < T ? P : T);
    locate_bin<T>(hist, misc, Pl > (uint32_t)tp.quota ? Pl - (uint32_t)tp.quota : 0u, tid);
    bin_lo = Pl ? misc[0] : 0u;
  }
  // does |x0| pattern u belong to this chunk's candidates?
  auto qual = [&](uint32_t u) { return pbound ? u >= pbound : digit(u) >= bin_lo; };
  DPM_R1STAMP(8)
  // 3. this chunk's candidates: a thread's are among its four largest values unless even the fourth qualifies
  {
    const int mine = vec ? (has ? 4 * ((n - tid * 4 + T * 4 - 1) / (T * 4)) : 0) : (has ? (n - tid + T - 1) / T : 0);
    const bool c1 = mine > 0 && qual(m1), c2 = mine > 1 && qual(m2);
    const bool c3 = mine > 2 && qual(m3), c4 = mine > 3 && qual(m4);
    if (__ballot(c4 && mine > 4)) {
      if (pbound)  // digit = the whole pattern: d >= bin is u >= pbound
        (void)compact_candidates<T, true>(sx0, n, pbound, misc, cand, tid, 0, 0u);
      else
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
    // smallest |x0| this workgroup would have published: the predicted bound, or the first pattern of the digit (0 = everything)
    const uint32_t bound = pbound ? pbound : (bin_lo ? (bin_lo + dbase) << THR_FSHIFT : 0u);
    __hip_atomic_store(&mine_slot[2], cmax | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mine_slot[1], bound | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&mine_slot[0], (over ? THR_OVERFLOW : ncl) | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  DPM_R1STAMP(9)
#if DPM_LAB
  // Elected reducer (experiment): workgroup 0 of the cluster goes on alone -- it polls the slots, gathers the union, selects
  // and publishes (a, b, valid); everybody else waits for that verdict: one wait, three words.  Workgroup 0 is elected
  // statically: a ticket ("the last to arrive reduces") would put a drain + a returning atomic in front of the first poll.
  const bool elect = tp.elect != 0 && k <= 64u;
  uint32_t* verdict = poison - THR_WS_POISON + THR_WS_RESULT + (pbound ? 4 : 0);
  if (elect && c != 0) {
    if (tid == 0) {
      uint32_t r0 = 0u, r1 = 0u, r2 = 0u, spins = 0u;
      for (;;) {
        if (!(r0 & THR_TAG)) r0 = __hip_atomic_load(&verdict[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(r1 & THR_TAG)) r1 = __hip_atomic_load(&verdict[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(r2 & THR_TAG)) r2 = __hip_atomic_load(&verdict[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((r0 & r1 & r2) & THR_TAG) || misc[30]) break;
        __builtin_amdgcn_s_sleep(1);
        if (wait_is_over(++spins, poison, tp)) give_up(misc + 30, poison, tp.fault);
      }
      misc[26] = r0;
      misc[27] = r1;
      misc[29] = r2;
    }
    __syncthreads();
    const bool got = !misc[30] && ((misc[26] & misc[27] & misc[29]) & THR_TAG);
    const bool ok_v = got && (misc[29] & 1u);
    if (ok_v) {
      a_out = misc[26] & ~THR_TAG;
      b_out = misc[27] & ~THR_TAG;
    } else {  // the next attempt / the general route expect their LDS state (hist is still all zero here)
      if (tid == 0) {
        misc[4] = 0u;
        misc[9] = 0u;
        misc[10] = 0u;
        misc[12] = 0u;
      }
      __syncthreads();
    }
    return ok_v;
  }
#endif
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
      if (wait_is_over(++spins, poison, tp, tid < 64)) give_up(misc + 30, poison, tp.fault);
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
          if (wait_is_over(++spins, poison, tp, tid < 64)) give_up(misc + 30, poison, tp.fault);
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
    if (total <= 192u) {  // rank counting is quadratic but three wavefronts' worth of it beats a histogram level
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
#if DPM_LAB
  if (elect && tid == 0 && !misc[30]) {  // the reducer's verdict for its peers (a workgroup that gave up publishes nothing)
    __hip_atomic_store(&verdict[0], a_out | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&verdict[1], b_out | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&verdict[2], (valid ? 1u : 0u) | THR_TAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
  if (!valid) {  // the next attempt / the general route expect their LDS state: hist all zero, no candidates
    if (leftovers) {
      hist[tid] = 0u;
      hist[tid + T] = 0u;
    }
    if (tid == 0) {
      misc[4] = 0u;
      misc[9] = 0u;   // k > 64 accumulates these with atomics
      misc[10] = 0u;
      misc[12] = 0u;
    }
    __syncthreads();
  }
  return valid;
}

// The order statistics of a whole sample by ONE workgroup from global memory, for a workgroup whose cluster cannot be
// relied on (a wait on a peer timed out, give_up): `bits_at(i)` recomputes |x0| of element i of the sample -- the
// same prologue arithmetic as phase 1, hence the same bits --, three radix levels (11 / 11 / 9 bits) find the element of
// ascending rank `rank`, one more pass its successor.  No peers, no workspace; hist may hold anything on entry and is
// left zeroed.  Slow (four passes over the sample through L2) and rare.
template <int T, typename F>
__device__ __forceinline__ void solo_select(F&& bits_at, int n, uint32_t rank, bool need_next, uint32_t* hist,
                                            uint32_t* misc, int tid, uint32_t& a, uint32_t& b) {
#pragma unroll
  for (int j = 0; j < THR_NB / T; ++j) hist[j * T + tid] = 0u;
  __syncthreads();
  uint32_t prefix = 0u, known = 0u, cnt_sel = 0u;
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = pass == 0 ? 20 : pass == 1 ? 9 : 0;
    const uint32_t dmask = pass == 2 ? 0x1ffu : 0x7ffu;
#pragma unroll 1
    for (int i = tid; i < n; i += T) {
      const uint32_t u = bits_at(i);
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
  if (need_next && rank + 1u >= cnt_sel) {  // the successor is the smallest element above a (if any)
    if (tid == 0) misc[12] = 0x7fffffffu;
    __syncthreads();
    uint32_t m = 0x7fffffffu;
#pragma unroll 1
    for (int i = tid; i < n; i += T) {
      const uint32_t u = bits_at(i);
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

// HOT != 0: the usual configuration fixed at compile time -- 16-byte accesses legal, no mask blend, form and guidance kind
// known -- so that the load and store loops are straight-line code without the wave-uniform branches of the run-time form
// and guidance and their operands (the kernel is as sensitive to its instruction count as to HBM):
//   HOT = 1  noise-prediction network with the division by the invariant alpha, top-K front end of the select;
//   HOT = 2  the same with the full level-0 histogram (quantiles far from 1; not instantiated since round 5: the catch-all
//            kernel serves them);
//   HOT = 3  like 1, the prologue chosen per sample at run time: x_start / v / score networks and divisors that fail the
//            guard of the division by an invariant (round 4: those ran the catch-all kernel, 14-19 % slower).
// Everything else runs the same source with HOT = 0 (run-time form, guidance, evaluation state, mask blend).
template <typename TS, typename TE, int FORM, int GUIDE, bool XE, int T, int HOT>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(HOT != 0 ? 4 : DPM_THR_CATCHALL_WAVES, 4))) void stage_thresh_kernel(
    const TS* __restrict__ x_1, const TS* __restrict__ xe_1, const TE* __restrict__ e0_1, const TE* __restrict__ e1_1,
    const TE* __restrict__ g, const TS* __restrict__ h1_1, const TS* __restrict__ h2_1, TS* __restrict__ xo_1,
    TS* __restrict__ mo_1, KParams p, ThrParams tp, KExt ext, const ThrTab tab) {
  // FORM / GUIDE may be FORM_RT / GUIDE_RT (HOT = 0: the general kernel, one per dtype pair): read from p then
  const bool nx = form_needs_x<FORM>(p), nh1 = form_needs_h1<FORM>(p), nh2 = form_needs_h2<FORM>(p);
  const bool g_cfg = guide_is<GUIDE>(DPM_GUIDE_CFG, p), g_cls = guide_is<GUIDE>(DPM_GUIDE_CLASSIFIER, p);
  constexpr int BPT = THR_NB / T;  // histogram bins per thread when all threads touch the histogram
  // tile rows a thread keeps in flight in the streaming phases: THR_ROWS in the specialised kernels; ONE in the run-time
  // dispatched catch-all kernel, whose extra operands (mask blend, run-time form / guidance) otherwise push it past the
  // 128 registers two workgroups per CU leave a wavefront (round 3: 14-19 registers spilled to scratch)
  constexpr int ROWS = HOT != 0 ? THR_ROWS : 1;
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
  const bool track = HOT == 1 || HOT == 3 || (HOT == 0 && (tp.topk > 0 || route1));  // phase 1 keeps every thread's four largest |x0|
  const bool topk = track && tp.topk > 0;                                 // top-K front end of the general route
  constexpr bool FAST_ONLY = HOT == 1 || HOT == 2;  // the launch guarantees tp.fastdiv there
  const bool fastdiv = FAST_ONLY || tp.fastdiv != 0;
  const int64_t eps_stride = ext.eps_stride;
  const int grp = k == 1 ? (int)blockIdx.x : (int)(blockIdx.x / k);
  const int c = k == 1 ? 0 : (int)(blockIdx.x % k);
  const TS* mask = HOT != 0 ? nullptr : static_cast<const TS*>(ext.mask);
  const TS* ba = HOT != 0 ? nullptr : static_cast<const TS*>(ext.ba);
  const TS* bb = HOT != 0 ? nullptr : static_cast<const TS*>(ext.bb);
  TS* xo2 = static_cast<TS*>(ext.xo2);
  // misc[30]: this workgroup gave up a wait on a peer and is out of the cluster protocol (give_up).  Fault injection (LAB
  // build only, ThrParams.debug_fault 2 / 3): workgroup 1 of every cluster is out from the start, with / without the mark.
#if DPM_LAB
  const int dbg_fault = tp.debug_fault;
#else
  constexpr int dbg_fault = 0;
#endif
  if (threadIdx.x == 0) misc[30] = (k > 1 && dbg_fault >= 2 && c == 1) ? 1u : 0u;
#if DPM_LAB
  if (tp.stagger) {  // experiment: phase offset between clusters (all workgroups of a cluster wait alike)
    const uint32_t ng = (uint32_t)tp.stagger >> 16 ? (uint32_t)tp.stagger >> 16 : 2u;
    const uint64_t ticks = (uint64_t)(((uint32_t)grp % ng) * ((uint32_t)tp.stagger & 0xffffu)) * 10u;  // wall clock: 100 MHz
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  }
#endif
  for (int s_idx = grp; s_idx < tp.batch; s_idx += tp.groups) {
    // The thread index is re-materialised per sample: otherwise the compiler hoists every per-thread predicate of the
    // body (dozens of 64-bit lane masks) out of this loop, runs out of SGPRs and pays v_readlane pairs all over the select.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    // the sample's tensors: the launch's own, or -- several requests fused into one launch (tp.bpr samples each,
    // launch_typed) -- those of request s_idx / bpr from the pointer table in the kernel arguments (scalar loads)
    const TS* __restrict__ x = x_1;
    const TS* __restrict__ xe = xe_1;
    const TE* __restrict__ e0 = e0_1;
    const TE* __restrict__ e1 = e1_1;
    const TS* __restrict__ h1 = h1_1;
    const TS* __restrict__ h2 = h2_1;
    TS* __restrict__ xo = xo_1;
    TS* __restrict__ mo = mo_1;
    uint32_t* ws_base = tp.ws;
    int64_t s_loc = s_idx;
    if (tp.bpr > 0) {
      // request r = s_idx / bpr, r < MULTI_MAX = 32: five scalar compare-and-add steps instead of a division (whose
      // reciprocal set-up the compiler hoists out of the sample loop into a register that lives through the whole kernel)
      static_assert(MULTI_MAX <= 32, "five binary-search steps");
      uint32_t r = 0u;
#pragma unroll
      for (uint32_t bit = 16u; bit; bit >>= 1)
        if ((uint64_t)(r + bit) * (uint32_t)tp.bpr <= (uint64_t)(uint32_t)s_idx) r += bit;
      s_loc = (int64_t)((uint32_t)s_idx - r * (uint32_t)tp.bpr);
      x = xe = static_cast<const TS*>(tab.x[r]);
      e0 = static_cast<const TE*>(tab.e0[r]);
      e1 = static_cast<const TE*>(tab.e1[r]);
      h1 = static_cast<const TS*>(tab.h1[r]);
      h2 = static_cast<const TS*>(tab.h2[r]);
      xo = static_cast<TS*>(tab.xo[r]);
      mo = static_cast<TS*>(tab.mo[r]);
      ws_base = tab.ws[r];
    }
    const int64_t base = s_loc * tp.per_sample + (int64_t)c * tp.chunk;
    const int64_t ebase = s_loc * (eps_stride ? eps_stride : tp.per_sample) + (int64_t)c * tp.chunk;
    const int64_t left = tp.per_sample - (int64_t)c * tp.chunk;
    const int n = left <= 0 ? 0 : (left < tp.chunk ? (int)left : tp.chunk);
    uint32_t* ws = k == 1 ? nullptr : ws_base + s_loc * tp.ws_stride;
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
      // predicted select bound of this sample (clusters with the single-exchange route only): the geometric extrapolation
      // of the selected order statistic of the previous two thresholded stages, lowered by THR_HINT_MARGIN
      uint32_t pb = 0u;
      if (route1 && tp.hint && !tp.hint_reset && tp.hint_predict) {
        const float h1v = tp.hint[(int64_t)s_idx * THR_HINT_W], h2v = tp.hint[(int64_t)s_idx * THR_HINT_W + 1];
        if (h1v > 1e-30f && h2v > 1e-30f && h1v < 1e30f && h2v < 1e30f) {
          const float r = h1v / h2v;
          if (r > 0.25f && r < 4.f) pb = __float_as_uint((h1v * r) * THR_HINT_MARGIN);
        }
      }
      misc[25] = pb;
      // a workgroup that gave up a wait on an earlier sample takes no part in this one either: tell the peers
      if (k > 1 && misc[30] && dbg_fault != 3)
        __hip_atomic_store(ws + THR_WS_POISON, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      // ROWS tile rows per iteration, the loads of all of them issued before the first use: the phase is bound by
      // the bytes one workgroup keeps in flight (two workgroups per CU, and while one of them selects only one streams).
      // FAST = noise-prediction network with the division by the invariant alpha (the common case, branch-free packed
      // arithmetic: the only loop of the HOT = 1 / 2 kernels); otherwise the general prologue (x_start / v / score
      // networks, divisors that fail the guard).  HOT = 3 and the catch-all kernel hold both loops and choose once per
      // sample (v-prediction at [1024,3,64,64]: 54.5 -> 44.3 us per stage against the catch-all kernel).
      auto load_phase = [&](auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      for (int i0 = tid * 4; i0 < n; i0 += ROWS * T * 4) {
        float vx[ROWS][4], v0[ROWS][4], v1[ROWS][4], vg[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const int ir = i0 + r * T * 4 < n ? i0 + r * T * 4 : i0;  // clamped: loads are unconditional
          load4(XE ? xe : x, base + ir, vx[r]);
          load4<true>(e0, ebase + ir, v0[r]);                     // the network outputs are dead after this kernel
          if (g_cfg) load4<true>(e1, ebase + ir, v1[r]);
          if (g_cls) load4<true>(g, base + ir, vg[r]);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const int i = i0 + r * T * 4;
          if (r == 0 || i < n) {
            float o[4];
            if constexpr (FAST) {  // the common parameterisation, division by the invariant alpha (3 VALU ops for ~12)
#pragma unroll
              for (int j = 0; j < 4; j += 2) {  // adjacent pairs: packed fp32 instructions
                const f32x2 z = {0.f, 0.f};
                const f32x2 rr = prologue<GUIDE, SPEC_NOISE_X0, f32x2, TE>(
                    f32x2{vx[r][j], vx[r][j + 1]}, f32x2{v0[r][j], v0[r][j + 1]},
                    g_cfg ? f32x2{v1[r][j], v1[r][j + 1]} : z, g_cls ? f32x2{vg[r][j], vg[r][j + 1]} : z, p);
                o[j] = rr[0];
                o[j + 1] = rr[1];
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                o[j] = prologue<GUIDE, PM_RT, float, TE>(vx[r][j], v0[r][j], g_cfg ? v1[r][j] : 0.f, g_cls ? vg[r][j] : 0.f, p);
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
      };
      if constexpr (FAST_ONLY) {
        load_phase(std::true_type{});
      } else {
        if (fastdiv)
          load_phase(std::true_type{});
        else
          load_phase(std::false_type{});
      }
    } else {
#pragma unroll 4
      for (int i = tid; i < n; i += T) {
        const float xev = to_f32(XE ? xe[base + i] : x[base + i]);
        const float o = prologue<GUIDE, PM_RT, float, TE>(xev, to_f32(e0[ebase + i]), g_cfg ? to_f32(e1[ebase + i]) : 0.f,
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

    // phase 3's first operand rows do not depend on the threshold: their loads are issued here and land while the
    // select runs (a sample's select is 4-9 us of barriers and exchanges during which this workgroup moves no data)
    float vx[ROWS][4], vh1[ROWS][4], vh2[ROWS][4];
    auto fetch_rows = [&](int i0) {  // rows i0, i0 + 4T, ... of this thread; lanes past the end re-read a valid group
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const int i = i0 + r * T * 4;
        const int64_t gi = base + (i < n ? i : (i0 < n ? i0 : 0));
        if (nx) load4<true>(x, gi, vx[r]);                 // last use of x and of the cached model values
        if (nh1) load4<true>(h1, gi, vh1[r]);
        if (nh2) load4<true>(h2, gi, vh2[r]);
      }
    };
    if (vec && n > 0) fetch_rows(tid * 4);

    // phase 2: the lo-th smallest |x0| of the whole sample.
    uint32_t prefix = 0u, known = 0u, rank = (uint32_t)tp.lo, cnt_sel = 0u;
    uint32_t hi = ABS, nc = 0u;
    bool use_cand = false, local_only = k == 1;  // local_only: no further cluster-wide step is needed
    bool hist_ready = !track;                    // the level-0 histogram of the whole chunk exists
    bool fast = false;                           // the candidates are few: finish by rank counting
    // clusters first try to settle the sample with ONE exchange (cluster_select_once); the general route below is the
    // fallback for samples whose large values sit in one chunk, and the only route when the quantile is not near 1
    uint32_t a1 = 0u, b1 = 0u;
    bool solved = false;
    uint32_t route = 4u;  // diagnostics (hint word 2): 1 predicted bound, 2 prediction rejected, then 3 single exchange / 4 general
    const uint32_t pbound = route1 ? misc[25] : 0u;  // cluster-uniform: every workgroup read the same hint words
    bool lost = k > 1 && misc[30] != 0u;  // out of the cluster protocol (give_up): straight to solo_select
    if (pbound && !lost) {  // the predicted attempt has a slot area of its own (a rejected one leaves its slots dirty)
      solved = cluster_select_once<T>(sx0, n, vec, m1, m2, m3, m4, has, hist, misc, cand,
                                      ws + THR_WS_WORDS + (size_t)k * THR_SLOTW, tp, k, c, tid, a1, b1, s_idx == grp,
                                      ws + THR_WS_POISON, pbound);
      route = solved ? 1u : 2u;
      lost = !solved && misc[30] != 0u;
    }
    const bool searched = route1 && !solved && !lost;  // the searched-bound attempt runs (and dirties the first slot area)
    if (searched) {
      solved = cluster_select_once<T>(sx0, n, vec, m1, m2, m3, m4, has, hist, misc, cand, ws + THR_WS_WORDS, tp, k, c, tid,
                                      a1, b1, s_idx == grp && !pbound, ws + THR_WS_POISON);
      if (solved && route != 2u) route = 3u;
      lost = !solved && misc[30] != 0u;
    }
    const bool general = !solved;                // the general route runs (for clusters: it dirties the merged histograms)

    // (a lambda for its early exits: a cluster wait that is given up ends the front end -- returns false)
    auto topk_front = [&]() -> bool {
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
        if (!cluster_barrier(ws + THR_WS_CNT + 4, k, misc + 30, ws, tp)) return false;
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
        if (!cluster_barrier(ws + THR_WS_CNT + 5, k, misc + 30, ws, tp)) return false;
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
      return true;
    };
    if (general && topk && !lost) lost = !topk_front();

    // 11 + 11 + 9-bit radix select over the candidates, or over the whole chunk.  In the latter case the elements
    // that share the selected top digit -- the only ones levels 1, 2 and the min-above search can still care about --
    // are compacted into `cand` after level 0 (wave-aggregated append); everything above that digit only matters
    // through its minimum, kept per lane in `hi`.
#pragma unroll 1
    for (int pass = 0; pass < 3 && !fast && general && !lost; ++pass) {
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
        if (!cluster_barrier(ws + THR_WS_CNT + pass, k, misc + 30, ws, tp)) {
          lost = true;
          break;
        }
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
          // (the counts of the cluster add up to cnt_sel <= THR_GCAP: every workgroup that gets here is in the protocol
          // and read the same complete histogram; the bound is defensive)
          const uint32_t room = slot0 < (uint32_t)THR_GCAP ? (uint32_t)THR_GCAP - slot0 : 0u;
          const uint32_t nput = nc < room ? nc : room;
          for (uint32_t i = tid; i < nput; i += T)
            __hip_atomic_store(&gl[slot0 + i], cand[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!cluster_barrier(ws + THR_WS_CNT + 1, k, misc + 30, ws, tp)) {
            lost = true;
            break;
          }
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
    } else if (lost) {
      a = b = 0.f;  // (solo_select below)
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
          if (cluster_barrier(ws + THR_WS_CNT + 3, k, misc + 30, ws, tp))
            b = __uint_as_float(ABS - __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          else
            lost = true;
        } else {
          b = __uint_as_float(misc[3]);
        }
      }
    }
    if (lost) {
      // This workgroup is out of the cluster protocol (a wait of its own timed out, on this sample or an earlier one): it
      // computes the sample's two order statistics alone (solo_select): same bits, no peers, no workspace.
      {
        // wave-uniform bases + a 32-bit element index (a clustered sample has at most THR_KMAX chunks: < 2^31 elements)
        const int64_t sx = s_loc * tp.per_sample, se = s_loc * (eps_stride ? eps_stride : tp.per_sample);
        const TS* __restrict__ xs = (XE ? xe : x) + sx;
        const TE* __restrict__ e0s = e0 + se;
        const TE* __restrict__ e1s = g_cfg ? e1 + se : e0 + se;
        const TE* __restrict__ gs = g_cls ? g + sx : e0 + se;
        auto bits_at = [&](int i) -> uint32_t {
          const float o = prologue<GUIDE, PM_RT, float, TE>(to_f32(xs[i]), to_f32(e0s[i]), g_cfg ? to_f32(e1s[i]) : 0.f,
                                          g_cls ? to_f32(gs[i]) : 0.f, p);
          return __float_as_uint(o) & ABS;
        };
        uint32_t sa, sb;
        solo_select<T>(bits_at, (int)tp.per_sample, (uint32_t)tp.lo, tp.hi != tp.lo, hist, misc, tid, sa, sb);
        a = __uint_as_float(sa);
        b = __uint_as_float(sb);
        // the operand rows fetched before the select are re-fetched here: not live across this (rare) detour, which
        // would otherwise cost every launch registers
        if (vec && n > 0) fetch_rows(tid * 4);
      }
    }
    DPM_TSTAMP(2)
    if (tp.hint && c == 0 && tid == 0) {  // this stage's statistic for the next stage's prediction
      float* hw = tp.hint + (int64_t)s_idx * THR_HINT_W;
      const float prev = tp.hint_reset ? 0.f : hw[0];
      hw[1] = prev;
      hw[0] = a;
      hw[2] = (float)(solved ? route : (route == 2u ? 2u : 4u));
      hw[3] = (float)misc[24];  // diagnostics: entries of the last union this workgroup gathered
    }
    // This workgroup is through with the sample's workspace.  The last of the cluster to say so puts every word it and
    // its peers dirtied back to zero (end of the sample loop): the workspace is all zero between launches, so no launch
    // has to clear it first.  The returning atomic is in flight during phase 3.
    // (a workgroup that gave up a wait says so in the upper half of the counter: it may have left words dirty that the
    // last workgroup's own route knows nothing about, see the clean-up)
    uint32_t done_old = 0u;
    if (k > 1 && tid == 0)
      done_old = __hip_atomic_fetch_add(ws + THR_WS_DONE, 1u + (misc[30] ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      // ROWS tile rows per iteration, the loads of all issued before the first use (explicit: the write-through
      // stores are assembly the loop unroller will not duplicate)
      for (int i0 = tid * 4; i0 < n; i0 += ROWS * T * 4) {  // (the first rows were fetched before the select)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
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
        if (i0 + ROWS * T * 4 < n) fetch_rows(i0 + ROWS * T * 4);  // the next iteration's operands
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
      if (tid == 0) {
        misc[11] = (done_old & 0xffffu) == k - 1u ? 1u : 0u;
        misc[28] = ((done_old >> 16) != 0u || misc[30]) ? 1u : 0u;  // some workgroup of the cluster gave up a wait
      }
      __syncthreads();
      if (misc[11]) {
        const bool any_dead = misc[28] != 0u;  // then: everything any route can have written
        // Only the slot areas an attempt of this sample published into, and of their slots the header and the slot_cap
        // value words (round 3 cleared both areas at their full stride whatever had run: 12.7 KB per sample of cfg5's
        // launch, 0.4 MB of its 3.5 MB of writes; a predicted attempt that solves the sample now leaves 1.7 KB to clear).
        // Flat, unconditional stores: clearing exactly max(slot_pub, count) words per slot was measured too -- 10 % fewer
        // bytes again, but the count comes from LDS and the loop sits on the critical path of the cluster's next sample
        // (+0.2 us per stage at cfg5's size, +1.5 us at [64,3,256,256], profiles/r04_thresholding.md).
        uint32_t* slots = ws + THR_WS_WORDS;
        const int zshift = tp.slot_shift;
        const uint32_t zw = 1u << zshift, zwords = k << zshift;  // slot_cap is a power of two
        for (uint32_t area = 0; area < 2u; ++area) {
          // (any_dead: a workgroup that gave up may have rewritten the hint words a late peer predicts from -- `pbound` and
          // `searched` are then no longer cluster-uniform, and a peer may have published into an area this workgroup's own
          // route never touched: clear both)
          const bool used = (any_dead && route1) || (area == 0u ? searched : pbound != 0u);
          if (!used) continue;
          uint32_t* as = slots + (size_t)area * k * THR_SLOTW;
          for (uint32_t q = tid; q < zwords; q += T) as[(size_t)(q >> zshift) * THR_SLOTW + THR_SLOT_HDR + (q & (zw - 1u))] = 0u;
          for (uint32_t i = tid; i < k * (uint32_t)THR_SLOT_HDR; i += T)
            as[(size_t)(i / THR_SLOT_HDR) * THR_SLOTW + (i % THR_SLOT_HDR)] = 0u;
        }
        if (general || !route1 || any_dead) {
          for (uint32_t i = tid; i < (uint32_t)THR_WS_WORDS; i += T) ws[i] = 0u;
        } else if (tid == 0) {
          ws[THR_WS_DONE] = 0u;
          ws[THR_WS_POISON] = 0u;
#if DPM_LAB
          if (tp.elect)
            for (int q = 0; q < 8; ++q) ws[THR_WS_RESULT + q] = 0u;
#endif
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

}  // namespace
