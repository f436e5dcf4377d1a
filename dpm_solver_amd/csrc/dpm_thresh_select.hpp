// dpm_thresh_select.hpp -- dynamic thresholding (ref :416-425), part 2 of 3: exact order statistics of |x0| -- the
// workgroup-local routines (top-4 lists, radix histograms, candidate compaction, rank counting / selection), the cluster's
// single-exchange select and the solo select of a workgroup that left its cluster.  Part of dpm_device.hpp (include that).
#pragma once

namespace {

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
// nan_tag != 0: the histogram is a level-0 one (digit = top 11 bits of the pattern) -- a non-empty last bin means the sample
// holds a NaN (canon_nan): misc[THR_NANW] = nan_tag.
template <int T>
__device__ __forceinline__ void locate_bin(uint32_t* hist, uint32_t* misc, uint32_t rank, int tid, uint32_t nan_tag = 0u) {
  static_assert(THR_NB == 4 * T, "one 16-byte histogram slice per thread");
  u32x4* h4 = reinterpret_cast<u32x4*>(hist);
  const u32x4 v = h4[tid];
  h4[tid] = u32x4{0u, 0u, 0u, 0u};
  if (nan_tag && tid == T - 1 && v[3]) misc[THR_NANW] = nan_tag;
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

// A WAVEFRONT-conditional sweep ("one of my threads holds more than four candidates": the callers' __ballot) must read the
// elements its own threads produced.  In the 16-byte layout those are the rows compact_candidates reads (thread t: elements
// 4t .. 4t+3 of every row of 4T); in the one-element-per-lane layout (unaligned or ragged samples) element i belongs to
// thread i % T, and the 16-byte rows of a wavefront hold other wavefronts' elements -- sweeping them there listed some
// elements twice and dropped others (a wrong order statistic in samples of 2049 .. 12288 elements whose length is not a
// multiple of 4, K close to T/4; found by tools/fuzz_gpu_thresh.py in round 6).  Same list protocol as compact_candidates.
template <int T>
__device__ __forceinline__ void compact_own_elements(const float* sx0, int n, uint32_t bin, uint32_t* misc, uint32_t* cand,
                                                     int tid, int shift = 20, uint32_t dbase = 0u) {
  constexpr uint32_t ABS = 0x7fffffffu;
  uint32_t cnt = 0u;
  for (int i = tid; i < n; i += T) {
    const uint32_t u = __float_as_uint(sx0[i]) & ABS;
    const uint32_t dr = u >> shift, d = dr > dbase ? dr - dbase : 0u;
    cnt += d >= bin ? 1u : 0u;
  }
  const uint32_t incl = wave_incl_scan(cnt);
  uint32_t slot = 0u;
  if ((tid & 63) == 63 && incl) slot = atomicAdd(&misc[4], incl);
  uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)slot, 63) + incl - cnt;
  for (int i = tid; i < n; i += T) {
    const uint32_t u = __float_as_uint(sx0[i]) & ABS;
    const uint32_t dr = u >> shift, d = dr > dbase ? dr - dbase : 0u;
    if (d >= bin) {
      if (off < (uint32_t)THR_CAP) cand[off] = u;
      ++off;
    }
  }
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
                                                    uint32_t* poison, uint32_t nan_tag, const uint32_t pbound = 0u) {
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
      if (!vec) {  // one element per lane: the sweep reads this wavefront's own elements (compact_own_elements)
        if (pbound)
          compact_own_elements<T>(sx0, n, pbound, misc, cand, tid, 0, 0u);
        else
          compact_own_elements<T>(sx0, n, bin_lo, misc, cand, tid, THR_FSHIFT, dbase);
      } else if (pbound)  // digit = the whole pattern: d >= bin is u >= pbound
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
  if (tid == 0 && umax == THR_NAN_KEY) misc[THR_NANW] = nan_tag;  // a chunk's maximum is a NaN (read behind the next barrier)
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
                                            uint32_t* misc, int tid, uint32_t& a, uint32_t& b, uint32_t nan_tag) {
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
    locate_bin<T>(hist, misc, rank, tid, pass == 0 ? nan_tag : 0u);
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

}  // namespace
