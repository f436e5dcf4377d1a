// dpm_thresh_kernel.hpp -- dynamic thresholding (ref :416-425), part 3 of 3: stage_thresh_kernel -- x0 of a sample's chunk
// computed once into LDS, the order statistics (dpm_thresh_select.hpp), clamp / divide / combine / store from LDS.
// Part of dpm_device.hpp (include that).
#pragma once

namespace {

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
  if (threadIdx.x == 0) misc[THR_NANW] = 0u;  // no sample's tag (the first barrier of the first sample is ahead)
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
                o[j] = canon_nan(rr[0]);
                o[j + 1] = canon_nan(rr[1]);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                o[j] = canon_nan(prologue<GUIDE, PM_RT, float, TE>(vx[r][j], v0[r][j], g_cfg ? v1[r][j] : 0.f, g_cls ? vg[r][j] : 0.f, p));
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
        const float o = canon_nan(prologue<GUIDE, PM_RT, float, TE>(xev, to_f32(e0[ebase + i]), g_cfg ? to_f32(e1[ebase + i]) : 0.f,
                                                  g_cls ? to_f32(g[base + i]) : 0.f, p));
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
    const uint32_t nan_tag = (uint32_t)s_idx + 1u;  // misc[THR_NANW] == nan_tag: this sample holds a NaN (dpm_thresh_common.hpp)
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
                                      ws + THR_WS_POISON, nan_tag, pbound);
      route = solved ? 1u : 2u;
      lost = !solved && misc[30] != 0u;
    }
    const bool searched = route1 && !solved && !lost;  // the searched-bound attempt runs (and dirties the first slot area)
    if (searched) {
      solved = cluster_select_once<T>(sx0, n, vec, m1, m2, m3, m4, has, hist, misc, cand, ws + THR_WS_WORDS, tp, k, c, tid,
                                      a1, b1, s_idx == grp && !pbound, ws + THR_WS_POISON, nan_tag);
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
      locate_bin<T>(hist, misc, (uint32_t)tp.mrank, tid, nan_tag);  // (the per-thread maxima: a NaN is some thread's maximum)
      const uint32_t bin_lo = misc[0];
      DPM_TSTAMP(5)
      {
        // The candidates of a thread are among its four largest values unless even the fourth reaches the digit (and
        // the thread has more elements): wavefronts where that happens anywhere sweep their LDS rows instead.
        const int mine = vec ? (has ? 4 * ((n - tid * 4 + T * 4 - 1) / (T * 4)) : 0) : (has ? (n - tid + T - 1) / T : 0);
        const bool c1 = mine > 0 && (m1 >> 20) >= bin_lo, c2 = mine > 1 && (m2 >> 20) >= bin_lo;
        const bool c3 = mine > 2 && (m3 >> 20) >= bin_lo, c4 = mine > 3 && (m4 >> 20) >= bin_lo;
        if (__ballot(c4 && mine > 4)) {
          if (vec)
            (void)compact_candidates<T, true>(sx0, n, bin_lo, misc, cand, tid);
          else  // one element per lane: the wavefront's own elements are not its 16-byte rows
            compact_own_elements<T>(sx0, n, bin_lo, misc, cand, tid);
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
      locate_bin<T>(hist, misc, rank, tid, pass == 0 ? nan_tag : 0u);
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
          return __float_as_uint(canon_nan(o)) & ABS;
        };
        uint32_t sa, sb;
        solo_select<T>(bits_at, (int)tp.per_sample, (uint32_t)tp.lo, tp.hi != tp.lo, hist, misc, tid, sa, sb, nan_tag);
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
    // torch.quantile 'linear' = ATen lerp(a, b, w), which is ONE fused multiply-add on every path torch takes (the CPU
    // kernel is vec::fmadd(w or w - 1, b - a, a or b) for whole vectors and tails alike, the device kernel's
    // "a + w * diff" / "b - diff * (1 - w)" is contracted by its compiler): with this library's -ffp-contract=off the fma
    // is written out.  (Rounds 1-5 had the two-rounding form: one ulp off torch.quantile in ~2 % of random samples, found by
    // tools/fuzz_gpu_thresh.py.)
    if (misc[THR_NANW] == nan_tag) a = b = __uint_as_float(THR_NAN_KEY);  // ref :420: a NaN anywhere in the sample
    const float diff = b - a;
    const float q = tp.w < 0.5f ? __builtin_fmaf(tp.w, diff, a) : __builtin_fmaf(tp.w - 1.f, diff, b);
    // ref :423, torch.maximum: a NaN quantile (a NaN among the two order statistics, inf - inf) stays NaN -- fmaxf drops it
    const float s = q != q ? q : fmaxf(q, tp.max_val);
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
            for (int j = 0; j < 4; ++j) om[j] = clamp_ref(sx0[i + j], s);  // ref :424
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
        const float mn = clamp_ref(sx0[i], s) / s;  // ref :424
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
