// dpm_launch.hpp -- launch plumbing of the stage kernels: device info, cluster shape, tuning, variant dispatch,
// the fused multi-request launcher, the catch-all kernel handles (part of dpm_device.hpp; include that)
#pragma once

namespace {

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
// (two slot areas: the attempt with a predicted bound and the one with a searched bound each publish into their own)
inline int64_t thr_ws_stride(int64_t k) { return (int64_t)THR_WS_WORDS + 2 * k * (int64_t)THR_SLOTW; }
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
// LAB build only: the lone-launch north-star kernels (2-byte state, 2M / first-order, inputs from HBM) with their read streams
// on the LDS-DMA path (stage_kernel_dma, DPM_TUNE_LDS_DMA).  Measured by rocprofv3 rows inside a network loop
// (profiles/r05_lone_floor.md): 8.32 us against 8.28 us through registers -- the no-arithmetic floor kernel gains 3-4 % from
// that path, the stage kernel nothing -- so the product keeps the register path and does not carry the variant.
#ifndef DPM_LDS_DMA_DEFAULT
#define DPM_LDS_DMA_DEFAULT 0
#endif
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
  // thresholded stage of n_multi requests fused into ONE launch (dpm_stage_launch_multi): `multi` = their buffer
  // records, all of the shape and dtypes of multi[0] (which is also the `b` the launcher is called with)
  const dpm_buffers* multi = nullptr;
  int n_multi = 0;
};

template <typename K, typename... Args>
void launch(K kern, dim3 grid, dim3 block, size_t lds, const LaunchCtx& c, Args... args) {
  if (c.start || c.stop)
    hipExtLaunchKernelGGL(kern, grid, block, lds, c.stream, c.start, c.stop, 0, args...);
  else
    hipLaunchKernelGGL(kern, grid, block, lds, c.stream, args...);
}

// what a fused multi-request launcher returns -- without setting an error text -- when this stage (form, guidance,
// prologue, buffers) has no fused variant: the caller then launches the requests one by one
constexpr int MULTI_NOT_BUILT = -1000;

// the operands of one single-request launch, typed: what both kernel families (streaming, thresholding) start from
template <typename TS, typename TE>
struct Operands {
  KParams p;
  const TS *x, *xe, *h1, *h2;
  const TE *e0, *e1, *g;
  TS *xo, *mo;
  KExt ext;       // duplicate output, mask blend, channel-sliced network output
  bool use_ext;
  int n_cu;
  Operands(const dpm_stage* st, const dpm_buffers* b)
      : p(make_params(st)),
        x(static_cast<const TS*>(b->x)),
        xe(static_cast<const TS*>(b->xe)),
        h1(static_cast<const TS*>(b->h1)),
        h2(static_cast<const TS*>(b->h2)),
        e0(static_cast<const TE*>(b->e0)),
        e1(static_cast<const TE*>(b->e1)),
        g(static_cast<const TE*>(b->g)),
        xo(static_cast<TS*>(b->x_out)),
        mo(static_cast<TS*>(b->m_out)) {
    const DeviceInfo& di = device_info();
    n_cu = di.n_cu > 0 ? di.n_cu : 256;
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
    use_ext = ext.xo2 || ext.mask || ext.eps_stride;
  }
};

// ---- a thresholded stage (stage_thresh_kernel): cluster shape, select parameters, kernel flavour, the launch
template <typename TS, typename TE, int FORM, int GUIDE, bool XE>
int launch_thresh(const dpm_stage* st, const dpm_buffers* b, const LaunchCtx& stream, const Operands<TS, TE>& op) {
  const KParams& p = op.p;
  const TS *x = op.x, *xe = op.xe, *h1 = op.h1, *h2 = op.h2;
  const TE *e0 = op.e0, *e1 = op.e1, *g = op.g;
  TS *xo = op.xo, *mo = op.mo;
  const KExt& ext = op.ext;
  const int n_cu = op.n_cu;
  const Tuning tn = tuning_for(b->opts);
  if (stream.dyn) return dpm_set_error(DPM_ERR_UNSUPPORTED, "dynamic thresholding with device-resident coefficients");
  const int64_t per_sample = b->n / b->batch;
  // several requests in one launch: one batch of n_multi * batch samples -- more samples per launch, smaller (or no)
  // clusters -- whose sample s lives in the tensors of request s / batch (ThrTab)
  const bool multi = stream.multi != nullptr;
  const int64_t batch = multi ? (int64_t)stream.n_multi * b->batch : b->batch;
  if (batch > 0x7fffffff || per_sample > ((int64_t)1 << 40))
    return dpm_set_error(DPM_ERR_UNSUPPORTED, "thresholding: batch / sample size out of range");
  ThrPlan pl = thr_plan(batch, per_sample, n_cu);
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream.stream, &cap_status);
  const bool capturing = cap_status != hipStreamCaptureStatusNone;
  // Clusters wait for each other inside the kernel, which is only safe while no OTHER clustered launch can hold part of
  // the chip at the same time.  Eager launches of this process are chained device-wide (below); a captured graph is
  // replayed outside that chain, possibly next to another graph on another stream.  Under capture a sample that fits
  // one workgroup's LDS therefore takes the cluster-free shape (one workgroup per sample) unless the caller opts in
  // (dpm_launch_opts.cluster_in_graph); larger samples have no such shape and keep their clusters, with bounded waits.
  if (capturing && pl.k > 1 && per_sample <= THR_CHUNK_MAX && !tn.cluster_in_graph) {
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
  tp.batch = (int32_t)batch;
  tp.fastdiv = st->model_type == DPM_MODEL_NOISE && (st->flags & DPM_F_TO_X0) && div_invariant_ok(st->alpha_e);
  const size_t a4s = sizeof(TS) * 4, a4e = sizeof(TE) * 4;
  tp.vec = per_sample % 4 == 0 && ext.eps_stride % 4 == 0 && ext.mask_period % 4 == 0 && aligned(x, a4s) &&
           aligned(xe, a4s) && aligned(h1, a4s) && aligned(h2, a4s) && aligned(xo, a4s) && aligned(mo, a4s) &&
           aligned(ext.xo2, a4s) && aligned(ext.mask, a4s) && aligned(ext.ba, a4s) && aligned(ext.bb, a4s) &&
           aligned(e0, a4e) && aligned(e1, a4e) && aligned(g, a4e);
  static const ThrTab no_tab = {};
  ThrTab tab_multi;
  if (multi) {
    // the fused launch serves plain requests: no extensions, the evaluation state is the state, distinct workspaces
    // (clusters of different requests run side by side); anything else is launched request by request
    if (XE || GUIDE == DPM_GUIDE_CLASSIFIER || stream.n_multi > MULTI_MAX) return MULTI_NOT_BUILT;
    std::memset(&tab_multi, 0, sizeof tab_multi);
    tp.bpr = (int32_t)b->batch;
    for (int r = 0; r < stream.n_multi; ++r) {
      const dpm_buffers& q = stream.multi[r];
      if (!q.x || (q.xe && q.xe != q.x) || q.x_out2 || (q.eps_stride && q.eps_stride != per_sample)) return MULTI_NOT_BUILT;
      if (pl.k > 1) {
        if (!q.workspace) return MULTI_NOT_BUILT;
        for (int r2 = 0; r2 < r; ++r2)
          if (stream.multi[r2].workspace == q.workspace) return MULTI_NOT_BUILT;
      }
      tp.vec = tp.vec && aligned(q.x, a4s) && aligned(q.h1, a4s) && aligned(q.h2, a4s) && aligned(q.x_out, a4s) &&
               aligned(q.m_out, a4s) && aligned(q.e0, a4e) && aligned(q.e1, a4e);
      tab_multi.x[r] = q.x;
      tab_multi.e0[r] = q.e0;
      tab_multi.e1[r] = q.e1;
      tab_multi.h1[r] = q.h1;
      tab_multi.h2[r] = q.h2;
      tab_multi.xo[r] = q.x_out;
      tab_multi.mo[r] = q.m_out;
      tab_multi.ws[r] = static_cast<uint32_t*>(q.workspace);
    }
  }
  const ThrTab& tab = multi ? tab_multi : no_tab;
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
      if (quota * 3 / 2 <= slot_cap && tn.cluster_one_hop) {
        tp.quota = (int32_t)quota;
        tp.kbig = (int32_t)K;
        tp.slot_cap = (int32_t)slot_cap;
        tp.slot_pub = (int32_t)std::min<int64_t>(slot_cap, quota + quota / 4 + 4);
        tp.slot_shift = slot_shift;
        tp.debug_reject = tn.cluster_one_hop == 2;
      }
    }
  }
#ifdef DPM_THR_TIMING
  static_assert(DPM_LAB, "DPM_THR_TIMING instruments the lab build only");
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
  // (classifier guidance -- the reference's own ImageNet-256 example samples with it AND thresholding, sample.sh:40-50 --
  // has the HOT = 3 flavour only: its two load loops cover the noise fast path and everything else)
  constexpr bool HOT_BUILT = (FORM == DPM_FORM_LIN1 || FORM == DPM_FORM_TWO || FORM == DPM_FORM_MS3) && !XE;
  constexpr bool HOT12_BUILT = HOT_BUILT && (GUIDE == DPM_GUIDE_NONE || GUIDE == DPM_GUIDE_CFG);
  const bool hot = HOT_BUILT && tp.vec && !ext.mask;
  const bool front = tp.topk > 0 || tp.quota > 0;  // the select starts from the per-thread maxima (quantile close to 1)
  // the general kernel reads form / guidance from the stage record and always takes the evaluation state through xe
  using ThrKernel = decltype(&stage_thresh_kernel<TS, TE, FORM_RT, GUIDE_RT, true, THR_THREADS, 0>);
  auto kern = reinterpret_cast<ThrKernel>(const_cast<void*>(dpm_catchall_thresh<TS, TE>()));
  if constexpr (HOT_BUILT) {
    // noise-prediction network + division by the invariant alpha: the compile-time prologue (HOT 1 / 2); any other
    // parameterisation with the usual near-1 quantile: the run-time prologue (HOT 3); the rest: the catch-all kernel
    bool chosen = false;
    if constexpr (HOT12_BUILT) {
      // (HOT 2 -- the same without the front end, for quantiles far from 1 -- was instantiated until round 5: no BASELINE
      // configuration and no reference example samples with such a ratio; those launches take the catch-all kernel)
      if (hot && front && tp.fastdiv && !tn.force_generic) {
        kern = stage_thresh_kernel<TS, TE, FORM, GUIDE, XE, THR_THREADS, 1>;
        chosen = true;
      }
    }
    if (!chosen && hot && front) kern = stage_thresh_kernel<TS, TE, FORM, GUIDE, XE, THR_THREADS, 3>;
  }
  if (!xe) xe = x;
  int64_t grid = batch;
  tp.groups = (int32_t)batch;
  if (pl.k > 1) {
    // clusters synchronise through spin barriers: every workgroup of the grid must be resident at once
    // (cached per thread for the last (device, kernel, LDS size): kernels of different flavours may differ in occupancy)
    static thread_local int occ_dev = -1, occ = 0;
    static thread_local size_t occ_lds = 0;
    static thread_local const void* occ_kern = nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != occ_dev || lds_bytes != occ_lds || occ_kern != reinterpret_cast<const void*>(kern)) {
      int nb = 0;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), THR_THREADS,
                                                                  lds_bytes);
      if (e != hipSuccess) return dpm_set_error((int)e, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
      occ_dev = dev;
      occ_lds = lds_bytes;
      occ_kern = reinterpret_cast<const void*>(kern);
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
    const int64_t groups = std::min<int64_t>(batch, cap / pl.k);
    tp.groups = (int32_t)groups;
    tp.ws = static_cast<uint32_t*>(b->workspace);
    tp.ws_stride = thr_ws_stride(pl.k);
    grid = groups * pl.k;
    // No clearing of the workspace here: the caller hands it over zero-filled once, the kernel leaves it zero-filled
    // (dpm_threshold_workspace_bytes).  A wait on a peer that times out is recovered from inside the kernel (solo_select:
    // same results, no error); the host-mapped word only records that it happened (dpm_cluster_timeout_poll).
    tp.fault = cluster_fault_word(dev, !capturing);
    tp.spin_limit = (uint32_t)tn.thr_spin_limit;
#if DPM_LAB
    if (tn.thr_debug_fault == 1) tp.spin_limit = 0u;
    tp.debug_fault = tn.thr_debug_fault;
    tp.elect = tn.thr_elect > 0 && tp.quota > 0;
    tp.stagger = tn.thr_stagger;
#endif
    // the select bound predicted from the previous stages (dpm_buffers.thr_hint): single requests on the one-exchange route
    // -- where it pays: a small K (the wanted rank from the top), so that the predicted union (~1.3-1.9 K entries instead
    // of k * quota) is finished by rank counting.  Measured (tools/thr_routes.py): [32,3,64,64] (K = 63) 11.3 -> 10.7 us per
    // stage, union 236 -> 116 entries, every stage from the third on predicted; [64,3,256,256] (K = 983) 53 -> 59 us -- a
    // 10-step trajectory changes the statistic by 2.5x per stage, the extrapolation lands low and the union GROWS.
    if (!multi && tp.quota > 0 && b->thr_hint && tp.kbig <= 128) {
      tp.hint = b->thr_hint;
      tp.hint_reset = st->index <= 0;
      tp.hint_predict = tn.thr_predict;
    }
    // Two clustered launches on different streams could each hold part of the CUs with spinning workgroups and
    // starve the other's missing peers.  Within this process they are therefore chained device-wide: wait for the
    // previous clustered launch (whatever its stream), record after this one.  (Not under stream capture, where an
    // event recorded outside the capture cannot be waited on; see above.)
    if (!capturing) {
      DeviceContext& ch = device_context(dev);
      std::lock_guard<std::mutex> lk(ch.mu);
      if (!ch.ev && hipEventCreateWithFlags(&ch.ev, hipEventDisableTiming) != hipSuccess) ch.ev = nullptr;
      if (ch.ev && ch.recorded) (void)hipStreamWaitEvent(stream.stream, ch.ev, 0);
      launch(kern, dim3((unsigned)grid), dim3(THR_THREADS), lds_bytes, stream, x, xe, e0, e1, g, h1, h2, xo, mo, p, tp, ext, tab);
      if (ch.ev && hipEventRecord(ch.ev, stream.stream) == hipSuccess) ch.recorded = true;
      t_dump(grid);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return dpm_set_error((int)e, "stage kernel launch failed: %s", hipGetErrorString(e));
      return DPM_OK;
    }
  }
  launch(kern, dim3((unsigned)grid), dim3(THR_THREADS), lds_bytes, stream, x, xe, e0, e1, g, h1, h2, xo, mo, p, tp, ext, tab);
  t_dump(grid);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "stage kernel launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

// ---- a stage of the streaming family (stage_kernel / the one-element-per-lane catch-all): variant and launch shape
template <typename TS, typename TE, int FORM, int GUIDE, bool XE>
int launch_stream(const dpm_stage* st, const dpm_buffers* b, const LaunchCtx& stream, const Operands<TS, TE>& op) {
  const KParams& p = op.p;
  const TS *x = op.x, *xe = op.xe, *h1 = op.h1, *h2 = op.h2;
  const TE *e0 = op.e0, *e1 = op.e1, *g = op.g;
  TS *xo = op.xo, *mo = op.mo;
  const KExt& ext = op.ext;
  const int n_cu = op.n_cu;
  const bool use_ext = op.use_ext;
  const size_t as = sizeof(TS) * EPT, ae = sizeof(TE) * EPT;
  bool vec = aligned(x, as) && aligned(xe, as) && aligned(h1, as) && aligned(h2, as) && aligned(xo, as) &&
             aligned(mo, as) && aligned(e0, ae) && aligned(e1, ae) && aligned(g, ae);
  if (use_ext)  // the extended vector kernel has no ragged tail and indexes whole 8-element groups
    vec = vec && aligned(ext.xo2, as) && aligned(ext.mask, as) && aligned(ext.ba, as) && aligned(ext.bb, as) &&
          b->n % EPT == 0 && ext.mask_period % EPT == 0 &&
          (!ext.eps_stride || (ext.per_sample % EPT == 0 && ext.eps_stride % EPT == 0));
  // what the streaming family instantiates (binary size, build time and first-call cost: one kernel per combination and
  // dtype pair).  Round 5 measured what a compile-time prologue is worth against the run-time one (SPEC_GENERIC: the mode is
  // chosen once per workgroup iteration, the same straight-line code) -- 0-7 % per launch, profiles/r05_kernel_budget.md --
  // and keeps it where samplers spend their time:
  //   * a separate evaluation state (xe != x) occurs in the singlestep mid / final stages -- forms TWO and SS3T -- and
  //     in the FIRST stage of a multistep run with a corrector on x_t (mask blend, any correcting_xt_fn): the network saw
  //     the raw x_T, the update starts from the corrected state (ref :1179-1183) -- form LIN1, unguided or CFG;
  //   * the compile-time prologues (noise-prediction network) for the forms samplers spend their time in -- LIN1, TWO,
  //     MS3 -- unguided and under classifier-free guidance; classifier guidance (an autograd pass through the classifier
  //     per step dwarfs 3 % of a stage kernel) and that one first stage take SPEC_GENERIC; SS3T and DENOISE run the general
  //     prologue anyway (true division: the same bits);
  //   everything else goes through the one-element-per-lane kernel.
  constexpr bool COMBO_BUILT = ((!XE && !(FORM == DPM_FORM_DENOISE && GUIDE == DPM_GUIDE_CLASSIFIER)) || FORM == DPM_FORM_TWO ||
                                FORM == DPM_FORM_SS3T || (FORM == DPM_FORM_LIN1 && GUIDE != DPM_GUIDE_CLASSIFIER)) &&
                               !(FORM == DPM_FORM_SS3T && GUIDE == DPM_GUIDE_CLASSIFIER) &&  // singlestep-3 'taylor' under classifier guidance: scalar kernel
                               !(FORM == DPM_FORM_SS3T && !XE);  // the samplers always hand SS3T its evaluation state (the rocprofv3 name
                                                                 // set of the suite, profiles/r05_kernels_launched_suite.md, has no launch without)
  // the one denoise_to_zero launch of a trajectory with a KExt extension (mask blend, strided network output): scalar kernel
  constexpr bool EXT_BUILT = FORM != DPM_FORM_DENOISE;
  constexpr bool SPEC_BUILT = (FORM == DPM_FORM_LIN1 || FORM == DPM_FORM_TWO || FORM == DPM_FORM_MS3) &&
                              GUIDE != DPM_GUIDE_CLASSIFIER && !(XE && FORM == DPM_FORM_LIN1);
  // device-resident coefficients (adaptive solver): DYN kernels exist for the forms it launches -- first-order,
  // second-order and the singlestep-3 'taylor' combination -- with a 4-byte state (the reference's adaptive solver keeps
  // fp32 with a discrete schedule), unguided or CFG, without the KExt extensions; anything else takes the
  // one-element-per-lane kernel
  constexpr bool DYN_BUILT = COMBO_BUILT && sizeof(TS) == 4 && GUIDE != DPM_GUIDE_CLASSIFIER &&
                             ((FORM == DPM_FORM_LIN1 && !XE) || (FORM == DPM_FORM_TWO && (XE || GUIDE == DPM_GUIDE_NONE)) ||
                              (FORM == DPM_FORM_SS3T && XE));  // the launches of the adaptive solver's device-side controller
  const bool dyn_vec = stream.dyn && DYN_BUILT && !use_ext;
  if (!vec || !COMBO_BUILT || (stream.dyn && !dyn_vec) || (use_ext && !EXT_BUILT)) {
    int64_t blocks = (b->n + 255) / 256;
    const int64_t cap = (int64_t)n_cu * 16;
    if (blocks > cap) blocks = cap;
    using ScalarKernel = decltype(&stage_kernel_scalar<TS, TE, false>);  // the DYN = true variant has the same signature
    const void* k = stream.dyn ? dpm_catchall_scalar<TS, TE, true>() : dpm_catchall_scalar<TS, TE, false>();
    launch(reinterpret_cast<ScalarKernel>(const_cast<void*>(k)), dim3((unsigned)blocks), dim3(256), 0, stream, x, xe ? xe : x,
           e0, e1, g, h1, h2, xo, mo, b->n, p, ext, stream.dyn, stream.skip);
  } else if constexpr (COMBO_BUILT) {
    const Tuning tn = tuning_for(b->opts);
    // the compile-time prologue exists for the data-prediction form of a noise network (dpmsolver++: eps -> x0 by the
    // invariant alpha) only.  The eps form (algorithm_type "dpmsolver") had one too until round 5: with inputs from HBM the
    // run-time prologue measures equal or faster on every eps-form launch (2M at cfg2 size fp16 / fp32, the unconditional
    // and the CFG singlestep-3 sampler at [64,3,256,256]; profiles/r05_kernel_budget.md), so it takes SPEC_GENERIC now
    const bool noise = SPEC_BUILT && !stream.dyn && !tn.force_generic && st->model_type == DPM_MODEL_NOISE &&
                       (st->flags & DPM_F_TO_X0) && div_invariant_ok(st->alpha_e);
    const int spec = noise ? SPEC_NOISE_X0 : SPEC_GENERIC;
    const int64_t ntiles = ((b->n / EPT) + 255) / 256;
    const bool big = ntiles >= 4 * (int64_t)n_cu;  // two tiles per iteration only when there is work for it
    // launch shape: one 256-lane group per U tiles, capped per CU; two groups per workgroup (stage_kernel) when that still
    // leaves two workgroups per CU: what larger workgroups save is dispatches ([256,4,64,64]: 2048 -> 1024), and a small
    // launch needs every CU more than it needs that.  (One step below -- 512 tiles, SD's [64,4,64,64] -- 512 threads measure
    // neutral inside the loop: CFG + duplicate store 6.87-6.98 us against 7.00-7.02, the plain fp16 kernel 4.81-4.86 against
    // 4.66-4.84; profiles/r04_block_threads.md.)
    auto shape_for = [&](int u) {
      const int64_t iters = (ntiles + u - 1) / u;
      int bt = 256;
      if (tn.block_threads > 0) bt = tn.block_threads;
      else if (iters >= 4 * (int64_t)n_cu) bt = STAGE_MAX_THREADS;
      const int64_t per = bt / 256;
      int64_t blocks = (iters + per - 1) / per;
      const int64_t cap = std::max<int64_t>(1, (int64_t)n_cu * tn.blocks_per_cu / per);
      if (blocks > cap) blocks = cap;
      return std::make_pair(dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3((unsigned)bt));
    };
#define DPM_LAUNCH(SPEC_, U_, NT_, EXT_)                                                                                 \
  do {                                                                                                                   \
    const auto sh_ = shape_for(U_);                                                                                      \
    launch(stage_kernel<TS, TE, FORM, GUIDE, XE, SPEC_, U_, NT_, EXT_>, sh_.first, sh_.second, 0, stream, x, xe, e0, e1, \
           g, h1, h2, xo, mo, b->n, p, ext, stream.dyn, stream.skip);                                                    \
  } while (0)
    if (dyn_vec) {
      if constexpr (DYN_BUILT) {
        const auto sh = shape_for(1);
        launch(stage_kernel<TS, TE, FORM, GUIDE, XE, SPEC_GENERIC, 1, DefNT<TS>::value, false, true>, sh.first, sh.second, 0,
               stream, x, xe, e0, e1, g, h1, h2, xo, mo, b->n, p, ext, stream.dyn, stream.skip);
      }
    } else if (use_ext) {
      // one tile per workgroup (round 2 launched two for an fp32 state with 2-byte outputs: CFG + duplicate store at
      // [256,4,64,64] 18.1 / 20.3 us back-to-back / evicted against 17.3 / 19.3 with one, tools/stage_bench.py); nt mask of
      // the inputs-from-HBM situation
      constexpr int ENT = sizeof(TS) == 2 ? 1 : (sizeof(TE) == 4 ? 5 : 1);
      if constexpr (EXT_BUILT) {
        if (spec == SPEC_GENERIC) {
          DPM_LAUNCH(SPEC_GENERIC, 1, ENT, true);
        } else if constexpr (SPEC_BUILT) {
          DPM_LAUNCH(SPEC_NOISE_X0, 1, ENT, true);
        }
      }
    } else if (spec == SPEC_GENERIC) {
      DPM_LAUNCH(SPEC_GENERIC, 1, DefNT<TS>::value, false);
    } else if constexpr (SPEC_BUILT) {
      if constexpr (HotCombo<FORM, GUIDE, XE>::value) {
        // the north-star kernels (2M / 1st-order update, no guidance): (tiles per iteration, nt mask) by situation and
        // dtypes, from profiles/r01_tuning_v3.txt / r01_tuning_v4.txt:
        //   inputs cache-resident: default policy, two tiles per iteration when there is work for it
        //   inputs from HBM:       one tile per iteration, nt loads (+ nt m store for fp32 + fp32), see below
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
            case 32 + 0: DPM_LAUNCH(SPEC_NOISE_X0, 4, 0, false); break;
            case 32 + 1: DPM_LAUNCH(SPEC_NOISE_X0, 4, 1, false); break;
            case 32 + 5: DPM_LAUNCH(SPEC_NOISE_X0, 4, 5, false); break;
            case 64 + 0: DPM_LAUNCH(SPEC_NOISE_X0, 8, 0, false); break;
            case 64 + 1: DPM_LAUNCH(SPEC_NOISE_X0, 8, 1, false); break;
            case 64 + 5: DPM_LAUNCH(SPEC_NOISE_X0, 8, 5, false); break;
            default: DPM_LAUNCH(SPEC_NOISE_X0, DEF_U, DefNT<TS>::value, false); break;
          }
        } else
#endif
        if (resident) {
          if (big) DPM_LAUNCH(SPEC_NOISE_X0, 2, 0, false); else DPM_LAUNCH(SPEC_NOISE_X0, 1, 0, false);
        } else {
          // from HBM: ONE tile per workgroup for every dtype pair.  Round 1 picked two tiles for 4-byte states from the
          // interleaved-requests emulation (15.4 vs 15.6 us); INSIDE a torch network loop (profiles/r03_in_loop.md,
          // rocprofv3 rows, 342 launches each) one tile is 15.0 us against 16.3, and four / eight tiles -- fewer, fatter
          // wavefronts with every load issued up front, the emulation's favourite at 14.4 us -- are 15.2 / 23.8 us.
          bool dma = false;
#if DPM_LAB  // the LDS-DMA variant (2-byte state and network output, no ragged tail): an experiment of the lab build
          if constexpr (sizeof(TS) == 2 && sizeof(TE) == 2) dma = (tn.lds_dma < 0 ? DPM_LDS_DMA_DEFAULT : tn.lds_dma) != 0 && b->n % EPT == 0;
          if (dma) {
            if constexpr (sizeof(TS) == 2 && sizeof(TE) == 2) {
              const auto sh = shape_for(1);
              launch(stage_kernel_dma<TS, TE, FORM, CNT>, sh.first, sh.second, (size_t)(sh.second.x / 64) * 3072, stream, x, e0, h1, xo, mo,
                     b->n, p);
            }
          }
#endif
          if (!dma) DPM_LAUNCH(SPEC_NOISE_X0, 1, CNT, false);
        }
      } else {
        DPM_LAUNCH(SPEC_NOISE_X0, DEF_U, DefNT<TS>::value, false);
      }
    }
#undef DPM_LAUNCH
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "stage kernel launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

template <typename TS, typename TE, int FORM, int GUIDE, bool XE>
int launch_typed(const dpm_stage* st, const dpm_buffers* b, const LaunchCtx& stream) {
  const Operands<TS, TE> op(st, b);
  return (st->flags & DPM_F_THRESH) ? launch_thresh<TS, TE, FORM, GUIDE, XE>(st, b, stream, op)
                                    : launch_stream<TS, TE, FORM, GUIDE, XE>(st, b, stream, op);
}

// Launch shape of the fused kernel (profiles/r02_tune_multi.txt, 32 x [256,4,64,64], kernel-only per request-stage):
// the inputs of a fused launch always come from HBM (R x 42 MB of other requests' traffic passed since they were
// written) -> streaming loads; one super-tile per workgroup -- a grid of all R x tiles workgroups, no grid-stride loop:
// fp16 7.95 us with the grid capped at 8 workgroups per CU, 7.7 / 7.5 at 16 / 32 per CU, 6.93 uncapped (0.757 of the
// HBM peak; fp32 15.4 -> 13.9, fp32 state + fp16 output 13.3 -> 12.6); two tiles per workgroup for 4-byte states.
template <typename TS, typename TE>
struct MultiShape {
  static constexpr int U = (sizeof(TS) == 4) ? 2 : 1;
  static constexpr int NT = 1;
  static constexpr int THREADS = 256;  // per workgroup (256 / 512: stage_kernel_multi)
};

// ---- fused multi-request launch of the streaming family (stage_kernel_multi); thresholded stages fuse inside
// launch_typed (LaunchCtx::multi)
template <typename TS, typename TE, int FORM, int GUIDE, int SPEC>
int launch_multi_spec(const dpm_stage* st, const dpm_buffers* bs, int n_req, const LaunchCtx& c) {
  const DeviceInfo& di = device_info();
  const int n_cu = di.n_cu > 0 ? di.n_cu : 256;
  const Tuning tn = tuning_for(bs[0].opts);
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
    tab.xo2[r] = bs[r].x_out2;
  }
  const KParams p = make_params(st);
  const int64_t n = bs[0].n;
  const int64_t ntiles = ((n / EPT) + 255) / 256;
  auto go = [&](auto kern, int u) {
    const int64_t spr = (ntiles + u - 1) / u;
    const int64_t groups = spr * n_req;                     // 256-lane groups of work: one super-tile each
    const int bt = tn.block_threads > 0 ? tn.block_threads : MultiShape<TS, TE>::THREADS;
    const int64_t per = bt / 256;
    const bool remap = tn.multi_xcd_remap < 0 ? sizeof(TS) == 2 : tn.multi_xcd_remap != 0;
    const uint32_t span = remap ? (uint32_t)((groups + 7) / 8) : 0u;   // super-tiles per XCD
    int64_t blocks = span ? 8 * (((int64_t)span + per - 1) / per) : (groups + per - 1) / per;
    if (tn.multi_blocks_per_cu > 0) {  // tuning hook: cap the grid, workgroups loop over the super-tiles
      const int64_t cap = (int64_t)n_cu * tn.multi_blocks_per_cu;
      if (blocks > cap) blocks = cap;
    }
    launch(kern, dim3((unsigned)blocks), dim3((unsigned)bt), 0, c, tab, n, (uint32_t)n_req, (uint32_t)spr, p, span);
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
    if (b.eps_stride && b.eps_stride != b.n / b.batch) return MULTI_NOT_BUILT;
    if (b.x_out2 && (st->guidance != DPM_GUIDE_CFG || !aligned(b.x_out2, as))) return MULTI_NOT_BUILT;
    if (b.xe && b.x && b.xe != b.x) return MULTI_NOT_BUILT;
    if (!(aligned(b.x, as) && aligned(b.xe, as) && aligned(b.h1, as) && aligned(b.h2, as) && aligned(b.x_out, as) &&
          aligned(b.m_out, as) && aligned(b.e0, ae) && aligned(b.e1, ae)))
      return MULTI_NOT_BUILT;
  }
  const bool x0 = (st->flags & DPM_F_TO_X0) != 0;
  const bool cfg = st->guidance == DPM_GUIDE_CFG;
  // x_start / v / score networks (and an alpha the division-by-invariant guard rejects) take the general prologue
  // ... and so does the eps form of a noise network (launch_stream: no compile-time prologue of its own since round 5)
  const bool generic = st->model_type != DPM_MODEL_NOISE || !x0 || !div_invariant_ok(st->alpha_e) ||
                       tuning_for(bs[0].opts).force_generic != 0;
#define DPM_MULTI(FORM_)                                                                                        \
  (generic ? (cfg ? launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_CFG, SPEC_GENERIC>(st, bs, n_req, c)             \
                  : launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_NONE, SPEC_GENERIC>(st, bs, n_req, c))           \
   : cfg   ? launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_CFG, SPEC_NOISE_X0>(st, bs, n_req, c)                   \
           : launch_multi_spec<TS, TE, FORM_, DPM_GUIDE_NONE, SPEC_NOISE_X0>(st, bs, n_req, c))
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
