// dpm_resident.hip -- EXPERIMENT: a RESIDENT stage kernel that is woken by
// a stream-ordered memory write instead of being dispatched.
//
// A lone [256,4,64,64] stage launch inside a real sampling loop spends ~2 of its 8.6 us ramping up and draining (dispatch,
// kernarg fetch, first-byte latency, last stores).  The one structural idea that had no measurement: keep the solver's
// workgroups on the chip for the whole trajectory.  One launch per trajectory on a side stream; every workgroup sleeps
// (s_sleep between polls) on a `go` word of the stage; the host enqueues, behind the network's last kernel,
//     hipStreamWriteValue64(stream, &go[s], <pointer of the network's output>)   -- wakes the workgroups, carries the pointer
//     hipStreamWaitValue32 (stream, &done[s], 1)                                  -- the next network call waits for the stage
// and the last workgroup to finish stage s writes done[s].  The tile body is the product's own (stage_tiles<>): unguided
// noise-prediction network, DPM-Solver++ multistep -- first-order and second-order forms -- which is everything the
// 20-step 2M trajectory of BASELINE configs[1] launches.  Nothing in DPM_Solver uses this; tools/in_loop.py --resident
// measures it against the dispatched kernel (profiles/r04_resident.md has the verdict).
#include "dpm_device.hpp"
#include "dpm_lab.h"

namespace {

constexpr int RES_MAX_STAGES = 64;

struct ResStage {
  KParams p;
  const void* x;
  const void* h1;
  void* xo;
  void* mo;
};

struct ResCtl {
  uint64_t go[RES_MAX_STAGES];      // 0, then the network output's address (hipStreamWriteValue64)
  uint32_t arrive[RES_MAX_STAGES];  // workgroups through with the stage
  uint32_t abort;                   // a wait ran out (a signal never came): every workgroup leaves, the done words are raised
};
// polls before a resident workgroup gives up on a stage's signal (~2 us each at sleep = 1: about ten seconds) -- a missed
// dpm_resident_signal must not hang the GPU
constexpr uint32_t RES_SPIN_LIMIT = 5u << 20;

template <typename TS, typename TE>
__global__ __launch_bounds__(256) void resident_kernel(const ResStage* __restrict__ stages, ResCtl* ctl, uint32_t** done,
                                                       int n_stages, int64_t n, int sleep) {
  __shared__ uint64_t eps_sh;
  const int64_t ngroups = n / EPT;
  const int64_t ntiles = (ngroups + 255) / 256;
  const KExt ext = {};
  for (int s = 0; s < n_stages; ++s) {
    if (threadIdx.x == 0) {
      uint64_t v;
      uint32_t spins = 0;
      // `sleep` x s_sleep 64 (64 x 64 clocks, ~2 us) between polls: pollers that sleep shorter take bandwidth from the
      // network that runs next to them (MI355X_MICROARCH.md prices busy pollers at up to -37 % of the chip's bandwidth)
      while ((v = __hip_atomic_load(&ctl->go[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) == 0ull) {
        for (int r = 0; r < sleep; ++r) __builtin_amdgcn_s_sleep(64);
        if (++spins > RES_SPIN_LIMIT || ((spins & 63u) == 0u && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      eps_sh = v;
    }
    __syncthreads();
    if (eps_sh == 0ull) {  // aborted: release every stream wait of the remaining stages and leave
      if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int q = s; q < n_stages; ++q) __hip_atomic_store(done[q], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    // the network's last kernel wrote eps (and this kernel may hold stale lines of the same address from an earlier stage)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const TE* e0 = reinterpret_cast<const TE*>(eps_sh);
    const ResStage& st = stages[s];
    const KParams p = st.p;
    const TS* x = static_cast<const TS*>(st.x);
    const TS* h1 = static_cast<const TS*>(st.h1);
    TS* xo = static_cast<TS*>(st.xo);
    TS* mo = static_cast<TS*>(st.mo);
    if (p.form == DPM_FORM_TWO) {
      for (int64_t t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x)
        stage_tiles<TS, TE, DPM_FORM_TWO, DPM_GUIDE_NONE, false, SPEC_NOISE_X0, 1, 5, false>(x, nullptr, e0, nullptr, nullptr, h1,
                                                                                            nullptr, xo, mo, ngroups, t0, p, ext);
    } else {
      for (int64_t t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x)
        stage_tiles<TS, TE, DPM_FORM_LIN1, DPM_GUIDE_NONE, false, SPEC_NOISE_X0, 1, 5, false>(x, nullptr, e0, nullptr, nullptr,
                                                                                             nullptr, nullptr, xo, mo, ngroups, t0, p, ext);
    }
    // this workgroup's stores are out (write-through) -> arrive; the last one tells the stream
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t old = __hip_atomic_fetch_add(&ctl->arrive[s], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old == gridDim.x - 1u) __hip_atomic_store(done[s], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

struct Resident {
  int n_stages = 0, wgs = 0, sleep = 1, state_dtype = 0;
  int64_t n = 0;
  ResStage* d_stages = nullptr;
  ResStage h_stages[RES_MAX_STAGES];
  ResCtl* d_ctl = nullptr;
  uint32_t** d_done = nullptr;          // device array of the signal words' addresses
  uint32_t* done[RES_MAX_STAGES] = {};  // hipMallocSignalMemory: one 8-byte allocation per stage
};

}  // namespace

extern "C" int dpm_resident_create(const dpm_stage* stages, const dpm_buffers* bufs, int n_stages, int workgroups, int sleep,
                                   void** out) {
  if (!stages || !bufs || !out || n_stages < 1 || n_stages > RES_MAX_STAGES || workgroups < 1)
    return dpm_set_error(DPM_ERR_ARG, "resident_create: bad arguments");
  Resident* r = new (std::nothrow) Resident();
  if (!r) return dpm_set_error(DPM_ERR_NOMEM, "resident_create: out of memory");
  {
    // every workgroup of the grid has to be resident at once (the last one to arrive raises done[s]): cap the grid at what
    // the device holds -- and at 7 of its 8 wavefront slots per SIMD, so that the network's kernels can still be scheduled
    // (2048 workgroups hung the loop in round 4)
    int occ = 0;
    const void* kf = bufs[0].state_dtype == DPM_DTYPE_F16 ? reinterpret_cast<const void*>(&resident_kernel<__half, __half>)
                                                            : reinterpret_cast<const void*>(&resident_kernel<float, float>);
    const DeviceInfo& di = device_info();
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kf, 256, 0) == hipSuccess && occ > 0 && di.n_cu > 0) {
      const int cap = di.n_cu * std::max(1, std::min(occ, 8) - 1);
      if (workgroups > cap) workgroups = cap;
    }
  }
  r->n_stages = n_stages;
  r->wgs = workgroups;
  r->sleep = sleep;
  r->n = bufs[0].n;
  r->state_dtype = bufs[0].state_dtype;
  for (int s = 0; s < n_stages; ++s) {
    const dpm_stage& st = stages[s];
    const dpm_buffers& b = bufs[s];
    const bool ok = (st.form == DPM_FORM_TWO || st.form == DPM_FORM_LIN1) && st.guidance == DPM_GUIDE_NONE &&
                    st.model_type == DPM_MODEL_NOISE && (st.flags & DPM_F_TO_X0) && !(st.flags & (DPM_F_THRESH | DPM_F_BLEND)) &&
                    (!b.xe || b.xe == b.x) && b.state_dtype == b.eps_dtype && b.n % (EPT * 256) == 0 && div_invariant_ok(st.alpha_e) &&
                    (b.state_dtype == DPM_DTYPE_F16 || b.state_dtype == DPM_DTYPE_F32);
    if (!ok) {
      delete r;  // (no device allocation yet)
      return dpm_set_error(DPM_ERR_UNSUPPORTED, "resident_create: stage %d is outside the experiment (unguided noise-prediction "
                           "2M++ stages, equal fp16 / fp32 dtypes, whole tiles)", s);
    }
    r->h_stages[s].p = make_params(&st);
    r->h_stages[s].x = b.x;
    r->h_stages[s].h1 = b.h1;
    r->h_stages[s].xo = b.x_out;
    r->h_stages[s].mo = b.m_out;
  }
  hipError_t e = hipMalloc(&r->d_stages, sizeof(ResStage) * RES_MAX_STAGES);
  if (e == hipSuccess) e = hipMalloc(&r->d_ctl, sizeof(ResCtl));
  if (e == hipSuccess) e = hipMalloc(&r->d_done, sizeof(uint32_t*) * RES_MAX_STAGES);
  for (int s = 0; s < n_stages && e == hipSuccess; ++s)
    e = hipExtMallocWithFlags(reinterpret_cast<void**>(&r->done[s]), 8, hipMallocSignalMemory);
  if (e == hipSuccess) e = hipMemcpy(r->d_done, r->done, sizeof(uint32_t*) * RES_MAX_STAGES, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    const int rc = dpm_set_error((int)e, "resident_create: %s", hipGetErrorString(e));
    dpm_resident_destroy(r);  // frees whatever was allocated
    return rc;
  }
  *out = r;
  return DPM_OK;
}

// start one trajectory: x_first = the state the first stage reads (the caller's x_T), x_last_out = where the last stage
// writes; clears the control words and launches the resident kernel on `side` (a stream other than the network's)
extern "C" int dpm_resident_start(void* h, const void* x_first, void* x_last_out, void* side) {
  Resident* r = static_cast<Resident*>(h);
  if (!r) return dpm_set_error(DPM_ERR_ARG, "resident_start: null handle");
  hipStream_t st = static_cast<hipStream_t>(side);
  if (x_first) r->h_stages[0].x = x_first;
  if (x_last_out) r->h_stages[r->n_stages - 1].xo = x_last_out;
  hipError_t e = hipMemcpyAsync(r->d_stages, r->h_stages, sizeof(ResStage) * r->n_stages, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(r->d_ctl, 0, sizeof(ResCtl), st);
  for (int s = 0; s < r->n_stages && e == hipSuccess; ++s) e = hipMemsetAsync(r->done[s], 0, 8, st);
  if (e != hipSuccess) return dpm_set_error((int)e, "resident_start: %s", hipGetErrorString(e));
  if (r->state_dtype == DPM_DTYPE_F16)
    hipLaunchKernelGGL((resident_kernel<__half, __half>), dim3(r->wgs), dim3(256), 0, st, r->d_stages, r->d_ctl, r->d_done, r->n_stages,
                       r->n, r->sleep);
  else
    hipLaunchKernelGGL((resident_kernel<float, float>), dim3(r->wgs), dim3(256), 0, st, r->d_stages, r->d_ctl, r->d_done, r->n_stages,
                       r->n, r->sleep);
  e = hipGetLastError();
  if (e != hipSuccess) return dpm_set_error((int)e, "resident_start: launch failed: %s", hipGetErrorString(e));
  return DPM_OK;
}

// behind the network's last kernel on `stream`: wake stage s with the output's address, make the stream wait for the stage
extern "C" int dpm_resident_signal(void* h, int s, const void* eps, void* stream) {
  Resident* r = static_cast<Resident*>(h);
  if (!r || s < 0 || s >= r->n_stages || !eps) return dpm_set_error(DPM_ERR_ARG, "resident_signal: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = hipStreamWriteValue64(st, &r->d_ctl->go[s], (uint64_t)reinterpret_cast<uintptr_t>(eps), 0);
  if (e == hipSuccess) e = hipStreamWaitValue32(st, r->done[s], 1u, hipStreamWaitValueEq, 0xffffffffu);
  if (e != hipSuccess) return dpm_set_error((int)e, "resident_signal: %s", hipGetErrorString(e));
  return DPM_OK;
}

extern "C" void dpm_resident_destroy(void* h) {
  Resident* r = static_cast<Resident*>(h);
  if (!r) return;
  for (int s = 0; s < RES_MAX_STAGES; ++s)
    if (r->done[s]) (void)hipFree(r->done[s]);
  if (r->d_stages) (void)hipFree(r->d_stages);
  if (r->d_ctl) (void)hipFree(r->d_ctl);
  if (r->d_done) (void)hipFree(r->d_done);
  delete r;
}
