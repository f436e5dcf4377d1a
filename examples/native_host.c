/* native_host.c -- the C ABI (include/dpm_hip.h) driven from plain C, no Python, no PyTorch.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/native_host.c \
 *       -Ldpm_solver_amd -ldpm_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN/../dpm_solver_amd' -o examples/native_host
 *   examples/native_host [out.bin]
 *
 * Builds the SD-v1 schedule, plans DPM-Solver++(2M) with 20 steps, and runs the trajectory with a model callback --
 * a stand-in "network" that writes eps = 0.5 * x with one hipMemcpy + the library's own add_noise kernel would be
 * overkill, so the callback simply enqueues a device-to-device copy of a frozen eps buffer (what a host that owns its
 * own network would replace with its inference call).  Prints a checksum and optionally dumps the final state, which
 * tests/test_gpu_extensions.py compares bit for bit with the Python host's result on the same inputs.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dpm_hip.h"

#define CHECK_HIP(x)                                                                  \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)
#define CHECK_DPM(x)                                                         \
  do {                                                                       \
    int rc_ = (x);                                                           \
    if (rc_ != DPM_OK) {                                                     \
      fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, dpm_last_error()); \
      return 3;                                                              \
    }                                                                        \
  } while (0)

typedef struct {
  const void* frozen_eps;
  size_t bytes;
  int calls;
} net_t;

/* dpm_model_cb: evaluate the "network" on x at st->t_input, write the raw output into e0 (enqueue only) */
static int model_cb(void* user, const dpm_stage* st, const void* x, void* e0, void* e1, void* stream) {
  net_t* net = (net_t*)user;
  (void)st; (void)x; (void)e1;
  net->calls++;
  return hipMemcpyAsync(e0, net->frozen_eps, net->bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : 1;
}

/* deterministic pseudo-normal inputs (sum of 4 LCG uniforms), reproduced by the test in numpy */
static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s; }
static float pseudo_normal(uint32_t* s) {
  float a = 0.f;
  for (int k = 0; k < 4; ++k) a += (float)(lcg(s) >> 8) * (1.0f / 16777216.0f);
  return (a - 2.0f) * 1.7320508f;
}

int main(int argc, char** argv) {
  const int B = 8, C = 4, H = 64, W = 64, steps = 20;
  const int64_t n = (int64_t)B * C * H * W;
  const size_t bytes = (size_t)n * sizeof(float);

  /* SD-v1 scaled-linear schedule: betas = linspace(sqrt(0.00085), sqrt(0.012), 1000)^2 in double */
  double betas[1000];
  const double b0 = sqrt(0.00085), b1 = sqrt(0.012);
  for (int i = 0; i < 1000; ++i) {
    const double v = b0 + (b1 - b0) * (double)i / 999.0;
    betas[i] = v * v;
  }
  dpm_schedule* sched = NULL;
  CHECK_DPM(dpm_schedule_create_betas_f64(betas, 1000, 1, &sched));

  dpm_plan_desc d;
  memset(&d, 0, sizeof d);
  d.algorithm_type = DPM_ALGO_DPMSOLVERPP;
  d.method = DPM_METHOD_MULTISTEP;
  d.order = 2;
  d.steps = steps;
  d.skip_type = DPM_SKIP_TIME_UNIFORM;
  d.solver_type = DPM_SOLVER_DPMSOLVER;
  d.lower_order_final = 1;
  d.model_type = DPM_MODEL_NOISE;
  d.guidance = DPM_GUIDE_NONE;
  d.t_start = 1.0;
  d.t_end = 1.0 / dpm_schedule_total_N(sched);
  d.guidance_scale = 1.0;
  d.thr_ratio = 0.995;
  d.thr_max = 1.0;
  dpm_plan* plan = NULL;
  CHECK_DPM(dpm_plan_create(sched, &d, &plan));
  const int n_stages = dpm_plan_num_stages(plan), n_slots = dpm_plan_num_slots(plan);

  float* hx = (float*)malloc(bytes);
  float* he = (float*)malloc(bytes);
  uint32_t seed = 12345u;
  for (int64_t i = 0; i < n; ++i) hx[i] = pseudo_normal(&seed);
  for (int64_t i = 0; i < n; ++i) he[i] = pseudo_normal(&seed);

  /* load-time checks of the ABI (include/dpm_hip.h): library not older than the header, struct layouts as compiled */
  if (dpm_version() < DPM_HIP_VERSION || dpm_sizeof(DPM_SIZEOF_RUN_BUFFERS) != sizeof(dpm_run_buffers) ||
      dpm_sizeof(DPM_SIZEOF_STAGE) != sizeof(dpm_stage) || dpm_sizeof(DPM_SIZEOF_PLAN_DESC) != sizeof(dpm_plan_desc)) {
    fprintf(stderr, "libdpm_hip.so (version %d) does not match the header this host was built against (%d)\n", dpm_version(),
            DPM_HIP_VERSION);
    return 5;
  }
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  dpm_run_buffers rb;
  memset(&rb, 0, sizeof rb);
  void* frozen = NULL;
  for (int i = 0; i < 4; ++i) CHECK_HIP(hipMalloc(&rb.xbuf[i], bytes));
  for (int i = 0; i < 3; ++i) CHECK_HIP(hipMalloc(&rb.hist[i], bytes));
  CHECK_HIP(hipMalloc(&rb.e0, bytes));
  CHECK_HIP(hipMalloc(&frozen, bytes));
  CHECK_HIP(hipMemcpy(rb.xbuf[0], hx, bytes, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(frozen, he, bytes, hipMemcpyHostToDevice));
  rb.n = n;
  rb.batch = B;
  rb.state_dtype = DPM_DTYPE_F32;
  rb.eps_dtype = DPM_DTYPE_F32;

  net_t net = {frozen, bytes, 0};
  int where = -1;
  CHECK_DPM(dpm_plan_run(plan, &rb, model_cb, &net, stream, &where));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(hx, rb.xbuf[where], bytes, hipMemcpyDeviceToHost));

  double sum = 0., asum = 0.;
  for (int64_t i = 0; i < n; ++i) {
    sum += hx[i];
    asum += fabs(hx[i]);
  }
  int n_cu = 0, lds = 0;
  char arch[64];
  CHECK_DPM(dpm_device_info(&n_cu, &lds, arch, (int)sizeof arch));
  printf("native_host: %s (%d CUs), library %d, %d stages, %d history slots, %d network calls, result in xbuf[%d]\n", arch,
         n_cu, dpm_version(), n_stages, n_slots, net.calls, where);
  printf("native_host: sum=%.9g abs_sum=%.9g\n", sum, asum);
  if (argc > 1) {
    FILE* f = fopen(argv[1], "wb");
    if (!f || fwrite(hx, 1, bytes, f) != bytes) {
      fprintf(stderr, "cannot write %s\n", argv[1]);
      return 4;
    }
    fclose(f);
  }
  dpm_plan_destroy(plan);
  dpm_schedule_destroy(sched);
  return 0;
}
