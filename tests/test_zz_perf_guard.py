"""Performance regression guard: the other GPU tests pin bits, this one pins time.

Four figures of the PRODUCT library, each measured in well under a second, against thresholds that are the figure measured
on MI355X boxes (profiles/r06_perf_guard.md: ten runs on two boxes, plus the spread over the boxes of rounds 3-5) + 8 %:

  fused           the headline launch -- 32 requests of [256,4,64,64] fp16 advanced by ONE stage_kernel_multi launch per stage,
                  inputs from HBM -- by HIP events around whole trajectories on the launch stream: >= FUSED_MIN_FRAC of 8 TB/s
  lone_cold       the same requests advanced with ONE launch each (dpm_launch_opts.no_fuse), kernel-only: <= LONE_COLD_MAX_US
  cfg5_stage      DPM-Solver++ 2M + dynamic thresholding [32,3,64,64] fp32, hipGraph-replayed stages: <= CFG5_STAGE_MAX_US
  small_stage     [8,4,64,64] fp32 2M stages back to back, hipGraph-replayed (latency-bound): <= SMALL_STAGE_MAX_US

The captured (hipGraph) form is used for the two small cases so that the figure is the GPU's, not the Python host's.  A
shared or throttled box can be slower than any kernel regression: every figure is the best of three short regions, a figure
that misses its threshold is measured again (up to three attempts, a second apart) before the test fails, and the test prints
what it measured (`pytest -s`) so that a failure shows by how much.  The file sorts LAST in the suite (test_zz_*): the
driver runs `pytest -x`, and a noisy box must not keep the bit-parity tests from running.
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

# thresholds = worst figure measured + 8 % (profiles/r06_perf_guard.md: ten runs on two boxes, plus the spread over the boxes of
# rounds 3-5 for the two headline figures)
FUSED_MIN_FRAC = 0.72           # 0.774-0.789 on three boxes here; 0.767-0.795 over the boxes of rounds 3-6
LONE_COLD_MAX_US = 9.3          # 8.39-8.50 (one run 8.995); 8.16-8.60 over the boxes of round 5
CFG5_STAGE_MAX_US = 13.1        # 12.09-12.16 captured (10.17 by rocprofv3 rows + the graph's node-to-node latency)
CFG5_CLUSTERED_MAX_US = 11.6    # 10.70-10.77: the clustered route (k = 6 workgroups per sample: what the eager loop launches) kept under capture
SMALL_STAGE_MAX_US = 3.0        # 2.51-2.61 captured

RESULTS = {}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from dpm_solver_amd import _lib as L
    if L.IS_LAB:
        pytest.skip("the guard times the product library")
    return torch.device("cuda", 0)


def _requests(R, dev):
    """R requests of [256,4,64,64] fp16 as dpm_run_buffers (x_T, frozen eps, 3 scratch states, 2 history slots), generated on
    the device"""
    from dpm_solver_amd import _lib as L
    g = torch.Generator(device=dev).manual_seed(7)
    keep, rbs = [], (L.RunBuffers * R)()
    for r in range(R):
        x = torch.randn((256, 4, 64, 64), generator=g, device=dev, dtype=torch.float16)
        eps = torch.randn((256, 4, 64, 64), generator=g, device=dev, dtype=torch.float16)
        xb = [x] + [torch.empty_like(x) for _ in range(3)]
        hb = [torch.empty_like(x) for _ in range(2)]
        rb = rbs[r]
        for i in range(4):
            rb.xbuf[i] = xb[i].data_ptr()
        for i in range(2):
            rb.hist[i] = hb[i].data_ptr()
        rb.e0 = eps.data_ptr()
        rb.n, rb.batch = x.numel(), 256
        rb.state_dtype = rb.eps_dtype = L.DTYPE_F16
        keep.append((xb, hb, eps))
    return rbs, keep


def _attempts(measure, ok, n=3):
    """measure() until ok(result) or n attempts; returns the last result"""
    res = None
    for k in range(n):
        res = measure()
        if ok(res):
            break
        time.sleep(1.0)
    return res


def test_fused_and_lone_launch_of_the_headline_workload(dev):
    import bench
    import dpm_solver_amd as D
    from dpm_solver_amd import _lib as L
    R, n_st = 32, 20
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, algorithm_type="dpmsolver++", state_dtype=torch.float16)
    plan = dpm._get_plan(method="multistep", order=2, steps=n_st, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    rbs, keep = _requests(R, dev)
    stream = torch.cuda.Stream(device=dev)
    sptr = C.c_void_p(stream.cuda_stream)
    res = (C.c_int * R)()
    n_el = 256 * 4 * 64 * 64
    traj_bytes = n_el * 2 * (18 * 5 + 4 + 4) * R                 # 18 steady stages (3 reads + 2 writes), first and last 4 streams
    with torch.cuda.stream(stream):
        for _ in range(2):
            L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, sptr, None, res))

        def fused():
            best = None
            for _ in range(3):
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(4):                                # 80 fused launches, ~17 ms
                    L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, sptr, None, res))
                e1.record(stream)
                torch.cuda.synchronize(dev)
                us = e0.elapsed_time(e1) * 1e3 / 4
                best = us if best is None else min(best, us)
            return best
        best = _attempts(fused, lambda us: traj_bytes / us / 1e3 / 8000.0 >= FUSED_MIN_FRAC)
        frac = traj_bytes / best / 1e3 / 8000.0
        RESULTS["fused_frac"] = round(frac, 4)
        RESULTS["fused_launch_us"] = round(best / n_st, 2)
        # the same requests, one launch each, kernel-only (start -> stop events of every launch)
        opts = L.LaunchOpts()
        opts.no_fuse = 1
        ms = (C.c_float * (R * n_st))()

        def lone_cold():
            rbs[0].opts = C.pointer(opts)
            cold = []
            for _ in range(2):
                L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, sptr, ms, res))
                cold.append(np.frombuffer(ms, dtype=np.float32).reshape(R, n_st)[:, 1:n_st - 1].astype(np.float64) * 1e3)
            rbs[0].opts = None
            cold = np.concatenate(cold)
            med = float(np.median(cold))
            return float(cold[cold < 50.0 * med].mean())
        lone = _attempts(lone_cold, lambda us: us <= LONE_COLD_MAX_US)
        RESULTS["lone_cold_us"] = round(lone, 3)
    print("\n[perf guard] fused: %.4f of 8 TB/s (%.1f us per launch); lone cold launch: %.2f us" % (frac, best / n_st, lone))
    assert frac >= FUSED_MIN_FRAC, "fused 32-request 2M launch: %.4f of peak < %.2f" % (frac, FUSED_MIN_FRAC)
    assert lone <= LONE_COLD_MAX_US, "lone cold [256,4,64,64] fp16 2M launch: %.2f us > %.1f" % (lone, LONE_COLD_MAX_US)


def test_small_and_thresholded_stages(dev):
    import config_bench
    t0 = time.perf_counter()
    cap = lambda r: r["captured"]["us_per_stage"]
    small = _attempts(lambda: config_bench.measure_frozen("cfg1", dev, captured=True, scale_k=0.25), lambda r: cap(r) <= SMALL_STAGE_MAX_US)
    thr = _attempts(lambda: config_bench.measure_frozen("cfg5", dev, captured=True, scale_k=0.25), lambda r: cap(r) <= CFG5_STAGE_MAX_US)
    # under capture a sample that fits one workgroup takes the cluster-free shape by default; the eager loop -- what
    # sample() runs -- uses k = 6 workgroup clusters per sample here: the guard watches that kernel too (cluster_in_graph)
    thr_k = _attempts(lambda: config_bench.measure_frozen("cfg5", dev, captured=True, scale_k=0.25, attrs=dict(cluster_in_graph=True)),
                      lambda r: cap(r) <= CFG5_CLUSTERED_MAX_US)
    RESULTS["cfg5_stage_clustered_us"] = thr_k["captured"]["us_per_stage"]
    RESULTS["small_stage_us"] = small["captured"]["us_per_stage"]
    RESULTS["small_stage_eager_us"] = small["us_per_stage"]
    RESULTS["cfg5_stage_us"] = thr["captured"]["us_per_stage"]
    RESULTS["cfg5_stage_eager_us"] = thr["us_per_stage"]
    print("\n[perf guard] [8,4,64,64] stage: %.2f us captured / %.2f us eager; cfg5 thresholded stage: %.2f us captured / %.2f us "
          "eager (%.1f s)" % (small["captured"]["us_per_stage"], small["us_per_stage"], thr["captured"]["us_per_stage"],
                              thr["us_per_stage"], time.perf_counter() - t0))
    assert small["captured"]["us_per_stage"] <= SMALL_STAGE_MAX_US, small["captured"]
    assert thr["captured"]["us_per_stage"] <= CFG5_STAGE_MAX_US, thr["captured"]
    print("[perf guard] cfg5 stage with its workgroup clusters kept under capture: %.2f us" % thr_k["captured"]["us_per_stage"])
    assert thr_k["captured"]["us_per_stage"] <= CFG5_CLUSTERED_MAX_US, thr_k["captured"]


def test_write_guard_record(dev):
    """the figures of this run as one JSON line (gpurun_out/perf_guard.jsonl when that directory exists): the record the
    thresholds are derived from"""
    import json
    if RESULTS and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        with open(os.path.join(ROOT, "gpurun_out", "perf_guard.jsonl"), "a") as f:
            f.write(json.dumps(dict(RESULTS, gpu=torch.cuda.get_device_name(0))) + "\n")
    assert "fused_frac" in RESULTS or "small_stage_us" in RESULTS
