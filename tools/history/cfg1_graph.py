#!/usr/bin/env python3
"""BASELINE cfg1 / cfg4's per-GPU shape ([8,4,64,64] fp32, 2M++, 20 steps) is launch-bound: 512 KiB per tensor, a stage
kernel runs for ~1.5 us.  This drives that trajectory three ways for a profiler (rocprofv3 --kernel-trace) or a stopwatch:
eager DPM_Solver.sample(), the hipGraph replay of DPM_Solver.capture(), and the native C loop (dpm_plan_run).
    python tools/cfg1_graph.py [--reps 200]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpm_solver_amd as D  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    dev = "cuda:0"
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(np.cumprod(1.0 - betas).astype(np.float32)))
    shape = (8, 4, 64, 64)
    e = torch.randn(shape, device=dev)
    x = torch.randn(shape, device=dev)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: e, ns), ns)
    kw = dict(steps=20, order=2)
    g = dpm.capture(x, **kw)
    plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    bufs = [torch.empty(shape, device=dev) for _ in range(6)]
    rb = L.RunBuffers()
    rb.xbuf[0], rb.e0 = x.data_ptr(), e.data_ptr()
    for j in range(3):
        rb.xbuf[1 + j] = bufs[j].data_ptr()
        rb.hist[j] = bufs[3 + j].data_ptr()
    rb.n, rb.batch, rb.state_dtype, rb.eps_dtype = x.numel(), shape[0], L.DTYPE_F32, L.DTYPE_F32
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = C.c_int(-1)

    def native():
        L.check(L.lib.dpm_plan_run(plan.handle, C.byref(rb), None, None, stream, C.byref(res)))
    for name, fn in (("eager sample()", lambda: dpm.sample(x, **kw)), ("captured graph replay", lambda: g(x)),
                     ("native dpm_plan_run", native)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        print("%-22s %7.1f us per 20-step trajectory (%.2f us per stage)" % (
            name, (time.perf_counter() - t0) / args.reps * 1e6, (time.perf_counter() - t0) / args.reps * 1e6 / 20))


if __name__ == "__main__":
    main()
