#!/usr/bin/env python3
"""Probe of the sporadic 50-85 ms start->stop interval of a launch timed with hipExtLaunchKernelGGL events
(profiles/r02_stall.md): many timed trajectories of tiny and large stage kernels; for every outlier the wall time of the
whole call (host clock around launch + synchronise) is printed next to the event interval, which tells a real stall of
the device from a glitch of the event pair."""
import ctypes as C
import sys
import time
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (tools run on the LAB build of the library: include/dpm_lab.h)
import dpm_solver_amd as D  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402


def main():
    dev = "cuda:0"
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(np.cumprod(1.0 - betas).astype(np.float32)))
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sp = C.c_void_p(stream.cuda_stream)
    for shape, runs in (((3, 3, 37, 41), 3000), ((256, 4, 64, 64), 1500)):
        x = torch.randn(shape, device=dev)
        eps = torch.randn(shape, device=dev)
        dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: eps, ns), ns)
        plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                             lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
        xb = [x.clone()] + [torch.empty_like(x) for _ in range(3)]
        hb = [torch.empty_like(x) for _ in range(3)]
        rb = L.RunBuffers()
        for i in range(4):
            rb.xbuf[i] = xb[i].data_ptr()
        for i in range(3):
            rb.hist[i] = hb[i].data_ptr()
        rb.e0 = eps.data_ptr()
        rb.n, rb.batch, rb.state_dtype, rb.eps_dtype = x.numel(), shape[0], L.DTYPE_F32, L.DTYPE_F32
        ms = (C.c_float * len(plan.stages))()
        res = C.c_int()
        walls, outliers = [], []
        for r in range(runs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            L.check(L.lib.dpm_plan_run_multi(plan.handle, C.byref(rb), 1, sp, ms, C.byref(res)))
            w = (time.perf_counter() - t0) * 1e3
            a = np.frombuffer(ms, dtype=np.float32)
            walls.append(w)
            if a.max() > 5.0:
                outliers.append((r, int(a.argmax()), float(a.max()), w))
        walls = np.array(walls)
        print("shape %s: %d timed trajectories x %d launches, wall per call median %.3f ms, p99 %.3f, max %.3f" % (
            shape, runs, len(plan.stages), np.median(walls), np.percentile(walls, 99), walls.max()))
        for r, i, m, w in outliers:
            print("   run %d stage %d: event interval %.2f ms, wall of the whole call %.2f ms" % (r, i, m, w))
        print("   %d outliers (> 5 ms) in %d launches" % (len(outliers), runs * len(plan.stages)), flush=True)


if __name__ == "__main__":
    main()
