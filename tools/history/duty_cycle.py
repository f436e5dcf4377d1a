#!/usr/bin/env python3
"""Does the fused 2M launch run faster when the chip is not streaming continuously?  bench.py times thousands of fused launches back
to back (a sustained 2.5 s region); inside a real network loop the same launch (tools/in_loop.py --requests R) reads 10-15 %
shorter by rocprofv3 rows.  Here: the bench's own launch (32 requests of [256,4,64,64] fp16, inputs from HBM: the requests rotate
through 3 sets = 4 GB) timed by events with an idle gap of G ms before every launch (host sleep, or a GEMM burst standing in for a
network).   python tools/duty_cycle.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dpm_solver_amd as D  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, algorithm_type="dpmsolver++", state_dtype=torch.float16)
    plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    R = 32
    groups = [bench.make_sets(R, torch.float16, dev, seed=5 + g) for g in range(3)]
    st = plan.stages[5]                                    # a steady-state 2M stage
    arrs = []
    for sets in groups:
        a = (L.Buffers * R)()
        for r, s_ in enumerate(sets):
            b = a[r]
            b.x, b.e0, b.h1 = s_["x"][0].data_ptr(), s_["eps"].data_ptr(), s_["h"][0].data_ptr()
            b.x_out, b.m_out = s_["x"][1].data_ptr(), s_["h"][1].data_ptr()
            b.n, b.batch = s_["x"][0].numel(), bench.B
            b.state_dtype = b.eps_dtype = L.DTYPE_F16
        arrs.append(a)
    stc = st.copy()
    stc.flags |= L.F_STORE_M
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    alg = 5 * groups[0][0]["x"][0].numel() * 2 * R
    w = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    scratch = torch.zeros(768 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    net = bench.LoopNet("gemm", 256, torch.float16, dev)
    xin = groups[0][0]["x"][0]
    tvec = torch.full((bench.B,), 500.0, device=dev)

    def launch(i):
        L.check(L.lib.dpm_stage_launch_multi(C.byref(stc), arrs[i % 3], R, sp))

    for i in range(6):
        launch(i)
    torch.cuda.synchronize()
    for label, gap in (("back to back", None), ("host idle 0.2 ms", 0.0002), ("host idle 1 ms", 0.001), ("host idle 5 ms", 0.005),
                       ("GEMM burst ~1 ms (8192^3 fp16) before every launch", "gemm"),
                       ("10 GEMM bursts (~10 ms) before every launch", "gemm10"),
                       ("768 MiB READ sweep (sum) before every launch", "read"),
                       ("768 MiB WRITE sweep (fill) before every launch", "write"),
                       ("1 LoopNet call (1.25 ms, ~3 GB of traffic) before every launch", "net1"),
                       ("16 LoopNet calls (20 ms) before every launch", "net16"), ("back to back again", None)):
        ts = []
        for i in range(60 if gap not in ("net16", "gemm10") else 30):
            if gap == "gemm":
                torch.mm(w, w)
            elif gap == "read":
                scratch.sum()
            elif gap == "write":
                scratch.fill_(1.0)
            elif gap == "gemm10":
                for _ in range(10):
                    torch.mm(w, w)
            elif gap in ("net1", "net16"):
                with torch.no_grad():
                    for _ in range(1 if gap == "net1" else 16):
                        net(xin, tvec)
            elif gap:
                torch.cuda.synchronize()
                time.sleep(gap)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(i)
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        us = np.array([a.elapsed_time(b) for a, b in ts[10:]]) * 1e3
        print("%-52s fused launch %7.1f us median (p10 %.1f, p90 %.1f)  %.3f of 8 TB/s" % (
            label, np.median(us), np.percentile(us, 10), np.percentile(us, 90), alg / np.median(us) / 1e3 / 8000.0), flush=True)


if __name__ == "__main__":
    main()
