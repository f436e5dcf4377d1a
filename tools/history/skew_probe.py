#!/usr/bin/env python3
"""Does the relative placement of a stage's five streams matter?  One 2M stage (x, eps, m_prev -> x_next, m) at
[256,4,64,64] with its buffers carved from one arena at offsets k * (size + skew): kernel time against the skew,
back to back and with the caches evicted before every launch.   usage (MI355X): python tools/skew_probe.py [fp32|fp16]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (tools run on the LAB build of the library: include/dpm_lab.h)
import dpm_solver_amd as D  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402

DEV = "cuda:0"


def main():
    dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "fp32") else torch.float16
    code = L.DTYPE_F32 if dt == torch.float32 else L.DTYPE_F16
    es = 4 if dt == torch.float32 else 2
    shape = (256, 4, 64, 64)
    n = int(np.prod(shape))
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(np.cumprod(1.0 - betas).astype(np.float32)))
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, state_dtype=dt)
    plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    st = plan.stages[5].copy()
    assert st.form == L.FORM_TWO
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scratch = torch.zeros(768 * 1024 * 1024 // 4, dtype=torch.float32, device=DEV)
    size = n * es
    print("%s, %d bytes per stream" % (dt, size))
    for skew in (0, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 4096 + 256, 1 << 20, (1 << 20) + 65536 + 4096 + 256, 3 << 19):
        arena = torch.empty(5 * (size + skew) + 4096, dtype=torch.uint8, device=DEV)
        base = (arena.data_ptr() + 4095) // 4096 * 4096
        ptr = [base + k * (size + skew) for k in range(5)]
        arena.zero_()
        b = L.Buffers()
        b.x, b.e0, b.h1, b.x_out, b.m_out = ptr
        b.n, b.batch, b.state_dtype, b.eps_dtype = n, shape[0], code, code
        res = {}
        for evict in (False, True):
            v = []
            for _ in range(40):
                if evict:
                    scratch.sum()
                ms = C.c_float()
                L.check(L.lib.dpm_stage_launch_timed(C.byref(st), C.byref(b), stream, C.byref(ms)))
                v.append(ms.value * 1e3)
            res[evict] = float(np.median(v[5:]))
        by = 5 * size
        pct = lambda us: by / us / 8e6 * 100          # bytes / us -> % of 8 TB/s
        print("skew %8d B: back to back %6.2f us = %4.1f %%   evicted %6.2f us = %4.1f %%" % (
            skew, res[False], pct(res[False]), res[True], pct(res[True])))
        del arena


if __name__ == "__main__":
    main()
