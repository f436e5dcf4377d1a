#!/usr/bin/env python3
"""Which kernels does the FIRST sample() of a process launch that are not the library's own?  (cold start: every first use of a
torch kernel loads one of torch's code objects.)  Run under rocprofv3 --kernel-trace --stats; prints nothing itself.

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o kt -- python tools/first_call_kernels.py [plain|thresholding|cfg]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpm_solver_amd as D      # noqa: E402

scenario = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda", 0)
betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(np.cumprod(1.0 - betas).astype(np.float32)))
e = torch.empty((8, 4, 64, 64) if scenario != "thresholding" else (32, 3, 64, 64), device=dev)   # no kernel: uninitialised values are fine
x = torch.empty_like(e)
kw = dict(steps=20, order=2)
if scenario == "cfg":
    e2 = torch.empty((16, 4, 64, 64), device=dev, dtype=torch.float16)
    cond = torch.empty(8, device=dev)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t, c: e2, ns, guidance_type="classifier-free", condition=cond,
                                       unconditional_condition=cond, guidance_scale=7.5), ns)
else:
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: e, ns), ns,
                       correcting_x0_fn="dynamic_thresholding" if scenario == "thresholding" else None)
dpm.sample(x, **kw)            # the network is a constant tensor: every kernel in the trace is the solver's or its host path's
torch.cuda.synchronize()
