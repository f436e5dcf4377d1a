#!/usr/bin/env python3
"""Per-stage time of thresholded trajectories that take the run-time dispatched CATCH-ALL thresholding kernel (a v-prediction
network, classifier guidance) next to the specialised kernel (noise prediction), frozen network, HIP events around whole
trajectories.  Run once per library build (DPM_SOLVER_AMD_LIB) to compare builds of the catch-all kernel.

    python tools/thr_catchall_ab.py [--label nowaves]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dpm_solver_amd as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--label", default="")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = "cuda:0"
    betas = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64)
    rows = []
    for shape in ((32, 3, 64, 64), (1024, 3, 64, 64), (64, 3, 256, 256)):
        g = torch.Generator().manual_seed(3)
        x = torch.randn(shape, generator=g).to(dev)
        eps = (torch.randn(shape, generator=g) * 0.5).to(dev)
        steps = 12
        for kind in ("noise", "v", "classifier"):
            ns = D.NoiseScheduleVP("discrete", betas=betas)
            if kind == "classifier":
                gfix = torch.randn(shape, generator=g).to(dev) * 0.01
                fn = D.model_wrapper(lambda xx, t: eps, ns, guidance_type="classifier", condition=torch.zeros(shape[0], device=dev),
                                     guidance_scale=1.0, classifier_fn=lambda xx, t, c: (xx * gfix).sum(dim=(1, 2, 3)))
            else:
                fn = D.model_wrapper(lambda xx, t: eps, ns, model_type=kind)
            dpm = D.DPM_Solver(fn, ns, correcting_x0_fn="dynamic_thresholding")
            for _ in range(3):
                dpm.sample(x, steps=steps, order=2)
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                dpm.sample(x, steps=steps, order=2)
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / steps)
            row = dict(label=args.label, shape=list(shape), network=kind, us_per_stage_median=round(float(np.median(ts)), 2),
                       us_per_stage_min=round(float(np.min(ts)), 2))
            rows.append(row)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
