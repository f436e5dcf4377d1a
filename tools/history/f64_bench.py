#!/usr/bin/env python3
"""Double-precision states (csrc/dpm_f64.hip: one run-time dispatched kernel, a compatibility path): microseconds per stage and
the fraction of the 8 TB/s peak of a 20-step 2M++ trajectory whose network is a constant tensor (events around whole
trajectories; 5 N 8 algorithmic bytes per steady-state stage).  Product library.

    python tools/f64_bench.py [--out profiles/r05_f64.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpm_solver_amd as D                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    betas = torch.from_numpy(np.linspace(1e-4, 0.02, 1000))
    rows = []
    for dtype in (torch.float64, torch.float32):
        ns = D.NoiseScheduleVP('discrete', betas=betas.to(dtype), dtype=dtype)
        for shape in ((256, 4, 64, 64), (64, 3, 256, 256)):
            for thr in (False, True):
                if thr and shape[1] != 3:
                    continue
                e = torch.randn(shape, device="cuda", dtype=dtype)
                x = torch.randn(shape, device="cuda", dtype=dtype)
                dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: e, ns), ns, algorithm_type="dpmsolver++",
                                   correcting_x0_fn="dynamic_thresholding" if thr else None)
                steps = 20
                for _ in range(2):
                    dpm.sample(x, steps=steps, order=2)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * a.reps)]
                for r in range(a.reps):
                    ev[2 * r].record()
                    dpm.sample(x, steps=steps, order=2)
                    ev[2 * r + 1].record()
                torch.cuda.synchronize()
                ms = float(np.median([ev[2 * r].elapsed_time(ev[2 * r + 1]) for r in range(a.reps)]))
                n = int(np.prod(shape))
                s = 8 if dtype == torch.float64 else 4
                byt = (18 * 5 + 2 * 4) * n * s
                row = dict(dtype=str(dtype).replace("torch.", ""), shape=list(shape), thresholding=thr, us_per_stage=round(ms * 1e3 / steps, 2),
                           algorithmic_MB_per_trajectory=round(byt / 1e6, 1), frac_of_8TBs=round(byt / (ms * 1e-3) / 8e12, 4))
                rows.append(row)
                print(json.dumps(row), flush=True)
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
