#!/usr/bin/env python3
"""Lab experiment ("what's weak": [64,3,256,256] thresholding at 0.55-0.59, 38 % of the kernel without memory
traffic): start the clusters of a thresholding launch out of phase (DPM_TUNE_THR_STAGGER), so that one group of clusters
streams while another selects.  Per shape and (groups, offset): median microseconds per stage of a 12-step 2M trajectory
(events around the whole trajectory, the network is a constant tensor), alternating with the undisturbed kernel; results
must be bit-identical.

    DPM_SOLVER_AMD_LIB=tools/_variants/lab/libdpm_lab.so python tools/thr_stagger.py [--out profiles/r05_thr_stagger.jsonl]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpm_solver_amd as D                      # noqa: E402
from dpm_solver_amd import _lib as L            # noqa: E402


def per_stage_us(dpm, x, steps, reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
    for r in range(reps):
        ev[2 * r].record()
        dpm.sample(x, steps=steps, order=2)
        ev[2 * r + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[2 * r].elapsed_time(ev[2 * r + 1]) for r in range(reps)])) * 1e3 / steps


def main():
    L.require_lab("tools/thr_stagger.py")
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--shapes", default="64x3x256x256,16x3x256x256,256x3x128x128,32x3x64x64,1024x3x64x64")
    ap.add_argument("--ticks", default="0,10,20,30,40,60,80,100,140")
    ap.add_argument("--groups", default="2,3,4")
    a = ap.parse_args()
    ns = D.NoiseScheduleVP('discrete', betas=torch.from_numpy(np.linspace(1e-4, 0.02, 1000).astype(np.float32)))
    rows = []
    for sh in a.shapes.split(","):
        shape = tuple(int(v) for v in sh.split("x"))
        g = torch.Generator(device="cuda").manual_seed(3)
        e = torch.randn(shape, device="cuda", generator=g)
        x = torch.randn(shape, device="cuda", generator=g)
        dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: e, ns), ns, correcting_x0_fn="dynamic_thresholding")
        ref = dpm.sample(x, steps=a.steps, order=2).clone()
        per_stage_us(dpm, x, a.steps, 3)
        for ng in (int(v) for v in a.groups.split(",")):
            for tk in (int(v) for v in a.ticks.split(",")):
                if tk == 0 and ng != 2:
                    continue
                L.check(L.lib.dpm_tuning_set(L.TUNE_THR_STAGGER, (ng << 16) | tk if tk else 0))
                try:
                    got = dpm.sample(x, steps=a.steps, order=2)
                    same = bool(torch.equal(got, ref))
                    us = per_stage_us(dpm, x, a.steps, a.reps)
                finally:
                    L.lib.dpm_tuning_set(L.TUNE_THR_STAGGER, 0)
                base = per_stage_us(dpm, x, a.steps, a.reps)
                row = dict(shape=list(shape), groups=ng, offset_us=tk / 10.0, us_per_stage=round(us, 2),
                           undisturbed_us_per_stage=round(base, 2), ratio=round(us / base, 4), bit_identical=same)
                rows.append(row)
                print(json.dumps(row), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
