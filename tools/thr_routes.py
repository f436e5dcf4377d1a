#!/usr/bin/env python3
"""Which route the clustered thresholding kernel takes per stage (dpm_buffers.thr_hint words 2, 3) and the kernel time per
stage with the predicted bound on / off, for the stage-table scenarios (frozen eps).   python tools/thr_routes.py"""
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
import dpm_solver_amd.solver as S  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402

DEV = "cuda:0"


def run(shape, steps, predict, model):
    ns = D.NoiseScheduleVP("discrete", betas=torch.from_numpy(np.linspace(1e-4, 0.02, 1000).astype(np.float32)))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape, generator=g).to(DEV)
    e = torch.randn(shape, generator=g).to(DEV)
    fn = (lambda xx, t: e) if model == "frozen" else (lambda xx, t: xx * 0.5)
    dpm = D.DPM_Solver(D.model_wrapper(fn, ns), ns, correcting_x0_fn="dynamic_thresholding")
    L.lib.dpm_tuning_set(L.TUNE_THR_PREDICT, int(predict))
    dpm.sample(x, steps=steps, order=2)
    rows = []
    real = S._stage_launch_raw

    def spy(st, b, stream):
        ms = C.c_float()
        rc = L.lib.dpm_stage_launch_timed(st, b, stream, C.byref(ms))
        fr = [v for v in dpm._fast.values() if getattr(v, "thr_hint", None) is not None][-1]
        h = fr.thr_hint.view(-1, L.THR_HINT_WORDS).cpu().numpy()
        rows.append((ms.value * 1e3, h[:, 2].copy(), h[:, 3].copy(), h[:, 0].copy()))
        return rc
    S._stage_launch_raw = spy
    try:
        dpm.sample(x, steps=steps, order=2)
    finally:
        S._stage_launch_raw = real
        L.lib.dpm_tuning_set(L.TUNE_THR_PREDICT, 1)
    return rows


for shape, steps in (((32, 3, 64, 64), 25), ((64, 3, 256, 256), 10)):
    for model in ("frozen", "half"):
        for predict in (1, 0):
            rows = run(shape, steps, predict, model)
            print("shape %s model %s predict %d   median of the stages >= 2: %.2f us" % (
                shape, model, predict, float(np.median([r[0] for r in rows[2:]]))))
            for i, (us, route, tot, a) in enumerate(rows):
                cnt = {int(k): int((route == k).sum()) for k in np.unique(route)}
                print("  stage %2d  %7.2f us  routes %s  union median %5.0f max %5.0f  a[0] %.4f" % (i, us, cnt, np.median(tot), tot.max(), a[0]))
