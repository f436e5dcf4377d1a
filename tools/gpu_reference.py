#!/usr/bin/env python3
"""The unmodified reference running ON THE SAME MI355X through PyTorch-ROCm (what a user of the reference has today) next
to the engine: solver time per 20-step DPM-Solver++(2M) trajectory with a frozen network, and a live parity check on
identical device inputs.  The reference file travels to the GPU box as git-ignored scratch ($DPM_REFERENCE_DIR, removed
after the call; tools/cpu_baseline.py does the same for the CPU timing) -- only this tool's output is committed.

    DPM_REFERENCE_DIR=_refscratch python tools/gpu_reference.py [--out gpurun_out/.../gpu_reference.json]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dpm_solver_amd as D  # noqa: E402


def load_reference():
    ref = bench.reference_dir()
    assert ref, "no reference checkout (DPM_REFERENCE_DIR)"
    spec = importlib.util.spec_from_file_location("_dpm_reference_gpu", os.path.join(ref, "dpm_solver_pytorch.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    return R


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    R = load_reference()
    dev = "cuda:0"
    ac = torch.from_numpy(bench.sd_alphas_cumprod())
    res = []
    for shape, dtype, kw, label in (
            ((256, 4, 64, 64), torch.float32, dict(), "cfg2 size, fp32, unguided"),
            ((256, 4, 64, 64), torch.float16, dict(), "cfg2 size, fp16 x_T (the reference promotes to fp32), unguided"),
            ((8, 4, 64, 64), torch.float32, dict(), "cfg1 / cfg4 per-GPU size, fp32, unguided"),
            ((8, 4, 64, 64), torch.float32, dict(cfg=7.5), "cfg4 per-GPU size, fp32, classifier-free guidance 7.5"),
            ((32, 3, 64, 64), torch.float32, dict(thr=True), "cfg5: dynamic thresholding, 25 steps")):
        g = torch.Generator().manual_seed(11)
        x = torch.randn(shape, generator=g).to(dev, dtype)
        eps = torch.randn(shape, generator=g).to(dev, dtype)
        steps = 25 if kw.get("thr") else 20
        cond = torch.ones(shape[0], device=dev)

        def build(M, sched):
            if kw.get("cfg"):
                net = lambda xx, t, c: eps.repeat(2, 1, 1, 1)[:xx.shape[0]] * (0.5 + 0.5 * c.reshape(-1, 1, 1, 1)).to(xx.dtype)
                fn = M.model_wrapper(net, sched, guidance_type="classifier-free", condition=cond, unconditional_condition=cond * 0,
                                     guidance_scale=kw["cfg"])
            else:
                fn = M.model_wrapper(lambda xx, t: eps, sched)
            return M.DPM_Solver(fn, sched, algorithm_type="dpmsolver++",
                                correcting_x0_fn="dynamic_thresholding" if kw.get("thr") else None)
        ref = build(R, R.NoiseScheduleVP("discrete", alphas_cumprod=ac.to(dev)))
        eng = build(D, D.NoiseScheduleVP("discrete", alphas_cumprod=ac))
        with torch.no_grad():
            yr = ref.sample(x, steps=steps, order=2)
            ye = eng.sample(x, steps=steps, order=2)
        torch.cuda.synchronize()
        err = float((ye.double() - yr.double()).abs().max() / yr.double().abs().max())
        t_ref = timed(lambda: ref.sample(x, steps=steps, order=2), 8)
        t_eng = timed(lambda: eng.sample(x, steps=steps, order=2), 20)
        row = dict(case=label, shape=list(shape), x_dtype=str(dtype).split(".")[-1], out_dtype_reference=str(yr.dtype).split(".")[-1],
                   out_dtype_engine=str(ye.dtype).split(".")[-1], steps=steps, reference_on_gpu_ms=round(t_ref, 3),
                   engine_ms=round(t_eng, 4), speedup=round(t_ref / t_eng, 1), rel_err_vs_reference_on_gpu=err)
        res.append(row)
        print(json.dumps(row), flush=True)
        assert err <= 1e-5, row
    out = dict(what="unmodified reference (dpm_solver_pytorch.py) on the MI355X through PyTorch-ROCm eager vs dpm_solver_amd, same "
                    "device inputs, frozen network (solver time only), wall per sample() call, median",
               gpu=torch.cuda.get_device_name(0), torch=torch.__version__, rows=res)
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
