#!/usr/bin/env python3
"""tools/fuzz_dropin.py --mode methods carried to the hardware: every random call of a PUBLIC method -- the per-update methods
(first / singlestep 2, 3 / multistep 2, 3 / the dispatchers), noise_prediction_fn / data_prediction_fn / model_fn,
denoise_to_zero_fn, add_noise, the time grids, dynamic_thresholding_fn, the schedule's functions, interpolate_fn; time tensors
0-dim / (1,)-shaped, fp32 / double; half / fp32 / double states; model values handed in or not; r1 / r2 floats, tensors or
None -- runs through the engine on the MI355X and again through the engine's host code on CPU tensors with the numpy double
of the kernels: same exception or same dtypes, shapes and values (fp32 / double bit-identical).  The build container's
tools/fuzz_dropin.py holds that double to the live reference over the same generator.

    python tools/fuzz_gpu_methods.py [--cases 4000] [--seed 0] [--out gpurun_out/.../fuzz_gpu_methods.json]
"""
import argparse
import contextlib
import faulthandler
import io
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("DPM_REFERENCE_DIR", "/nonexistent")
import fuzz_dropin as FD  # noqa: E402  (its `R` is whatever `dpm_solver_pytorch` resolves to: not used here)
import dpm_solver_amd as D  # noqa: E402
import dpm_solver_amd.solver as S  # noqa: E402
import dpm_solver_amd.utils as U  # noqa: E402
from kernel_double import install_cpu_double  # noqa: E402


class _MP:
    def setattr(self, o, n, v):
        setattr(o, n, v)


def one(call, cfg, device):
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = call(D, FD.eng_schedule(cfg["schedule"]), U, device=device)
        return ("ok", [t.detach().cpu() for t in FD._flatten(out)])
    except Exception as ex:                             # noqa: BLE001
        return ("raise", (type(ex).__name__, str(ex)[:160]), traceback.format_exc(limit=4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=4000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--case-timeout", type=int, default=60)
    args = ap.parse_args()
    FD.WIDE_NET = True
    rng = np.random.default_rng(args.seed)
    calls = [FD.random_method_call(rng) for _ in range(args.cases)]
    if args.device == "cpu":
        install_cpu_double(_MP(), S, D)
    cur = (os.path.splitext(args.out)[0] if args.out else "/tmp/fuzz_gpu_methods") + "_current_case.txt"
    t0 = time.perf_counter()
    gpu = []
    for i, (cfg, call) in enumerate(calls):
        with open(cur, "w") as f:
            f.write("%d %s\n" % (i, cfg))
        faulthandler.dump_traceback_later(args.case_timeout, exit=True, file=sys.__stderr__)
        gpu.append(one(call, cfg, args.device))
        faulthandler.cancel_dump_traceback_later()
    os.remove(cur)
    if args.device != "cpu":
        torch.cuda.synchronize()
        install_cpu_double(_MP(), S, D)
    t_gpu = time.perf_counter() - t0
    torch.set_num_threads(1)
    n_bad = n_raise = n_ok = n_same = 0
    kinds, per = {}, {}
    worst = {"float32": 0.0, "float64": 0.0, "half": 0.0}
    for i, ((cfg, call), g) in enumerate(zip(calls, gpu)):
        c = one(call, cfg, "cpu")
        bad = []
        a = per.setdefault(cfg["what"], dict(calls=0, returned=0, bit_identical=0, disagreements=0))
        a["calls"] += 1
        if g[0] != c[0]:
            bad.append("GPU %s, double %s: %s | %s" % (g[0], c[0], g[1] if g[0] == "raise" else "", c[1] if c[0] == "raise" else ""))
        elif g[0] == "raise":
            n_raise += 1
            if g[1] != c[1]:
                bad.append("exception %s vs %s" % (g[1], c[1]))
        else:
            n_ok += 1
            a["returned"] += 1
            ga, ca = g[1], c[1]
            same = len(ga) == len(ca)
            if not same:
                bad.append("%d vs %d tensors returned" % (len(ga), len(ca)))
            for k, (x, y) in enumerate(zip(ga, ca)):
                if x.dtype != y.dtype or tuple(x.shape) != tuple(y.shape):
                    bad.append("tensor %d: %s %s vs %s %s" % (k, x.dtype, tuple(x.shape), y.dtype, tuple(y.shape)))
                    same = False
                    break
                eq = bool(((x == y) | (x.isnan() & y.isnan())).all()) if x.is_floating_point() else bool(torch.equal(x, y))
                same = same and eq
                if not eq:
                    pk = float(y.double().abs().max()) or 1.0
                    err = float((x.double() - y.double()).nan_to_num().abs().max()) / pk
                    key = "half" if x.dtype in (torch.float16, torch.bfloat16) else str(x.dtype)[6:]
                    worst[key] = max(worst.get(key, 0.0), err)
                    tol = 4e-3 if x.dtype is torch.float16 else (3e-2 if x.dtype is torch.bfloat16 else (1e-12 if x.dtype is torch.float64 else 2e-6))
                    if err > tol:
                        bad.append("tensor %d values: %.3g (tolerance %.1g)" % (k, err, tol))
                        break
            n_same += same
            a["bit_identical"] += same
        if bad:
            n_bad += 1
            a["disagreements"] += 1
            kinds[cfg["what"]] = kinds.get(cfg["what"], 0) + 1
            print("call %d: %s\n    %s" % (i, cfg, "\n    ".join(bad)), flush=True)
            if g[0] == "raise" and c[0] != "raise":
                print("    " + g[2].replace("\n", "\n    "))
    rec = dict(calls=args.cases, seed=args.seed, returned=n_ok, raised_alike=n_raise, bit_identical=n_same, disagreements=n_bad, kinds=kinds,
               worst_fraction_of_peak=worst, per_method=per, gpu_seconds=round(t_gpu, 1),
               device=(torch.cuda.get_device_name(0) if args.device != "cpu" else "cpu (self-check)"),
               what="the public methods of DPM_Solver / NoiseScheduleVP / interpolate_fn on the MI355X vs the engine's host code on the "
                    "numpy double, tools/fuzz_dropin.py's method-call generator")
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
