#!/usr/bin/env python3
"""Fuzz of the dynamic-thresholding kernels against `torch.quantile` itself (ref :416-425, the reference's own three lines
evaluated on the CPU): `DPM_Solver.dynamic_thresholding_fn(x0)` on the GPU must return the reference's bits for ANY batch,
sample size, ratio and max_val -- sample sizes drawn around the kernels' boundaries (a workgroup's LDS chunk of 12288 elements
+- 1, cluster splits, odd and prime sizes), ratios drawn from the continuum (not only 0.995), value distributions with heavy
ties (values quantised to a bf16 / fp16 / integer grid, constants, mostly zeros), outliers, fp32 and fp64.

    python tools/fuzz_gpu_thresh.py [--cases 3000] [--seed 0] [--out gpurun_out/.../fuzz_gpu_thresh.json]
"""
import argparse
import faulthandler
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import dpm_solver_amd as D  # noqa: E402
from engine_cases import make_schedule  # noqa: E402

DEV = "cuda:0"
SIZES = [1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 3071, 3072, 4095, 4096, 4097,
         6143, 6144, 6145, 12287, 12288, 12289, 12290, 16383, 16384, 16385, 24575, 24576, 24577, 36864, 38450, 49151, 49152, 49153,
         65535, 65536, 65537, 98304, 196608, 196609]


def random_case(rng):
    per = int(SIZES[int(rng.integers(0, len(SIZES)))]) if rng.random() < 0.75 else int(rng.integers(1, 200000))
    B = int(rng.choice([1, 1, 2, 3, 4, 5, 6, 8, 13, 17, 32, 40, 130, 300]))
    while B * per > 2_500_000:
        B = max(1, B // 2)
    p = float(rng.choice([0.5, 0.9, 0.95, 0.99, 0.995, 0.999, 1.0, 0.0])) if rng.random() < 0.35 else float(rng.uniform(0.0, 1.0) ** 0.25)
    return dict(B=B, per=per, p=p, max_val=float(rng.choice([0.5, 1.0, 2.0, 0.0])), dist=str(rng.choice(
        ["randn", "randn", "randn", "bf16grid", "f16grid", "intgrid", "const", "zeros", "outliers", "uniform", "nonfinite"])),
        scale=float(rng.choice([0.3, 1.0, 2.0, 10.0])), f64=bool(rng.integers(0, 8) == 0), seed=int(rng.integers(0, 1 << 30)))


def make(cfg):
    g = np.random.default_rng(cfg["seed"])
    shape = (cfg["B"], cfg["per"])
    a = g.standard_normal(shape) * cfg["scale"]
    d = cfg["dist"]
    if d == "bf16grid":
        a = torch.from_numpy(a).to(torch.bfloat16).double().numpy()
    elif d == "f16grid":
        a = torch.from_numpy(a).to(torch.float16).double().numpy()
    elif d == "intgrid":
        a = np.round(a * 2) / 2
    elif d == "const":
        a = np.full(shape, cfg["scale"] * 1.7) * np.sign(g.standard_normal(shape))
    elif d == "zeros":
        a = a * (g.random(shape) < 0.02)
    elif d == "outliers":
        a = a * np.where(g.random(shape) < 0.001, 1000.0, 1.0)
    elif d == "uniform":
        a = g.uniform(-cfg["scale"], cfg["scale"], size=shape)
    elif d == "nonfinite":                               # per row: nothing / one NaN / one inf / infs around the wanted rank / NaN + infs
        n = shape[1]
        K = max(1, int(n - np.floor(np.float32(cfg["p"]) * np.float32(n - 1))))
        for b in range(shape[0]):
            kind, idx = int(g.integers(0, 6)), g.permutation(n)
            if kind in (1, 5):
                a[b, idx[0]] = np.nan
            if kind == 2:
                a[b, idx[0]] = np.inf
            if kind == 3:
                a[b, idx[:K + 1]] = np.inf
            if kind == 4:
                a[b, idx[:K]] = -np.inf
            if kind == 5:
                a[b, idx[1:K + 1]] = np.inf
    return torch.from_numpy(a).to(torch.float64 if cfg["f64"] else torch.float32)


def reference(x0, p, max_val):
    """ref :416-425, verbatim in effect: torch.quantile over the flattened sample, the max with max_val, clamp, divide"""
    dims = x0.dim()
    pq = torch.quantile(torch.abs(x0).reshape((x0.shape[0], -1)), p, dim=1)
    s = torch.maximum(pq, max_val * torch.ones_like(pq))[(...,) + (None,) * (dims - 1)]
    return torch.clamp(x0, -s, s) / s, pq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--case-timeout", type=int, default=60)
    ap.add_argument("--only", type=int, default=None)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    cfgs = [random_case(rng) for _ in range(args.cases)]
    idx = range(args.cases) if args.only is None else [args.only]
    ns = make_schedule("ddpm")
    cur = (os.path.splitext(args.out)[0] if args.out else "/tmp/fuzz_gpu_thresh") + "_current_case.txt"
    n_bad = 0
    per_dist = {}
    t0 = time.perf_counter()
    for i in idx:
        cfg = cfgs[i]
        with open(cur, "w") as f:
            f.write("%d %s\n" % (i, cfg))
        faulthandler.dump_traceback_later(args.case_timeout, exit=True, file=sys.__stderr__)
        x0 = make(cfg)
        dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=cfg["p"],
                           thresholding_max_val=cfg["max_val"])
        got = dpm.dynamic_thresholding_fn(x0.to(DEV), None).cpu()
        faulthandler.cancel_dump_traceback_later()
        want, pq = reference(x0, cfg["p"], cfg["max_val"])
        # rows holding a NaN: the reference's whole row is NaN (torch.quantile), and so is the engine's (canon_nan + the level-0
        # histograms' last bin, dpm_thresh_common.hpp); counted for the record
        nan_rows = x0.isnan().any(dim=1)
        if bool(nan_rows.any()):
            a = per_dist.setdefault("rows holding a NaN", dict(rows=0, whole_row_nan_like_the_reference=0))
            a["rows"] += int(nan_rows.sum())
            a["whole_row_nan_like_the_reference"] += int((got.isnan().all(dim=1) & nan_rows).sum())
        ok = got.dtype == want.dtype and bool(((got == want) | (got.isnan() & want.isnan())).all())
        a = per_dist.setdefault(cfg["dist"] + (" f64" if cfg["f64"] else ""), dict(cases=0, disagreements=0))
        a["cases"] += 1
        if not ok:
            a["disagreements"] += 1
            n_bad += 1
            rows = [b for b in range(cfg["B"]) if not bool(((got[b] == want[b]) | (got[b].isnan() & want[b].isnan())).all())]
            b = rows[0]
            # the scale the kernel divided by, read off an unclamped element
            j = int(torch.argmin((want[b].abs() - 0.5).abs()))
            s_gpu = float(x0[b, j] / got[b, j]) if float(got[b, j]) != 0 else float("nan")
            srt = torch.sort(x0[b].abs())[0]
            rk = int(torch.searchsorted(srt, torch.tensor(abs(s_gpu), dtype=srt.dtype)))
            print("case %d: %s\n    %d of %d samples differ (first: %d); torch.quantile %.9g, the kernel divided by ~%.9g (rank %d of %d, the "
                  "reference's rank %.3f); max |d| %.3g" % (i, {k: v for k, v in cfg.items() if k != "seed"}, len(rows), cfg["B"], b, float(pq[b]),
                                                           s_gpu, rk, cfg["per"], float(np.float32(cfg["p"])) * (cfg["per"] - 1),
                                                           float((got - want).nan_to_num().abs().max())), flush=True)
    os.remove(cur)
    rec = dict(cases=len(list(idx)), seed=args.seed, disagreements=n_bad, per_distribution=per_dist, seconds=round(time.perf_counter() - t0, 1),
               device=torch.cuda.get_device_name(0),
               what="DPM_Solver.dynamic_thresholding_fn on the GPU vs the reference's three lines (torch.quantile, maximum, clamp / divide) on the CPU, bit for bit")
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
