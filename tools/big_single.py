#!/usr/bin/env python3
"""Launch shape of ONE large stage launch (round 6): plain DPM_Solver.sample() on a [B,4,64,64] tensor with B >> 256.

The single-request streaming kernel caps its grid at 8 workgroups per CU and walks the tiles in a grid-stride loop; the fused
multi-request kernel launches one workgroup per (super-)tile, uncapped, with an XCD-contiguous tile mapping for 2-byte states
(profiles/r02_tune_multi.txt: 7.95 us capped at 8 per CU, 6.93 uncapped, per 42 MB).  tools/config_bench.py's `one8192` case
showed the gap on the drop-in path: 226.8 us per stage through sample() against 208 us for the same bytes through
sample_requests.  This tool measures, per batch size and dtype pair (frozen network, HIP events around whole trajectories,
lab build for the knobs):

  default          DPM_Solver.sample(x) with the single-request shape: grid capped at 8 workgroups per CU, grid-stride loop
  capN             the same with the cap at N workgroups per CU              (DPM_TUNE_BLOCKS_PER_CU)
  multi            the launch handed to the fused kernel as a group of one   (DPM_TUNE_BIG_TILES = 1: Tuning::big_tiles)
each `frozen` (stages back to back: inputs of the smaller sizes sit in the Infinity Cache) and `evicted` (768 MiB streamed through
the chip before every launch, kernel-only by dpm_stage_launch_timed: inputs from HBM, what a network between the stages does).

    python tools/big_single.py [--sizes 256,512,1024,2048,8192] [--out FILE]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401
import torch  # noqa: E402
import bench  # noqa: E402
import dpm_solver_amd as D  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402

PEAK = 8000.0


def events(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


_SCRATCH = None


def evict(dev):
    """stream 768 MiB through the chip (read-only): nothing the previous stage wrote is left in L2 / the Infinity Cache"""
    global _SCRATCH
    if _SCRATCH is None:
        _SCRATCH = torch.zeros(768 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    _SCRATCH.sum()


def evicted_us(dpm, x, dev):
    """kernel-only median of the steady-state stages with the caches evicted before every launch (dpm_stage_launch_timed)"""
    import ctypes as C
    import numpy as np
    import dpm_solver_amd.solver as S
    real = S._stage_launch_raw
    rows = []

    def shim(st, b, stream):
        evict(dev)
        ms = C.c_float()
        rc = L.lib.dpm_stage_launch_timed(st, b, stream, C.byref(ms))
        rows.append(ms.value * 1e3)
        return rc
    S._stage_launch_raw = shim
    try:
        for _ in range(3):
            dpm.sample(x, steps=20, order=2)
    finally:
        S._stage_launch_raw = real
    us = np.array(rows).reshape(3, 20)[1:, 1:19]
    return float(np.median(us))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256,384,512,768,1024,2048,4096,8192")
    ap.add_argument("--pairs", default="fp16:fp16,fp32:fp32,fp32:fp16")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    L.require_lab("tools/big_single.py")
    dev = torch.device("cuda", 0)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    variants = [("default", {}), ("cap16", {L.TUNE_BLOCKS_PER_CU: 16}), ("cap64", {L.TUNE_BLOCKS_PER_CU: 64}),
                ("multi", {L.TUNE_BIG_TILES: 1})]
    base = {L.TUNE_BLOCKS_PER_CU: 8, L.TUNE_BIG_TILES: 0}
    rows = []
    for pair in args.pairs.split(","):
        sname, ename = pair.split(":")
        sd, ed = bench._DT[sname], bench._DT[ename]
        for B in (int(v) for v in args.sizes.split(",")):
            g = torch.Generator(device=dev).manual_seed(B)
            x = torch.randn((B, 4, 64, 64), generator=g, device=dev, dtype=torch.float32).to(sd)
            eps = torch.randn((B, 4, 64, 64), generator=g, device=dev, dtype=torch.float32).to(ed)
            kwargs = dict(algorithm_type="dpmsolver++")
            if sd is not torch.float32:
                kwargs["state_dtype"] = sd
            dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: eps, ns), ns, **kwargs)
            n = x.numel()
            ssz, esz = x.element_size(), eps.element_size()
            traj_bytes = n * (18 * (4 * ssz + esz) + 2 * (3 * ssz + esz))
            stage_bytes = n * (4 * ssz + esz)
            reps = max(3, min(40, int(2e9 / traj_bytes)))
            run = lambda: dpm.sample(x, steps=20, order=2)
            res = dict(batch=B, state=sname, eps=ename, tiles=n // 2048, MB_per_stage=round(stage_bytes / 1e6, 1))
            with torch.no_grad():
                for k, v in base.items():
                    L.check(L.lib.dpm_tuning_set(k, v))
                want = run()
                for name, knobs in variants:
                    for k, v in {**base, **knobs}.items():
                        L.check(L.lib.dpm_tuning_set(k, v))
                    got = run()
                    assert torch.equal(got, want), name
                    res[name + "_frozen_us"] = round(min(events(run, reps) for _ in range(3)) / 20, 2)
                    res[name + "_evicted_us"] = round(evicted_us(dpm, x, dev), 2)
                for k, v in base.items():
                    L.check(L.lib.dpm_tuning_set(k, v))
            for mode in ("frozen", "evicted"):
                res["best_" + mode] = min((res[nm + "_" + mode + "_us"], nm) for nm, _ in variants)[1]
            res["default_evicted_frac"] = round(stage_bytes / res["default_evicted_us"] / 1e3 / PEAK, 4)
            res["multi_evicted_frac"] = round(stage_bytes / res["multi_evicted_us"] / 1e3 / PEAK, 4)
            rows.append(res)
            print(json.dumps(res), flush=True)
            del x, eps, dpm
            torch.cuda.empty_cache()
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
