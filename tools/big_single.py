#!/usr/bin/env python3
"""Launch shape of ONE large stage launch (round 6): plain DPM_Solver.sample() on a [B,4,64,64] tensor with B >> 256.

The single-request streaming kernel caps its grid at 8 workgroups per CU and walks the tiles in a grid-stride loop; the fused
multi-request kernel launches one workgroup per (super-)tile, uncapped, with an XCD-contiguous tile mapping for 2-byte states
(profiles/r02_tune_multi.txt: 7.95 us capped at 8 per CU, 6.93 uncapped, per 42 MB).  tools/config_bench.py's `one8192` case
showed the gap on the drop-in path: 226.8 us per stage through sample() against 208 us for the same bytes through
sample_requests.  This tool measures, per batch size and dtype pair (frozen network, HIP events around whole trajectories,
lab build for the knobs):

  default          DPM_Solver.sample(x)                                    (the library's own choice of shape)
  capN             the same with the grid cap at N workgroups per CU       (DPM_TUNE_BLOCKS_PER_CU)
  requests         sample_requests over 32 (or fewer) equal batch slices of the same tensor: the fused kernel

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256,512,1024,2048,8192")
    ap.add_argument("--pairs", default="fp16:fp16,fp32:fp32,fp32:fp16")
    ap.add_argument("--caps", default="16,32,4096")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    L.require_lab("tools/big_single.py")
    dev = torch.device("cuda", 0)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
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
            reps = max(3, min(40, int(2e9 / traj_bytes)))
            run = lambda: dpm.sample(x, steps=20, order=2)
            res = dict(batch=B, state=sname, eps=ename, MB_per_stage=round(traj_bytes / 20 / 1e6, 1))
            with torch.no_grad():
                want = run()
                run()
                res["default_us"] = round(min(events(run, reps) for _ in range(3)) / 20, 2)
                for cap in (int(v) for v in args.caps.split(",")):
                    L.check(L.lib.dpm_tuning_set(L.TUNE_BLOCKS_PER_CU, cap))
                    got = run()
                    assert torch.equal(got, want)
                    res["cap%d_us" % cap] = round(min(events(run, reps) for _ in range(3)) / 20, 2)
                L.check(L.lib.dpm_tuning_set(L.TUNE_BLOCKS_PER_CU, 8))
                R = 32
                while R > 1 and (B % R or (n // R) % 4096):
                    R //= 2
                if R > 1:
                    xs = list(x.chunk(R))
                    es = list(eps.chunk(R))
                    calls = [0]

                    def frozen(xx, t):
                        calls[0] += 1
                        return es[(calls[0] - 1) % R]
                    dpr = D.DPM_Solver(D.model_wrapper(frozen, ns), ns, **kwargs)

                    def runr():
                        calls[0] = 0
                        return dpr.sample_requests(xs, steps=20, order=2)
                    outs = runr()
                    assert torch.equal(torch.cat(outs), want)
                    runr()
                    res["requests_us"] = round(min(events(runr, reps) for _ in range(3)) / 20, 2)
                    res["requests_R"] = R
            best = min(v for k, v in res.items() if k.endswith("_us"))
            res["default_frac"] = round(traj_bytes / 20 / res["default_us"] / 1e3 / PEAK, 4)
            res["best_frac"] = round(traj_bytes / 20 / best / 1e3 / PEAK, 4)
            rows.append(res)
            print(json.dumps(res), flush=True)
            del x, eps, dpm
            torch.cuda.empty_cache()
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
