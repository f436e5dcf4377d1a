#!/usr/bin/env python3
"""Launch-shape sweeps of the stage kernels on the GPU (run via gpurun).  One tool, two sweeps; both need a library
built with every (tiles per iteration, nt mask) variant:

    python -c "import __graft_entry__ as g; g.build_variant('tune', ['-DDPM_TUNING_VARIANTS'], lab=True)"   # a LAB variant: the knobs
    export DPM_SOLVER_AMD_LIB=tools/_variants/tune/libdpm_lab.so
    python tools/tune.py multi  [--requests 32]      > profiles/rNN_tune_multi.txt     # fused multi-request launches
    python tools/tune.py single [--dtypes fp16,fp32] > profiles/rNN_tune_single.txt    # one request per launch

`multi`: for every (state dtype, eps dtype) x (tiles per iteration, nt mask) x grid cap, 20-stage trajectories of R
requests through dpm_plan_run_multi: mean kernel-only time of a steady-state 2M launch per request-stage and the fraction
of the 8 TB/s HBM peak on its algorithmic bytes; `unfused` rows = one launch per request, requests interleaved.
`single`: the same variants for single launches, inputs cache-resident (trajectories back to back) and from HBM
(32 requests interleaved, one launch each).  (Round 1's three sweep scripts -- nt-mask split, address skew between the
buffers of a set, re-check after kernel changes; profiles/r01_tuning*.txt -- are folded into this one.)
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (tools run on the LAB build of the library: include/dpm_lab.h)
import bench  # noqa: E402
import dpm_solver_amd as D  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402

_DT = {"fp16": torch.float16, "fp32": torch.float32, "bf16": torch.bfloat16}


def sweep_multi(args):
    dev = torch.device("cuda", 0)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = C.c_void_p(stream.cuda_stream)
    n_el = bench.B * int(np.prod(bench.SHAPE))
    pairs = [(torch.float16, torch.float16), (torch.float32, torch.float32), (torch.float32, torch.float16),
             (torch.bfloat16, torch.bfloat16)]
    if args.quick:
        pairs = pairs[:2]
    for sd, ed in pairs:
        dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, algorithm_type="dpmsolver++", state_dtype=sd)
        plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                             lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
        nst = len(plan.stages)
        sets = bench.make_sets(args.requests, sd, dev, seed=7, eps_dtype=ed)
        rbs = (L.RunBuffers * len(sets))(*[s["rb"] for s in sets])
        res = (C.c_int * len(sets))()
        ms = (C.c_float * (len(sets) * nst))()
        ssz = torch.empty((), dtype=sd).element_size()
        esz = torch.empty((), dtype=ed).element_size()
        alg = n_el * (4 * ssz + esz)

        def run(label):
            vals = []
            for _ in range(args.reps):
                L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, len(sets), sptr, ms, res))
                a = np.frombuffer(ms, dtype=np.float32).reshape(len(sets), nst)[:, 1:nst - 1]
                vals.append(a.astype(np.float64).mean() * 1e3)
            us = float(np.median(vals))
            print("%-9s %-9s %-34s %8.3f us/request-stage  %7.1f GB/s  %.3f of peak" % (
                str(sd).split(".")[1], str(ed).split(".")[1], label, us, alg / us / 1e3, alg / us / 1e3 / 8000.0), flush=True)

        L.lib.dpm_tuning_set(L.TUNE_MULTI_FUSE, 0)
        run("unfused (1 launch per request)")
        L.lib.dpm_tuning_set(L.TUNE_MULTI_FUSE, 1)
        run("fused default")
        for u in (1, 2):
            for nt in (0, 1, 5):
                for bpc in (8, 16, 32, 4096):
                    L.lib.dpm_tuning_set(L.TUNE_UNROLL, u)
                    L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, nt)
                    L.lib.dpm_tuning_set(L.TUNE_MULTI_BLOCKS_PER_CU, bpc)
                    run("fused U=%d nt=%d blocks/CU=%d" % (u, nt, bpc))
        L.lib.dpm_tuning_set(L.TUNE_UNROLL, 0)
        L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, -1)
        L.lib.dpm_tuning_set(L.TUNE_MULTI_BLOCKS_PER_CU, 0)
        if sd == torch.float16:
            for r in (2, 4, 8, 16):
                sub = (L.RunBuffers * r)(*[s["rb"] for s in sets[:r]])
                vals = []
                for _ in range(args.reps + 2):
                    L.check(L.lib.dpm_plan_run_multi(plan.handle, sub, r, sptr, ms, res))
                    a = np.frombuffer(ms, dtype=np.float32)[: r * nst].reshape(r, nst)[:, 1:nst - 1]
                    vals.append(a.astype(np.float64).mean() * 1e3)
                us = float(np.median(vals))
                print("float16   float16   fused default, R=%-2d (partly cache-resident) %8.3f us/request-stage  %.3f of peak" % (
                    r, us, alg / us / 1e3 / 8000.0), flush=True)
        del sets, rbs
        torch.cuda.empty_cache()




def sweep_single(args):
    dev = torch.device("cuda", 0)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = C.c_void_p(stream.cuda_stream)
    n_el = bench.B * int(np.prod(bench.SHAPE))
    L.lib.dpm_tuning_set(L.TUNE_MULTI_FUSE, 0)
    for name in args.dtypes.split(","):
        sname, _, ename = name.partition("/")
        sd, ed = _DT[sname], _DT[ename or sname]
        dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, state_dtype=sd)
        plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                             lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
        nst = len(plan.stages)
        sets = bench.make_sets(args.requests, sd, dev, seed=7, eps_dtype=ed)
        rbs = (L.RunBuffers * len(sets))(*[s["rb"] for s in sets])
        res, resm = C.c_int(-1), (C.c_int * len(sets))()
        buf, msb = (C.c_float * nst)(), (C.c_float * (len(sets) * nst))()
        alg = n_el * (4 * torch.empty((), dtype=sd).element_size() + torch.empty((), dtype=ed).element_size())
        for u in (1, 2, 4, 8):     # tiles per workgroup, all loads issued up front (4, 8: fewer, fatter wavefronts)
            for nt in (0, 1, 5):
                L.lib.dpm_tuning_set(L.TUNE_UNROLL, u)
                L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, nt)
                warm = []
                for i in range(16):
                    L.check(L.lib.dpm_plan_run_multi(plan.handle, C.byref(sets[i % 8]["rb"]), 1, sptr, buf, C.byref(res)))
                    warm.append(np.frombuffer(buf, dtype=np.float32)[1:nst - 1].astype(np.float64).mean() * 1e3)
                cold = []
                for i in range(args.reps):
                    L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, len(sets), sptr, msb, resm))
                    cold.append(np.frombuffer(msb, dtype=np.float32).reshape(len(sets), nst)[:, 1:nst - 1].astype(np.float64).mean() * 1e3)
                w, c = float(np.mean(warm[4:])), float(np.median(cold))
                print("%-9s U=%d nt=%d   cache-resident %7.3f us (%.3f of peak)   from HBM %7.3f us (%.3f of peak)" % (
                    name, u, nt, w, alg / w / 1e3 / 8000.0, c, alg / c / 1e3 / 8000.0), flush=True)
        L.lib.dpm_tuning_set(L.TUNE_UNROLL, 0)
        L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, -1)
        del sets, rbs
        torch.cuda.empty_cache()
    L.lib.dpm_tuning_set(L.TUNE_MULTI_FUSE, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sweep", choices=["multi", "single"])
    ap.add_argument("--requests", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--dtypes", default="fp16,fp32,fp32/fp16")
    args = ap.parse_args()
    (sweep_multi if args.sweep == "multi" else sweep_single)(args)


if __name__ == "__main__":
    main()
