#!/usr/bin/env python3
"""Second sweep: (a) non-temporal split (loads / x_out store / m_out store), (b) address skew between the 8
state-sized arrays of a buffer set (the caching allocator hands out 2 MiB-aligned blocks, so all five streams of
a launch would otherwise hit the same HBM channel/bank bits at the same time)."""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench as BN
import dpm_solver_amd as D
from dpm_solver_amd import _lib as L

B, N_EL = 256, 256 * 4 * 64 * 64


def arena_sets(n_sets, dtype, dev, pad, eps_dtype=None):
    eps_dtype = eps_dtype or dtype
    esz = torch.empty((), dtype=dtype).element_size()
    size = N_EL * esz
    slot = size + pad
    arena = torch.empty(n_sets * 8 * slot + 4096, dtype=torch.uint8, device=dev)
    base = (arena.data_ptr() + 255) // 256 * 256
    g = torch.Generator(device="cpu").manual_seed(3)
    sets = []
    views = []
    for s in range(n_sets):
        ptrs = [base + (s * 8 + k) * slot for k in range(8)]
        off = [p - arena.data_ptr() for p in ptrs]
        ts = [arena[o:o + size].view(dtype) for o in off]
        ts[0].copy_(torch.randn(N_EL, generator=g).to(dev, dtype))
        if eps_dtype != dtype:       # network output narrower than the state: it occupies the front of its slot
            ts[1] = arena[off[1]:off[1] + N_EL * torch.empty((), dtype=eps_dtype).element_size()].view(eps_dtype)
        ts[1].copy_(torch.randn(N_EL, generator=g).to(dev, eps_dtype))
        rb = L.RunBuffers()
        rb.xbuf[0] = ptrs[0]
        rb.e0 = ptrs[1]
        for i in range(3):
            rb.xbuf[1 + i] = ptrs[2 + i]
        for i in range(2):
            rb.hist[i] = ptrs[5 + i]
        rb.n, rb.batch = N_EL, B
        dmap = {torch.float16: L.DTYPE_F16, torch.float32: L.DTYPE_F32, torch.bfloat16: L.DTYPE_BF16}
        rb.state_dtype, rb.eps_dtype = dmap[dtype], dmap[eps_dtype]
        sets.append(rb)
        views.append(ts)
    return arena, sets, views


def measure(plan, sets, nst, sptr):
    res = C.c_int(-1)
    n = len(sets)
    rbs = (L.RunBuffers * n)(*sets)
    out = {}
    for i in range(n):
        L.check(L.lib.dpm_plan_run(plan.handle, C.byref(sets[i]), None, None, sptr, C.byref(res)))
    buf = (C.c_float * nst)()
    ks = []
    for i in range(16):
        L.check(L.lib.dpm_plan_run_timed(plan.handle, C.byref(sets[i % n]), sptr, buf, C.byref(res)))
        ks.append(np.frombuffer(buf, dtype=np.float32)[1:nst - 1].copy())
    out["k_seq_us"] = round(float(np.mean(ks) * 1e3), 3)
    resm = (C.c_int * n)()
    L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, n, sptr, None, resm))
    msb = (C.c_float * (n * nst))()
    kc = []
    for i in range(3):
        L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, n, sptr, msb, resm))
        kc.append(np.frombuffer(msb, dtype=np.float32).reshape(n, nst)[:, 1:nst - 1].copy())
    out["k_cold_us"] = round(float(np.mean(kc) * 1e3), 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4):
        L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, n, sptr, None, resm))
    torch.cuda.synchronize()
    out["wall_cold_us"] = round((time.perf_counter() - t0) / (4 * n * nst) * 1e6, 3)
    return out


def main():
    dev = torch.device("cuda", 0)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(BN.sd_alphas_cumprod()))
    sptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for dname in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp16", "fp32"]):
        dtype = {"fp16": torch.float16, "fp32": torch.float32}[dname]
        esz = 2 if dname == "fp16" else 4
        alg = 5 * N_EL * esz
        dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, state_dtype=dtype)
        plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                             lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1e-3)
        nst = len(plan.stages)
        for pad in (0, 256, 4096, 4096 + 256, 65536 + 4096 + 256, 2 * 1024 * 1024 // 8 + 4096):
            arena, sets, views = arena_sets(8, dtype, dev, pad)
            for U in (1, 2):
                for NT in (0, 1, 5, 6, 7):
                    for bpc in ((8,) if U == 1 else (4,)):
                        L.check(L.lib.dpm_tuning_set(L.TUNE_UNROLL, U))
                        L.check(L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, NT))
                        L.check(L.lib.dpm_tuning_set(L.TUNE_BLOCKS_PER_CU, bpc))
                        r = measure(plan, sets, nst, sptr)
                        r.update(dtype=dname, pad=pad, U=U, NT=NT, bpc=bpc, seq_GBs=round(alg / r["k_seq_us"] / 1e3),
                                 cold_GBs=round(alg / r["k_cold_us"] / 1e3))
                        print(json.dumps(r), flush=True)
            del arena, sets, views
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
