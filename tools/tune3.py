#!/usr/bin/env python3
"""Re-check of the launch-shape defaults after the kernel changes (division by invariant, split-tile fp32 layout):
unroll x non-temporal mask x blocks per CU, warm (sequential trajectories) and HBM-cold (8 requests interleaved)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import bench as BN
import dpm_solver_amd as D
import tune2 as T2
from dpm_solver_amd import _lib as L


N_SETS = 32   # 32 x 8 state-sized arrays: more than 1 GB between two uses of a buffer, far beyond the 256 MiB Infinity Cache


def main():
    dev = torch.device("cuda", 0)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(BN.sd_alphas_cumprod()))
    sptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for dname in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp16", "fp32", "fp32/fp16"]):
        dtype = {"fp16": torch.float16, "fp32": torch.float32, "fp32/fp16": torch.float32}[dname]
        edt = torch.float16 if dname == "fp32/fp16" else dtype
        alg = T2.N_EL * (4 * (2 if dname == "fp16" else 4) + (2 if edt == torch.float16 else 4))
        dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, state_dtype=dtype)
        plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                             lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1e-3)
        nst = len(plan.stages)
        arena, sets, views = T2.arena_sets(N_SETS, dtype, dev, 0, edt)
        for U in (1, 2):
            for NT in (0, 1, 5, 6, 7):
                for bpc in (8,):
                    L.check(L.lib.dpm_tuning_set(L.TUNE_UNROLL, U))
                    L.check(L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, NT))
                    L.check(L.lib.dpm_tuning_set(L.TUNE_BLOCKS_PER_CU, bpc))
                    r = T2.measure(plan, sets, nst, sptr)
                    print("%s U=%d NT=%d bpc=%-2d  warm %.2f us (%d GB/s)  cold %.2f us (%d GB/s)" % (
                        dname, U, NT, bpc, r["k_seq_us"], alg / r["k_seq_us"] / 1e3, r["k_cold_us"], alg / r["k_cold_us"] / 1e3),
                        flush=True)
        del arena, sets, views
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
