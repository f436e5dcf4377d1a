#!/usr/bin/env python3
"""Memory-system ceilings for the 2M stage's access pattern on this GPU (no arithmetic), warm and cold."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (tools run on the LAB build of the library: include/dpm_lab.h)
from dpm_solver_amd import _lib as L


def main():
    dev = torch.device("cuda", 0)
    sptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for nbytes, tag in [(8 << 20, "8MiB/stream (fp16 [256,4,64,64])"), (16 << 20, "16MiB/stream (fp32)"), (256 << 20, "256MiB/stream")]:
        nsets = 8 if nbytes <= (16 << 20) else 2
        sets = [[torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255) for _ in range(5)] for _ in range(nsets)]
        for kind, streams in [(0, 2), (1, 5), (2, 5)]:
            for block in (256, 512, 1024):
                for bpc in (2048 // block, 4096 // block):
                    for nt in (0, 1, 5, 7):
                        ms = C.c_float()
                        # warm: same set repeatedly; cold: rotate sets (footprint > 256 MiB Infinity Cache when 8 sets of 5x16MiB.. )
                        res = {}
                        for mode in ("warm", "cold"):
                            t = []
                            for it in range(24):
                                s = sets[0] if mode == "warm" else sets[it % nsets]
                                L.check(L.lib.dpm_calib_launch(kind, block, bpc, nt, s[0].data_ptr(), s[1].data_ptr(), s[2].data_ptr(),
                                                               s[3].data_ptr(), s[4].data_ptr(), nbytes, sptr, C.byref(ms)))
                                if it >= 8:
                                    t.append(ms.value)
                            res[mode] = float(np.mean(t) * 1e3)
                        print(json.dumps(dict(size=tag, kind=("copy", "3r2w", "4r1w")[kind], block=block, bpc=bpc, nt=nt,
                                              warm_us=round(res["warm"], 2), cold_us=round(res["cold"], 2),
                                              warm_GBs=round(streams * nbytes / res["warm"] / 1e3), cold_GBs=round(streams * nbytes / res["cold"] / 1e3))),
                              flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
