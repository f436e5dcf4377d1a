#!/usr/bin/env python3
"""Clocks while the fused 2M launch streams back to back vs behind compute (tools/duty_cycle.py found 217 us vs 187 us for the same
cold 1.34 GB launch): sample the SMI clocks (sysfs pp_dpm_* / rocm-smi) from a thread while each mode runs."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_sysfs():
    out = {}
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_*"):
        try:
            cur = [l for l in open(f).read().splitlines() if l.strip().endswith("*")]
            if cur:
                out[os.path.basename(f)] = cur[0].strip()
        except Exception:
            pass
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try:
            out["power_W"] = round(int(open(f).read()) / 1e6, 1)
        except Exception:
            pass
    return out


def read_smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        c = d.get("card0", d[next(iter(d))])
        return {k: v for k, v in c.items() if "clock" in k.lower() or "power" in k.lower()}
    except Exception as e:
        return {"error": str(e)}



if __name__ == "__main__":
    print("sysfs idle:", read_sysfs())
    print("rocm-smi idle:", read_smi())
    import numpy as np
    import torch
    import ctypes as C
    import bench
    import dpm_solver_amd as D
    from dpm_solver_amd import _lib as L
    dev = torch.device("cuda", 0)
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, algorithm_type="dpmsolver++", state_dtype=torch.float16)
    plan = dpm._get_plan(method="multistep", order=2, steps=20, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    R = 32
    sets = bench.make_sets(R, torch.float16, dev, seed=5)
    rbs = (L.RunBuffers * R)(*[s_["rb"] for s_ in sets])
    res = (C.c_int * R)()
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    w = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    stop = [False]
    samples = []

    def sampler():
        while not stop[0]:
            samples.append((time.perf_counter(), read_sysfs()))
            time.sleep(0.05)

    def run(label, fn, seconds=3.0):
        samples.clear()
        stop[0] = False
        th = threading.Thread(target=sampler)
        th.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            fn()
            n += 1
        torch.cuda.synchronize()
        smi = read_smi()
        stop[0] = True
        th.join()
        keys = sorted({k for _, s in samples for k in s})
        print("== %s (%d iterations in %.1f s)" % (label, n, seconds))
        for k in keys:
            vals = [s.get(k) for _, s in samples[len(samples) // 3:]]
            uniq = {}
            for v in vals:
                uniq[v] = uniq.get(v, 0) + 1
            print("   %-16s %s" % (k, sorted(uniq.items(), key=lambda kv: -kv[1])[:4]))
        print("   rocm-smi at the end:", smi)

    def streaming():
        L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, sp, None, res))
        torch.cuda.synchronize()

    def gemms():
        for _ in range(20):
            torch.mm(w, w)
        torch.cuda.synchronize()

    def mixed():
        for _ in range(4):
            torch.mm(w, w)
        L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, sp, None, res))
        torch.cuda.synchronize()
    run("fused stage launches back to back (the bench's timed region)", streaming)
    run("GEMMs only", gemms)
    run("4 GEMMs + one 20-launch trajectory, alternating", mixed)
    run("fused stage launches back to back again", streaming)
