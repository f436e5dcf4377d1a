#!/usr/bin/env python3
"""Cold-start cost of the product library: what a txt2img-style caller -- one trajectory per
prompt batch, scripts/txt2img.py:251-311 -- pays before its first result, on a FRESH process:

    import torch | import dpm_solver_amd (dlopen of the library: its code objects are registered with the HIP runtime)
    | first sample() at [8,4,64,64] (first launches: the runtime loads the code object that holds the kernel)
    | second sample() (steady state) -- and the same with dynamic thresholding and with classifier-free guidance.

Each scenario runs in its own subprocess (nothing is warm), `--repeat` times; medians are reported.

    python tools/cold_start.py [--repeat 3] [--out gpurun_out/cold_start.json]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
t0 = time.perf_counter()
import torch
torch.cuda.init()
x_probe = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t1 = time.perf_counter()
sys.path.insert(0, %(root)r)
import dpm_solver_amd as D
t2 = time.perf_counter()
import numpy as np
scenario = %(scenario)r
betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(np.cumprod(1.0 - betas).astype(np.float32)))
dev = torch.device("cuda", 0)
kw = dict(steps=20, order=2)
if scenario == "plain":
    x = torch.randn(8, 4, 64, 64, device=dev)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns, algorithm_type="dpmsolver++")
elif scenario == "thresholding":
    x = torch.randn(32, 3, 64, 64, device=dev)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns, algorithm_type="dpmsolver++",
                       correcting_x0_fn="dynamic_thresholding")
    kw = dict(steps=25, order=2)
else:                                   # classifier-free guidance, fp16 network under an fp32 state (SD under autocast)
    x = torch.randn(8, 4, 64, 64, device=dev)
    cond = torch.ones(8, device=dev)
    model = D.model_wrapper(lambda xx, t, c: (xx * (1 + 0.1 * c.view(-1, 1, 1, 1))).half(), ns, guidance_type="classifier-free",
                            condition=cond, unconditional_condition=cond * 0, guidance_scale=7.5)
    dpm = D.DPM_Solver(model, ns, algorithm_type="dpmsolver++")
torch.cuda.synchronize()
net_ms = None
if %(warm_net)r:
    # the stand-in network's own first call (torch loads the code objects of ITS kernels lazily too): not the solver's cost
    tn = time.perf_counter()
    tv = torch.full((x.shape[0],), 0.5, device=dev)
    _ = dpm._wrapped(x, tv); torch.cuda.synchronize()
    net_ms = (time.perf_counter() - tn) * 1e3
t3 = time.perf_counter()
y = dpm.sample(x, **kw); torch.cuda.synchronize()
t4 = time.perf_counter()
y = dpm.sample(x, **kw); torch.cuda.synchronize()
t5 = time.perf_counter()
for _ in range(10):
    y = dpm.sample(x, **kw)
torch.cuda.synchronize()
t6 = time.perf_counter()
print(json.dumps(dict(scenario=scenario, import_torch_and_context_ms=(t1 - t0) * 1e3, import_dpm_solver_amd_ms=(t2 - t1) * 1e3,
                      first_sample_ms=(t4 - t3) * 1e3, network_first_call_ms=net_ms, second_sample_ms=(t5 - t4) * 1e3, steady_sample_ms=(t6 - t5) * 1e2,
                      library=D.LIB_PATH, library_bytes=os.path.getsize(D.LIB_PATH))))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cold_start.json"))
    ap.add_argument("--scenarios", default="plain,thresholding,cfg")
    args = ap.parse_args()
    rows = []
    for scenario in args.scenarios.split(","):
        def child(warm_net):
            runs = []
            for _ in range(args.repeat):
                r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, scenario=scenario, warm_net=warm_net)], cwd=ROOT,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
                if r.returncode != 0:
                    print(r.stderr[-2000:], file=sys.stderr)
                    raise SystemExit(1)
                runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
            med = {k: (sorted(v[k] for v in runs)[len(runs) // 2] if isinstance(runs[0][k], float) else runs[0][k]) for k in runs[0]}
            return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in med.items()}
        med = child(False)
        med["cold_start_ms"] = round(med["import_dpm_solver_amd_ms"] + med["first_sample_ms"], 3)
        # the same with the network's own torch kernels loaded first (one call of model_fn before the first sample()): what
        # of the first call is the solver's -- its code object(s), the plan, the launch records
        w = child(True)
        med["network_first_call_ms"] = w["network_first_call_ms"]
        med["first_sample_network_warm_ms"] = w["first_sample_ms"]
        med["cold_start_network_warm_ms"] = round(w["import_dpm_solver_amd_ms"] + w["first_sample_ms"], 3)
        med["runs"] = args.repeat
        rows.append(med)
        print(json.dumps(med), flush=True)
    out = dict(what="fresh process per scenario, medians of %d runs; cold_start_ms = import dpm_solver_amd + first sample() "
                    "(the process's torch import and HIP context are the caller's own and listed apart); "
                    "cold_start_network_warm_ms = the same in a second fresh process whose stand-in network was called once "
                    "first (network_first_call_ms: torch loading its own kernels)" % args.repeat, rows=rows)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
