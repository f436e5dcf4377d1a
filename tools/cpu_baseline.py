#!/usr/bin/env python3
"""Time the unmodified reference (DPM_Solver.sample of $DPM_REFERENCE_DIR/dpm_solver_pytorch.py, default
/root/reference) on this machine's host cores on bench.py's workload and write profiles/cpu_baseline_reference.json.
The reference checkout exists in the build container only; bench.py attaches this file to its `cpu_baseline` when it
runs where the reference is absent (the GPU box).

On a GPU box: the reference file travels there as untracked, git-ignored scratch
(`_refscratch/`, removed after the call; never committed) and only the RESULT is committed:
    DPM_REFERENCE_DIR=_refscratch python tools/cpu_baseline.py --out gpurun_out/.../cpu_baseline_reference_gpubox.json \
        --where "MI355X box host cores (gpurun)"
"""
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import argparse
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_baseline_reference.json"))
    ap.add_argument("--where", default="build container (no GPU)")
    ap.add_argument("--budget", type=float, default=30.0)
    ap.add_argument("--no-port", action="store_true", help="do not time the numpy port beside the reference")
    args = ap.parse_args()
    ref = bench.reference_dir()
    assert ref, "no reference checkout (DPM_REFERENCE_DIR)"
    out = bench.cpu_baseline_reference(ref, bench.sd_alphas_cumprod(), budget_s=args.budget)
    if not args.no_port:
        # the numpy port (what bench.py times where the reference is absent) on the SAME cores in the SAME run: the ratio
        # bench.py's line carries as cpu_baseline.port_over_reference
        port = bench.cpu_baseline_port(bench.sd_alphas_cumprod(), budget_s=args.budget / 2)
        try:
            cport = bench.cpu_baseline_port_c(bench.sd_alphas_cumprod(), budget_s=args.budget / 2)
            out["c_port_same_box"] = cport
            out["c_port_over_reference"] = dict(single_thread=round(cport["single_thread"]["value"] / out["single_thread"]["value"], 3),
                                                best=round(cport["value"] / out["value"], 3),
                                                what="plain-C fused port (oracle/dpm_oracle_kernels.c) / unmodified reference, same cores, same run")
        except Exception as e:
            out["c_port_same_box"] = dict(error="%s: %s" % (type(e).__name__, e))
        out["port_same_box"] = port
        out["port_over_reference"] = dict(
            single_thread=round(port["single_thread"]["value"] / out["single_thread"]["value"], 4),
            best=round(port["value"] / out["value"], 4),
            what="numpy port (oracle/dpm_oracle.py) Msamples/s / unmodified reference Msamples/s on the same host cores in the same "
                 "run: at one thread each, and best thread count of each (port %d threads, reference %d)" % (port["cores"], out["cores"]))
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    gpu = torch.cuda.get_device_name(0) if torch.cuda.is_available() else None
    out["host"] = dict(cpu=cpu, cores=os.cpu_count(), machine=platform.machine(), where=args.where, gpu=gpu,
                       torch=torch.__version__)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
