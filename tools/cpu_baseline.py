#!/usr/bin/env python3
"""Time the unmodified reference (DPM_Solver.sample of $DPM_REFERENCE_DIR/dpm_solver_pytorch.py, default
/root/reference) on this machine's host cores on bench.py's workload and write profiles/cpu_baseline_reference.json.
The reference checkout exists in the build container only; bench.py attaches this file to its `cpu_baseline` when it
runs where the reference is absent (the GPU box)."""
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ref = bench.reference_dir()
    assert ref, "no reference checkout (DPM_REFERENCE_DIR)"
    out = bench.cpu_baseline_reference(ref, bench.sd_alphas_cumprod(), budget_s=30.0)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    out["host"] = dict(cpu=cpu, cores=os.cpu_count(), machine=platform.machine(), where="build container (no GPU)")
    p = os.path.join(ROOT, "profiles", "cpu_baseline_reference.json")
    json.dump(out, open(p, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
