#!/usr/bin/env python3
"""Per-workgroup phase timeline of the thresholding stage kernel (the analysis behind DESIGN.md section 5).

    python tools/thr_timeline.py --build                 # here or on the GPU box: a -DDPM_THR_TIMING library under tools/_thr_timing/
    python tools/thr_timeline.py --run [--batch 1024 --chw 3 64 64]     # MI355X (through gpurun)

The debug library stamps `wall_clock64()` (100 MHz) at the phase boundaries of the first sample every workgroup
processes; the 40th thresholding launch of the process is dumped.  Printed: median / p10 / p90 of every phase and how
many workgroups sit in the load phase, the select and the store phase every 4 us."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_thr_timing")
LIB = os.path.join(OUT, "libdpm_hip_timing.so")


def build():
    """the instrumented library: the LAB build's sources with -DDPM_THR_TIMING (the stamps are lab-only code)"""
    sys.path.insert(0, ROOT)
    import shutil
    import __graft_entry__ as G
    lib = G.build_variant("thr_timing", ["-DDPM_THR_TIMING"], lab=True)
    os.makedirs(OUT, exist_ok=True)
    shutil.copy(lib, LIB)
    print("built", LIB)


def run(batch, chw):
    import numpy as np
    dump = os.path.join(OUT, "stamps.txt")
    if os.path.exists(dump):
        os.remove(dump)
    env = dict(os.environ, DPM_SOLVER_AMD_LIB=LIB, DPM_THR_TIMING_FILE=dump)
    code = (
        "import numpy as np, torch, dpm_solver_amd as D\n"
        "ns = D.NoiseScheduleVP('discrete', betas=torch.from_numpy(np.linspace(1e-4, 0.02, 1000).astype(np.float32)))\n"
        "shape = (%d, %d, %d, %d)\n"
        "e = torch.randn(shape, device='cuda')\n"
        "s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, ns), ns, correcting_x0_fn='dynamic_thresholding')\n"
        "x = torch.randn(shape, device='cuda')\n"
        "for _ in range(3): s.sample(x, steps=25, order=2)\n"
        "torch.cuda.synchronize()\n" % ((batch,) + tuple(chw)))
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT, env=env)
    a = np.loadtxt(dump, dtype=np.float64) / 100.0          # microseconds
    t0 = a[:, 0].min()
    names = [(0, "start"), (1, "x0 in LDS (load phase)"), (8, "one-hop: chunk max, maxima histogram, bound"),
             (9, "one-hop: candidates compacted and published"), (10, "one-hop: headers of all slots arrived"),
             (11, "one-hop: union gathered"), (12, "one-hop: select on the union"),
             (4, "maxima histogram"), (5, "bin located"),
             (6, "candidates compacted / exchanged"), (7, "rank counting"), (2, "threshold known"), (3, "end (store phase)")]
    used = [(j, nm) for j, nm in names if np.all(a[:, j] > 0)]
    print("%d workgroups, first-sample span %.1f us" % (len(a), a[:, 3].max() - t0))
    for (j0, n0), (j1, n1) in zip(used[:-1], used[1:]):
        d = a[:, j1] - a[:, j0]
        print("  %-36s %6.2f us  (p10 %.2f, p90 %.2f)" % (n1, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    print("workgroups per phase over time:")
    for ts in np.arange(0.0, a[:, 3].max() - t0, 4.0):
        x = ts + t0
        print("  t = %5.1f us   load %4d   select %4d   store %4d" % (
            ts, ((a[:, 0] <= x) & (x < a[:, 1])).sum(), ((a[:, 1] <= x) & (x < a[:, 2])).sum(), ((a[:, 2] <= x) & (x < a[:, 3])).sum()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--chw", type=int, nargs=3, default=[3, 64, 64])
    args = ap.parse_args()
    if args.build or not os.path.exists(LIB):
        build()
    if args.run:
        run(args.batch, args.chw)
