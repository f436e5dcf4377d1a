#!/usr/bin/env python3
"""The 2M stage kernel inside a real torch network loop: DPM_Solver.sample() on one
[256,4,64,64] request with a random-init torch network as model_fn (bench.LoopNet), the stage kernel's duration taken
between the network's kernels, with and without the prefetch launch (x and the cached model value pulled towards the
memory-side cache from a side stream while the network's last layer runs).

    python tools/in_loop.py [--kinds gemm,conv] [--dtype fp16] [--out gpurun_out/in_loop.json]   # events, all variants
    rocprofv3 --kernel-trace --stats -d DIR -o kt -- python tools/in_loop.py --trace-only [--prefetch 0]
    python tools/in_loop.py --summarise DIR/kt [--md profiles/r03_in_loop.md]                     # rows of the trace
"""
import argparse
import glob
import json
import os
import sqlite3
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarise(d, md=None, title="", pattern="stage_kernel"):
    """stage-kernel rows of a rocprofv3 kernel trace of `--trace-only`: duration, the kernel that ran before each, the gap"""
    f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    assert f, "no *_results.db under %s" % d
    cur = sqlite3.connect(f[0]).cursor()
    rows = cur.execute("select name, start, duration from kernels order by start").fetchall()
    out = ["# %s\n\n" % (title or "stage kernel inside a torch network loop (rocprofv3 --kernel-trace)")]
    names = [r[0] for r in rows]
    st = np.array([r[1] for r in rows], dtype=np.float64)
    du = np.array([r[2] for r in rows], dtype=np.float64)
    is_stage = np.array([pattern in n for n in names])
    idx = np.nonzero(is_stage)[0]
    out.append("%d kernel rows, %d stage-kernel rows\n\n" % (len(rows), len(idx)))
    # per distinct stage kernel
    out.append("| stage kernel | rows | mean us | median us | p10 | p90 |\n|---|---|---|---|---|---|\n")
    for n in sorted(set(names[i] for i in idx)):
        v = du[[i for i in idx if names[i] == n]] / 1e3
        out.append("| `%s` | %d | %.3f | %.3f | %.3f | %.3f |\n" % (n[:150].replace("|", "/"), len(v), v.mean(), np.median(v),
                                                                  np.percentile(v, 10), np.percentile(v, 90)))
    # what runs right before / after a stage kernel, and the gaps
    prev = {}
    gaps_b, gaps_a = [], []
    for i in idx:
        if i > 0:
            prev.setdefault(names[i - 1][:110], []).append(du[i - 1] / 1e3)
            gaps_b.append((st[i] - (st[i - 1] + du[i - 1])) / 1e3)
        if i + 1 < len(rows):
            gaps_a.append((st[i + 1] - (st[i] + du[i])) / 1e3)
    out.append("\nkernel that ends right before a stage kernel starts (the network's last kernel):\n\n| kernel | times | its mean us |\n|---|---|---|\n")
    for n, v in sorted(prev.items(), key=lambda kv: -len(kv[1]))[:6]:
        out.append("| `%s` | %d | %.2f |\n" % (n.replace("|", "/"), len(v), float(np.mean(v))))
    if gaps_b:
        out.append("\ngap previous kernel end -> stage kernel start: median %.2f us; stage kernel end -> next kernel start: "
                   "median %.2f us\n" % (float(np.median(gaps_b)), float(np.median(gaps_a))))
    # one steady-state step, row by row
    if len(idx) > 12:
        a, b = idx[10], idx[11]
        out.append("\none solver step of the trace (stage kernel, the network call, next stage kernel):\n\n| kernel | start us (rel) | duration us |\n|---|---|---|\n")
        for i in range(a, b + 1):
            out.append("| `%s` | %.2f | %.2f |\n" % (names[i][:110].replace("|", "/"), (st[i] - st[a]) / 1e3, du[i] / 1e3))
    pf = [i for i, n in enumerate(names) if "prefetch_kernel" in n]
    if pf:
        v = du[pf] / 1e3
        out.append("\nprefetch kernel: %d rows, mean %.2f us\n" % (len(v), v.mean()))
    text = "".join(out)
    if md:
        open(md, "w").write(text)
    print(text)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="gemm")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--trajectories", type=int, default=8)
    ap.add_argument("--prefetch", default=None, help="trace-only mode: None | 0 | 1")
    ap.add_argument("--trace-only", action="store_true")
    ap.add_argument("--summarise", default=None)
    ap.add_argument("--md", default=None)
    ap.add_argument("--title", default="")
    ap.add_argument("--out", default=None)
    ap.add_argument("--requests", type=int, default=0, help="trace-only: R requests advanced together by DPM_Solver.sample_requests "
                    "with the real network (one fused stage launch per stage): rows of stage_kernel_multi inside the loop")
    ap.add_argument("--calib", action="store_true", help="the NO-ARITHMETIC kernel of the same five streams (dpm_calib_launch) in "
                    "the stage kernel's place in the loop: what the memory system alone charges a lone launch there")
    ap.add_argument("--pattern", default="stage_kernel", help="--summarise: substring of the kernel rows to report")
    ap.add_argument("--sweep", action="store_true", help="tuning build: (tiles per workgroup, nt mask) of the 2M kernel INSIDE "
                    "the loop, for fp16, fp32 and fp32 state + fp16 network (events)")
    ap.add_argument("--unroll", type=int, default=0, help="tuning build only: tiles per workgroup of the 2M stage kernel")
    ap.add_argument("--nt", type=int, default=-1, help="tuning build only: nt mask")
    args = ap.parse_args()
    if args.summarise:
        return summarise(args.summarise, args.md, args.title, args.pattern)
    import torch
    import bench
    import dpm_solver_amd as D
    from dpm_solver_amd import _lib as L
    dev = torch.device("cuda", 0)
    dtype = bench._DT[args.dtype]
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    if args.requests:
        net = bench.LoopNet(args.kinds.split(",")[0], args.width, dtype, dev)
        g = torch.Generator(device="cpu").manual_seed(99)
        xs = [torch.randn((bench.B,) + bench.SHAPE, generator=g).to(dev, dtype) for _ in range(args.requests)]
        dpm = D.DPM_Solver(D.model_wrapper(net, ns), ns, algorithm_type="dpmsolver++", state_dtype=dtype)
        with torch.no_grad():
            want = [dpm.sample(x, steps=20, order=2) for x in xs[:2]]
            for _ in range(max(2, args.trajectories // 2)):
                got = dpm.sample_requests(xs, steps=20, order=2)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got[:2], want))
        print("sample_requests loop done: %d requests of [%d,4,64,64] %s, real network between the fused stage launches" % (args.requests, bench.B, args.dtype))
        return
    if args.calib:
        import ctypes as C
        net = bench.LoopNet(args.kinds.split(",")[0], args.width, dtype, dev)
        g = torch.Generator(device="cpu").manual_seed(4321)
        x = torch.randn((bench.B,) + bench.SHAPE, generator=g).to(dev, dtype)
        bufs = [torch.empty_like(x) for _ in range(4)]          # x / m ping-pong: read the pair the previous launch wrote
        tvec = torch.full((bench.B,), 500.0, device=dev)
        sptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        nbytes = x.numel() * x.element_size()
        with torch.no_grad():
            xi, mi = x, bufs[0]
            for it in range(20 * args.trajectories):
                eps = net(xi, tvec)
                xo, mo = (bufs[1], bufs[2]) if it % 2 == 0 else (bufs[3], bufs[0])
                L.check(L.lib.dpm_calib_launch(1, 256, 8, 1, xi.data_ptr(), eps.data_ptr(), mi.data_ptr(), xo.data_ptr(), mo.data_ptr(),
                                               nbytes, sptr, None))
                xi, mi = xo, mo
        torch.cuda.synchronize()
        print("calib loop done: %d launches of the no-arithmetic kernel (3 read + 2 write streams of %d bytes)" % (20 * args.trajectories, nbytes))
        return
    if args.sweep:
        for sname, ename in (("fp16", "fp16"), ("fp32", "fp32"), ("fp32", "fp16")):
            for u in (1, 2, 4):
                for nt in (0, 1, 5):
                    L.check(L.lib.dpm_tuning_set(L.TUNE_UNROLL, u))
                    L.check(L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, nt))
                    r = bench.in_network_loop(D, L, ns, dev, bench._DT[sname], kind=args.kinds.split(",")[0], width=args.width,
                                              trajectories=args.trajectories, net_dtype=bench._DT[ename])
                    print("%s state / %s network  U=%d nt=%d   stage kernel in the loop %7.3f us (events; %.3f of peak)   "
                          "added wall per stage %7.3f us" % (sname, ename, u, nt, r["stage_kernel_us"], r["frac"],
                                                            r["stage_added_wall_us"]), flush=True)
        return
    if args.unroll:
        L.check(L.lib.dpm_tuning_set(L.TUNE_UNROLL, args.unroll))
        L.check(L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, args.nt if args.nt >= 0 else 1))
    res = []
    for kind in args.kinds.split(","):
        if args.trace_only:
            pf = None if args.prefetch in (None, "None", "none") else int(args.prefetch)
            r = bench.in_network_loop(D, L, ns, dev, dtype, kind=kind, width=args.width, trajectories=args.trajectories, prefetch=pf)
            r["variant"] = "%s prefetch=%s (under the profiler)" % (kind, pf)
            res.append(r)
            continue
        for pf in (None, 0, 1):
            r = bench.in_network_loop(D, L, ns, dev, dtype, kind=kind, width=args.width, trajectories=args.trajectories, prefetch=pf)
            r["variant"] = "%s prefetch=%s" % (kind, pf)
            res.append(r)
            print(json.dumps(r), flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    if args.trace_only:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
