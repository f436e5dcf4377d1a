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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (tools run on the LAB build of the library: include/dpm_lab.h)


def summarise(d, md=None, title="", pattern="stage_"):
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


# The kernels of the other BASELINE configs inside a real torch network loop: DPM_Solver.sample() with a random-init conv
# network (MIOpen) as the model, to be run under `rocprofv3 --kernel-trace --stats` and summarised with --summarise.
#   shape, state dtype, network dtype, CFG scale (None: unguided), thresholding, sample() kwargs, schedule, network width,
#   memory format of network and x_T
CASES = {
    # classifier-free guidance 7.5 + the duplicate store of the [2B,...] network input, SD under autocast (ref :322-330)
    "cfg_sd64": dict(shape=(64, 4, 64, 64), state="fp32", net="fp16", cfg=7.5, thr=False, kw=dict(steps=20, order=2), sched="sd", width=256),
    "cfg_sd8": dict(shape=(8, 4, 64, 64), state="fp32", net="fp16", cfg=7.5, thr=False, kw=dict(steps=20, order=2), sched="sd", width=256),
    # BASELINE cfg5: dynamic thresholding, pixel space, 25 steps (ref :416-425)
    "cfg5": dict(shape=(32, 3, 64, 64), state="fp32", net="fp32", cfg=None, thr=True, kw=dict(steps=25, order=2), sched="ddpm", width=128),
    # BASELINE cfg3: DPM-Solver-3 singlestep, 15 NFE, CFG 7.5 (ref :675-794)
    "cfg3": dict(shape=(64, 3, 256, 256), state="fp32", net="fp32", cfg=7.5, thr=False,
                 kw=dict(steps=15, order=3, method="singlestep"), sched="ddpm", algo="dpmsolver", width=32),
    # the plain 2M kernel behind a conv network in the default layout and in channels_last
    # SD-style autocast at the north-star size: fp32 state, fp16 network (the split layout + lane exchange of the 2-byte streams)
    "autocast256": dict(shape=(256, 4, 64, 64), state="fp32", net="fp16", cfg=None, thr=False, kw=dict(steps=20, order=2), sched="sd", width=256),
    # the plain 2M kernel at SD's batch (512 tiles: the smallest launch that takes 512-thread workgroups)
    "plain64": dict(shape=(64, 4, 64, 64), state="fp16", net="fp16", cfg=None, thr=False, kw=dict(steps=20, order=2), sched="sd", width=256),
    "nchw": dict(shape=(256, 4, 64, 64), state="fp16", net="fp16", cfg=None, thr=False, kw=dict(steps=20, order=2), sched="sd", width=256),
    "nhwc": dict(shape=(256, 4, 64, 64), state="fp16", net="fp16", cfg=None, thr=False, kw=dict(steps=20, order=2), sched="sd", width=256,
                 channels_last=True),
}


def run_case(name, trajectories):
    import time
    import torch
    import bench
    import dpm_solver_amd as D
    c = CASES[name]
    dev = torch.device("cuda", 0)
    sd, nd = bench._DT[c["state"]], bench._DT[c["net"]]
    shape = c["shape"]
    if c["sched"] == "sd":
        ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    else:
        ns = D.NoiseScheduleVP("discrete", betas=torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64))
    net = bench.LoopNet("conv", c["width"], nd, dev, channels=shape[1])
    cl = bool(c.get("channels_last"))
    if cl:
        net = net.to(memory_format=torch.channels_last)
    scale = 0.5 if c["thr"] else 1.0
    if c["cfg"] is not None:
        cond = torch.ones(shape[0], device=dev)
        model = D.model_wrapper(lambda x, t, cc: net(x.to(nd), t), ns, guidance_type="classifier-free", condition=cond,
                                unconditional_condition=cond * 0, guidance_scale=c["cfg"])
    elif c["thr"]:
        model = D.model_wrapper(lambda x, t: net(x.to(nd), t) * scale, ns)
    else:
        model = D.model_wrapper(lambda x, t: net(x.to(nd), t), ns)    # the network's last kernel is its last convolution
    kwargs = dict(algorithm_type=c.get("algo", "dpmsolver++"))
    if c["thr"]:
        kwargs["correcting_x0_fn"] = "dynamic_thresholding"
    if sd is not torch.float32:
        kwargs["state_dtype"] = sd
    dpm = D.DPM_Solver(model, ns, **kwargs)
    g = torch.Generator(device="cpu").manual_seed(4321)
    x = torch.randn(shape, generator=g).to(dev, sd)
    if cl:
        x = x.to(memory_format=torch.channels_last)
    with torch.no_grad():
        out = dpm.sample(x, **c["kw"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(trajectories):
            out = dpm.sample(x, **c["kw"])
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / trajectories
        # the same network calls alone (same inputs, same number): what the solver adds to a trajectory
        calls = []
        inner = model.model
        model.model = lambda *a: (calls.append(a), inner(*a))[1]
        dpm.sample(x, **c["kw"])
        model.model = inner
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(trajectories):
            for a in calls:
                inner(*a)
        torch.cuda.synchronize()
        wall_net = (time.perf_counter() - t0) / trajectories
    n = 1
    for d in shape:
        n *= d
    assert torch.isfinite(out.float()).all()
    print(json.dumps(dict(case=name, shape=list(shape), state=c["state"], network=c["net"], cfg=c["cfg"], thresholding=c["thr"],
                          sample_kwargs=c["kw"], channels_last=cl, elements=n, network_calls=len(calls),
                          trajectory_ms=round(wall * 1e3, 4), network_calls_alone_ms=round(wall_net * 1e3, 4),
                          solver_added_us_per_call=round((wall - wall_net) / max(len(calls), 1) * 1e6, 2),
                          network_ms_per_call=round(wall_net / max(len(calls), 1) * 1e3, 4),
                          out_channels_last=bool(out.dim() == 4 and out.is_contiguous(memory_format=torch.channels_last)
                                                 and not out.is_contiguous()))), flush=True)


def resident_experiment(kind, dtype_name, width, trajectories, configs):
    """EXPERIMENT: the resident stage kernel (dpm_resident_*: woken by
    hipStreamWriteValue64 behind the network's last kernel, the next network call held by hipStreamWaitValue32) against the
    dispatched stage kernel of DPM_Solver.sample(), same network, same [256,4,64,64] request, same arithmetic (the results
    must be equal).  Reports per configuration (workgroups, sleep): the wall time a solver stage adds to a network call and
    what the parked pollers cost the network.  Keep criterion: added wall per stage drops by >= 1.5 us AND the network
    call gets < 0.5 % slower."""
    import ctypes as C
    import torch
    import bench
    import dpm_solver_amd as D
    from dpm_solver_amd import _lib as L
    dev = torch.device("cuda", 0)
    dtype = bench._DT[dtype_name]
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    net = bench.LoopNet(kind, width, dtype, dev)
    g = torch.Generator(device="cpu").manual_seed(4321)
    x_T = torch.randn((bench.B,) + bench.SHAPE, generator=g).to(dev, dtype)
    dpm = D.DPM_Solver(D.model_wrapper(net, ns), ns, algorithm_type="dpmsolver++", state_dtype=dtype)
    n_st = bench.STEPS_SOLVER
    rows = []
    with torch.no_grad():
        want = dpm.sample(x_T, steps=n_st, order=2)
        torch.cuda.synchronize()
        fr = next(iter(dpm._fast.values()))
        plan = dpm._get_plan(method="multistep", order=2, steps=n_st, skip_type="time_uniform", solver_type="dpmsolver",
                             lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
        roles = plan.roles
        tin = plan.time_views(dev, bench.B, False)["t_input_b"]
        stages = (L.Stage * n_st)(*fr.stages)
        bufs = (L.Buffers * n_st)(*fr.bufs)
        out = torch.empty_like(want)
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)

        def timed(fn):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3

        def net_only():
            for i in range(n_st):
                net(x_T, tin[i])
        dispatched = lambda: dpm.sample(x_T, steps=n_st, order=2)
        for wgs, sleep in configs:
            h = C.c_void_p()
            L.check(L.lib.dpm_resident_create(stages, bufs, n_st, wgs, sleep, C.byref(h)))

            def start():
                side.wait_stream(main)
                L.check(L.lib.dpm_resident_start(h, x_T.data_ptr(), out.data_ptr(), C.c_void_p(side.cuda_stream)))

            def resident():
                start()
                keep = []
                for i in range(n_st):
                    xe = x_T if roles[i][1] == 0 else fr.xbuf[roles[i][1]]
                    eps = net(xe, tin[i])
                    keep.append(eps)
                    L.check(L.lib.dpm_resident_signal(h, i, eps.data_ptr(), C.c_void_p(main.cuda_stream)))
                main.wait_stream(side)
                return keep

            net_ms = [None]

            def parked():
                """the network calls alone while the resident kernel sits on the chip polling (then let it run out)"""
                start()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                net_only()
                e1.record()
                eps = net(x_T, tin[0])
                for i in range(n_st):
                    L.check(L.lib.dpm_resident_signal(h, i, eps.data_ptr(), C.c_void_p(main.cuda_stream)))
                main.wait_stream(side)
                torch.cuda.synchronize()
                net_ms[0] = e0.elapsed_time(e1) * 1e3

            resident()
            torch.cuda.synchronize()
            equal = bool(torch.equal(out, want))
            t_res, t_dis, t_net, t_park = [], [], [], []
            for _ in range(max(6, trajectories)):
                t_res.append(timed(resident))
                t_dis.append(timed(dispatched))
                t_net.append(timed(net_only))
                parked()
                t_park.append(net_ms[0])
            med = lambda v: float(np.median(v))
            row = dict(workgroups=wgs, sleep_x_s_sleep64=sleep, result_equals_dispatched=equal,
                       network_ms_per_call=round(med(t_net) / n_st / 1e3, 4),
                       network_ms_per_call_with_pollers_parked=round(med(t_park) / n_st / 1e3, 4),
                       network_slowdown_pct=round((med(t_park) / med(t_net) - 1.0) * 100.0, 3),
                       stage_added_wall_us_dispatched=round((med(t_dis) - med(t_net)) / n_st, 3),
                       stage_added_wall_us_resident=round((med(t_res) - med(t_net)) / n_st, 3),
                       trajectory_ms_dispatched=round(med(t_dis) / 1e3, 4), trajectory_ms_resident=round(med(t_res) / 1e3, 4))
            row["gain_us_per_stage"] = round(row["stage_added_wall_us_dispatched"] - row["stage_added_wall_us_resident"], 3)
            row["keep"] = bool(equal and row["gain_us_per_stage"] >= 1.5 and row["network_slowdown_pct"] < 0.5)
            rows.append(row)
            print(json.dumps(row), flush=True)
            L.lib.dpm_resident_destroy(h)
    return rows


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
    ap.add_argument("--block-threads", type=int, default=-1, help="DPM_TUNE_BLOCK_THREADS for the run (0 = by size, the default)")
    ap.add_argument("--calib-shape", default="256:8:1", help="--calib: threads per workgroup : workgroups per CU : nt mask")
    ap.add_argument("--pattern", default="stage_", help="--summarise: substring of the kernel rows to report")
    ap.add_argument("--sweep", action="store_true", help="tuning build: (tiles per workgroup, nt mask) of the 2M kernel INSIDE "
                    "the loop, for fp16, fp32 and fp32 state + fp16 network (events)")
    ap.add_argument("--case", default=None, choices=sorted(CASES), help="trace-only runs of the OTHER kernels the BASELINE configs "
                    "launch, each inside a real torch network loop (VERDICT round 3, item 5): see CASES")
    ap.add_argument("--resident", default=None, help="EXPERIMENT: resident stage kernel vs the dispatched one; comma list of "
                    "workgroups:sleep, e.g. 512:1,1024:1,2048:1,1024:4")
    ap.add_argument("--unroll", type=int, default=0, help="tuning build only: tiles per workgroup of the 2M stage kernel")
    ap.add_argument("--nt", type=int, default=-1, help="tuning build only: nt mask")
    ap.add_argument("--lds-dma", type=int, default=-1, help="DPM_TUNE_LDS_DMA: 1 / 0 = the lone 2-byte 2M launch reads by LDS-DMA / "
                    "through registers (default: the library's choice)")
    args = ap.parse_args()
    if args.summarise:
        return summarise(args.summarise, args.md, args.title, args.pattern)
    if args.block_threads >= 0:
        from dpm_solver_amd import _lib as L0
        L0.check(L0.lib.dpm_tuning_set(L0.TUNE_BLOCK_THREADS, args.block_threads))
    if args.case:
        return run_case(args.case, args.trajectories)
    if args.resident:
        cfgs = [tuple(int(v) for v in c.split(":")) for c in args.resident.split(",")]
        rows = resident_experiment(args.kinds.split(",")[0], args.dtype, args.width, args.trajectories, cfgs)
        if args.out:
            json.dump(rows, open(args.out, "w"), indent=1)
        return
    import torch
    import bench
    import lab_secondary as LS
    import dpm_solver_amd as D
    from dpm_solver_amd import _lib as L
    if args.lds_dma >= 0:
        L.check(L.lib.dpm_tuning_set(L.TUNE_LDS_DMA, args.lds_dma))
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
        cblk, cbpc, cnt = (int(v) for v in args.calib_shape.split(":"))
        with torch.no_grad():
            xi, mi = x, bufs[0]
            for it in range(20 * args.trajectories):
                eps = net(xi, tvec)
                xo, mo = (bufs[1], bufs[2]) if it % 2 == 0 else (bufs[3], bufs[0])
                L.check(L.lib.dpm_calib_launch(1, cblk, cbpc, cnt, xi.data_ptr(), eps.data_ptr(), mi.data_ptr(), xo.data_ptr(), mo.data_ptr(),
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
                    r = LS.in_network_loop(D, L, ns, dev, bench._DT[sname], kind=args.kinds.split(",")[0], width=args.width,
                                              trajectories=args.trajectories, net_dtype=bench._DT[ename])
                    print("%s state / %s network  U=%d nt=%d   stage kernel in the loop %7.3f us (events; %.3f of peak)   "
                          "added wall per stage %7.3f us" % (sname, ename, u, nt, r["stage_kernel_us"], r["frac"],
                                                            r["trajectory_minus_network_alone_us_per_stage"]), flush=True)
        return
    if args.unroll:
        L.check(L.lib.dpm_tuning_set(L.TUNE_UNROLL, args.unroll))
        L.check(L.lib.dpm_tuning_set(L.TUNE_NONTEMPORAL, args.nt if args.nt >= 0 else 1))
    res = []
    for kind in args.kinds.split(","):
        if args.trace_only:
            pf = None if args.prefetch in (None, "None", "none") else int(args.prefetch)
            r = LS.in_network_loop(D, L, ns, dev, dtype, kind=kind, width=args.width, trajectories=args.trajectories, prefetch=pf)
            r["variant"] = "%s prefetch=%s (under the profiler)" % (kind, pf)
            res.append(r)
            continue
        for pf in (None, 0, 1):
            r = LS.in_network_loop(D, L, ns, dev, dtype, kind=kind, width=args.width, trajectories=args.trajectories, prefetch=pf)
            r["variant"] = "%s prefetch=%s" % (kind, pf)
            res.append(r)
            print(json.dumps(r), flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    if args.trace_only:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
