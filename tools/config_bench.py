#!/usr/bin/env python3
"""Every BASELINE.json configuration on the clock, through the drop-in API.

bench.py's headline times BASELINE configs[1] with 32 requests fused per launch.  This module times the OTHER
configurations -- and the headline's bytes through plain `DPM_Solver.sample()` on one tensor -- on the PRODUCT library, in
the same process, and returns one dict per case for the `configs` block of bench.py's JSON line:

  cfg1       DPM-Solver++ 2M, 20 steps, [8,4,64,64] fp32 (configs[0] on the GPU; configs[3]'s per-GPU size)  ref :1171-1213
  cfg3       DPM-Solver-3 singlestep, 15 steps, [64,3,256,256] fp32, CFG 7.5                                 ref :675-794, :322-330
  cfg5       DPM-Solver++ 2M + dynamic thresholding, 25 steps, [32,3,64,64] fp32                             ref :416-425
  cfg_sd64   DPM-Solver++ 2M, 20 steps, [64,4,64,64] fp32 state / fp16 network, CFG 7.5 (SD under autocast)  ref :322-330
  one8192    DPM-Solver++ 2M, 20 steps, ONE [8192,4,64,64] fp16 tensor: the headline's bytes per stage       ref :1047
             through plain DPM_Solver.sample() / dpm_stage_launch (which hands a launch this large to the fused kernel as a
             group of one, Tuning::big_tiles, profiles/r06_big_single.md), not sample_requests

How a case is timed (`frozen`, the mode BASELINE's "dummy model_fn" describes): the network is a function that returns a
pre-staged tensor, so a trajectory is the solver's launches and nothing else; K trajectories of `sample()` are bracketed by
HIP events on the launch stream (eager: the Python host loop launches every stage; captured: `DPM_Solver.capture()`, one
graph launch per trajectory).  `us_per_stage` = that time / (K x stages) -- dispatch gaps and host time included, which is
what a loop pays.  `algorithmic_bytes` are counted from the launch records the run really issued (state / output / cached
value streams read + streams written, per launch; the thresholding order statistics move no extra algorithmic bytes).
Stages of the small cases follow each other within microseconds, so their inputs sit in L2 / the 256 MiB Infinity Cache --
those cases are latency-bound and carry `x_latency_bound_stage` (ratio to cfg1's stage in the same mode) instead of a
roofline fraction; cfg3 and one8192 stream > 256 MiB per stage and carry `frac` of the 8 TB/s HBM peak ("effective": the
whole loop's bytes over the whole loop's time).

`in_loop` (lab build only; tools/lab_secondary.py calls it in bench.py's lab subprocess): the same `sample()` call with a
random-init conv network as model_fn, start/stop events attached to every stage launch (dpm_stage_launch_traced): the
kernel-only duration of each stage kernel behind a real network, inputs from HBM.

    python tools/config_bench.py [--cases cfg1,cfg3,...] [--out FILE]          # frozen, product library
    rocprofv3 --kernel-trace --stats -d DIR -o kt -- python tools/config_bench.py --cases cfg3 --trace-only
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0

# shape, state dtype, network-output dtype, CFG scale (None: unguided), thresholding, algorithm, sample() kwargs, schedule,
# trajectories timed (eager / captured), conv-network width of the in-loop mode
CASES = {
    "cfg1": dict(shape=(8, 4, 64, 64), state="fp32", net="fp32", cfg=None, thr=False, algo="dpmsolver++",
                 kw=dict(steps=20, order=2), sched="sd", K=(200, 400), width=256,
                 workload="DPM-Solver++ 2M, 20 steps, [8,4,64,64] fp32 (BASELINE configs[0] on the GPU = configs[3] per-GPU size)",
                 ref="dpm_solver_pytorch.py:1171-1213"),
    "cfg3": dict(shape=(64, 3, 256, 256), state="fp32", net="fp32", cfg=7.5, thr=False, algo="dpmsolver",
                 kw=dict(steps=15, order=3, method="singlestep"), sched="ddpm", K=(30, 30), width=32,
                 workload="DPM-Solver-3 singlestep, 15 steps, [64,3,256,256] fp32, CFG 7.5 (BASELINE configs[2])",
                 ref="dpm_solver_pytorch.py:675-794, :322-330"),
    "cfg5": dict(shape=(32, 3, 64, 64), state="fp32", net="fp32", cfg=None, thr=True, algo="dpmsolver++",
                 kw=dict(steps=25, order=2), sched="ddpm", K=(100, 200), width=128,
                 workload="DPM-Solver++ 2M + dynamic thresholding, 25 steps, [32,3,64,64] fp32 (BASELINE configs[4])",
                 ref="dpm_solver_pytorch.py:416-425"),
    "cfg_sd64": dict(shape=(64, 4, 64, 64), state="fp32", net="fp16", cfg=7.5, thr=False, algo="dpmsolver++",
                     kw=dict(steps=20, order=2), sched="sd", K=(100, 200), width=256,
                     workload="DPM-Solver++ 2M, 20 steps, [64,4,64,64] fp32 state / fp16 network output, CFG 7.5 (SD under autocast)",
                     ref="dpm_solver_pytorch.py:322-330"),
    "one8192": dict(shape=(8192, 4, 64, 64), state="fp16", net="fp16", cfg=None, thr=False, algo="dpmsolver++",
                    kw=dict(steps=20, order=2), sched="sd", K=(20, 20), width=0,
                    workload="DPM-Solver++ 2M, 20 steps, ONE [8192,4,64,64] fp16 tensor through plain DPM_Solver.sample(): the "
                             "headline's bytes per stage through the drop-in call, not sample_requests",
                    ref="dpm_solver_pytorch.py:1047"),
}
ORDER = ("cfg1", "cfg3", "cfg5", "cfg_sd64", "one8192")
_SZ = {0: 4, 1: 2, 2: 2, 3: 8}            # DPM_DTYPE_* -> bytes


def launch_bytes(so, bo, L):
    """algorithmic bytes of ONE stage launch from its records (dpm_stage, dpm_buffers): streams read + streams written"""
    ss, es, n = _SZ[bo.state_dtype], _SZ[bo.eps_dtype], bo.n
    needs_x = so.form != L.FORM_DENOISE
    need_xe = bool(so.flags & L.F_TO_X0) or so.model_type in (1, 2)
    rd = ss if needs_x else 0
    if need_xe and ((bo.xe and bo.xe != bo.x) or not needs_x):
        rd += ss
    rd += es * (1 + (1 if so.guidance == 1 else 0) + (1 if so.guidance == 2 else 0))
    rd += ss * ((1 if so.form in (1, 2, 3) else 0) + (1 if so.form in (2, 3) else 0))
    wr = ss * (1 + (1 if bo.x_out2 else 0) + (1 if so.flags & L.F_STORE_M else 0))
    dup = ss * n if bo.x_out2 else 0
    return n * (rd + wr), dup


def build(name, dev, network="frozen"):
    """(solver, x_T, sample kwargs, case) of a case; network = 'frozen' (returns a pre-staged tensor) | 'conv' (bench.LoopNet)"""
    import torch
    import bench
    import dpm_solver_amd as D
    c = CASES[name]
    sd, nd = bench._DT[c["state"]], bench._DT[c["net"]]
    shape = c["shape"]
    if c["sched"] == "sd":
        ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    else:
        ns = D.NoiseScheduleVP("discrete", betas=torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64))
    g = torch.Generator(device="cpu").manual_seed(4321)
    x = torch.randn(shape, generator=g).to(dev, sd)
    scale = 0.5 if c["thr"] else 1.0
    if network == "frozen":
        B2 = (2 * shape[0],) + tuple(shape[1:])
        eps = (torch.randn(B2 if c["cfg"] is not None else shape, generator=g) * scale).to(dev, nd)
        net = lambda xx, t, *cond: eps
    else:
        cnet = bench.LoopNet("conv", c["width"], nd, dev, channels=shape[1])
        net = (lambda xx, t, *cond: cnet(xx.to(nd), t) * scale) if c["thr"] else (lambda xx, t, *cond: cnet(xx.to(nd), t))
    if c["cfg"] is not None:
        cond = torch.ones(shape[0], device=dev)
        model = D.model_wrapper(net, ns, guidance_type="classifier-free", condition=cond, unconditional_condition=cond * 0,
                                guidance_scale=c["cfg"])
    else:
        model = D.model_wrapper(net, ns)
    kwargs = dict(algorithm_type=c["algo"])
    if c["thr"]:
        kwargs["correcting_x0_fn"] = "dynamic_thresholding"
    if sd is not torch.float32:
        kwargs["state_dtype"] = sd
    return D.DPM_Solver(model, ns, **kwargs), x, dict(c["kw"]), c


def count_launches(dpm, x, kw):
    """one trajectory with a counting shim in front of the launch: [(algorithmic bytes, of which duplicate store)] per stage"""
    import dpm_solver_amd.solver as S
    from dpm_solver_amd import _lib as L
    rows = []
    real = S._stage_launch_raw

    def shim(st, b, stream):
        rows.append(launch_bytes(st._obj, b._obj, L))
        return real(st, b, stream)
    S._stage_launch_raw = shim
    try:
        dpm.sample(x, **kw)
    finally:
        S._stage_launch_raw = real
    return rows


def _events(dev, fn, reps):
    import torch
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e3 / reps                      # us per call


def measure_frozen(name, dev, captured=True, scale_k=1.0, attrs=None):
    """the `frozen` figures of a case on whatever library the process loaded (bench.py: the product library); attrs: engine
    options set on the solver (e.g. cluster_in_graph=True: thresholding keeps its workgroup clusters under capture)"""
    import torch
    dpm, x, kw, c = build(name, dev, "frozen")
    for k, v in (attrs or {}).items():
        setattr(dpm, k, v)
    with torch.no_grad():
        for _ in range(3):
            out = dpm.sample(x, **kw)
        rows = count_launches(dpm, x, kw)
        n_st = len(rows)
        alg = float(sum(r[0] for r in rows))
        dup = float(sum(r[1] for r in rows))
        ke, kc = (max(3, int(k * scale_k)) for k in c["K"])
        best = None
        for _ in range(3):                                        # best of three regions: the first may still page the graph in
            t = _events(dev, lambda: dpm.sample(x, **kw), ke)
            best = t if best is None else min(best, t)
        eager_us = best
        res = dict(case=name, workload=c["workload"], reference=c["ref"], shape=list(c["shape"]), state_dtype=c["state"],
                   network_output_dtype=c["net"], stages_per_trajectory=n_st,
                   algorithmic_bytes_per_trajectory=int(alg), algorithmic_bytes_per_stage=int(alg / n_st),
                   us_per_stage=round(eager_us / n_st, 3), us_per_trajectory=round(eager_us, 2), trajectories_timed=ke,
                   achieved_gbs=round(alg / eager_us / 1e3, 1), frac=round(alg / eager_us / 1e3 / HBM_PEAK_GBS, 4),
                   mode="frozen model_fn (pre-staged output), DPM_Solver.sample() eager: HIP events around %d trajectories on the "
                        "launch stream, dispatch gaps and host time included" % ke,
                   library="product", measured_in_this_run=True)
        if dup:
            # classifier-free guidance: the stage kernel also writes the second half of the [2B,...] network input (the
            # reference's torch.cat([x] * 2), ref :326) -- bytes the reference's update formulas do not contain
            res["algorithmic_bytes_without_duplicate_store"] = int(alg - dup)
            res["frac_without_duplicate_store"] = round((alg - dup) / eager_us / 1e3 / HBM_PEAK_GBS, 4)
        if name == "cfg3":
            # 4 order-3 steps + one order-2 + one order-1 (ref :499-516): 3 stages per order-3 step
            res["us_per_order3_step"] = round(3 * eager_us / n_st, 2)
            res["frac_label"] = "effective"
        if captured:
            g = dpm.capture(x, **kw)
            g.replay()
            torch.cuda.synchronize(dev)
            assert torch.equal(g.static_out, out), "captured trajectory differs from the eager one"
            best = None
            for _ in range(3):
                t = _events(dev, g.replay, kc)
                best = t if best is None else min(best, t)
            res["captured"] = dict(us_per_stage=round(best / n_st, 3), us_per_trajectory=round(best, 2), trajectories_timed=kc,
                                   frac=round(alg / best / 1e3 / HBM_PEAK_GBS, 4),
                                   how="DPM_Solver.capture(): the trajectory as one hipGraph, replayed back to back")
            del g
    del dpm, x
    torch.cuda.empty_cache()
    return res


def measure_in_loop(name, dev, trajectories=4):
    """LAB build only: the case's sample() with a random-init conv network as model_fn, start/stop events attached to every
    stage launch (dpm_stage_launch_traced): kernel-only duration of each stage kernel behind a real network"""
    import torch
    import dpm_solver_amd.solver as S
    from dpm_solver_amd import _lib as L
    L.require_lab("tools/config_bench.py --in-loop")
    dpm, x, kw, c = build(name, dev, "conv")
    with torch.no_grad():
        dpm.sample(x, **kw)
        rows = count_launches(dpm, x, kw)
        n_st = len(rows)
        n = n_st * trajectories
        trace = C.c_void_p()
        L.check(L.lib.dpm_trace_create(n, C.byref(trace)))
        real = S._stage_launch_raw
        count = [0]

        def traced(st, b, stream):
            k = count[0]
            count[0] += 1
            return L.lib.dpm_stage_launch_traced(st, b, stream, trace, k)
        try:
            S._stage_launch_raw = traced
            for _ in range(trajectories):
                dpm.sample(x, **kw)
            ms = (C.c_float * n)()
            L.check(L.lib.dpm_trace_read(trace, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), ms, n))
        finally:
            S._stage_launch_raw = real
            L.lib.dpm_trace_destroy(trace)
    us = np.frombuffer(ms, dtype=np.float32).reshape(trajectories, n_st).astype(np.float64)[1:] * 1e3   # first: warm-up
    by = np.array([r[0] for r in rows], dtype=np.float64)
    med = np.median(us, axis=0)                                   # per stage position
    steady = slice(1, n_st - 1) if n_st > 2 else slice(0, n_st)
    alg, t = float(by[steady].sum()), float(med[steady].sum())
    del dpm, x
    torch.cuda.empty_cache()
    return dict(case=name, stage_kernel_us=round(float(np.median(us[:, steady])), 3),
                stage_kernel_mean_us=round(float(us[:, steady].mean()), 3),
                algorithmic_bytes_per_stage=int(alg / max(len(med[steady]), 1)),
                achieved_gbs=round(alg / t / 1e3, 1), frac=round(alg / t / 1e3 / HBM_PEAK_GBS, 4),
                network="bench.LoopNet(conv, width %d) as model_fn" % c["width"], library="lab", measured_in_this_run=True,
                how="kernel-only: start/stop events attached to each stage launch inside the network loop "
                    "(dpm_stage_launch_traced), median of the steady-state stages of %d trajectories; frac = their bytes / their "
                    "time" % (trajectories - 1))


def run_all(dev, cases=ORDER, captured=True, scale_k=1.0):
    """bench.py's `configs` block: every case, frozen, on the loaded (product) library"""
    out = {}
    for name in cases:
        try:
            out[name] = measure_frozen(name, dev, captured=captured, scale_k=scale_k)
        except Exception as e:                                   # a secondary never takes the headline down
            out[name] = dict(case=name, error="%s: %s" % (type(e).__name__, e), measured_in_this_run=False)
    base = out.get("cfg1", {})
    if "us_per_stage" in base:
        for name, r in out.items():
            if "us_per_stage" in r:
                r["x_latency_bound_stage"] = round(r["us_per_stage"] / base["us_per_stage"], 3)
                if "captured" in r and "captured" in base:
                    r["captured"]["x_latency_bound_stage"] = round(r["captured"]["us_per_stage"] / base["captured"]["us_per_stage"], 3)
        out["latency_bound_stage"] = dict(
            eager_us=base["us_per_stage"], captured_us=base.get("captured", {}).get("us_per_stage"),
            what="cfg1's stage ([8,4,64,64] fp32, 1.3 MB): a launch whose duration is dispatch latency, not bytes -- the unit "
                 "x_latency_bound_stage is expressed in")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(ORDER))
    ap.add_argument("--in-loop", action="store_true", help="lab build: kernel-only durations behind a conv network")
    ap.add_argument("--trace-only", action="store_true", help="a few eager trajectories per case and nothing else (rocprofv3 runs)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.in_loop:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import _lab  # noqa: F401
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cases = [c for c in args.cases.split(",") if c]
    if args.trace_only:
        with torch.no_grad():
            for name in cases:
                dpm, x, kw, c = build(name, dev, "frozen")
                for _ in range(12):
                    dpm.sample(x, **kw)
                torch.cuda.synchronize(dev)
                del dpm, x
                torch.cuda.empty_cache()
        print("traced: %s" % ",".join(cases))
        return
    if args.in_loop:
        res = {name: measure_in_loop(name, dev) for name in cases if CASES[name]["width"]}
    else:
        res = run_all(dev, cases)
    for k, v in res.items():
        print(json.dumps({k: v}), flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
