#!/usr/bin/env python3
"""bench.py's secondary measurements that need the LAB build of the library (include/dpm_lab.h) -- run by bench.py in a
subprocess (DPM_SOLVER_AMD_LIB = tools/_variants/lab/libdpm_lab.so), so that the process the headline is timed in loads the
product library only.  Prints ONE JSON line:

  in_network_loop        the stage kernel inside a REAL torch network loop: DPM_Solver.sample() on one [256,4,64,64] request
                         with a random-init torch network as model_fn; kernel-only durations by start/stop events attached to
                         each launch (dpm_stage_launch_traced), the wall time the 20 solver stages add to the 20 network calls,
                         and -- `floor` -- the best no-arithmetic kernel of the floor sweep (tools/floor.py,
                         profiles/r05_lone_floor.md) in the same slot of the same loop: frac_of_floor = floor / stage kernel
  no_arithmetic_ceiling  three read + two write streams with no arithmetic: one request warm / cold, the fused launch's size
  configs_in_loop        the stage kernels of the other BASELINE configurations (tools/config_bench.py: cfg1, cfg3, cfg5, cfg_sd64)
                         inside a conv-network loop, kernel-only by traced launches
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401
import torch  # noqa: E402
import bench  # noqa: E402
from bench import B, SHAPE, STEPS_SOLVER, HBM_PEAK_GBS, LoopNet  # noqa: E402,F401

# the floor sweep's best configuration (profiles/r05_lone_floor.md); measured live here in the stage kernel's slot
BEST_FLOOR = dict(load_path=0, rows=2, block=512, blocks_per_cu=0, nt=1, prio=0, store=0)
try:
    BEST_FLOOR = json.load(open(os.path.join(ROOT, "profiles", "r05_lone_floor.json")))["best_floor_config"]
except Exception:
    pass


def in_network_loop(D, L, ns, dev, dtype, kind="gemm", width=256, trajectories=6, prefetch=None, net_dtype=None):
    """DPM_Solver.sample() (2M++, 20 steps) on one [256,4,64,64] request with LoopNet as the network.  Returns the
    kernel-only duration of the steady-state stage kernel inside the loop (start/stop events attached to each launch,
    no synchronisation between launches: dpm_stage_launch_traced) and the wall time the solver stages add to the network
    calls.  prefetch = None | 0 | 1: pull the next stage's x and cached model value towards the memory-side cache from a
    side stream while the network's last layer runs (dpm_prefetch_launch, default / streaming loads)."""
    import dpm_solver_amd.solver as S
    net_dtype = net_dtype or dtype                  # fp16 network under an fp32 state: SD under autocast
    net = LoopNet(kind, width, net_dtype, dev)
    g = torch.Generator(device="cpu").manual_seed(4321)
    x_T = torch.randn((B,) + SHAPE, generator=g).to(dev, dtype)
    model = net if net_dtype == dtype else (lambda x, t: net(x.to(net_dtype), t))
    dpm = D.DPM_Solver(D.model_wrapper(model, ns), ns, algorithm_type="dpmsolver++", state_dtype=dtype)
    with torch.no_grad():
        out0 = dpm.sample(x_T, steps=STEPS_SOLVER, order=2)              # builds the launch records, warms the allocator
        torch.cuda.synchronize(dev)
        n_st = STEPS_SOLVER
        trace = C.c_void_p()
        L.check(L.lib.dpm_trace_create(n_st * trajectories, C.byref(trace)))
        raw = S._stage_launch_raw
        count = [0]
        fr = next(iter(dpm._fast.values()))
        side = torch.cuda.Stream(device=dev)
        ev = torch.cuda.Event()

        def traced(st, b, stream):
            k = count[0]
            count[0] += 1
            return L.lib.dpm_stage_launch_traced(st, b, stream, trace, k)

        def pull():                                                       # called by the network before its last layer
            i = count[0] % n_st                                           # the stage this network call feeds
            b = fr.bufs[i]
            ptrs = [p for p in (b.x, b.h1, b.h2) if p]
            if not ptrs:
                return
            ev.record()
            side.wait_event(ev)
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            nb = (C.c_int64 * len(ptrs))(*[x_T.numel() * x_T.element_size()] * len(ptrs))
            L.check(L.lib.dpm_prefetch_launch(arr, nb, len(ptrs), int(prefetch), C.c_void_p(side.cuda_stream)))

        net.before_last = pull if prefetch is not None else None
        if prefetch is not None:          # x and the cached model value are expected in the memory-side cache then
            for b in fr.bufs[1:]:
                b.inputs_resident = 1
        try:
            S._stage_launch_raw = traced
            for _ in range(trajectories):
                out = dpm.sample(x_T, steps=STEPS_SOLVER, order=2)
            ms = (C.c_float * (n_st * trajectories))()
            L.check(L.lib.dpm_trace_read(trace, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), ms, n_st * trajectories))
        finally:
            S._stage_launch_raw = raw
            L.lib.dpm_trace_destroy(trace)
        assert torch.equal(out, out0), "traced / prefetching runs changed the result"
        us = np.frombuffer(ms, dtype=np.float32).reshape(trajectories, n_st).astype(np.float64) * 1e3
        steady = us[1:, 1:n_st - 1]                                       # first trajectory: warm-up
        # wall: K trajectories with the solver vs the same network calls alone
        tb = dpm._get_plan(method="multistep", order=2, steps=STEPS_SOLVER, skip_type="time_uniform", solver_type="dpmsolver",
                           lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N).time_views(dev, B, False)
        tin = tb["t_input_b"]

        def timed(fn):
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1) * 1e3                              # us

        def net_only():
            for i in range(n_st):
                model(x_T, tin[i])
        with_solver = lambda: dpm.sample(x_T, steps=STEPS_SOLVER, order=2)
        with_solver()
        net_only()
        # alternate the two (A B A B ...) and take the median of the PAIRED differences: the network's own time drifts by more
        # than the 20 stage kernels cost (40 ms +- 0.5 % is +- 10 us per stage), so medians of the two series taken
        # separately measure the drift -- round 4's 4.5 ... 37.6 us spread over boxes was that, not host overhead: by
        # rocprofv3 rows the stage kernel starts 0.00 us behind the network's last kernel and the next network kernel 0.12 us
        # behind it (profiles/r03_in_loop.md).  Adjacent runs share the drift; their difference does not carry it.
        ts, tn = [], []
        for _ in range(max(24, 3 * trajectories)):
            ts.append(timed(with_solver))
            tn.append(timed(net_only))
        ts, tn = np.array(ts), np.array(tn)
        pair = np.concatenate([ts - tn, ts[1:] - tn[:-1]])               # each solver run against both of its neighbours
        t_net = float(np.median(tn))
        t_solver = t_net + float(np.median(pair))
        added_iqr = [float(np.percentile(pair, 25)) / n_st, float(np.percentile(pair, 75)) / n_st]
        net.before_last = None
        for b in fr.bufs:
            b.inputs_resident = 0
    n_el = B * int(np.prod(SHAPE))
    ssz = x_T.element_size()
    alg = n_el * (4 * ssz + torch.empty((), dtype=net_dtype).element_size())
    med = float(np.median(steady))
    added = (t_solver - t_net) / n_st
    return dict(network="LoopNet(%s, width %d): %.2f ms per call" % (kind, width, t_net / n_st / 1e3),
                network_ms_per_call=round(t_net / n_st / 1e3, 4),
                stage_kernel_us=round(med, 3), stage_kernel_mean_us=round(float(steady.mean()), 3),
                stage_kernel_p10_p90_us=[round(float(np.percentile(steady, 10)), 3), round(float(np.percentile(steady, 90)), 3)],
                frac=round(alg / med / 1e3 / HBM_PEAK_GBS, 4), achieved=round(alg / med / 1e3, 1),
                first_stage_us=round(float(np.median(us[1:, 0])), 3), last_stage_us=round(float(np.median(us[1:, -1])), 3),
                # (trajectory - the same 20 network calls alone) / 20, median of paired alternating runs -- NOT the solver's
                # cost: the two runs feed the network different data (evolving states vs the constant x_T) and a 40 ms
                # network-dominated pair differs by +-0.5 ms for that reason alone.  By rocprofv3 rows the stage kernel starts
                # 0.00 us behind the network's last kernel and the network's next kernel 0.00 us behind it
                # (profiles/r05_in_loop_trace_conv_fp16.md): on the GPU's timeline a stage adds its kernel, nothing else.
                trajectory_minus_network_alone_us_per_stage=round(added, 3),
                trajectory_minus_network_alone_iqr_us_per_stage=[round(v, 3) for v in added_iqr],
                solver_share_of_trajectory=round(n_st * med / t_solver, 5),
                trajectory_ms=round(t_solver / 1e3, 4), prefetch=prefetch,
                how="DPM_Solver.sample() on one [%d,4,64,64] %s request, 2M++ 20 steps, torch network as model_fn; "
                    "stage_kernel_us = median start->stop event interval of the steady-state stage launches inside the "
                    "loop (dpm_stage_launch_traced)"
                    % (B, str(dtype).split(".")[-1]))


def floor_in_loop(D, L, ns, dev, dtype, kind, trajectories=5):
    """the best floor kernel (BEST_FLOOR) and the stage kernel alternately in the slot behind the network's last kernel"""
    import floor as F
    lp = F.Loop({torch.float16: "fp16", torch.float32: "fp32", torch.bfloat16: "bf16"}[dtype], kind=kind)
    f = F.desc(L, BEST_FLOOR)
    n = trajectories * STEPS_SOLVER
    res = {}
    for label in ("stage", "floor", "stage", "floor"):
        trace = C.c_void_p()
        L.check(L.lib.dpm_trace_create(n, C.byref(trace)))

        def slot(k, st, b, stream):
            if label == "stage":
                return L.lib.dpm_stage_launch_traced(st, b, stream, trace, k)
            bb = b._obj
            if not F.steady(k) or not bb.h1:
                return 0
            return L.lib.dpm_floor_launch_traced(C.byref(f), bb.x, bb.e0, bb.h1, lp.scratch[0].data_ptr(), lp.scratch[1].data_ptr(),
                                                 lp.nbytes, stream, trace, k)
        try:
            lp.run(slot, trajectories)
            ms = (C.c_float * n)()
            L.check(L.lib.dpm_trace_read(trace, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), ms, n))
        finally:
            L.lib.dpm_trace_destroy(trace)
        us = np.frombuffer(ms, dtype=np.float32).astype(np.float64) * 1e3
        res.setdefault(label, []).append([us[k] for k in range(STEPS_SOLVER, n) if F.steady(k) and us[k] > 0])
    st_us = float(np.median(np.concatenate(res["stage"])))
    fl_us = float(np.median(np.concatenate(res["floor"])))
    return dict(config=F.cfg_id(BEST_FLOOR), floor_kernel_us=round(fl_us, 3), stage_kernel_us=round(st_us, 3),
                frac_of_floor=round(fl_us / st_us, 4),
                how="no-arithmetic kernel (3 read + 2 write streams, the best configuration of the floor sweep, "
                    "profiles/r05_lone_floor.md) and the stage kernel alternately in the slot behind the network's last kernel, "
                    "median event interval of the steady-state launches; frac_of_floor = floor / stage kernel")


def ceilings(L, dev, dtype, R):
    """what the memory system sustains for these streams with no arithmetic at all (3 read + 2 write streams)"""
    ssz = torch.empty((), dtype=dtype).element_size()
    n_el = B * int(np.prod(SHAPE))
    nb = n_el * ssz
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    sets = [[torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(5)] for _ in range(R)]
    cal = {}
    msv = C.c_float()
    for mode in ("warm", "cold"):
        ts = []
        for it in range(24 if mode == "warm" else 3 * R):
            s_ = sets[0] if mode == "warm" else sets[it % R]
            L.check(L.lib.dpm_calib_launch(1, 256, 8, 0 if mode == "warm" else 5, s_[0].data_ptr(), s_[1].data_ptr(), s_[2].data_ptr(),
                                           s_[3].data_ptr(), s_[4].data_ptr(), nb, stream, C.byref(msv)))
            if it >= 8:
                ts.append(msv.value)
        cal[mode] = float(np.mean(ts) * 1e3)
    del sets
    big = [torch.empty(R * nb, dtype=torch.uint8, device=dev) for _ in range(5)]
    ts = []
    for it in range(6):
        L.check(L.lib.dpm_calib_launch(1, 256, 4096, 1, big[0].data_ptr(), big[1].data_ptr(), big[2].data_ptr(),
                                       big[3].data_ptr(), big[4].data_ptr(), R * nb, stream, C.byref(msv)))
        if it >= 2:
            ts.append(msv.value)
    del big
    cal["fused"] = float(np.mean(ts) * 1e3)
    return dict(pattern="3 read + 2 write streams, same bytes, 256-thread workgroups (dpm_calib_launch, lab build)",
                one_request_warm_us=round(cal["warm"], 3), one_request_cold_us=round(cal["cold"], 3),
                fused_size_us=round(cal["fused"], 3),
                fused_size_frac_of_peak=round(5 * nb * R / cal["fused"] / 1e3 / HBM_PEAK_GBS, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32", "bf16"])
    ap.add_argument("--eps-dtype", default=None, choices=["fp16", "fp32", "bf16"])
    ap.add_argument("--loop-net", default="conv", choices=["gemm", "conv", "none"])
    ap.add_argument("--requests", type=int, default=32)
    ap.add_argument("--configs", default="cfg1,cfg3,cfg5,cfg_sd64",
                    help="cases of tools/config_bench.py whose stage kernels are timed inside a conv-network loop (traced launches)")
    args = ap.parse_args()
    import dpm_solver_amd as D
    from dpm_solver_amd import _lib as L
    L.require_lab("tools/lab_secondary.py")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dtype = bench._DT[args.dtype]
    eps_dtype = bench._DT[args.eps_dtype] if args.eps_dtype else dtype
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
    out = {"library": L.LIB_PATH}
    if args.loop_net != "none":
        for kind in ([args.loop_net, "gemm"] if args.loop_net != "gemm" else ["gemm"]):
            try:
                out["in_network_loop"] = in_network_loop(D, L, ns, dev, dtype, kind=kind, net_dtype=eps_dtype)
                out["in_network_loop"]["measured_in_this_run"] = True
                if dtype == eps_dtype:
                    out["in_network_loop"]["floor"] = floor_in_loop(D, L, ns, dev, dtype, kind)
                    out["in_network_loop"]["frac_of_floor"] = out["in_network_loop"]["floor"]["frac_of_floor"]
                break
            except Exception as e:
                out["in_network_loop"] = dict(error="%s: %s" % (type(e).__name__, e))
    if dtype == eps_dtype:
        try:
            out["no_arithmetic_ceiling"] = ceilings(L, dev, dtype, args.requests)
        except Exception as e:
            out["no_arithmetic_ceiling"] = dict(error="%s: %s" % (type(e).__name__, e))
    if args.configs:
        import config_bench
        out["configs_in_loop"] = {}
        for name in args.configs.split(","):
            try:
                out["configs_in_loop"][name] = config_bench.measure_in_loop(name, dev)
            except Exception as e:
                out["configs_in_loop"][name] = dict(error="%s: %s" % (type(e).__name__, e))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
