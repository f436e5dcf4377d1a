#!/usr/bin/env python3
"""Fuzz of the C ABI's native sample loop -- what a host WITHOUT Python drives (include/dpm_hip.h: dpm_plan_run,
dpm_plan_run_multi, dpm_graph_create / dpm_graph_launch) -- against `DPM_Solver.sample()` on the same GPU, over random
plans: method, order, steps, skip type, solver type, algorithm, parameterisation, time range, lower_order_final,
denoise_to_zero, dynamic thresholding, unguided / classifier-free guidance (with and without `dup_state`: the stage
kernel writing the callback's [2B,...] input), fp32 / fp16 / bf16 states, a half-precision network next to an fp32
state, state shapes of 1 to 5 dimensions.  The model callback (ctypes -> the same stand-in torch network) is what a
C host's callback would enqueue.  Three variants per drawn case:

  run        dpm_plan_run with the callback                                   == sample(x)            bit for bit
  multi      dpm_plan_run_multi over R requests with FROZEN network outputs   == R x dpm_plan_run     bit for bit
  graph      dpm_graph_create (capturable callback) + two dpm_graph_launch    == dpm_plan_run         bit for bit

    python tools/fuzz_gpu_capi.py [--cases 1500] [--seed 0] [--out gpurun_out/.../fuzz_gpu_capi.json]
"""
import argparse
import contextlib
import ctypes as C
import faulthandler
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_gpu as FG  # noqa: E402
from fuzz_gpu import D, DT, make_schedule  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402

DEV = "cuda:0"
DCODE = {torch.float32: L.DTYPE_F32, torch.float16: L.DTYPE_F16, torch.bfloat16: L.DTYPE_BF16}


def random_case(rng):
    cfg = FG.random_case(rng)
    cfg["variant"] = str(rng.choice(["run", "run", "multi", "graph"]))
    cfg["dup"] = bool(rng.integers(0, 2))
    cfg["n_req"] = int(rng.integers(2, 7))
    cfg["sdt"] = str(rng.choice(["f32", "f32", "f16", "bf16"]))
    # what the native loop covers: fixed-grid methods, no Python correctors, no classifier (its gradient has no buffer here)
    if cfg["method"] == "adaptive":
        cfg["method"] = str(rng.choice(["multistep", "singlestep"]))
    if cfg["guidance"] == "classifier":
        cfg["guidance"] = "classifier-free"
    cfg["cxt"] = cfg["cx0"] = cfg["ret_inter"] = cfg["noncontig"] = False
    cfg["call"] = "sample"
    cfg["xdt"] = cfg["sdt"]
    if cfg["sdt"] != "f32":
        cfg["net_dt"] = "same"
    elif cfg["net_dt"] == "f32":
        cfg["net_dt"] = "same"
    if cfg["thresholding"] and cfg["algorithm_type"] == "dpmsolver":
        cfg["thresholding"] = False
    return cfg


def same(a, b):
    """bit-for-bit up to NaNs (a half state that overflows: NaN on both sides)"""
    return a.shape == b.shape and a.dtype == b.dtype and bool(((a == b) | (a.isnan() & b.isnan())).all())


def one_case(cfg):
    """(status, detail): 'ok' / 'raise' (sample() raised: nothing to drive) / 'bad'"""
    g = torch.Generator().manual_seed(cfg["seed"])
    sdt = DT[cfg["sdt"]]
    shape = tuple(cfg["shape"])
    x = torch.randn(shape, generator=g).to(sdt).to(DEV)
    ns = make_schedule(cfg["schedule"])
    dpm = FG.build(ns, cfg, x, None, solver_kwargs=(dict(state_dtype=sdt) if sdt is not torch.float32 else None))
    kw = dict(steps=cfg["steps"], order=cfg["order"], method=cfg["method"], skip_type=cfg["skip_type"],
              solver_type=cfg["solver_type"], lower_order_final=cfg["lower_order_final"], denoise_to_zero=cfg["denoise_to_zero"])
    try:
        want = dpm.sample(x, t_start=cfg["t_start"], t_end=cfg["t_end"], **kw)
    except Exception as e:                              # noqa: BLE001
        return "raise", "%s: %s" % (type(e).__name__, str(e)[:80])
    if want.dtype is not sdt:
        return "raise", "result dtype %s (promoted): not a plan the native loop is handed" % want.dtype
    t_T = float(ns.T if cfg["t_start"] is None else cfg["t_start"])
    t_0 = float(1. / ns.total_N if cfg["t_end"] is None else cfg["t_end"])
    skw = dict(kw)
    if skw["solver_type"] not in L.SOLVER:
        skw["solver_type"] = "dpmsolver"                 # sample() ran: the run holds no update that reads it
    if cfg["method"] == "singlestep_fixed" and cfg["order"] not in (1, 2, 3):
        return "raise", "no-op plan"
    plan = dpm._get_plan(precision=0, t_T=t_T, t_0=t_0, **skw)
    n_st = len(plan.stages)
    if n_st == 0:
        return "raise", "empty plan"
    B = shape[0]
    wrapped = dpm._wrapped
    cfg_on = wrapped.effective_guidance == "classifier-free"
    dup = bool(cfg["dup"] and cfg_on)
    net_dt = sdt if cfg["net_dt"] == "same" else DT[cfg["net_dt"]]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    R = cfg["n_req"] if cfg["variant"] == "multi" else 1

    def buffers(x_T):
        full = (2 * B,) + shape[1:] if dup else shape
        xb = [torch.cat([x_T, x_T]) if dup else x_T.clone()] + [torch.empty(full, dtype=sdt, device=DEV) for _ in range(3)]
        hb = [torch.empty(shape, dtype=sdt, device=DEV) for _ in range(3)]
        out2 = torch.empty(((2 * B,) + shape[1:]) if cfg_on else shape, dtype=net_dt, device=DEV)
        rb = L.RunBuffers()
        for i in range(4):
            rb.xbuf[i] = xb[i].data_ptr()
        for i in range(3):
            rb.hist[i] = hb[i].data_ptr()
        if cfg_on:
            rb.e1, rb.e0 = out2[:B].data_ptr(), out2[B:].data_ptr()
        else:
            rb.e0 = out2.data_ptr()
        rb.n, rb.batch, rb.state_dtype, rb.eps_dtype = x_T.numel(), B, DCODE[sdt], DCODE[net_dt]
        rb.dup_state = 1 if dup else 0
        ws = None
        if cfg["thresholding"]:
            nb = L.lib.dpm_threshold_workspace_bytes(B, x_T.numel() // B)
            if nb:
                ws = torch.zeros(nb, dtype=torch.uint8, device=DEV)
                rb.workspace = ws.data_ptr()
        return rb, xb, hb, out2, ws

    # the callback a C host would write: enqueue-only, no allocation (everything it touches exists before the run -- the same
    # arithmetic as FG.build's stand-in network: fp32 products, one rounding to the state's dtype, one to the network's)
    cond_on = cfg["guidance"] == "classifier-free"
    nb_ = 2 * B if cfg_on else B
    bshape = (nb_,) + (1,) * (len(shape) - 1)
    f_time = [(torch.full((nb_,), st.t_input, dtype=torch.float32, device=DEV) * 0.0005 + 0.25).reshape(bshape) for st in plan.stages]
    f_cond = None
    if cond_on:
        cvec = torch.cat([wrapped.unconditional_condition, wrapped.condition]) if cfg_on else wrapped.condition
        f_cond = (cvec.to(torch.float32) * 0.1 + 1.0).reshape(bshape)
    full_shape = (nb_,) + shape[1:]

    def make_cb(xb, out2, log=None):
        by_ptr = {t.data_ptr(): t for t in xb}
        tmp32 = torch.empty(full_shape, dtype=torch.float32, device=DEV)
        mid = torch.empty(full_shape, dtype=sdt, device=DEV)

        def cb(user, st, xptr, e0, e1, strm):
            xin = by_ptr[xptr]
            i = st.contents.index
            if cfg_on and not dup:
                tmp32[:B].copy_(xin)
                tmp32[B:].copy_(xin)
            else:
                tmp32.copy_(xin)
            tmp32.mul_(f_time[i])
            if f_cond is not None:
                tmp32.mul_(f_cond)
            mid.copy_(tmp32)
            out2.copy_(mid)
            if log is not None:
                log.append(i)
            return 0
        return L.MODEL_CB(cb)

    res = C.c_int(-1)
    rb, xb, hb, out2, ws = buffers(x)
    log = []
    cb = make_cb(xb, out2, log)
    rc = L.lib.dpm_plan_run(plan.handle, C.byref(rb), C.cast(cb, C.c_void_p), None, stream, C.byref(res))
    if rc:
        return "bad", "dpm_plan_run rc %d: %s" % (rc, L.lib.dpm_last_error().decode("utf-8", "replace"))
    torch.cuda.synchronize()
    got = xb[res.value][:B] if dup else xb[res.value]
    if log != list(range(n_st)):
        return "bad", "callback order %s for %d stages" % (log[:8], n_st)
    if not same(got, want):
        d = float((got.double() - want.double()).abs().max()) / (float(want.double().abs().max()) or 1.0)
        return "bad", "dpm_plan_run != sample(): %.3g of the peak" % d
    if not same(xb[0][:B] if dup else xb[0], x):
        return "bad", "dpm_plan_run wrote the caller's x_T"
    if cfg["variant"] == "multi":
        # frozen outputs: every request keeps ONE output tensor for all stages (the HBM-cold mode of bench.py); per request
        # dpm_plan_run without a callback is the yardstick
        xs = [torch.randn(shape, generator=g).to(sdt).to(DEV) for _ in range(R)]
        sets = [buffers(xr) for xr in xs]
        for (rbr, xbr, hbr, o2, wsr), xr in zip(sets, xs):
            o2.copy_((torch.cat([xr, xr]) if cfg_on else xr).float().mul(0.31).add(0.01).to(net_dt))
        singles = []
        for rbr, xbr, hbr, o2, wsr in sets:
            rc = L.lib.dpm_plan_run(plan.handle, C.byref(rbr), None, None, stream, C.byref(res))
            if rc:
                return "bad", "dpm_plan_run (frozen) rc %d" % rc
            torch.cuda.synchronize()
            singles.append((xbr[res.value][:B] if dup else xbr[res.value]).clone())
        rbs = (L.RunBuffers * R)()
        for r in range(R):
            C.memmove(C.byref(rbs, r * C.sizeof(L.RunBuffers)), C.byref(sets[r][0]), C.sizeof(L.RunBuffers))
        resm = (C.c_int * R)()
        rc = L.lib.dpm_plan_run_multi(plan.handle, rbs, R, stream, None, resm)
        if rc:
            return "bad", "dpm_plan_run_multi rc %d: %s" % (rc, L.lib.dpm_last_error().decode("utf-8", "replace"))
        torch.cuda.synchronize()
        for r in range(R):
            gm = sets[r][1][resm[r]][:B] if dup else sets[r][1][resm[r]]
            if not same(gm, singles[r]):
                return "bad", "dpm_plan_run_multi request %d of %d != dpm_plan_run" % (r, R)
    if cfg["variant"] == "graph":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sp = C.c_void_p(side.cuda_stream)
            rb2, xb2, hb2, o22, ws2 = buffers(x)
            cb2 = make_cb(xb2, o22)
            for _ in range(2):                          # warm: allocator pools, torch kernels
                L.lib.dpm_plan_run(plan.handle, C.byref(rb2), C.cast(cb2, C.c_void_p), None, sp, C.byref(res))
            side.synchronize()
            gh = C.c_void_p()
            rc = L.lib.dpm_graph_create(plan.handle, C.byref(rb2), C.cast(cb2, C.c_void_p), None, sp, C.byref(gh))
            if rc:
                return "bad", "dpm_graph_create rc %d: %s" % (rc, L.lib.dpm_last_error().decode("utf-8", "replace"))
            try:
                for rep in range(2):
                    x_new = x if rep == 0 else (x.float() * 0.5 + 0.25).to(sdt)
                    (xb2[0][:B] if dup else xb2[0]).copy_(x_new)
                    if dup:
                        xb2[0][B:].copy_(x_new)
                    rc = L.lib.dpm_graph_launch(gh, sp)
                    if rc:
                        return "bad", "dpm_graph_launch rc %d" % rc
                    side.synchronize()
                    gg = xb2[L.lib.dpm_graph_result(gh)]
                    gg = gg[:B] if dup else gg
                    if rep == 0 and not same(gg, want):
                        return "bad", "dpm_graph_launch != sample()"
                    if rep == 1:
                        rb3, xb3, hb3, o23, ws3 = buffers(x_new)
                        cb3 = make_cb(xb3, o23)
                        L.lib.dpm_plan_run(plan.handle, C.byref(rb3), C.cast(cb3, C.c_void_p), None, sp, C.byref(res))
                        side.synchronize()
                        w3 = xb3[res.value][:B] if dup else xb3[res.value]
                        if not same(gg, w3):
                            return "bad", "second dpm_graph_launch (new x_T) != dpm_plan_run"
            finally:
                L.lib.dpm_graph_destroy(gh)
        torch.cuda.current_stream().wait_stream(side)
    return "ok", ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--case-timeout", type=int, default=60)
    ap.add_argument("--only", type=int, default=None)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    cfgs = [random_case(rng) for _ in range(args.cases)]
    idx = range(args.cases) if args.only is None else [args.only]
    cur = (os.path.splitext(args.out)[0] if args.out else "/tmp/fuzz_gpu_capi") + "_current_case.txt"
    per = {}
    n_bad = 0
    t0 = time.perf_counter()
    for i in idx:
        cfg = cfgs[i]
        with open(cur, "w") as f:
            f.write("%d %s\n" % (i, cfg))
        faulthandler.dump_traceback_later(args.case_timeout, exit=True, file=sys.__stderr__)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                status, detail = one_case(cfg)
        except Exception as e:                          # noqa: BLE001
            import traceback
            status, detail = "bad", "tool / library exception %s: %s\n%s" % (type(e).__name__, str(e)[:200], traceback.format_exc(limit=4))
        faulthandler.cancel_dump_traceback_later()
        a = per.setdefault(cfg["variant"], dict(cases=0, driven=0, not_driven=0, disagreements=0))
        a["cases"] += 1
        a["driven"] += status == "ok"
        a["not_driven"] += status == "raise"
        if args.only is not None:
            print(status, detail)
        if status == "bad":
            a["disagreements"] += 1
            n_bad += 1
            print("case %d: %s\n    %s" % (i, {k: v for k, v in cfg.items() if k != "seed"}, detail), flush=True)
    os.remove(cur)
    rec = dict(cases=len(list(idx)), seed=args.seed, disagreements=n_bad, per_variant=per, seconds=round(time.perf_counter() - t0, 1),
               device=torch.cuda.get_device_name(0),
               what="the C ABI's native sample loop (dpm_plan_run / dpm_plan_run_multi / dpm_graph_*) with a model callback vs "
                    "DPM_Solver.sample() on the same GPU, bit for bit; not_driven = sample() raised or promoted the state")
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
