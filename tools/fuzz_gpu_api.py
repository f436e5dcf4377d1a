#!/usr/bin/env python3
"""tools/fuzz_gpu.py for the EXTENSION entry points: every random sample() configuration is run on the MI355X through one of
the engine's GPU-side extensions and compared with the plain `sample()` of the engine's host code on the numpy double of the
stage kernel (tests/kernel_double.py; the CPU suite holds that to the live reference):

  requests      sample_requests([x_0 .. x_R-1])  (one fused launch per stage)          == [sample(x_r)]
  capture       g = capture(x, ...); g(x); g(x')  (hipGraph replay, network included)    == sample(x), sample(x')
  auto_capture  auto_capture = 1, three calls (eager, eager, replay)                     == sample(x) three times
  nhwc          a network that answers in channels_last                                  == the same network, default layout
  stream        the call on a non-default stream                                         == sample(x)
  state_half    DPM_Solver(state_dtype=fp16 / bf16)                                      == the same on the double
  maskblend     correcting_xt_fn = MaskBlend(...) folded into the stage kernel           == the same on the double
  device_adapt  method='adaptive' with the controller on the device                      ~= the host loop on the double

fp32 / double results are expected bit-identical except device_adapt (another sequence of accept / reject decisions is
possible: tolerance).  Needs no reference checkout.

    python tools/fuzz_gpu_api.py [--cases 1200] [--seed 0] [--out gpurun_out/.../fuzz_gpu_api.json]
"""
import argparse
import contextlib
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
from fuzz_gpu import D, S, DT, make_schedule, install_cpu_double  # noqa: E402

APIS = ["requests", "requests", "capture", "capture", "auto_capture", "nhwc", "stream", "state_half", "maskblend", "device_adapt"]


def random_case(rng):
    cfg = FG.random_case(rng)
    cfg["api"] = str(rng.choice(APIS))
    cfg["n_req"] = int(rng.integers(2, 6))
    cfg["half"] = str(rng.choice(["f16", "bf16"]))
    cfg["mask_kind"] = int(rng.integers(0, 3))
    api = cfg["api"]
    # what the variant needs to take its own path (everything else stays as drawn)
    if api in ("capture", "auto_capture"):
        cfg["cxt"] = cfg["cx0"] = False                 # Python callbacks are never captured
        cfg["ret_inter"] = False
        if cfg["method"] == "adaptive":
            cfg["method"] = "multistep"
        if cfg["guidance"] == "classifier":              # autograd inside a stream capture: not what this tool is about
            cfg["guidance"] = "uncond"
    if api == "nhwc":
        cfg["shape"] = [(2, 3, 4, 4), (1, 3, 4, 4), (3, 3, 32, 32), (5, 4, 16, 16)][int(rng.integers(0, 4))]
        cfg["noncontig"] = False
    if api in ("requests", "capture"):
        cfg["call"] = "sample"                           # (both take sample()'s arguments)
    if api == "device_adapt":
        cfg["method"] = "adaptive"
        cfg["order"] = int(rng.integers(2, 4))
        cfg["thresholding"] = cfg["cx0"] = cfg["cxt"] = False
        cfg["ret_inter"] = False
        cfg["call"] = "sample"
        if cfg["xdt"] != "f32":
            cfg["xdt"] = "f32"
    if api == "maskblend":
        cfg["cxt"] = False
        if cfg["method"] == "adaptive":
            cfg["method"] = "singlestep"
        cfg["shape"] = [(2, 3, 4, 4), (1, 3, 4, 4), (3, 3, 32, 32), (5, 4, 16, 16), (2, 3, 4)][int(rng.integers(0, 5))]
        if cfg["xdt"] == "f64":
            cfg["xdt"] = "f32"
    if api == "state_half":
        if cfg["xdt"] == "f64":
            cfg["xdt"] = "f32"
        if cfg["method"] == "adaptive":
            cfg["method"] = "multistep"
    if cfg["thresholding"] and cfg["algorithm_type"] == "dpmsolver":
        cfg["thresholding"] = False
    if cfg["thresholding"] and cfg["xdt"] in ("f16", "bf16") and cfg["schedule"] == "vp_linear":
        cfg["thresholding"] = False
    return cfg


def inputs(cfg, device):
    g = torch.Generator().manual_seed(cfg["seed"])
    n = cfg["n_req"] if cfg["api"] == "requests" else (2 if cfg["api"] == "capture" else 1)
    xs = []
    for _ in range(n):
        x = torch.randn(cfg["shape"], generator=g).to(DT[cfg["xdt"]])
        if cfg["noncontig"] and x.dim() >= 2:
            x = x.transpose(0, 1).contiguous().transpose(0, 1)
        x = x.to(device)
        if cfg["noncontig"] and x.dim() >= 2 and x.is_contiguous():
            x = x.transpose(0, 1).contiguous().transpose(0, 1)
        xs.append(x)
    extra = {}
    if cfg["api"] == "maskblend":
        shp = tuple(cfg["shape"])
        mshape = [shp[-2:], shp[1:], shp][cfg["mask_kind"]] if len(shp) >= 3 else shp
        extra["mask"] = (torch.rand(mshape, generator=g) > 0.5).float().to(device)
        extra["x0"] = torch.randn(shp, generator=g).to(device)
        extra["noise"] = torch.randn(shp, generator=g).to(device)
    return xs, extra


def solver(cfg, x, trace, extra, on_gpu):
    ns = make_schedule(cfg["schedule"])
    api = cfg["api"]
    quiet = api in ("capture", "auto_capture")          # the tracing stand-in network reads its time on the host: not capturable
    dpm = FG.build(ns, cfg, x, trace if not quiet else None,
                   solver_kwargs=dict(state_dtype=DT[cfg["half"]]) if api == "state_half" else None)
    if api == "maskblend":
        dpm.correcting_xt_fn = D.MaskBlend(ns, extra["mask"], x0=extra["x0"], noise=extra["noise"])
    if api == "device_adapt":
        dpm.adaptive_on_device = bool(on_gpu)
    return dpm


def run(cfg, device):
    on_gpu = device != "cpu"
    xs, extra = inputs(cfg, device)
    trace = []
    api = cfg["api"]
    try:
        dpm = solver(cfg, xs[0], trace, extra, on_gpu)
        kw = dict(steps=cfg["steps"], order=cfg["order"], method=cfg["method"], skip_type=cfg["skip_type"],
                  solver_type=cfg["solver_type"], lower_order_final=cfg["lower_order_final"],
                  denoise_to_zero=cfg["denoise_to_zero"], return_intermediate=cfg["ret_inter"], t_start=cfg["t_start"], t_end=cfg["t_end"])
        if cfg["method"] == "adaptive":
            kw.update(atol=0.05, rtol=0.1)
        if cfg["call"] == "inverse" and cfg["method"] != "adaptive":
            kw["t_start"], kw["t_end"] = cfg["t_end"], cfg["t_start"]
            call = dpm.inverse
        else:
            call = dpm.sample
        if not on_gpu or api in ("state_half", "maskblend", "device_adapt"):
            outs = [call(x, **kw) for x in xs]
            if api == "auto_capture":
                outs = outs * 3
        elif api == "requests":
            outs = dpm.sample_requests(xs, **kw)
        elif api == "capture":
            kwc = dict(kw)
            g = dpm.capture(xs[0], **kwc)
            outs = [g(x).clone() for x in xs]
        elif api == "auto_capture":
            dpm.auto_capture = 1
            outs = [call(xs[0], **kw) for _ in range(3)]
        elif api == "nhwc":
            outs = [call(x, **kw) for x in xs]
        elif api == "stream":
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs = [call(x, **kw) for x in xs]
            torch.cuda.current_stream().wait_stream(st)
        else:
            raise AssertionError(api)
        flat = []
        for o in outs:
            if isinstance(o, tuple):
                flat.append(o[0].cpu())
                flat.extend(t.cpu() for t in o[1])
            else:
                flat.append(o.cpu())
        return ("ok", (flat[0], flat[1:]), trace)
    except Exception as e:                              # noqa: BLE001
        import traceback
        return ("raise", (type(e).__name__, str(e)[:160]), trace, traceback.format_exc(limit=6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--case-timeout", type=int, default=60)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--debug-case", type=int, default=None, help="replay ONE case and print where the two runs part")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    cfgs = [random_case(rng) for _ in range(max(args.cases, (args.debug_case or 0) + 1))]
    if args.debug_case is not None:
        cfg = cfgs[args.debug_case]
        print({k: v for k, v in cfg.items() if k != "seed"})
        FG.RECORD = []
        if cfg["api"] == "nhwc":
            cfg["_nhwc"] = True
        g = run(cfg, args.device)
        cfg.pop("_nhwc", None)
        rg, FG.RECORD = FG.RECORD, []
        install_cpu_double(FG._MP(), S, D)
        c = run(cfg, "cpu")
        rc = FG.RECORD
        print("GPU:", g[0], g[1] if g[0] == "raise" else "", "double:", c[0], "boundary tensors:", len(rg), len(rc))
        for k, ((wa, a), (wb, b)) in enumerate(zip(rg, rc)):
            same = wa == wb and a.dtype == b.dtype and a.shape == b.shape and bool(torch.equal(a, b))
            d = float((a.double() - b.double()).abs().max()) if a.shape == b.shape and a.numel() else float("nan")
            print("%3d %-26s %-10s %-10s %s max|d| %.3g of %.3g %s" % (k, wa, str(a.dtype)[6:], str(b.dtype)[6:], "same" if same else "DIFFERENT",
                  d, float(b.double().abs().max()) if b.numel() else 0.0, "" if wa == wb else "(vs %s)" % wb))
        if g[0] == "ok" and c[0] == "ok":
            for k, (a, b) in enumerate(zip([g[1][0]] + g[1][1], [c[1][0]] + c[1][1])):
                print("result %d" % k, a.dtype, b.dtype, "equal" if torch.equal(a, b) else "max|d| %.3g of %.3g" % (float((a.double() - b.double()).abs().max()), float(b.double().abs().max())))
        return 0
    if args.only is not None:
        cfgs = [cfgs[args.only]]
    if args.device == "cpu":
        install_cpu_double(FG._MP(), S, D)
    cur = (os.path.splitext(args.out)[0] if args.out else "/tmp/fuzz_gpu_api") + "_current_case.txt"
    t0 = time.perf_counter()
    gpu = []
    with contextlib.redirect_stdout(io.StringIO()):
        for i, cfg in enumerate(cfgs):
            with open(cur, "w") as f:
                f.write("%d %s\n" % (i, cfg))
            faulthandler.dump_traceback_later(args.case_timeout, exit=True, file=sys.__stderr__)
            if cfg["api"] == "nhwc":
                cfg["_nhwc"] = True
            gpu.append(run(cfg, args.device))
            faulthandler.cancel_dump_traceback_later()
    os.remove(cur)
    if args.device != "cpu":
        torch.cuda.synchronize()
        install_cpu_double(FG._MP(), S, D)
    t_gpu = time.perf_counter() - t0
    torch.set_num_threads(1)
    per_api = {}
    n_bad = 0
    for i, (cfg, g) in enumerate(zip(cfgs, gpu)):
        cfg.pop("_nhwc", None)
        with contextlib.redirect_stdout(io.StringIO()):
            c = run(cfg, "cpu")
        cmp_cfg = dict(cfg)
        if cfg["api"] in ("capture", "auto_capture"):
            g, c = g[:2] + ([],) + g[3:], c[:2] + ([],) + c[3:]          # no trace on the captured side
        if cfg["api"] == "requests":
            g, c = g[:2] + (sorted(g[2]),) + g[3:], c[:2] + (sorted(c[2]),) + c[3:]  # stage-major vs request-major call order
        if cfg["api"] == "device_adapt":
            g, c = g[:2] + ([],) + g[3:], c[:2] + ([],) + c[3:]          # the device controller looks ahead: more network calls
        bad, worst, same = FG.compare(cmp_cfg, g, c)
        a = per_api.setdefault(cfg["api"], dict(cases=0, returned=0, bit_identical=0, disagreements=0, worst=0.0))
        a["cases"] += 1
        if g[0] == "ok":
            a["returned"] += 1
            a["bit_identical"] += bool(same)
            a["worst"] = max(a["worst"], worst)
        if bad:
            a["disagreements"] += 1
            n_bad += 1
            print("case %d: %s\n    %s" % (i if args.only is None else args.only, {k: v for k, v in cfg.items() if k != "seed"}, "\n    ".join(bad)), flush=True)
            if g[0] == "raise" and c[0] != "raise" and len(g) > 3:
                print("    " + g[3].replace("\n", "\n    "))
    rec = dict(cases=len(cfgs), seed=args.seed, disagreements=n_bad, per_api=per_api, gpu_seconds=round(t_gpu, 1),
               device=(torch.cuda.get_device_name(0) if args.device != "cpu" else "cpu (self-check)"),
               what="the engine's GPU-side extensions on the MI355X vs plain sample() of the engine's host code on the numpy double")
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
