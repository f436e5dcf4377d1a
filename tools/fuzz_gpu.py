#!/usr/bin/env python3
"""The drop-in fuzz carried to the hardware: random cases of the kind tools/fuzz_dropin.py draws (the same generator plus
larger shapes and networks that answer in another dtype) run TWICE through the engine -- once on the MI355X through libdpm_hip.so (the product kernels), once on CPU tensors with
the stage kernel replaced by its numpy double (tests/kernel_double.py) -- and the two must agree: raised or returned, the
exception, dtype, shape, the network-call trace, and the values (the same host code drives both; what differs is the kernel,
so fp32 / double results are expected to be BIT-identical; the tolerances below only catch what is not).

tools/fuzz_dropin.py (build container: needs /root/reference) shows  reference == engine-with-the-double  over these cases;
this tool (GPU box: needs no reference) shows  engine-with-the-double == engine-on-the-GPU  over the same cases.

    python tools/fuzz_gpu.py [--cases 1500] [--seed 0] [--out gpurun_out/.../fuzz_gpu.json]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
os.environ.setdefault("DPM_REFERENCE_DIR", "/nonexistent")

import cases as C  # noqa: E402,F401
import dpm_solver_amd as D  # noqa: E402
import dpm_solver_amd.solver as S  # noqa: E402
from engine_cases import make_schedule  # noqa: E402
from kernel_double import install_cpu_double  # noqa: E402

SHAPES = [(3,), (2, 5), (2, 3, 4), (2, 3, 4, 4), (1, 3, 4, 4), (1, 2, 3, 2, 2), (4, 1, 1, 1), (2, 12), (3, 3, 32, 32), (5, 4, 16, 16)]
DT = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}


def random_case(rng):
    """tools/fuzz_dropin.py's generator plus two larger shapes and the network's dtype"""
    method = str(rng.choice(["multistep", "singlestep", "singlestep_fixed", "adaptive"], p=[0.4, 0.3, 0.2, 0.1]))
    order = int(rng.choice([1, 2, 3, 3, 2, 4, 0])) if rng.random() < 0.15 else int(rng.integers(1, 4))
    steps = int(rng.integers(1, 13))
    guidance = str(rng.choice(["uncond", "uncond", "classifier-free", "classifier"]))
    return dict(method=method, order=order, steps=steps, shape=SHAPES[int(rng.integers(0, len(SHAPES)))],
                schedule=str(rng.choice(["sd", "ddpm", "vp_linear", "cosine1000"])),
                skip_type=str(rng.choice(["time_uniform", "logSNR", "time_quadratic", "bogus"], p=[0.4, 0.3, 0.27, 0.03])),
                solver_type=str(rng.choice(["dpmsolver", "taylor", "bogus"], p=[0.55, 0.42, 0.03])),
                algorithm_type=str(rng.choice(["dpmsolver++", "dpmsolver"])),
                model_type=str(rng.choice(["noise", "x_start", "v", "score"])), guidance=guidance,
                scale=float(rng.choice([1.0, 2.5, 7.5])), thresholding=bool(rng.integers(0, 4) == 0),
                cxt=bool(rng.integers(0, 5) == 0), cx0=bool(rng.integers(0, 6) == 0),
                lower_order_final=bool(rng.integers(0, 2)), denoise_to_zero=bool(rng.integers(0, 3) == 0),
                t_end=(None if rng.random() < 0.5 else float(rng.choice([1e-3, 1e-2, 0.05]))),
                t_start=(None if rng.random() < 0.5 else float(rng.choice([1.0, 0.8, 0.5]))),
                call=str(rng.choice(["sample", "sample", "sample", "inverse"])),
                ret_inter=bool(rng.integers(0, 2)), xdt=str(rng.choice(["f32", "f32", "f32", "f64", "f16", "bf16"])),
                noncontig=bool(rng.integers(0, 6) == 0), net_dt=str(rng.choice(["same", "same", "same", "f16", "f32"])),
                seed=int(rng.integers(0, 1 << 30)))


RECORD = None          # --debug-case: list receiving (what, tensor on the host) at the network / callback boundary


def _rec(what, t):
    if RECORD is not None:
        RECORD.append((what, t.detach().cpu().clone()))


def build(ns, cfg, x, trace, solver_kwargs=None):
    """trace: list receiving one record per network call, or None (a network without host reads: capturable)"""
    B = x.shape[0]
    dev = x.device
    cond = torch.arange(1, B + 1, dtype=torch.float32, device=dev) * 0.5
    kw = dict(model_type=cfg["model_type"], guidance_type=cfg["guidance"], guidance_scale=cfg["scale"])
    ndt = None if cfg["net_dt"] == "same" else DT[cfg["net_dt"]]

    def base(xx, t, c=None):
        if trace is not None:
            trace.append((tuple(xx.shape), str(xx.dtype), str(t.dtype), tuple(t.shape), round(float(t.reshape(-1)[0]), 4)))
        _rec("network input", xx)
        _rec("network time", t)
        # (half inputs: arithmetic in fp32, ONE rounding -- torch's half operations with Python scalars do not round alike on
        # the CPU and on the GPU, and this tool compares a CPU run with a GPU run)
        wd = xx.dtype if xx.dtype in (torch.float32, torch.float64) else torch.float32
        tt = t.to(wd).reshape((-1,) + (1,) * (xx.dim() - 1))
        out = xx.to(wd) * (tt * 0.0005 + 0.25)
        if c is not None:
            out = out * (c.to(wd).reshape((-1,) + (1,) * (xx.dim() - 1)) * 0.1 + 1.0)
        out = out.to(xx.dtype)
        # (a network that answers in its own dtype: Stable Diffusion under autocast hands fp16 to an fp32 state)
        out = out if (ndt is None or xx.dtype is torch.float64) else out.to(ndt)
        if cfg.get("_nhwc") and out.dim() == 4:          # (tools/fuzz_gpu_api.py: a network that works in channels_last)
            out = out.contiguous(memory_format=torch.channels_last)
        _rec("network output", out)
        return out
    if cfg["guidance"] == "classifier-free":
        net = lambda xx, t, c: base(xx, t, c)
        kw.update(condition=cond, unconditional_condition=torch.zeros(B, device=dev))
    elif cfg["guidance"] == "classifier":
        net = lambda xx, t, c=None: base(xx, t)
        kw.update(condition=cond, classifier_fn=lambda xx, t, c: -0.5 * (xx.reshape(xx.shape[0], -1) ** 2).sum(dim=1) * 0.01)
    else:
        net = lambda xx, t: base(xx, t)
    fn = D.model_wrapper(net, ns, **kw)
    skw = dict(algorithm_type=cfg["algorithm_type"])
    if cfg["thresholding"]:
        skw["correcting_x0_fn"] = "dynamic_thresholding"
    elif cfg["cx0"]:
        def cx0(x0, t):
            _rec("correcting_x0_fn input", x0)
            return torch.clamp(x0, -2.0, 2.0)
        skw["correcting_x0_fn"] = cx0
    if cfg["cxt"]:
        def cxt(xt, t, step):
            _rec("correcting_xt_fn input", xt)
            out = (xt.to(torch.float32 if xt.dtype is not torch.float64 else xt.dtype) * 0.99 + 0.001 * step).to(xt.dtype)
            _rec("correcting_xt_fn output", out)
            return out
        skw["correcting_xt_fn"] = cxt
    skw.update(solver_kwargs or {})
    dpm = D.DPM_Solver(fn, ns, **skw)
    dpm.adaptive_on_device = False          # the reference's host loop on both sides: the same sequence of launches
    return dpm


def run(cfg, device):
    g = torch.Generator().manual_seed(cfg["seed"])
    x = torch.randn(cfg["shape"], generator=g).to(DT[cfg["xdt"]])
    if cfg["noncontig"] and x.dim() >= 2:
        x = x.transpose(0, 1).contiguous().transpose(0, 1)
    x = x.to(device)
    if cfg["noncontig"] and x.dim() >= 2 and x.is_contiguous():
        x = x.transpose(0, 1).contiguous().transpose(0, 1)
    trace = []
    try:
        dpm = build(make_schedule(cfg["schedule"]), cfg, x, trace)
        kw = dict(steps=cfg["steps"], order=cfg["order"], method=cfg["method"], skip_type=cfg["skip_type"],
                  solver_type=cfg["solver_type"], lower_order_final=cfg["lower_order_final"],
                  denoise_to_zero=cfg["denoise_to_zero"], return_intermediate=cfg["ret_inter"])
        if cfg["method"] == "adaptive":
            kw.update(atol=0.05, rtol=0.1)
        if cfg["call"] == "sample" or cfg["method"] == "adaptive":
            out = dpm.sample(x, t_start=cfg["t_start"], t_end=cfg["t_end"], **kw)
        else:
            out = dpm.inverse(x, t_start=cfg["t_end"], t_end=cfg["t_start"], **kw)
        if isinstance(out, tuple):
            out = (out[0].cpu(), [t.cpu() for t in out[1]])
        else:
            out = out.cpu()
        return ("ok", out, trace)
    except Exception as e:                              # noqa: BLE001 -- the comparison is about what is raised
        return ("raise", (type(e).__name__, str(e)[:160]), trace)


def compare(cfg, g, c):
    """(disagreements, largest deviation of the values as a fraction of the peak, bit-identical?)"""
    if g[0] != c[0]:
        return ["GPU %s, double %s: %s | %s" % (g[0], c[0], g[1] if g[0] == "raise" else "", c[1] if c[0] == "raise" else "")], 0.0, False
    if g[0] == "raise":
        return ([] if g[1] == c[1] else ["exception %s vs %s" % (g[1], c[1])]), 0.0, True
    go, co = g[1], c[1]
    gi, ci = [], []
    if isinstance(go, tuple):
        go, gi = go
        co, ci = co
    bad = []
    if go.dtype != co.dtype or tuple(go.shape) != tuple(co.shape):
        return ["result %s %s vs %s %s" % (go.dtype, tuple(go.shape), co.dtype, tuple(co.shape))], 0.0, False
    if len(gi) != len(ci):
        return ["%d vs %d intermediates" % (len(gi), len(ci))], 0.0, False
    if not bool(torch.isfinite(co.double()).all()):
        return [], 0.0, bool(torch.equal(torch.isfinite(go), torch.isfinite(co)))
    peak = max([float(co.double().abs().max())] + [float(t.double().abs().max()) for t in ci]) or 1.0
    worst, same = 0.0, True
    for a, b in zip([go] + gi, [co] + ci):
        if a.dtype != b.dtype:
            bad.append("intermediate dtype %s vs %s" % (a.dtype, b.dtype))
            break
        same = same and bool(torch.equal(a, b))
        worst = max(worst, float((a.double() - b.double()).abs().max()) / peak)
    half = go.dtype in (torch.float16, torch.bfloat16)
    # adaptive: the accept / reject decisions read an error norm whose reduction order differs between the device kernel and numpy
    tol = (4e-3 if go.dtype is torch.float16 else 3e-2) if half else (1e-12 if go.dtype is torch.float64 else 2e-6)
    if cfg["method"] == "adaptive":
        tol = max(tol, 1e-4)
    if worst > tol:
        bad.append("values: %.3g of the peak (tolerance %.2g)" % (worst, tol))
    gt, ct = g[2], c[2]
    if len(gt) != len(ct):
        bad.append("network calls %d vs %d" % (len(gt), len(ct)))
    else:
        for k, (a, b) in enumerate(zip(gt, ct)):
            if a[:4] != b[:4] or abs(a[4] - b[4]) > 2e-4:
                bad.append("network call %d: %s vs %s" % (k, a, b))
                break
    return bad, worst, same


class _MP:
    def __init__(self):
        self.undo = []

    def setattr(self, o, n, v):
        self.undo.append((o, n, getattr(o, n)))
        setattr(o, n, v)


def debug_case(args):
    """the tensors at the network / callback boundary of both runs, in order: the first that differs says whether a stage kernel
    (an input differs after equal outputs) or a torch operation of the stand-in network (an output differs for equal inputs)"""
    global RECORD
    rng = np.random.default_rng(args.seed)
    for _ in range(args.debug_case + 1):
        cfg = random_case(rng)
    if cfg["thresholding"] and cfg["algorithm_type"] == "dpmsolver":
        cfg["thresholding"] = False
    print({k: v for k, v in cfg.items() if k != "seed"})
    RECORD = []
    g = run(cfg, args.device)
    rg, RECORD = RECORD, []
    install_cpu_double(_MP(), S, D)
    c = run(cfg, "cpu")
    rc = RECORD
    print("GPU:", g[0], "double:", c[0], "boundary tensors:", len(rg), len(rc))
    for k, ((wa, a), (wb, b)) in enumerate(zip(rg, rc)):
        same = wa == wb and a.dtype == b.dtype and a.shape == b.shape and bool(torch.equal(a, b))
        d = float((a.double() - b.double()).abs().max()) if a.shape == b.shape and a.numel() else float("nan")
        print("%3d %-26s %-15s %-15s %s max|d| %.3g of %.3g" % (k, wa, str(a.dtype)[6:], str(b.dtype)[6:], "same" if same else "DIFFERENT", d,
                                                               float(b.double().abs().max()) if b.numel() else 0.0))
    bad, worst, same = compare(cfg, g, c)
    print(bad, worst, same)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--case-timeout", type=int, default=60)
    ap.add_argument("--debug-case", type=int, default=None, help="replay ONE case and print where the two runs part")
    args = ap.parse_args()
    if args.debug_case is not None:
        return debug_case(args)
    rng = np.random.default_rng(args.seed)
    cfgs = []
    for _ in range(args.cases):
        cfg = random_case(rng)
        if cfg["thresholding"] and cfg["algorithm_type"] == "dpmsolver":
            cfg["thresholding"] = False
        if cfg["thresholding"] and cfg["xdt"] in ("f16", "bf16") and cfg["schedule"] == "vp_linear":
            cfg["thresholding"] = False       # torch.quantile rejects half tensors: the reference raises, nothing to compare
        cfgs.append(cfg)
    t0 = time.perf_counter()
    gpu = []
    if args.device == "cpu":                  # plumbing check of this tool where there is no GPU: the double on both sides
        install_cpu_double(_MP(), S, D)
    import faulthandler
    cur = (os.path.splitext(args.out)[0] if args.out else "/tmp/fuzz_gpu") + "_current_case.txt"
    with contextlib.redirect_stdout(io.StringIO()):
        for i, cfg in enumerate(cfgs):
            # a case that does not come back (a host loop that never ends, a kernel that never finishes) must say which one it
            # is: the stack of every thread after --case-timeout seconds, then exit
            with open(cur, "w") as f:
                f.write("%d %s\n" % (i, cfg))
            faulthandler.dump_traceback_later(args.case_timeout, exit=True, file=sys.__stderr__)
            gpu.append(run(cfg, args.device))
            faulthandler.cancel_dump_traceback_later()
    os.remove(cur)
    if args.device != "cpu":
        torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    if args.device != "cpu":
        install_cpu_double(_MP(), S, D)
    torch.set_num_threads(1)
    n_bad = n_same = n_raise = n_ok = 0
    worst_all = {"f32": 0.0, "f64": 0.0, "half": 0.0, "adaptive": 0.0}
    kinds = {}
    t0 = time.perf_counter()
    for i, (cfg, g) in enumerate(zip(cfgs, gpu)):
        with contextlib.redirect_stdout(io.StringIO()):
            c = run(cfg, "cpu")
        bad, worst, same = compare(cfg, g, c)
        n_raise += g[0] == "raise"
        if g[0] == "ok":
            n_ok += 1
            n_same += bool(same)
            go = g[1][0] if isinstance(g[1], tuple) else g[1]
            k = "adaptive" if cfg["method"] == "adaptive" else ("f64" if go.dtype is torch.float64 else ("f32" if go.dtype is torch.float32 else "half"))
            worst_all[k] = max(worst_all[k], worst)
        if bad:
            n_bad += 1
            kinds[bad[0].split(":")[0][:40]] = kinds.get(bad[0].split(":")[0][:40], 0) + 1
            print("case %d: %s\n    %s" % (i, {k: v for k, v in cfg.items() if k != "seed"}, "\n    ".join(bad)), flush=True)
    rec = dict(cases=args.cases, seed=args.seed, returned=n_ok, raised=n_raise, bit_identical=n_same,
               disagreements=n_bad, kinds=kinds, worst_fraction_of_peak=worst_all, gpu_seconds=round(t_gpu, 1),
               double_seconds=round(time.perf_counter() - t0, 1), device=(torch.cuda.get_device_name(0) if args.device != "cpu" else "cpu (self-check)"),
               library=os.path.basename(getattr(D._lib, "LIB_PATH", "libdpm_hip.so")),
               what="engine on the GPU (libdpm_hip.so) vs the engine's host code on CPU tensors with the numpy double of the stage "
                    "kernel, tools/fuzz_dropin.py's case generator")
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
