#!/usr/bin/env python3
"""Kernel-only timing of every stage-kernel variant the BASELINE configs use (MI355X).

Runs DPM_Solver.sample() with a frozen network and replaces dpm_stage_launch by dpm_stage_launch_timed
(hipExtLaunchKernelGGL start/stop events around the kernel itself), then groups the launches by kernel signature
and prints algorithmic bytes, median microseconds and GB/s against the 8 TB/s HBM peak.

    python tools/stage_bench.py [--md profiles/rNN_stage_table.md]
"""
import argparse
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (tools run on the LAB build of the library: include/dpm_lab.h)
import dpm_solver_amd as D  # noqa: E402
import dpm_solver_amd.solver as S  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402

DEV = "cuda:0"
PEAK = 8000.0
FORM = {0: "LIN1", 1: "TWO", 2: "MS3", 3: "SS3T", 4: "DENOISE"}
GUIDE = {0: "-", 1: "cfg", 2: "clsg"}


def sd_schedule():
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    return D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(np.cumprod(1.0 - betas).astype(np.float32)))


def ddpm_schedule():
    return D.NoiseScheduleVP("discrete", betas=torch.from_numpy(np.linspace(1e-4, 0.02, 1000, dtype=np.float64).astype(np.float32)))


_SCRATCH = None


def evict_caches():
    """stand-in for the network that runs between two solver stages in real use: stream 768 MiB through the chip so
    that nothing the previous stage wrote is left in L2 or the 256 MiB Infinity Cache"""
    global _SCRATCH
    if _SCRATCH is None:
        _SCRATCH = torch.zeros(768 * 1024 * 1024 // 4, dtype=torch.float32, device=DEV)
    _SCRATCH.sum()          # read-only: leaves clean lines behind (a dirty evictor adds its own write-backs to the kernel)


class Timed:
    def __init__(self, evict=False):
        self.rows = []
        self.real = L.lib.dpm_stage_launch
        self.evict = evict

    def __call__(self, st, b, stream):
        so, bo = st._obj, b._obj
        ms = C.c_float()
        if self.evict:
            evict_caches()
        rc = L.lib.dpm_stage_launch_timed(st, b, stream, C.byref(ms))
        ss = {L.DTYPE_F32: 4, L.DTYPE_F16: 2, L.DTYPE_BF16: 2}[bo.state_dtype]
        es = {L.DTYPE_F32: 4, L.DTYPE_F16: 2, L.DTYPE_BF16: 2}[bo.eps_dtype]
        n = bo.n
        needs_x = so.form != L.FORM_DENOISE
        need_xe = bool(so.flags & L.F_TO_X0) or so.model_type in (1, 2)
        rd = 0
        if needs_x:
            rd += ss
        if need_xe and (bo.xe and bo.xe != bo.x or not needs_x):
            rd += ss
        rd += es * (1 + (1 if so.guidance == 1 else 0) + (1 if so.guidance == 2 else 0))
        rd += ss * ((1 if so.form in (1, 2, 3) else 0) + (1 if so.form in (2, 3) else 0))
        if so.flags & L.F_BLEND:
            rd += ss * (1 + (1 if bo.blend_b else 0)) + ss * bo.mask_period / n
        wr = ss * (1 + (1 if bo.x_out2 else 0) + (1 if so.flags & L.F_STORE_M else 0))
        sig = "%s %s%s%s%s%s%s" % (FORM[so.form], GUIDE[so.guidance], " thr" if so.flags & L.F_THRESH else "",
                                  " +m" if so.flags & L.F_STORE_M else "", " dup" if bo.x_out2 else "",
                                  " strided" if bo.eps_stride else "", " blend" if so.flags & L.F_BLEND else "")
        self.rows.append((sig, n * (rd + wr), ms.value * 1e3))
        return rc


ONLY = None


def run(name, solver, x, reps=5, **kw):
    """every launch of `solver.sample(x, **kw)` timed twice: back to back (inputs of a stage still in the caches) and
    with the caches evicted before each launch (what a real network between the stages does)"""
    if ONLY and ONLY not in name:
        return []
    res = {}
    for evict in (False, True):
        t = Timed(evict)
        L.lib.dpm_stage_launch = t
        S._stage_launch_raw = t            # the prebuilt launch records of DPM_Solver.sample()
        try:
            for _ in range(reps if not evict else max(reps - 2, 2)):
                solver.sample(x, **kw)
        finally:
            L.lib.dpm_stage_launch = t.real
            S._stage_launch_raw = t.real
        n_rep = reps if not evict else max(reps - 2, 2)
        per = len(t.rows) // n_rep
        groups = collections.OrderedDict()
        for sig, by, us in t.rows[per:]:          # first repetition = warm-up
            groups.setdefault((sig, by), []).append(us)
        for (sig, by), v in groups.items():
            res.setdefault((sig, by), {})[evict] = (float(np.median(v)), len(v) // (n_rep - 1))
    out = []
    for (sig, by), d in res.items():
        w, c = d[False][0], d[True][0]
        out.append((name, sig, d[False][1], by, w, by / w / 1e3, c, by / c / 1e3))
    return out


def fused_thresholding(ns, shape, n_req, steps=10):
    """median kernel time per request-stage (steady-state 2M stages) of n_req thresholded requests advanced by
    dpm_plan_run_multi (a workspace per request): fused launches, and DPM_TUNE_MULTI_FUSE = 0 (a launch per request)"""
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, correcting_x0_fn="dynamic_thresholding")
    plan = dpm._get_plan(method="multistep", order=2, steps=steps, skip_type="time_uniform", solver_type="dpmsolver",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    n = int(np.prod(shape))
    keep, rbs = [], (L.RunBuffers * n_req)()
    for i in range(n_req):
        t = [torch.randn(shape, device=DEV) for _ in range(2)] + [torch.empty(shape, device=DEV) for _ in range(6)]
        keep.append(t)
        rb = rbs[i]
        rb.xbuf[0], rb.e0 = t[0].data_ptr(), t[1].data_ptr()
        for j in range(3):
            rb.xbuf[1 + j] = t[2 + j].data_ptr()
            rb.hist[j] = t[5 + j].data_ptr()
        rb.n, rb.batch, rb.state_dtype, rb.eps_dtype = n, shape[0], L.DTYPE_F32, L.DTYPE_F32
    nb = L.lib.dpm_threshold_workspace_bytes(shape[0], n // shape[0])
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = (C.c_int * n_req)()
    ms = (C.c_float * (n_req * len(plan.stages)))()
    out = []
    ws = [torch.zeros(max(nb, 16), dtype=torch.uint8, device=DEV) for _ in range(n_req)]
    for i in range(n_req):
        rbs[i].workspace = ws[i].data_ptr() if nb else None
    for fuse in (1, 0):
        L.lib.dpm_tuning_set(L.TUNE_MULTI_FUSE, fuse)
        try:
            per = []
            for rep in range(4):
                L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, n_req, stream, ms, res))
                torch.cuda.synchronize()
                a = np.array(list(ms)).reshape(n_req, len(plan.stages))
                if rep:
                    per.append(np.median(a[:, 2:-1]))      # the steady-state stages (TWO + m store)
        finally:
            L.lib.dpm_tuning_set(L.TUNE_MULTI_FUSE, 1)
        out.append(float(np.median(per)) * 1e3)
    return tuple(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default=None)
    ap.add_argument("--only", default=None, help="substring filter on scenario names (skips the others and the loop timing)")
    ap.add_argument("--block-threads", type=int, default=-1, help="DPM_TUNE_BLOCK_THREADS for the run (default: by size)")
    ap.add_argument("--force-generic", action="store_true",
                    help="DPM_TUNE_FORCE_GENERIC: the run-time-prologue kernels everywhere (the A/B behind the kernel-count budget)")
    args = ap.parse_args()
    L.require_lab("tools/stage_bench.py")
    if args.block_threads >= 0:
        L.check(L.lib.dpm_tuning_set(L.TUNE_BLOCK_THREADS, args.block_threads))
    if args.force_generic:
        L.check(L.lib.dpm_tuning_set(L.TUNE_FORCE_GENERIC, 1))
    torch.manual_seed(0)
    global ONLY
    ONLY = args.only
    rows = []
    sd, dd = sd_schedule(), ddpm_schedule()

    def frozen(shape, dtype, n=1):
        outs = [torch.randn(shape, device=DEV).to(dtype) for _ in range(n)]
        return outs

    # cfg2: 2M++, [256,4,64,64], fp16 / fp32 state
    for dt in (torch.float16, torch.float32):
        shape = (256, 4, 64, 64)
        e, = frozen(shape, dt)
        s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, sd), sd, state_dtype=dt)
        rows += run("cfg2 2M++ %s" % str(dt)[6:], s, torch.randn(shape, device=DEV).to(dt), steps=20, order=2)
    # autocast: fp32 state, fp16 network output
    shape = (256, 4, 64, 64)
    e, = frozen(shape, torch.float16)
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, sd), sd)
    rows += run("cfg2-size 2M++ f32 state / f16 eps", s, torch.randn(shape, device=DEV), steps=20, order=2)
    # other parameterisations of the network (SD 2.x is a v-prediction model; x_start: consistency-style / imagen heads)
    for mt in ("v", "x_start"):
        shape = (256, 4, 64, 64)
        e, = frozen(shape, torch.float16)
        s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, sd, model_type=mt), sd, state_dtype=torch.float16)
        rows += run("cfg2-size 2M++ float16 %s-prediction" % mt, s, torch.randn(shape, device=DEV).half(), steps=20, order=2)
    # eps-form (algorithm_type="dpmsolver": no eps -> x0 conversion): multistep at cfg2 size, and the unconditional singlestep-3
    # sampler the reference recommends for pixel-space models without guidance
    for dt in (torch.float16, torch.float32):
        shape = (256, 4, 64, 64)
        e, = frozen(shape, dt)
        s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, sd), sd, algorithm_type="dpmsolver", state_dtype=dt)
        rows += run("cfg2-size 2M dpmsolver (eps form) %s" % str(dt)[6:], s, torch.randn(shape, device=DEV).to(dt), steps=20, order=2)
    shape = (64, 3, 256, 256)
    e, = frozen(shape, torch.float32)
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, dd), dd, algorithm_type="dpmsolver")
    rows += run("uncond 3S dpmsolver (eps form) [64,3,256,256]", s, torch.randn(shape, device=DEV), reps=3, steps=15, order=3, method="singlestep")
    del e
    # 3M++
    shape = (256, 4, 64, 64)
    e, = frozen(shape, torch.float16)
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, sd), sd, state_dtype=torch.float16)
    rows += run("3M++ float16", s, torch.randn(shape, device=DEV).half(), steps=20, order=3)
    # SD-like: 2M++ with CFG, fp32 state + fp16 network output, [64,4,64,64]
    shape = (64, 4, 64, 64)
    e2, = frozen((128, 4, 64, 64), torch.float16)
    c = torch.zeros(64, device=DEV)
    s = D.DPM_Solver(D.model_wrapper(lambda x, t, cc: e2, sd, guidance_type="classifier-free", condition=c,
                                     unconditional_condition=c, guidance_scale=7.5), sd)
    rows += run("SD 2M++ cfg, f32 state / f16 eps", s, torch.randn(shape, device=DEV), steps=20, order=2)
    # the same two at cfg2 size ([256,4,64,64]): large enough for the two-tile variant / not bounded by launch ramp-up
    shape = (256, 4, 64, 64)
    e2, = frozen((512, 4, 64, 64), torch.float16)
    c = torch.zeros(256, device=DEV)
    s = D.DPM_Solver(D.model_wrapper(lambda x, t, cc: e2, sd, guidance_type="classifier-free", condition=c,
                                     unconditional_condition=c, guidance_scale=7.5), sd)
    rows += run("cfg2-size 2M++ cfg, f32 state / f16 eps", s, torch.randn(shape, device=DEV), steps=20, order=2)
    del e2
    e, = frozen(shape, torch.float32)
    mb = D.MaskBlend(sd, torch.rand(64, 64, device=DEV), x0=torch.randn(shape, device=DEV), noise=torch.randn(shape, device=DEV))
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, sd), sd, correcting_xt_fn=mb)
    rows += run("cfg2-size inpaint 2M++ MaskBlend f32", s, torch.randn(shape, device=DEV), steps=20, order=2)
    # cfg3: DPM-Solver-3 singlestep, 15 NFE, [64,3,256,256] fp32, CFG 7.5
    shape = (64, 3, 256, 256)
    e2, = frozen((128, 3, 256, 256), torch.float32)
    c = torch.zeros(64, device=DEV)
    for algo in ("dpmsolver", "dpmsolver++"):
        s = D.DPM_Solver(D.model_wrapper(lambda x, t, cc: e2, dd, guidance_type="classifier-free", condition=c,
                                         unconditional_condition=c, guidance_scale=7.5), dd, algorithm_type=algo)
        rows += run("cfg3 3S %s" % algo, s, torch.randn(shape, device=DEV), reps=3, steps=15, order=3, method="singlestep")
    del e2
    # cfg5: 2M++ with dynamic thresholding, pixel space; batch 32 (BASELINE) and 1024
    for B in (32, 1024):
        shape = (B, 3, 64, 64)
        e, = frozen(shape, torch.float32)
        s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, dd), dd, correcting_x0_fn="dynamic_thresholding")
        rows += run("cfg5 2M++ thr B=%d" % B, s, torch.randn(shape, device=DEV), steps=25, order=2)
    # learned-variance output (6 channels) + thresholding, B=1024
    shape = (1024, 3, 64, 64)
    e6 = torch.randn((1024, 6, 64, 64), device=DEV)
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e6[:, :3], dd), dd, correcting_x0_fn="dynamic_thresholding")
    rows += run("guided-diffusion 2M++ thr 6ch B=1024", s, torch.randn(shape, device=DEV), steps=25, order=2)
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e6[:, :3], dd), dd)
    rows += run("guided-diffusion 2M++ 6ch B=1024", s, torch.randn(shape, device=DEV), steps=25, order=2)
    del e6
    # the reference's own ImageNet-256 example (examples/ddpm_and_guided-diffusion/sample.sh:40-50): dpmsolver++ 2M, 20 steps,
    # classifier guidance scale 8, dynamic thresholding, learned-variance (6-channel) network output read in place
    shape = (16, 3, 256, 256)
    e6 = torch.randn((16, 6, 256, 256), device=DEV)
    gfix = torch.randn(shape, device=DEV) * 0.01
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e6[:, :3], dd, guidance_type="classifier", condition=torch.zeros(16, device=DEV),
                                     guidance_scale=8.0, classifier_fn=lambda x, t, c: (x * gfix).sum(dim=(1, 2, 3))), dd,
                     correcting_x0_fn="dynamic_thresholding")
    rows += run("guided-diffusion ImageNet-256 example: 2M++ thr 6ch classifier [16,3,256,256]", s, torch.randn(shape, device=DEV),
                reps=3, steps=20, order=2)
    del e6, gfix
    # large-sample thresholding [64,3,256,256]
    shape = (64, 3, 256, 256)
    e, = frozen(shape, torch.float32)
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, dd), dd, correcting_x0_fn="dynamic_thresholding")
    rows += run("2M++ thr [64,3,256,256]", s, torch.randn(shape, device=DEV), reps=3, steps=10, order=2)
    # inpainting: 2M++ + MaskBlend, [64,4,64,64] fp32
    shape = (64, 4, 64, 64)
    e, = frozen(shape, torch.float32)
    mb = D.MaskBlend(sd, torch.rand(64, 64, device=DEV), x0=torch.randn(shape, device=DEV), noise=torch.randn(shape, device=DEV))
    s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, sd), sd, correcting_xt_fn=mb)
    rows += run("inpaint 2M++ MaskBlend f32", s, torch.randn(shape, device=DEV), steps=20, order=2)

    # Python host loop: eager sample() vs DPM_Solver.capture() replay (frozen network), wall per trajectory
    import time
    loop = []
    for label, shape, dt, kw in [] if ONLY else [
            ("cfg2 [256,4,64,64] f16 2M++ 20 steps", (256, 4, 64, 64), torch.float16, dict(steps=20, order=2)),
            ("cfg1 [8,4,64,64] f32 2M++ 20 steps", (8, 4, 64, 64), torch.float32, dict(steps=20, order=2))]:
        e, = frozen(shape, dt)
        s = D.DPM_Solver(D.model_wrapper(lambda x, t: e, sd), sd, state_dtype=dt)
        x = torch.randn(shape, device=DEV).to(dt)
        g = s.capture(x, **kw)
        res = {}
        for mode, fn in (("eager", lambda: s.sample(x, **kw)), ("graph", lambda: g(x))):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / 50 * 1e6
        loop.append((label, res["eager"], res["graph"]))

    # adaptive solver (DPM-Solver-12 / -23): host-side control loop (one .item() per iteration) vs the controller on the device
    adaptive = []
    if not ONLY:
        import contextlib
        import io
        lin = D.NoiseScheduleVP("linear")
        for shape in ((8, 4, 64, 64), (256, 4, 64, 64)):
            xa = torch.randn(shape, device=DEV)
            for order in (2, 3):
                row = [str(shape), order]
                for on_dev in (False, True):
                    s = D.DPM_Solver(D.model_wrapper(lambda x, t: x * 0.5, lin), lin, algorithm_type="dpmsolver")
                    s.adaptive_on_device = on_dev
                    buf = io.StringIO()
                    with contextlib.redirect_stdout(buf):
                        for _ in range(3):
                            s.sample(xa, method="adaptive", order=order, t_end=1e-3)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(10):
                            s.sample(xa, method="adaptive", order=order, t_end=1e-3)
                        torch.cuda.synchronize()
                    nfe = int(buf.getvalue().strip().splitlines()[-1].split()[-1])
                    row += [(time.perf_counter() - t0) / 10 * 1e3, nfe]
                adaptive.append(row)

    # R thresholded requests in flight (a server; dpm_plan_run_multi): one thresholding launch per stage over all
    # requests' samples vs one launch per request
    fused_thr = []
    if not ONLY or "in flight" in ONLY:
        for shape, n_req in (((32, 3, 64, 64), 32), ((32, 3, 64, 64), 8), ((8, 3, 256, 256), 8)):
            fused_thr.append((shape, n_req) + fused_thresholding(dd, shape, n_req))

    # requests in flight through the Python API: DPM_Solver.sample_requests vs sample() per request (wall per step)
    py_req = []
    if not ONLY:
        import time as _t
        for label, shape, n_req, cfg_on, edt in (
                ("SD-style CFG 7.5, [64,4,64,64] fp32 state / fp16 outputs", (64, 4, 64, 64), 16, True, torch.float16),
                ("cfg2 [256,4,64,64] fp16", (256, 4, 64, 64), 16, False, torch.float16)):
            nb = shape[0] * (2 if cfg_on else 1)
            outs = [torch.randn((nb,) + shape[1:], device=DEV).to(edt) for _ in range(n_req)]
            calls = [0]

            def pick():
                calls[0] += 1
                return outs[(calls[0] - 1) % n_req]
            if cfg_on:
                cnd = torch.zeros(shape[0], device=DEV)
                fn = D.model_wrapper(lambda x, t, cc: pick(), sd, guidance_type="classifier-free", condition=cnd,
                                     unconditional_condition=cnd, guidance_scale=7.5)
                slv = D.DPM_Solver(fn, sd)
                xs = [torch.randn(shape, device=DEV) for _ in range(n_req)]
            else:
                slv = D.DPM_Solver(D.model_wrapper(lambda x, t: pick(), sd), sd, state_dtype=torch.float16)
                xs = [torch.randn(shape, device=DEV).half() for _ in range(n_req)]
            row = [label, n_req]
            for fnc in (lambda: [slv.sample(x, steps=20, order=2) for x in xs], lambda: slv.sample_requests(xs, steps=20, order=2)):
                for _ in range(2):
                    calls[0] = 0
                    fnc()
                torch.cuda.synchronize()
                t0 = _t.perf_counter()
                for _ in range(5):
                    calls[0] = 0
                    fnc()
                torch.cuda.synchronize()
                row.append((_t.perf_counter() - t0) / 5 * 1e3)
            py_req.append(row)
            del outs, xs, slv

    hdr = ("| scenario | kernel (form guidance flags) | launches | alg. MB | back-to-back us | GB/s | % of 8 TB/s "
           "| caches evicted us | GB/s | % of 8 TB/s |\n|---|---|---|---|---|---|---|---|---|---|")
    lines = [hdr]
    for name, sig, cnt, by, us, gbs, cus, cgbs in rows:
        lines.append("| %s | %s | %d | %.2f | %.2f | %.0f | %.1f | %.2f | %.0f | %.1f |" % (
            name, sig, cnt, by / 1e6, us, gbs, 100 * gbs / PEAK, cus, cgbs, 100 * cgbs / PEAK))
    txt = "\n".join(lines)
    print(txt)
    if args.md:
        with open(args.md, "w") as f:
            f.write("# Stage-kernel table (kernel-only, hipExtLaunchKernelGGL events; tools/stage_bench.py)\n\n"
                    "`DPM_Solver.sample()` with a frozen network, the library's default cache policy (streaming loads: in real "
                    "use a network runs between two stages).  *back-to-back*: the launches follow each other, a stage's inputs "
                    "are still in L2 / the 256 MiB Infinity Cache -- the policy is then the wrong one for the small working "
                    "sets.  *caches evicted*: 768 MiB are streamed through the chip before every launch, as a network would: "
                    "every stream comes from HBM; this is the column that describes real sampling loops.\n\n")
            f.write(txt + "\n")
            f.write("\n## Python host loop (DPM_Solver.sample, frozen network): eager vs hipGraph replay (DPM_Solver.capture)\n\n"
                    "| workload | eager us / trajectory | captured us / trajectory |\n|---|---|---|\n")
            for label, a, b in loop:
                f.write("| %s | %.1f | %.1f |\n" % (label, a, b))
            f.write("\n## Adaptive solver, frozen network x*0.5, 'linear' schedule, t_end = 1e-3: wall per sample() call\n\n"
                    "| state | order | host control loop ms | NFE | controller on the device ms | NFE |\n|---|---|---|---|---|---|\n")
            for shp, order, th, nh, td, nd in adaptive:
                f.write("| %s | %d | %.3f | %d | %.3f | %d |\n" % (shp, order, th, nh, td, nd))
            f.write("\n## Dynamic thresholding with R requests in flight (dpm_plan_run_multi, fp32, 2M++): kernel time per "
                    "request-stage\n\n| requests x shape | one launch per stage over all requests' samples, us | one launch per "
                    "request, us |\n|---|---|---|\n")
            for shape, n_req, t_f, t_s in fused_thr:
                f.write("| %d x %s | %.2f | %.2f |\n" % (n_req, shape, t_f, t_s))
            f.write("\n## Requests in flight through the Python API (frozen network outputs, one per request): wall per 20-stage "
                    "step of all requests\n\n| workload | requests | `sample()` per request, ms | `sample_requests()`, ms |\n|---|---|---|---|\n")
            for label, n_req, t_one, t_grp in py_req:
                f.write("| %s | %d | %.3f | %.3f |\n" % (label, n_req, t_one, t_grp))
    for label, n_req, t_one, t_grp in py_req:
        print("python API, %d requests of %s: sample() per request %.3f ms per step, sample_requests %.3f ms" % (n_req, label, t_one, t_grp))
    for shape, n_req, t_f, t_s in fused_thr:
        print("thresholding, %d requests of %s in flight: %.2f us per request-stage in one launch per stage, %.2f us launched "
              "request by request" % (n_req, shape, t_f, t_s))
    for label, a, b in loop:
        print("python loop %s: eager %.1f us, captured %.1f us" % (label, a, b))
    for shp, order, th, nh, td, nd in adaptive:
        print("adaptive %s order %d: host loop %.3f ms (nfe %d), device controller %.3f ms (nfe %d)" % (shp, order, th, nh, td, nd))


if __name__ == "__main__":
    main()
