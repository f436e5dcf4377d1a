#!/usr/bin/env python3
"""The FLOOR of the lone [256,4,64,64] 2M launch inside a real torch network loop.

A lone 42 MB stage launch behind a network's last kernel runs at 0.61-0.63 of the 8 TB/s peak (8.3-8.6 us), the same launch
fused with 31 others at 0.77.  Round 4 showed the stage kernel within 3-6 % of a no-arithmetic kernel of the same five
streams -- but that floor kernel had only ever been varied over workgroup size, workgroups per CU and the nt policy.  This
tool sweeps the floor itself (`dpm_floor_launch`, lab build: three read streams, two write streams, no arithmetic) over

    load path        global_load_dwordx4 into registers | LDS-DMA (global_load_lds_dwordx4 + ds_read_b128)
    rows in flight   1 / 2 / 4 16-byte loads per lane and stream before the first use (bytes in flight per CU)
    workgroup        256 / 512 / 1024 threads; grid cap 8 / 16 workgroups per CU or none (one tile-row set per 256 lanes)
    load policy      default | nt;   store policy  write-through | plain | nt
    wave priority    s_setprio 0 / 3 (against the tail of the network's last kernel)

in the SAME slot of the SAME loop the stage kernel runs in: DPM_Solver.sample() (2M++, 20 steps) on one request with a
random-init conv network (bench.LoopNet, MIOpen) as model_fn; per step the network, then -- directly behind its last kernel,
inputs cold -- the kernel under test on the stage's own x / eps / m_prev (outputs into scratch), then the real stage kernel
(untimed, so that the trajectory's data stays real: power and clocks depend on it).  Reference rows: the product's stage
kernel in that slot, and the page-touch side-stream helper (`dpm_pagetouch_launch`: one load per 4 KiB of x and m_prev from
a side stream, started before the network's last kernel) in front of it.

    python tools/floor.py --check                                    # every variant moves the right bytes (d = a^b, e = b^c)
    python tools/floor.py --sweep [--out gpurun_out/floor.json]      # start/stop events attached to each launch
    rocprofv3 --kernel-trace -d DIR -o kt -- python tools/floor.py --trace-only --configs <ids> [--seq DIR/seq.json]
    python tools/floor.py --summarise DIR/kt --seq DIR/seq.json      # kernel rows of the trace -> per-config table
"""
import argparse
import ctypes as C
import glob
import itertools
import json
import os
import sqlite3
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

B, SHAPE, STEPS = 256, (4, 64, 64), 20
PEAK = 8000.0


def cfg_id(c):
    return "p%d_r%d_b%d_g%d_nt%d_pr%d_st%d" % (c["load_path"], c["rows"], c["block"], c["blocks_per_cu"], c["nt"], c["prio"], c["store"])


def parse_id(s):
    v = dict(zip(("load_path", "rows", "block", "blocks_per_cu", "nt", "prio", "store"),
                 [int("".join(ch for ch in tok if ch.isdigit())) for tok in s.split("_")]))
    return v


def all_configs():
    out = []
    for path, rows, block, bpc, nt, prio in itertools.product((0, 1), (1, 2, 4), (256, 512, 1024), (0, 8, 16), (1, 0), (0, 3)):
        if path == 1 and (block // 64) * 3 * rows * 1024 > 65536:
            continue
        out.append(dict(load_path=path, rows=rows, block=block, blocks_per_cu=bpc, nt=nt, prio=prio, store=0))
    return out


def desc(L, c):
    f = L.FloorDesc()
    for k, v in c.items():
        setattr(f, k, v)
    return f


def check():
    """every variant writes d = a ^ b and e = b ^ c, ragged sizes included"""
    import _lab  # noqa: F401
    import torch
    from dpm_solver_amd import _lib as L
    L.require_lab("tools/floor.py")
    dev = torch.device("cuda", 0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bad = 0
    cfgs = all_configs() + [dict(c, store=s) for c in all_configs()[:6] for s in (1, 2)]
    for nbytes in (16 * 1000 * 13, 8 << 20):
        a, b, c = (torch.randint(0, 2 ** 31 - 1, (nbytes // 4,), dtype=torch.int32, device=dev) for _ in range(3))
        for cf in cfgs:
            d, e = torch.zeros_like(a), torch.zeros_like(a)
            L.check(L.lib.dpm_floor_launch(C.byref(desc(L, cf)), a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(),
                                           e.data_ptr(), nbytes, stream, None))
            torch.cuda.synchronize()
            ok = bool(torch.equal(d, a ^ b) and torch.equal(e, b ^ c))
            bad += not ok
            if not ok:
                print("WRONG", cfg_id(cf), nbytes)
    print("floor variants checked: %d launches, %d wrong" % (2 * len(cfgs), bad))
    return bad == 0


class Loop:
    """DPM_Solver.sample() on one [256,4,64,64] request with a conv network; the slot behind the network's last kernel is
    handed to `slot_fn(k, st, b, stream)` (k = running launch index), then the real stage kernel runs (untimed)."""

    def __init__(self, dtype_name="fp16", kind="conv", width=256):
        import _lab  # noqa: F401
        import torch
        import bench
        import dpm_solver_amd as D
        import dpm_solver_amd.solver as S
        from dpm_solver_amd import _lib as L
        L.require_lab("tools/floor.py")
        self.torch, self.S, self.L, self.D = torch, S, L, D
        self.dev = torch.device("cuda", 0)
        dtype = bench._DT[dtype_name]
        self.dtype = dtype
        ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(bench.sd_alphas_cumprod()))
        self.net = bench.LoopNet(kind, width, dtype, self.dev)
        g = torch.Generator(device="cpu").manual_seed(4321)
        self.x_T = torch.randn((B,) + SHAPE, generator=g).to(self.dev, dtype)
        self.dpm = D.DPM_Solver(D.model_wrapper(self.net, ns), ns, algorithm_type="dpmsolver++", state_dtype=dtype)
        with torch.no_grad():
            self.out0 = self.dpm.sample(self.x_T, steps=STEPS, order=2)
        torch.cuda.synchronize()
        self.nbytes = self.x_T.numel() * self.x_T.element_size()
        self.scratch = [torch.empty_like(self.x_T) for _ in range(2)]
        self.fr = next(iter(self.dpm._fast.values()))
        self.side = torch.cuda.Stream(device=self.dev)

    def run(self, slot_fn, trajectories, before_last=None):
        torch, S = self.torch, self.S
        raw = S._stage_launch_raw
        count = [0]

        def patched(st, b, stream):
            k = count[0]
            count[0] += 1
            rc = slot_fn(k, st, b, stream)
            if rc:
                return rc
            return raw(st, b, stream)
        self.net.before_last = (lambda: before_last(count[0])) if before_last else None
        try:
            S._stage_launch_raw = patched
            with torch.no_grad():
                for _ in range(trajectories):
                    out = self.dpm.sample(self.x_T, steps=STEPS, order=2)
        finally:
            S._stage_launch_raw = raw
            self.net.before_last = None
        assert torch.equal(out, self.out0), "the slot kernel changed the trajectory"
        return count[0]


def steady(k):
    """launch index k is a steady-state 2M stage (reads x, eps, m_prev; writes x', m): not the first / last stage"""
    return 0 < k % STEPS < STEPS - 1


def sweep(args):
    lp = Loop(args.dtype)
    L, torch = lp.L, lp.torch
    cfgs = all_configs()
    if args.quick:
        cfgs = [c for c in cfgs if c["blocks_per_cu"] in (0, 8) and c["prio"] == 0][::3]
    T = args.trajectories
    results = {}
    stream_of = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def measure(label, slot_builder, before_last=None, reps=1):
        vals = []
        for _ in range(reps):
            trace = C.c_void_p()
            n = T * STEPS
            L.check(L.lib.dpm_trace_create(n, C.byref(trace)))
            try:
                lp.run(slot_builder(trace), T, before_last=before_last)
                ms = (C.c_float * n)()
                L.check(L.lib.dpm_trace_read(trace, stream_of(), ms, n))
            finally:
                L.lib.dpm_trace_destroy(trace)
            us = np.frombuffer(ms, dtype=np.float32).astype(np.float64) * 1e3
            vals.append(np.array([us[k] for k in range(STEPS, n) if steady(k) and us[k] > 0]))     # first trajectory: warm-up
        v = np.concatenate(vals)
        results[label] = dict(median_us=round(float(np.median(v)), 3), mean_us=round(float(v.mean()), 3),
                              p10_us=round(float(np.percentile(v, 10)), 3), p90_us=round(float(np.percentile(v, 90)), 3), n=int(v.size))
        return results[label]

    # the product's stage kernel in the slot (traced); the real (untimed) launch behind it re-runs the same stage
    def stage_slot(trace):
        def fn(k, st, b, stream):
            return L.lib.dpm_stage_launch_traced(st, b, stream, trace, k)
        return fn

    def floor_slot(cf):
        f = desc(L, cf)

        def builder(trace):
            def fn(k, st, b, stream):
                bb = b._obj
                if not steady(k) or not bb.h1:
                    return 0
                return L.lib.dpm_floor_launch_traced(C.byref(f), bb.x, bb.e0, bb.h1, lp.scratch[0].data_ptr(),
                                                     lp.scratch[1].data_ptr(), lp.nbytes, stream, trace, k)
            return fn
        return builder

    def touch(stride):
        ev = torch.cuda.Event()

        def before_last(k):                      # the network is about to enqueue its last kernel; stage k % STEPS follows
            b = lp.fr.bufs[k % STEPS]
            ptrs = [p for p in (b.x, b.h1) if p]
            if not ptrs:
                return
            ev.record()
            lp.side.wait_event(ev)
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            nb = (C.c_int64 * len(ptrs))(*[lp.nbytes] * len(ptrs))
            L.check(L.lib.dpm_pagetouch_launch(arr, nb, len(ptrs), stride, C.c_void_p(lp.side.cuda_stream)))
        return before_last

    measure("stage_kernel", stage_slot, reps=2)
    print("stage kernel in the slot: %s" % results["stage_kernel"], flush=True)
    for stride in (4096, 65536, 2 << 20):
        measure("stage_kernel+pagetouch_%d" % stride, stage_slot, before_last=touch(stride))
        print("  + page touch every %d B: %s" % (stride, results["stage_kernel+pagetouch_%d" % stride]), flush=True)
    for i, cf in enumerate(cfgs):
        r = measure(cfg_id(cf), floor_slot(cf))
        if i % 12 == 0:
            print("%3d/%d %s %s" % (i, len(cfgs), cfg_id(cf), r), flush=True)
    measure("stage_kernel_again", stage_slot, reps=2)
    floors = sorted(((v["median_us"], k) for k, v in results.items() if k.startswith("p")))
    best = floors[:12]
    # the store policies on the best load-side configurations
    for _, k in best[:4]:
        for st in (1, 2):
            cf = dict(parse_id(k), store=st)
            measure(cfg_id(cf), floor_slot(cf))
    alg = 5 * lp.nbytes
    out = dict(what="floor of the lone [%d,4,64,64] %s 2M launch inside a conv-network loop: median start->stop event interval "
                    "(us) of the kernel in the slot right behind the network's last kernel; events carry a constant ~1.2-1.5 us "
                    "dispatch offset over rocprofv3 rows (profiles/README.md): ranking and ratios, not absolutes" % (B, args.dtype),
               algorithmic_bytes=alg, results=results, best=[k for _, k in best])
    json.dump(out, open(args.out, "w"), indent=1)
    print("\nstage kernel: %.3f / %.3f us;  best floor configurations:" % (results["stage_kernel"]["median_us"],
                                                                          results["stage_kernel_again"]["median_us"]))
    for v, k in best:
        print("  %-34s %.3f us" % (k, v))
    return out


def trace_only(args):
    """for rocprofv3 --kernel-trace: passes of `trajectories` trajectories each -- the stage kernel as the library launches it
    (read streams by LDS-DMA), the stage kernel forced onto the register path (DPM_TUNE_LDS_DMA = 0), every floor
    configuration of --configs in the slot (followed by the real stage kernel), then the two stage-kernel passes again.  The
    order goes to --seq so that --summarise can attribute the rows (run-time parameters do not show in a kernel's name)."""
    lp = Loop(args.dtype)
    L = lp.L
    ids = [s for s in args.configs.split(",") if s]
    seq = []
    T = args.trajectories

    def nothing(k, st, b, stream):
        return 0          # the real launch behind the slot IS the stage kernel: nothing extra in the slot

    def stage_pass(dma):
        L.check(L.lib.dpm_tuning_set(L.TUNE_LDS_DMA, dma))
        lp.run(nothing, T)
        L.check(L.lib.dpm_tuning_set(L.TUNE_LDS_DMA, -1))
        seq.append(dict(id="stage_kernel" if dma != 0 else "stage_kernel_register_path", trajectories=T, floor=False))
    stage_pass(-1)
    stage_pass(0)
    for s in ids:
        f = desc(L, parse_id(s))

        def fn(k, st, b, stream, f=f):
            bb = b._obj
            if not steady(k) or not bb.h1:
                return 0
            return L.lib.dpm_floor_launch(C.byref(f), bb.x, bb.e0, bb.h1, lp.scratch[0].data_ptr(), lp.scratch[1].data_ptr(),
                                          lp.nbytes, stream, None)
        lp.run(fn, T)
        seq.append(dict(id=s, trajectories=T, floor=True))
    stage_pass(-1)
    stage_pass(0)
    lp.torch.cuda.synchronize()
    json.dump(seq, open(args.seq, "w"))
    print("traced %d configurations" % len(ids))


def summarise(args):
    f = glob.glob(os.path.join(args.summarise, "**", "*_results.db"), recursive=True)
    assert f, "no *_results.db under %s" % args.summarise
    cur = sqlite3.connect(f[0]).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
    kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
    rows = cur.execute("select name, start, duration from %s order by start" % kt).fetchall()
    seq = json.load(open(args.seq))
    names = [r[0] for r in rows]
    du = np.array([r[2] for r in rows], dtype=np.float64) / 1e3
    floor_rows = [i for i, n in enumerate(names) if "floor_kernel" in n]
    stage_rows = [i for i, n in enumerate(names) if "stage_kernel" in n]
    per = STEPS - 2
    alg = 5 * B * int(np.prod(SHAPE)) * (2 if args.dtype != "fp32" else 4)
    # the Loop's constructor ran one trajectory (STEPS stage rows) before the first pass; every pass launches T x STEPS stage
    # kernels (in the floor passes: behind the floor kernel, inputs warm -- not reported)
    out, stage = {}, {}
    fpos, spos = 0, STEPS
    for s in seq:
        T = s["trajectories"]
        srows = stage_rows[spos:spos + T * STEPS]
        spos += T * STEPS
        if s["floor"]:
            n = T * per
            v = du[floor_rows[fpos:fpos + n]][per:]               # first trajectory of the configuration: warm-up
            fpos += n
            out[s["id"]] = dict(median_us=round(float(np.median(v)), 3), mean_us=round(float(v.mean()), 3), rows=int(v.size),
                                frac_of_peak=round(alg / float(np.median(v)) / 1e3 / PEAK, 4))
        else:
            keep = [srows[t * STEPS + k] for t in range(1, T) for k in range(1, STEPS - 1)]
            stage.setdefault(s["id"], []).extend(du[keep].tolist())
            stage.setdefault(s["id"] + "|name", names[keep[0]][:160])
    assert fpos == len(floor_rows) and spos == len(stage_rows), (fpos, len(floor_rows), spos, len(stage_rows))
    res = dict(algorithmic_bytes=alg, floor=out,
               what="rocprofv3 --kernel-trace rows: kernel durations in the slot right behind the network's last kernel "
                    "(conv network, one [%d,4,64,64] %s request, DPM_Solver.sample 2M++ 20 steps)" % (B, args.dtype))
    for k in ("stage_kernel", "stage_kernel_register_path"):
        if k in stage:
            v = np.array(stage[k])
            res[k] = dict(median_us=round(float(np.median(v)), 3), mean_us=round(float(v.mean()), 3), rows=int(v.size),
                          frac_of_peak=round(alg / float(np.median(v)) / 1e3 / PEAK, 4), kernel=stage[k + "|name"])
    best = min(out.items(), key=lambda kv: kv[1]["median_us"]) if out else None
    if best:
        res["best_floor"] = dict(id=best[0], **best[1])
        res["stage_kernel_over_best_floor"] = round(res["stage_kernel"]["median_us"] / best[1]["median_us"], 4)
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--trace-only", action="store_true")
    ap.add_argument("--summarise", default=None)
    ap.add_argument("--configs", default="")
    ap.add_argument("--seq", default="gpurun_out/floor_seq.json")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--trajectories", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/floor.json")
    args = ap.parse_args()
    if args.summarise:
        return summarise(args)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    if args.check:
        sys.exit(0 if check() else 1)
    if args.sweep:
        return sweep(args)
    if args.trace_only:
        return trace_only(args)
    ap.print_help()


if __name__ == "__main__":
    main()
