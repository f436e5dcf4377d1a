#!/usr/bin/env python3
"""Kernel-level fuzz: ONE `dpm_stage_launch` per case with a randomly built stage record -- combination form (LIN1 / TWO / MS3
/ SS3T / DENOISE), eps -> x0, stored model value, history base, dynamic thresholding, parameterisation (noise / x_start / v /
score), guidance (none / classifier-free / classifier), random coefficients -- on random geometry: batches of 1 to 40, sample
sizes around every boundary of the kernels' tiling (1 element ... whole tiles +- 1 ... the thresholding LDS chunk +- 1 ...
several tiles), every dtype pair that has a kernel ((fp32 | fp16 | bf16 | fp64 state) x network dtype), separate evaluation
state, unaligned views (storage offsets of 1 .. 7 elements), channel-sliced network outputs read in place (eps_stride), the
duplicated [2B,...] store, channels_last operands -- against the numpy double of the stage arithmetic
(tests/kernel_double.py: launch_stage_double) from the SAME inputs.  fp32 / fp64 / half results bit for bit.

    python tools/fuzz_gpu_kernel.py [--cases 3000] [--seed 0] [--out gpurun_out/.../fuzz_gpu_kernel.json]
"""
import argparse
import faulthandler
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import dpm_solver_amd._device as DV  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402
import kernel_double as KD  # noqa: E402

DEV = "cuda:0"
PAIRS = [("f32", "f32"), ("f32", "f32"), ("f32", "f16"), ("f32", "bf16"), ("f16", "f16"), ("bf16", "bf16"), ("f64", "f64")]
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "f64": torch.float64}
# per-sample sizes: 1 element-per-lane kernel, the 8-element access group, one 2048-element tile (256 lanes x 8), the
# thresholding chunk of one workgroup (12288 fp32 elements = 3x64x64), several tiles -- each with its neighbours
SIZES = [1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 33, 48, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096,
         4097, 6143, 6144, 12287, 12288, 12289, 16384, 19200, 24576, 24577, 49152, 65536, 65537]


def random_case(rng):
    sd, ed = PAIRS[int(rng.integers(0, len(PAIRS)))]
    form = int(rng.choice([L.FORM_LIN1, L.FORM_TWO, L.FORM_TWO, L.FORM_MS3, L.FORM_SS3T, L.FORM_DENOISE]))
    per = int(SIZES[int(rng.integers(0, len(SIZES)))]) if rng.random() < 0.8 else int(rng.integers(1, 70000))
    B = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 13, 32, 40]))
    while B * per > 1_600_000:
        B = max(1, B // 2)
    to_x0 = bool(rng.integers(0, 2))
    thr = bool(to_x0 and sd in ("f32", "f64") and rng.random() < 0.3)
    layout = str(rng.choice(["plain", "plain", "plain", "offset", "slice", "nhwc"]))
    values = str(rng.choice(["normal"] * 5 + ["extreme"])) if EXTREME else "normal"
    if values == "extreme":
        thr = False          # (rows holding a NaN under thresholding: by design not the reference's whole-row NaN)
    return dict(sd=sd, ed=ed, form=form, per=per, B=B, to_x0=to_x0, thr=thr, store_m=bool(rng.integers(0, 2)),
                base_hist=bool(form == L.FORM_TWO and rng.integers(0, 2)), model_type=int(rng.integers(0, 4)),
                guidance=int(rng.choice([0, 0, 1, 2])), xe_sep=bool(rng.integers(0, 3) == 0), dup=bool(rng.integers(0, 5) == 0),
                layout=layout, offset=int(rng.integers(1, 8)), seed=int(rng.integers(0, 1 << 30)),
                values=values)


EXTREME = False   # --extreme: one case in six draws magnitudes 1e-45 .. 1e38 and sprinkles inf / NaN over its operands


def shape_of(cfg):
    B, per = cfg["B"], cfg["per"]
    if cfg["layout"] in ("slice", "nhwc"):
        for c in (4, 3, 2):                              # a [B,C,H,W] view of the same size when it factorises
            if per % c == 0:
                hw = per // c
                for h in (64, 32, 16, 8, 4, 2, 1):
                    if hw % h == 0:
                        return (B, c, h, hw // h)
    return (B, per)


def make(cfg):
    g = np.random.default_rng(cfg["seed"])
    sd, ed = DT[cfg["sd"]], DT[cfg["ed"]]
    shape = shape_of(cfg)
    lay = cfg["layout"] if len(shape) == 4 or cfg["layout"] in ("plain", "offset") else "plain"

    def tens(dt, scale=1.0, kind="state"):
        a = g.standard_normal(shape) * scale
        if cfg.get("values") == "extreme":
            a = a * 10.0 ** g.uniform(-45, 38, size=shape) * (g.random(shape) < 0.5) + a * (g.random(shape) < 0.5)
            r = g.random(shape)
            a = np.where(r < 0.002, np.inf, np.where(r < 0.004, -np.inf, np.where(r < 0.006, np.nan, a)))
        a = torch.from_numpy(a.astype(np.float64)).to(dt)
        if lay == "offset":                              # a view whose storage starts off the 16-byte grid
            flat = torch.empty(a.numel() + 8, dtype=dt)
            flat[cfg["offset"]:cfg["offset"] + a.numel()] = a.reshape(-1)
            return flat.to(DEV)[cfg["offset"]:cfg["offset"] + a.numel()].reshape(shape)
        if lay == "nhwc" and len(shape) == 4:
            return a.to(DEV).contiguous(memory_format=torch.channels_last)
        if lay == "slice" and kind == "eps" and len(shape) == 4:
            wide = torch.from_numpy(g.standard_normal((shape[0], 2 * shape[1]) + shape[2:])).to(dt).to(DEV)
            wide[:, :shape[1]] = a.to(DEV)
            return wide[:, :shape[1]]                    # the first C channels of a 2C-channel output (learned variance)
        return a.to(DEV)
    st = L.Stage()
    st.h1_slot = st.h2_slot = st.m_slot = -1
    st.form, st.model_type, st.guidance = cfg["form"], cfg["model_type"], cfg["guidance"]
    fl = 0
    if cfg["to_x0"]:
        fl |= L.F_TO_X0
    if cfg["thr"]:
        fl |= L.F_THRESH
    if cfg["base_hist"]:
        fl |= L.F_BASE_HIST
    st.flags = fl
    a = float(g.uniform(0.05, 0.999))
    st.alpha_e, st.sigma_e = a, float(np.sqrt(1.0 - a * a))
    st.cfg_scale = float(g.choice([1.0, 2.5, 7.5, 7.3]))
    st.cg_scale = float(g.uniform(0.1, 3.0))
    st.cx, st.c0, st.c1, st.c2 = (float(v) for v in g.uniform(-1.5, 1.5, size=4))
    for i in range(5):
        st.k[i] = float(g.uniform(0.2, 3.0))
    st.thr_ratio = float(g.uniform(0.9, 0.999))
    st.thr_max = float(g.choice([1.0, 0.5, 2.0]))
    x = tens(sd)
    xe = tens(sd) if cfg["xe_sep"] else None
    e0 = tens(ed, 1.0, "eps")
    e1 = tens(ed, 1.0, "eps") if cfg["guidance"] == 1 else None
    gr = tens(torch.float32 if sd is not torch.float64 else sd, 0.3, "grad") if cfg["guidance"] == 2 else None
    if gr is not None and sd in (torch.float16, torch.bfloat16):
        gr = gr.to(sd)                                   # (a half state: the classifier's gradient w.r.t. a half input is half)
    h1 = tens(sd) if cfg["form"] in (L.FORM_TWO, L.FORM_MS3, L.FORM_SS3T) else None
    h2 = tens(sd) if cfg["form"] in (L.FORM_MS3, L.FORM_SS3T) else None
    return st, x, xe, e0, e1, gr, h1, h2, sd


def cpu(t):
    return None if t is None else t.detach().cpu()


def same_bits(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and bool(((a == b) | (a.isnan() & b.isnan())).all())


def same_extreme(a, b):
    """--extreme cases: equal bits, or both non-finite (the division by a launch-invariant alpha / sigma is the exact quotient
    for finite operands with a normal quotient, dpm_stage_kernel.hpp: div_by_alpha; an infinite numerator comes out NaN where a
    true division gives inf), or both below 2^-98 (3e-30; half: below 64 x the smallest normal number) and one ulp apart: the
    correction's residual x - q alpha is exact only while it is a normal number, i.e. for |x| above ~2^-102, and subnormal
    quotients carry fewer bits -- magnitudes no sampler state or network output has)"""
    return a.shape == b.shape and a.dtype == b.dtype and bool(extreme_ok(a, b).all())


def extreme_ok(a, b):
    af, bf = a.double(), b.double()
    tiny = 2.0 ** -14 if a.dtype == torch.float16 else 2.0 ** -126
    small = tiny * 64 if a.dtype == torch.float16 else 2.0 ** -98
    close = ((af - bf).abs() < tiny / 4) | ((af - bf).abs() <= bf.abs() * 2.0 ** -21)
    return (a == b) | (~af.isfinite() & ~bf.isfinite()) | ((af.abs() < small) & (bf.abs() < small) & close)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--case-timeout", type=int, default=60)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--extreme", action="store_true")
    args = ap.parse_args()
    global EXTREME
    EXTREME = args.extreme
    rng = np.random.default_rng(args.seed)
    cfgs = [random_case(rng) for _ in range(args.cases)]
    idx = range(args.cases) if args.only is None else [args.only]
    cur = (os.path.splitext(args.out)[0] if args.out else "/tmp/fuzz_gpu_kernel") + "_current_case.txt"
    n_bad = n_rej = n_ok = 0
    rejected, per_pair, per_form = {}, {}, {}
    t0 = time.perf_counter()
    for i in idx:
        cfg = cfgs[i]
        with open(cur, "w") as f:
            f.write("%d %s\n" % (i, cfg))
        faulthandler.dump_traceback_later(args.case_timeout, exit=True, file=sys.__stderr__)
        st, x, xe, e0, e1, gr, h1, h2, sd = make(cfg)
        ext = {"dup": True} if cfg["dup"] else None
        st_c = st.copy()
        try:
            out, m = DV._launch_stage(st, x, xe, e0, e1, gr, h1, h2, sd, want_m=cfg["store_m"], ext=ext)
            torch.cuda.synchronize()
        except (ValueError, RuntimeError, NotImplementedError) as e:
            faulthandler.cancel_dump_traceback_later()
            n_rej += 1
            key = "%s: %s" % (type(e).__name__, str(e)[:90])
            rejected[key] = rejected.get(key, 0) + 1
            continue
        faulthandler.cancel_dump_traceback_later()
        ext_c = {"dup": True} if cfg["dup"] else None
        want, wm = KD.launch_stage_double(st_c, cpu(x), cpu(xe), cpu(e0), cpu(e1), cpu(gr), cpu(h1), cpu(h2), sd,
                                          want_m=cfg["store_m"], ext=ext_c)
        same = same_extreme if cfg.get("values") == "extreme" else same_bits
        bad = []
        if not same(out.cpu(), want):
            d = (out.cpu().double() - want.double()).nan_to_num().abs()
            bad.append("x_out: %d of %d elements differ, max %.3g of %.3g" % (int((out.cpu() != want).sum()), want.numel(), float(d.max()),
                                                                             float(want.double().abs().max())))
        if cfg["store_m"] and (m is None or wm is None or not same(m.cpu(), wm)):
            bad.append("m_out differs" if m is not None and wm is not None else "m_out missing on one side")
        if cfg["dup"] and not (same(ext["x2"][:cfg["B"]].cpu(), want) and same(ext["x2"][cfg["B"]:].cpu(), want)):
            bad.append("the duplicated [2B,...] store differs from x_out")
        n_ok += 1
        for d_, k in ((per_pair, "%s/%s" % (cfg["sd"], cfg["ed"])), (per_form, "form %d%s" % (cfg["form"], " thr" if cfg["thr"] else ""))):
            a = d_.setdefault(k, dict(launches=0, disagreements=0))
            a["launches"] += 1
            a["disagreements"] += bool(bad)
        if bad and args.only is not None:
            o, w = out.cpu().reshape(-1), want.reshape(-1)
            ii = torch.nonzero(~((o == w) | (o.isnan() & w.isnan()))).reshape(-1)[:6].tolist()
            if cfg.get("values") == "extreme":
                ii = torch.nonzero(~extreme_ok(o, w)).reshape(-1)[:6].tolist()
                if m is not None and wm is not None:
                    ii += torch.nonzero(~extreme_ok(m.cpu().reshape(-1), wm.reshape(-1))).reshape(-1)[:6].tolist()
            print("   stage: cx %r c0 %r c1 %r c2 %r k %r alpha %r sigma %r cfg %r cg %r thr %r %r flags %#x" % (
                st_c.cx, st_c.c0, st_c.c1, st_c.c2, list(st_c.k), st_c.alpha_e, st_c.sigma_e, st_c.cfg_scale, st_c.cg_scale, st_c.thr_ratio, st_c.thr_max, st_c.flags))
            flat = lambda t: None if t is None else t.detach().cpu().reshape(t.shape[0], -1) if False else t.detach().cpu().contiguous().reshape(-1)
            for j in ii:
                print("   [%d] gpu %r double %r | x %r xe %r e0 %r e1 %r g %r h1 %r h2 %r%s" % (
                    j, o[j].item(), w[j].item(), flat(x)[j].item(), None if xe is None else flat(xe)[j].item(), flat(e0)[j].item(),
                    None if e1 is None else flat(e1)[j].item(), None if gr is None else flat(gr)[j].item(),
                    None if h1 is None else flat(h1)[j].item(), None if h2 is None else flat(h2)[j].item(),
                    "" if m is None or wm is None else " | m gpu %r double %r" % (m.cpu().reshape(-1)[j].item(), wm.reshape(-1)[j].item())))
        if bad:
            n_bad += 1
            print("case %d: %s shape %s\n    %s" % (i, {k: v for k, v in cfg.items() if k != "seed"}, shape_of(cfg), "\n    ".join(bad)), flush=True)
    os.remove(cur)
    rec = dict(cases=len(list(idx)), seed=args.seed, launched=n_ok, rejected_by_the_library=n_rej, rejections=rejected, disagreements=n_bad,
               per_dtype_pair=per_pair, per_form=per_form, seconds=round(time.perf_counter() - t0, 1), device=torch.cuda.get_device_name(0),
               what="one dpm_stage_launch per case (random stage record, geometry around every tiling boundary, dtype pairs, layouts) vs "
                    "the numpy double of the stage arithmetic from the same inputs, bit for bit")
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    return n_bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
