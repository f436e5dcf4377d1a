#!/usr/bin/env python3
"""The reference's OWN example call sites, source files unchanged, running on the engine -- the proof of the "drops into
the existing examples" clause of BASELINE.json's north_star that does not go through dpm_solver_amd/adapters/.

Three call sites, each executed twice on identical inputs:

  own   the file imports its own vendored copy of the solver, exactly as in the reference tree;
  shim  the same file, byte for byte, with `shims/` resolving the import line to dpm_solver_amd.

  A  examples/stable-diffusion/ldm/models/diffusion/dpm_solver/sampler.py   (`from .dpm_solver import ...`, :5)
     DPMSolverSampler.sample / .stochastic_encode / .encode and both DiffEdit variants of scripts/diffedit_inpaint.ipynb
  B  examples/score_sde_pytorch/sampling.py:get_dpm_solver_sampler              (`from dpm_solver import ...`, :29)
     default singlestep-3 / logSNR, denoise, dpmsolver++ multistep, adaptive
  C  examples/ddpm_and_guided-diffusion/runners/diffusion.py:Diffusion.sample_image, the dpmsolver / dpmsolver++ branch
     (:594-639, `from dpm_solver.sampler import ...`, :595): the method itself runs, on a bare Diffusion object carrying
     the args / config fields the branch reads; third-party packages the module imports at its top and this image does
     not have (tkinter, blobfile, torchvision, lmdb, pytorch_fid) are served as inert placeholders -- the branch touches none.

Networks are the small deterministic stand-ins of tests/golden/cases.py (the real UNets are out of scope).
Per call site the tool reports max |own - shim| / max |own| and whether the network-call traces (batch shapes, times)
agree; the bar is 1e-5 (BASELINE.json).  The reference tree is read from $DPM_REFERENCE_DIR (default /root/reference):
it is NOT part of this repository and nothing here is imported by the product.

    python tools/dropin_examples.py --device cuda:0 --out profiles/r05_dropin.json
    (tests/test_dropin_examples.py runs the same functions on CPU with tests/kernel_double.py behind the launch records)
"""
import argparse
import contextlib
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import io
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import cases as CS  # noqa: E402  (stand-in networks and seeded inputs shared with the golden generator)

SHIMS = os.path.join(ROOT, "shims")
TOL = 1e-5
# top-level module names of the example applications: purged from sys.modules around every run so that `own` and `shim`
# (and the three applications, which reuse names like `models`) never see each other's modules
_APP_ROOTS = ("dpm_solver", "sampling", "sde_lib", "models", "runners", "functions", "datasets", "evaluate", "utils",
              "losses", "likelihood", "refsd_pkg")


def reference_examples():
    d = os.environ.get("DPM_REFERENCE_DIR", "/root/reference")
    ex = os.path.join(d, "examples")
    return ex if os.path.isdir(ex) else None


# ---------------------------------------------------------------------------------------------------------------
# import plumbing
# ---------------------------------------------------------------------------------------------------------------
class _AnyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any()


class _Any(metaclass=_AnyMeta):
    """inert placeholder object: callable, iterable (empty), any attribute"""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any()

    def __call__(self, *a, **k):
        return _Any()

    def __iter__(self):
        return iter(())


class _Placeholder(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _AnyMeta(name, (_Any,), {})


class _AbsentPackages(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """serves placeholders for the third-party packages named in `roots` (absent from this image)"""

    def __init__(self):
        self.roots = set()

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Placeholder(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


@contextlib.contextmanager
def app_tree(*path_entries):
    """sys.path = path_entries + sys.path and a clean slate for the applications' top-level module names; undone on exit"""
    is_app = lambda k: k.split(".")[0] in _APP_ROOTS
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if is_app(k)}
    old_path = list(sys.path)
    sys.path[:0] = list(path_entries)
    importlib.invalidate_caches()
    try:
        yield
    finally:
        sys.path[:] = old_path
        for k in [k for k in sys.modules if is_app(k)]:
            sys.modules.pop(k)
        sys.modules.update(saved)
        importlib.invalidate_caches()


def _np(t):
    return t.detach().float().cpu().numpy()


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _solver_module_file(cls):
    return os.path.abspath(sys.modules[cls.__module__].__file__)


# ---------------------------------------------------------------------------------------------------------------
# A. Stable-Diffusion adapter: the reference's sampler.py inside a package whose `.dpm_solver` is its own file / the shim
# ---------------------------------------------------------------------------------------------------------------
def run_stable_diffusion(mode, device, ex):
    sd_pkg = os.path.join(ex, "stable-diffusion", "ldm", "models", "diffusion", "dpm_solver")
    with app_tree():
        pkg = types.ModuleType("refsd_pkg")
        # the package directory as the integration puts it: the reference's files, with dpm_solver.py replaced by
        # shims/dpm_solver/dpm_solver.py in shim mode
        pkg.__path__ = [os.path.join(SHIMS, "dpm_solver")] if mode == "shim" else [sd_pkg]
        sys.modules["refsd_pkg"] = pkg
        spec = importlib.util.spec_from_file_location("refsd_pkg.sampler", os.path.join(sd_pkg, "sampler.py"))
        RS = importlib.util.module_from_spec(spec)
        sys.modules["refsd_pkg.sampler"] = RS
        spec.loader.exec_module(RS)                     # runs `from .dpm_solver import NoiseScheduleVP, model_wrapper, DPM_Solver`
        info = dict(sampler_file=spec.origin, solver_file=_solver_module_file(RS.DPM_Solver))
        if not str(device).startswith("cuda"):
            # the class insists on torch.device("cuda") for its buffers (sampler.py:17-19): lifted for the CPU test only
            RS.DPMSolverSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
        inp = CS.sampler_inputs()
        model = CS.FakeLatentDiffusion(torch, device)
        smp = RS.DPMSolverSampler(model)
        x_T, x0, noise, mask = (_t(inp[k], device) for k in ("x_T", "x0", "noise", "mask"))
        cond, uncond = _t(inp["cond"], device), _t(inp["uncond"], device)
        B = x_T.shape[0]
        out = {}
        x, inter = smp.sample(10, B, x_T.shape[1:], conditioning=cond, unconditional_guidance_scale=7.5,
                              unconditional_conditioning=uncond, x_T=x_T, verbose=False)
        out["sample/final"] = _np(x)
        out["sample/intermediates"] = np.stack([_np(v) for v in inter])
        out["stochastic_encode"] = _np(smp.stochastic_encode(x0, 0.6, noise=noise.unsqueeze(0)))
        enc, einter = smp.encode(10, x0, 0.6, conditioning=cond, unconditional_guidance_scale=7.5,
                                 unconditional_conditioning=uncond)
        out["encode/final"] = _np(enc)
        out["encode/intermediates"] = np.stack([_np(v) for v in einter])
        rev = list(reversed(einter))
        det = lambda xt, t, step: xt * mask + (1 - mask) * rev[step]            # diffedit_inpaint.ipynb, deterministic
        x, _ = smp.sample(10, B, x_T.shape[1:], conditioning=cond * 0.5, unconditional_guidance_scale=7.5,
                          unconditional_conditioning=uncond, lower_order_final=False, t_start=smp.ratio_to_time(0.6),
                          x_T=enc, correcting_xt_fn=det)
        out["diffedit_det"] = _np(x)
        noised = smp.stochastic_encode(x0, 0.6, noise=noise.unsqueeze(0))

        def sto(xt, t, step):                                                    # diffedit_inpaint.ipynb, stochastic
            return xt * mask + (1 - mask) * smp.stochastic_encode(x0, smp.time_to_ratio(t), noise=noise.unsqueeze(0))
        x, _ = smp.sample(10, B, x_T.shape[1:], conditioning=cond * 0.5, unconditional_guidance_scale=7.5,
                          unconditional_conditioning=uncond, lower_order_final=False, t_start=smp.ratio_to_time(0.6),
                          x_T=noised, correcting_xt_fn=sto)
        out["diffedit_sto"] = _np(x)
        trace = [(s, t) for s, t in model.calls]
    return out, trace, info


# ---------------------------------------------------------------------------------------------------------------
# B. ScoreSDE: sampling.get_dpm_solver_sampler with its own sde_lib / models.utils
# ---------------------------------------------------------------------------------------------------------------
def run_score_sde(mode, device, ex):
    app = os.path.join(ex, "score_sde_pytorch")
    entries = ([SHIMS] if mode == "shim" else []) + [app]
    out, trace = {}, []
    with app_tree(*entries):
        import sampling as SM       # the reference file; line 29: from dpm_solver import NoiseScheduleVP, model_wrapper, DPM_Solver
        import sde_lib
        info = dict(sampling_file=os.path.abspath(SM.__file__), solver_file=_solver_module_file(SM.DPM_Solver))
        sde = sde_lib.VPSDE(beta_min=0.1, beta_max=20., N=1000)
        inverse_scaler = lambda x: (x + 1.) / 2.
        base = CS.score_sde_model(torch)

        class Traced(torch.nn.Module):
            def forward(self, x, labels):
                trace.append((tuple(x.shape), labels.detach().float().cpu().numpy().copy()))
                return base(x, labels)
        model = Traced()
        for i, (tag, kw) in enumerate(CS.SCORE_SDE_RUNS):
            torch.manual_seed(100 + i)           # sde.prior_sampling draws on the CPU generator, then .to(device)
            fn = SM.get_dpm_solver_sampler(sde, CS.SCORE_SDE_SHAPE, inverse_scaler, device=device, **kw)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):          # 'adaptive solver nfe N' is printed by both solvers
                y, nfe = fn(model)
            out["%s/x" % tag] = _np(y)
            out["%s/nfe" % tag] = np.int64(nfe)
            if "nfe" in buf.getvalue():
                out["%s/printed_nfe" % tag] = np.int64(int(buf.getvalue().strip().split()[-1]))
            trace.append(("run", tag))
        # the adaptive run once more with the reference's host-side control loop (one .item() per iteration): the
        # network-call trace must then equal the reference's call for call
        if mode == "shim":
            import dpm_solver_amd
            saved_flag = dpm_solver_amd.DPM_Solver.adaptive_on_device
            dpm_solver_amd.DPM_Solver.adaptive_on_device = False
        try:
            torch.manual_seed(100 + len(CS.SCORE_SDE_RUNS) - 1)
            fn = SM.get_dpm_solver_sampler(sde, CS.SCORE_SDE_SHAPE, inverse_scaler, device=device, method="adaptive", order=2)
            with contextlib.redirect_stdout(io.StringIO()):
                y, nfe = fn(model)
            out["adaptive_host_loop/x"] = _np(y)
            trace.append(("run", "adaptive_host_loop"))
        finally:
            if mode == "shim":
                dpm_solver_amd.DPM_Solver.adaptive_on_device = saved_flag
    return out, trace, info


# ---------------------------------------------------------------------------------------------------------------
# C. guided-diffusion: Diffusion.sample_image, dpmsolver / dpmsolver++ branch
# ---------------------------------------------------------------------------------------------------------------
def run_guided_diffusion(mode, device, ex):
    app = os.path.join(ex, "ddpm_and_guided-diffusion")
    entries = ([SHIMS] if mode == "shim" else []) + [app]
    finder = _AbsentPackages()
    out, trace = {}, []
    with app_tree(*entries):
        sys.meta_path.append(finder)
        try:
            while True:
                try:
                    RD = importlib.import_module("runners.diffusion")
                    break
                except ModuleNotFoundError as e:         # a third-party package this image lacks: placeholder, retry
                    root = (e.name or "").split(".")[0]
                    if not root or root in finder.roots or root in _APP_ROOTS or os.path.exists(os.path.join(app, root)):
                        raise
                    finder.roots.add(root)
                    for k in [k for k in sys.modules if k.split(".")[0] in _APP_ROOTS]:
                        sys.modules.pop(k)
            inp = CS.gd_inputs()
            x = _t(inp["x"], device)
            net = CS.gd_network(torch, _t(inp["junk"], device))
            clf = CS.gd_classifier(torch, _t(inp["w"], device))

            def model(xx, t, **kw):
                trace.append((tuple(xx.shape), t.detach().float().cpu().numpy().copy()))
                return net(xx, t, **kw)
            betas = _t(CS.schedule_inputs("ddpm")["betas"], device)
            for i, (tag, kw) in enumerate(CS.GD_RUNS):
                r = object.__new__(RD.Diffusion)          # the Runner's __init__ wants the whole app (datasets, checkpoints)
                r.betas = betas
                r.num_timesteps = betas.shape[0]
                r.args = types.SimpleNamespace(
                    skip=1, scale=None, fixed_class=None, sample_type=kw["sample_type"], thresholding=kw["thresholding"],
                    denoise=kw["denoise"], timesteps=kw.get("timesteps", 12), dpm_solver_order=kw.get("order", 2),
                    skip_type="time_uniform", dpm_solver_method=kw.get("method", "multistep"), lower_order_final=True,
                    dpm_solver_type="dpmsolver", dpm_solver_atol=0.0078, dpm_solver_rtol=0.05, eta=0.0)
                r.config = types.SimpleNamespace(
                    sampling=types.SimpleNamespace(classifier_scale=kw["scale"], cond_class=True),
                    data=types.SimpleNamespace(num_classes=CS.GD_CLASSES),
                    model=types.SimpleNamespace(out_channels=6))
                torch.manual_seed(200 + i)               # classes = torch.randint(...) on the CPU generator (:534)
                y, classes = r.sample_image(x, model, classifier=clf if kw["use_clf"] else None)
                out["%s/x" % tag] = _np(y)
                out["%s/classes" % tag] = classes.cpu().numpy()
                trace.append(("run", tag))
            import dpm_solver.sampler as used            # what the branch's import line (:595) resolved to
            info = dict(runner_file=os.path.abspath(RD.__file__), import_line_resolved_to_file=os.path.abspath(used.__file__),
                        solver_file=_solver_module_file(used.DPM_Solver),
                        placeholder_packages=sorted(finder.roots))
        finally:
            sys.meta_path.remove(finder)
    return out, trace, info


SITES = [
    ("stable-diffusion DPMSolverSampler", "examples/stable-diffusion/ldm/models/diffusion/dpm_solver/sampler.py:5,75-87,95,122-136",
     run_stable_diffusion),
    ("ScoreSDE get_dpm_solver_sampler", "examples/score_sde_pytorch/sampling.py:29,505-555", run_score_sde),
    ("guided-diffusion Diffusion.sample_image (dpmsolver branch)", "examples/ddpm_and_guided-diffusion/runners/diffusion.py:594-639",
     run_guided_diffusion),
]


# ---------------------------------------------------------------------------------------------------------------
# comparison
# ---------------------------------------------------------------------------------------------------------------
def _segments(trace):
    """[(tag, [calls])]: the trace cut at its ("run", tag) markers (a trace without markers is one segment)"""
    segs, cur = [], []
    for u in trace:
        if u[0] == "run":
            segs.append((u[1], cur))
            cur = []
        else:
            cur.append(u)
    if cur or not segs:
        segs.append(("", cur))
    return segs


def _traces_equal(a, b):
    """same number of network calls, same batch shapes, same time labels, run by run.  Fixed time grids: labels equal to
    1e-6 relative (they are, bit for bit, on every grid of these runs).  method='adaptive': the step sizes are computed
    from the error norm of the data (ref :1001-1006), so a 1e-7 difference in x moves the later times by a few 1e-6
    relative: 5e-5.  The engine's device-side adaptive controller (the default) keeps the host `adaptive_lookahead`
    iterations ahead of the device's accept / reject decisions: its trace is the reference's trace plus at most
    (lookahead + 1) * order calls after t_end was reached, whose outputs no kernel reads (the reported
    NFE is equal).  The run tagged 'host_loop' repeats the adaptive run with DPM_Solver.adaptive_on_device = False, where
    the traces must be equal call for call."""
    sa, sb = _segments(a), _segments(b)
    if [t for t, _ in sa] != [t for t, _ in sb]:
        return False, "runs %r vs %r" % ([t for t, _ in sa], [t for t, _ in sb])
    worst, extra, n = 0.0, 0, 0
    for (tag, ca), (_, cb) in zip(sa, sb):
        adaptive = "adaptive" in tag
        lookahead_calls = 4 if (adaptive and "host_loop" not in tag) else 0
        if not (len(ca) <= len(cb) <= len(ca) + lookahead_calls):
            return False, "run %r: %d vs %d network calls" % (tag, len(ca), len(cb))
        extra += len(cb) - len(ca)
        n += len(ca)
        for i, (u, v) in enumerate(zip(ca, cb)):
            if u[0] != v[0]:
                return False, "run %r call %d: %r vs %r" % (tag, i, u[0], v[0])
            tu, tv = np.asarray(u[1], dtype=np.float64), np.asarray(v[1], dtype=np.float64)
            if tu.shape != tv.shape:
                return False, "run %r call %d: time vectors of shape %r vs %r" % (tag, i, tu.shape, tv.shape)
            rel = float((np.abs(tu - tv) / np.maximum(np.abs(tu), 1e-30)).max(initial=0.0))
            if rel > (5e-5 if adaptive else 1e-6):
                return False, "run %r call %d: time labels differ by %.3g relative" % (tag, i, rel)
            worst = max(worst, rel)
    msg = "%d network calls, same batch shapes, time labels equal to %.2g relative" % (n, worst)
    if extra:
        msg += "; +%d look-ahead calls of the device-side adaptive controller after t_end" % extra
    return True, msg


def compare(own, shim):
    """per result: max |shim - own| / max |own| (integers: equality)"""
    rows = {}
    assert sorted(own) == sorted(shim), (sorted(own), sorted(shim))
    for k in sorted(own):
        a, b = np.asarray(own[k]), np.asarray(shim[k])
        if a.dtype.kind in "iu":
            rows[k] = dict(equal=bool(a.shape == b.shape and np.array_equal(a, b)), own=a.tolist(), shim=b.tolist())
        else:
            assert a.shape == b.shape, (k, a.shape, b.shape)
            den = max(float(np.abs(a.astype(np.float64)).max()), 1e-30)
            rows[k] = dict(rel_err=float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) / den, shape=list(a.shape))
    return rows


def run_site(name, where, fn, device, ex):
    own, tr_o, info_o = fn("own", device, ex)
    shim, tr_s, info_s = fn("shim", device, ex)
    rows = compare(own, shim)
    ok_t, msg = _traces_equal(tr_o, tr_s)
    worst = max([r["rel_err"] for r in rows.values() if "rel_err" in r] or [0.0])
    ints_ok = all(r["equal"] for r in rows.values() if "equal" in r)
    rel = lambda p: os.path.relpath(p, os.path.dirname(ex)) if p.startswith(os.path.dirname(ex)) else os.path.relpath(p, ROOT)
    engine_side = {k: rel(v) for k, v in info_s.items() if k.endswith("_file")}
    assert engine_side["solver_file"].startswith("dpm_solver_amd"), engine_side     # the shim run really ran the engine
    assert not rel(info_o["solver_file"]).startswith("dpm_solver_amd"), info_o      # ... and the own run did not
    return dict(call_site=name, reference_lines=where, device=str(device),
                own={k: rel(v) for k, v in info_o.items() if k.endswith("_file")}, shim=engine_side,
                placeholder_packages=info_s.get("placeholder_packages", []),
                max_rel_err=worst, integers_equal=ints_ok, network_trace_equal=ok_t, network_trace=msg,
                passed=bool(worst <= TOL and ints_ok and ok_t), results=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    ex = reference_examples()
    assert ex, "no reference checkout: set DPM_REFERENCE_DIR to a tree that holds examples/"
    rows = []
    for name, where, fn in SITES:
        row = run_site(name, where, fn, args.device, ex)
        rows.append(row)
        print("%-62s max rel-err %.3g  %s  %s" % (name, row["max_rel_err"], row["network_trace"],
                                                  "PASS" if row["passed"] else "FAIL"), flush=True)
    out = dict(what="the reference's example call sites, source files unchanged, once on their own vendored solver and once with "
                    "shims/ resolving the same import line to dpm_solver_amd; identical inputs, stand-in networks of "
                    "tests/golden/cases.py; rel_err = max |shim - own| / max |own|",
               tolerance=TOL, device=args.device,
               gpu=torch.cuda.get_device_name(0) if args.device.startswith("cuda") else None, torch=torch.__version__,
               sites=rows, passed=all(r["passed"] for r in rows))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
    assert out["passed"], [r["call_site"] for r in rows if not r["passed"]]


if __name__ == "__main__":
    main()
