"""Per-launch differential (debugging aid of tools/fuzz_gpu_api.py): one case of its seeded sequence runs on the GPU with every
DV._launch_stage call repeated on the numpy double (tests/kernel_double.py) from the SAME inputs -- the first launch that
differs is printed with its stage record, dtypes and first elements.

    python tools/launch_diff.py SEED CASE
"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import dpm_solver_amd as D
import dpm_solver_amd._device as DV
import dpm_solver_amd.solver as S
from dpm_solver_amd import _lib as L
import kernel_double as KD
import fuzz_gpu_api as A
import fuzz_gpu as FG

seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
cfgs = [A.random_case(rng) for _ in range(idx + 1)]
cfg = cfgs[idx]
print({k: v for k, v in cfg.items() if k != "seed"})
orig = DV._launch_stage
n = [0]
def cpu(t):
    return None if t is None else t.detach().cpu()
def wrapped(st, x, xe, e0, e1, g, h1, h2, state_dtype, want_m=None, ext=None, opts=None, coef64=None):
    st_copy = st.copy() if hasattr(st, "copy") else st
    ext_c = None
    if ext is not None:
        ext_c = {k: (tuple(cpu(v) if torch.is_tensor(v) else v for v in val) if isinstance(val, tuple) else val) for k, val in ext.items()}
    args_c = [cpu(v) for v in (x, xe, e0, e1, g, h1, h2)]
    out = orig(st, x, xe, e0, e1, g, h1, h2, state_dtype, want_m=want_m, ext=ext, opts=opts, coef64=coef64)
    want = KD.launch_stage_double(st_copy, *args_c, state_dtype, want_m=want_m, ext=ext_c, opts=None, coef64=coef64)
    n[0] += 1
    for name, a, b in (("x_out", out[0], want[0]), ("m_out", out[1], want[1])):
        if a is None and b is None:
            continue
        a = a.cpu()
        same = torch.equal(a, b)
        d = float((a.double() - b.double()).abs().max())
        print("launch %d %s form %d flags %#x model %d guid %d sd %s eps %s x %s xe %s shape %s: %s max|d| %.3g of %.3g" % (
            n[0], name, st_copy.form, st_copy.flags, st_copy.model_type, st_copy.guidance, str(state_dtype)[6:], str(e0.dtype)[6:],
            None if x is None else str(x.dtype)[6:], None if xe is None else str(xe.dtype)[6:], tuple(a.shape), "same" if same else "DIFFERENT", d, float(b.double().abs().max())))
        if not same and n[0] <= 3:
            print("   gpu   ", a.flatten()[:8].tolist())
            print("   double", b.flatten()[:8].tolist())
            print("   coef cx %r c0 %r c1 %r c2 %r k %r alpha_e %r sigma_e %r cfg %r" % (st_copy.cx, st_copy.c0, st_copy.c1, st_copy.c2, list(st_copy.k), st_copy.alpha_e, st_copy.sigma_e, st_copy.cfg_scale))
            for nm, v in zip(("x", "xe", "e0", "e1", "h1", "h2"), (args_c[0], args_c[1], args_c[2], args_c[3], args_c[5], args_c[6])):
                if v is not None:
                    print("   ", nm, v.dtype, v.flatten()[:8].tolist())
    return out
DV._launch_stage = wrapped
S._launch_stage = wrapped
r = A.run(cfg, "cuda:0")
print(r[0], r[1] if r[0] == "raise" else "")
if r[0] == "raise": print(r[3])
