"""numpy test double of `dpm_stage_launch` -- TEST INFRASTRUCTURE, never imported by the product.

`-m "not gpu"` tests monkeypatch `dpm_solver_amd.solver._launch_stage` with `launch_stage_double` so that
the complete host side (C planner coefficients, plan walking, buffer roles, wrapper batching, callbacks,
dtype policy) can be run on CPU tensors and compared with the goldens generated from the reference.
The arithmetic restates the device functions `prologue<>` / `combine<>` of
dpm_solver_amd/csrc/dpm_kernels.hip one to one, in np.float32; the GPU parity tests then only have to show
kernel == this double (they show more: kernel vs oracle and vs goldens directly).
"""
import numpy as np
import torch

from dpm_solver_amd import _lib as L
from oracle import dpm_oracle as O

F32 = np.float32


F64 = np.float64


def _np(t):
    if t is None:
        return None
    t = t.detach().cpu()
    return t.numpy() if t.dtype == torch.float64 else t.float().numpy()


class _Coef:
    """the scalars of a stage in the arithmetic type FT of the launch: np.float32 (the stage record's floats), or -- a
    double-precision state, dpm_f64.hip -- np.float64 from the dpm_stage_f64 record (coef64) when the plan is a
    double-precision one, else the record's floats converted exactly (torch's type promotion of fp32 scalars)"""

    def __init__(self, st, FT=F32, c64=None):
        src = c64 if (FT is F64 and c64 is not None) else st
        for f in ("alpha_e", "sigma_e", "cfg_scale", "cg_scale", "cx", "c0", "c1", "c2", "thr_ratio", "thr_max"):
            setattr(self, f, FT(getattr(src, f)))
        self.k = [FT(v) for v in src.k]
        self.model_type, self.guidance, self.flags, self.form = st.model_type, st.guidance, st.flags, st.form
        self.FT = FT


def _as_coef(st):
    return st if isinstance(st, _Coef) else _Coef(st)


def to_noise(o, xe, st):
    st = _as_coef(st)
    a, s = st.alpha_e, st.sigma_e
    if st.model_type == L.MODEL["x_start"]:
        return (xe - a * o) / s
    if st.model_type == L.MODEL["v"]:
        return a * o + s * xe
    if st.model_type == L.MODEL["score"]:
        return (-s) * o
    return o


def half_rounder(dtype):
    """fp32 numpy array -> the same values rounded through the half type `dtype` (torch dtype or DPM_DTYPE code); None for
    4- and 8-byte types"""
    if dtype in (torch.float16, L.DTYPE_F16):
        return lambda a: np.asarray(a, dtype=F32).astype(np.float16).astype(F32)
    if dtype in (torch.bfloat16, L.DTYPE_BF16):
        return lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=F32)).to(torch.bfloat16).float().numpy()
    return None


def prologue(st, xe, e0, e1, g, rnd=None):
    """rnd: rounding through the network output's half storage type (half_rounder) -- the classifier-free blend of a
    noise-prediction network runs on the network's own tensors in the reference (ref :326-330): three half operations"""
    st = _as_coef(st)
    FT = st.FT
    if st.guidance == L.GUIDE["classifier-free"]:
        nu, nc = to_noise(e1, xe, st), to_noise(e0, xe, st)
        if rnd is not None and st.model_type == L.MODEL["noise"] and FT is F32:
            eps = rnd(nu + rnd(st.cfg_scale * rnd(nc - nu)))
        else:
            eps = nu + st.cfg_scale * (nc - nu)
    elif st.guidance == L.GUIDE["classifier"]:
        eps = to_noise(e0, xe, st) - st.cg_scale * g
    else:
        eps = to_noise(e0, xe, st)
    if st.flags & L.F_TO_X0:
        return ((xe - st.sigma_e * eps) / st.alpha_e).astype(FT)
    return eps.astype(FT)


def combine(st, x, mn, h1, h2, rnd=None):
    """rnd: rounding through the network's half storage type (half_rounder) -- with a half-precision noise network in the
    noise-prediction form the reference's model values are half tensors and every difference of two of them is a half
    operation (ref :636-903); thresholding implies the data-prediction form, so never there"""
    st = _as_coef(st)
    F32 = st.FT                                  # (the literals below in the launch's arithmetic type)
    cx, c0, c1, c2 = st.cx, st.c0, st.c1, st.c2
    k = st.k
    hm = (rnd is not None and st.FT is np.float32 and not (st.flags & L.F_TO_X0) and st.model_type == L.MODEL["noise"]
          and st.guidance != L.GUIDE["classifier"])
    md = (lambda a, b: rnd(a - b)) if hm else (lambda a, b: a - b)
    if st.form == L.FORM_LIN1:
        return cx * x - c0 * mn
    if st.form == L.FORM_TWO:
        D = k[0] * md(mn, h1)
        P = h1 if (st.flags & L.F_BASE_HIST) else mn
        return (cx * x - c0 * P) - c1 * D
    if st.form == L.FORM_MS3:
        D1_0 = k[0] * md(mn, h1)
        D1_1 = k[1] * md(h1, h2)
        dd = D1_0 - D1_1
        D1 = D1_0 + k[2] * dd
        D2 = k[3] * dd
        return ((cx * x - c0 * mn) - c1 * D1) - c2 * D2
    if st.form == L.FORM_SS3T:
        D1_0 = k[0] * md(h2, h1)
        D1_1 = k[1] * md(mn, h1)
        D1 = (k[2] * D1_0 - k[3] * D1_1) / k[4]
        D2 = (F32(2.0) * (D1_1 - D1_0)) / k[4]
        return ((cx * x - c0 * h1) - c1 * D1) - c2 * D2
    if st.form == L.FORM_DENOISE:
        return mn
    raise AssertionError(st.form)


def blend(st_alpha, st_sigma, v, mask, a, b):
    """KExt epilogue / dpm_blend_launch: x*mask + (1 - mask)*(alpha*a + sigma*b), fp32, one rounding per op"""
    m = np.broadcast_to(mask.reshape((1,) * (v.ndim - mask.ndim) + mask.shape), v.shape) if mask.size != v.size \
        else mask.reshape(v.shape)
    r = a if b is None else F32(st_alpha) * a + F32(st_sigma) * b
    return (v * m + (F32(1.0) - m) * r).astype(F32)


def threshold64(x0, ratio, max_val):
    """dynamic thresholding in double (stage_thresh_kernel_f64; ref :416-425 with torch.quantile's semantics: ascending
    rank q (n - 1) in double, ATen's lerp)"""
    rows = x0.reshape(x0.shape[0], -1)
    srt = np.sort(np.abs(rows), axis=1)
    n = rows.shape[1]
    rank = F64(ratio) * F64(n - 1)
    lo, hi = int(np.floor(rank)), int(np.ceil(rank))
    w = rank - F64(lo)
    a, b = srt[:, lo], srt[:, hi]
    q = O.fma_rows(w, b - a, a, F64) if w < 0.5 else O.fma_rows(w - F64(1.0), b - a, b, F64)   # one fused multiply-add
    q = np.where(np.isnan(rows).any(axis=1), np.nan, q)        # torch.quantile: a row holding a NaN gives NaN
    s = np.maximum(q, F64(max_val)).reshape((-1,) + (1,) * (x0.ndim - 1))
    with np.errstate(invalid="ignore"):
        return np.minimum(np.maximum(x0, -s), s) / s


def launch_stage_double(st, x, xe, e0, e1, g, h1, h2, state_dtype, want_m=None, ext=None, opts=None, coef64=None):
    ref_t = x if x is not None else xe
    FT = F64 if state_dtype == torch.float64 else F32
    c = _Coef(st, FT, coef64.contents if coef64 is not None and hasattr(coef64, "contents") else coef64)
    cast = (lambda a: None if a is None else a.astype(FT)) if FT is F64 else (lambda a: a)
    xn, xen = cast(_np(x)), cast(_np(xe))
    if xen is None:
        xen = xn
    if xn is None:
        xn = xen
    ed = e0.dtype          # the eps dtype the launch binds (solver._launch_stage): converted to the state's when there is no kernel pair
    if ed not in (torch.float32, torch.float16, torch.bfloat16) or (state_dtype != torch.float32 and ed != state_dtype):
        ed = state_dtype
    if g is not None and g.dtype != ed and state_dtype == torch.float32:
        ed = torch.float32  # an fp32 classifier gradient next to a half network output: the output is widened (_device._launch_stage)
    # operands are brought to the launch's dtypes like _device._launch_stage does (`_conv`): the states and cached model
    # values to the state dtype, the network outputs to `ed` -- a rounding where that narrows (an explicit half state_dtype
    # meeting an fp32 x_T or an fp32 network), exact everywhere else
    to = lambda t, dt: None if t is None else (t if t.dtype == dt else t.to(dt))
    x, xe, h1, h2 = to(x, state_dtype), to(xe, state_dtype), to(h1, state_dtype), to(h2, state_dtype)
    e0, e1, g = to(e0, ed), to(e1, ed), to(g, ed)
    xn, xen = cast(_np(x)), cast(_np(xe))
    if xen is None:
        xen = xn
    if xn is None:
        xn = xen
    mn = prologue(c, xen, cast(_np(e0)), cast(_np(e1)), cast(_np(g)), half_rounder(ed))
    if st.flags & L.F_THRESH:
        mn = threshold64(mn, c.thr_ratio, c.thr_max) if FT is F64 else O.dynamic_threshold(mn, F32(st.thr_ratio), F32(st.thr_max))
    out = combine(c, xn, mn, cast(_np(h1)), cast(_np(h2)), half_rounder(ed)).astype(FT)
    store = bool(st.flags & L.F_STORE_M) if want_m is None else want_m
    conv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(state_dtype).reshape(ref_t.shape)
    if ext is not None and ext.get("blend") is not None:
        mask, period, ba, bb, alpha, sigma = ext["blend"]
        if FT is F64:
            m_, a_, b_ = _np(mask).astype(F64), _np(ba).astype(F64), (None if bb is None else _np(bb).astype(F64))
            mm = np.broadcast_to(m_.reshape((1,) * (out.ndim - m_.ndim) + m_.shape), out.shape) if m_.size != out.size else m_.reshape(out.shape)
            r = a_ if b_ is None else F64(alpha) * a_ + F64(sigma) * b_
            out = out * mm + (F64(1.0) - mm) * r
        else:
            out = blend(alpha, sigma, conv(out).float().numpy(), _np(mask), _np(ba), _np(bb))
    x_out = conv(out)
    if ext is not None and ext.get("dup"):
        ext["x2"] = torch.cat([x_out, x_out])
        x_out = ext["x2"][:x_out.shape[0]]
    return x_out, (conv(mn) if store else None)


def maskblend_apply_double(self, x, t_host, step):
    """numpy double of MaskBlend.apply (dpm_blend_launch)"""
    mask, period, ba, bb, alpha, sigma = self.operands(x.shape, x.dtype, x.device, t_host, step)
    out = blend(alpha, sigma, _np(x), _np(mask), _np(ba), _np(bb))
    return torch.from_numpy(np.ascontiguousarray(out)).to(x.dtype).reshape(x.shape)


def adaptive_error_double(x_lower, x_higher, x_prev, atol, rtol):
    """numpy double of dpm_adaptive_error_launch: per-sample RMS of (xh - xl)/delta (fp32 terms, double accumulation),
    then the batch maximum, returned as a 0-dim tensor"""
    if x_lower.shape[0] == 0:
        return torch.tensor(0.0, dtype=torch.float32)
    if x_lower.dtype is torch.float64:        # a double state: the product evaluates the reference's tensor expression itself
        delta = torch.max(torch.ones_like(x_lower) * atol, rtol * torch.max(torch.abs(x_lower), torch.abs(x_prev.to(x_lower.dtype))))
        v = ((x_higher - x_lower) / delta).reshape((x_lower.shape[0], -1))
        return torch.sqrt(torch.square(v).mean(dim=-1)).max()
    l, h, p = _np(x_lower), _np(x_higher), _np(x_prev)
    delta = np.maximum(F32(atol), F32(rtol) * np.maximum(np.abs(l), np.abs(p))).astype(F32)
    v = ((h - l) / delta).astype(F32)
    sq = (v * v).astype(F32).reshape(v.shape[0], -1).astype(np.float64)
    e = np.sqrt((sq.sum(axis=1) / sq.shape[1]).astype(F32)).astype(F32)
    return torch.tensor(float(e.max()), dtype=torch.float32)


def add_noise_double(sched_handle, x, noise, t_host):
    """numpy double of dpm_add_noise_launch (ref :1012-1030): alpha*x + sigma*noise, fp32, no fused multiply-add"""
    import ctypes as C
    if x.dtype is torch.float64:          # double tensors: the schedule in double at double times, else fp32 values promoted
        if t_host.dtype == np.float64:
            def ev(what):
                o = np.empty(len(t_host), dtype=F64)
                L.check(L.lib.dpm_schedule_eval_f64(sched_handle, what, t_host.ctypes.data_as(C.POINTER(C.c_double)), len(t_host),
                                                    o.ctypes.data_as(C.POINTER(C.c_double))))
                return o
        else:
            ev = lambda what: np.array([_eval1(sched_handle, what, t) for t in t_host], dtype=F32).astype(F64)
        a, s = ev(L.EVAL_ALPHA), ev(L.EVAL_STD)
        xn, nz = x.numpy(), noise.numpy()
        return torch.from_numpy(np.stack([a[j] * xn + s[j] * nz[j] for j in range(len(t_host))]))
    ev = lambda what: np.array([_eval1(sched_handle, what, t) for t in t_host], dtype=F32)
    a, s = ev(L.EVAL_ALPHA), ev(L.EVAL_STD)
    xn, nz = _np(x), _np(noise)
    out = np.stack([(a[j] * xn).astype(F32) + (s[j] * nz[j]).astype(F32) for j in range(len(t_host))]).astype(F32)
    return torch.from_numpy(out).to(x.dtype)


def _eval1(h, what, t):
    import ctypes as C
    i = np.array([t], dtype=F32)
    o = np.empty(1, dtype=F32)
    L.check(L.lib.dpm_schedule_eval(h, what, i.ctypes.data_as(C.POINTER(C.c_float)), 1, o.ctypes.data_as(C.POINTER(C.c_float))))
    return o[0]


# ------------------------------------------------------------------------------------------------
# pointer-level double of the C entry point dpm_stage_launch(dpm_stage*, dpm_buffers*, stream): what the prebuilt
# launch records of DPM_Solver's fast path call (solver._stage_launch_raw).  Works on the raw addresses of CPU tensors,
# so the buffer choreography (roles, history slots, duplicate store, strided outputs) is exercised exactly as the
# library sees it.
# ------------------------------------------------------------------------------------------------
import ctypes as _C

_SIZES = {L.DTYPE_F32: 4, L.DTYPE_F16: 2, L.DTYPE_BF16: 2, L.DTYPE_F64: 8}


def _rd(ptr, n, code):
    if not ptr:
        return None
    raw = np.frombuffer((_C.c_char * (n * _SIZES[code])).from_address(ptr), dtype=np.uint8)
    if code == L.DTYPE_F64:
        return raw.view(np.float64).copy()
    if code == L.DTYPE_F32:
        return raw.view(np.float32).copy()
    if code == L.DTYPE_F16:
        return raw.view(np.float16).astype(F32)
    return (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32).copy()


def _wr(ptr, arr, code):
    a = np.ascontiguousarray(arr, dtype=F64 if code == L.DTYPE_F64 else F32).reshape(-1)
    n = a.size
    dst = np.frombuffer((_C.c_char * (n * _SIZES[code])).from_address(ptr), dtype=np.uint8)
    if code == L.DTYPE_F64:
        dst.view(np.float64)[:] = a
    elif code == L.DTYPE_F32:
        dst.view(np.float32)[:] = a
    elif code == L.DTYPE_F16:
        dst.view(np.float16)[:] = a.astype(np.float16)
    else:
        dst.view(np.uint16)[:] = torch.from_numpy(a).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def launch_raw_double(st_ref, b_ref, stream):
    st, b = st_ref._obj, b_ref._obj
    n, B = int(b.n), int(b.batch)
    sd, ed = b.state_dtype, b.eps_dtype
    per = n // B

    def eps(ptr):
        if not ptr:
            return None
        if b.eps_stride and b.eps_stride != per:
            full = _rd(ptr, (B - 1) * int(b.eps_stride) + per, ed)
            return np.concatenate([full[i * int(b.eps_stride): i * int(b.eps_stride) + per] for i in range(B)])
        return _rd(ptr, n, ed)

    x, xe = _rd(b.x, n, sd), _rd(b.xe, n, sd)
    if xe is None:
        xe = x
    if x is None:
        x = xe
    FT = F64 if sd == L.DTYPE_F64 else F32
    assert FT is F32 or ed == L.DTYPE_F64, "a double state needs double network outputs"
    c = _Coef(st, FT, b.coef64.contents if b.coef64 else None)
    mn = prologue(c, xe, eps(b.e0), eps(b.e1), _rd(b.g, n, ed), half_rounder(ed))
    if st.flags & L.F_THRESH:
        if FT is F64:
            mn = threshold64(mn.reshape(B, per), c.thr_ratio, c.thr_max).reshape(-1)
        else:
            mn = O.dynamic_threshold(mn.reshape(B, per), F32(st.thr_ratio), F32(st.thr_max)).reshape(-1)
    assert not (st.flags & L.F_BLEND), "the fast path never carries a blend"
    out = combine(c, x, mn, _rd(b.h1, n, sd), _rd(b.h2, n, sd), half_rounder(ed)).astype(FT)
    _wr(b.x_out, out, sd)
    if b.x_out2:
        _wr(b.x_out2, out, sd)
    if st.flags & L.F_STORE_M:
        _wr(b.m_out, mn, sd)
    return 0


class _Ref:
    """stand-in for ctypes.byref(obj): launch_raw_double only looks at ._obj"""

    def __init__(self, obj):
        self._obj = obj


def launch_multi_double(st_ref, bufs, n_req, stream):
    """dpm_stage_launch_multi: the same stage of n_req requests, one after the other"""
    for r in range(int(n_req)):
        rc = launch_raw_double(st_ref, _Ref(bufs[r]), stream)
        if rc:
            return rc
    return 0


def install_cpu_double(monkeypatch, S, D):
    """route every device entry point of dpm_solver_amd.solver to its numpy double (CPU tensors)"""
    monkeypatch.setattr(S, "_launch_stage", launch_stage_double)
    monkeypatch.setattr(S, "_stage_launch_raw", launch_raw_double)
    monkeypatch.setattr(S, "_stage_launch_multi_raw", launch_multi_double)
    monkeypatch.setattr(S, "_launch_ctx", lambda dev: (None, 0, False, False))
    monkeypatch.setattr(S, "_require_gpu", lambda x: None)
    monkeypatch.setattr(D.MaskBlend, "apply", maskblend_apply_double)
    monkeypatch.setattr(S, "_adaptive_error", adaptive_error_double)
    monkeypatch.setattr(S, "_add_noise", add_noise_double)
