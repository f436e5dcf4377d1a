"""CPU oracle for the DPM-Solver / DPM-Solver++ sampling path  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the algorithm in the reference
`dpm_solver_pytorch.py` (LuChengTHU/dpm-solver).  It exists so that the HIP engine in
`dpm_solver_amd/` can be checked on the GPU box, where the reference is not present.

  * Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
    it -- as the checker / the timed CPU baseline, never as part of the product path.
  * Parity pin: every function here is checked against golden vectors produced by running the
    real reference (tests/golden/make_golden.py -> tests/golden/*.npz) in
    tests/test_oracle_golden.py.  The reference itself ships no tests or golden vectors for
    this path, so those generated fixtures ARE the pin.

Numerics.  The reference evaluates everything in fp32: the state tensors, and also every scalar
of the noise schedule (alpha_t, sigma_t, lambda_t, h, phi_k ...), which it carries as 0-dim or
(1,)-shaped fp32 tensors.  The oracle follows the same operation order in np.float32.
Elementary functions (exp, log, expm1, log1p, sqrt) are evaluated in float64 and rounded once to
fp32, i.e. correctly rounded fp32 functions; torch's CPU kernels (SLEEF, <=1 ulp) may differ from
that by one ulp on individual scalars, which bounds oracle-vs-reference agreement at ~1e-6
relative rather than bit-exact.  Tensor arithmetic (+,-,*,/) is IEEE and bit-identical.

Citations `ref:NNN` are line numbers in /root/reference/dpm_solver_pytorch.py.
"""
import ctypes
import ctypes.util
import math

import numpy as np

F32 = np.float32
F64 = np.float64

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.fmaf.restype, _libm.fmaf.argtypes = ctypes.c_float, [ctypes.c_float] * 3
_libm.fma.restype, _libm.fma.argtypes = ctypes.c_double, [ctypes.c_double] * 3


def fma_rows(a, b, c, dtype=F32):
    """a * b + c with ONE rounding (the C library's fmaf / fma), elementwise over broadcast operands"""
    f = _libm.fmaf if dtype is F32 else _libm.fma
    a, b, c = np.broadcast_arrays(np.asarray(a, dtype), np.asarray(b, dtype), np.asarray(c, dtype))
    out = np.empty(a.shape, dtype)
    of, af, bf, cf = out.reshape(-1), a.reshape(-1), b.reshape(-1), c.reshape(-1)
    for i in range(of.size):
        of[i] = f(af[i], bf[i], cf[i])
    return out


# --------------------------------------------------------------------------------------------
# correctly-rounded fp32 elementary functions (scalar or array)
# --------------------------------------------------------------------------------------------
def _f(v):
    return np.asarray(v, dtype=F32)


def exp32(v):
    return np.exp(_f(v).astype(F64)).astype(F32)


def log32(v):
    return np.log(_f(v).astype(F64)).astype(F32)


def expm1_32(v):
    return np.expm1(_f(v).astype(F64)).astype(F32)


def log1p_32(v):
    return np.log1p(_f(v).astype(F64)).astype(F32)


def cos32(v):
    return np.cos(_f(v).astype(F64)).astype(F32)


def arccos32(v):
    return np.arccos(_f(v).astype(F64)).astype(F32)


def sqrt32(v):
    return np.sqrt(_f(v))  # IEEE sqrt is correctly rounded in fp32 already


def logaddexp32(a, b):
    """torch.logaddexp in fp32: max(a,b) + log1p(exp(-|a-b|))."""
    a, b = _f(a), _f(b)
    m = np.maximum(a, b)
    return (m + log1p_32(exp32(-np.abs(a - b)))).astype(F32)


def linspace32(start, end, n):
    """torch.linspace(start, end, n) for fp32 on CPU: fp32 step, symmetric fill from both ends,
    each point one fused multiply-add (verified bitwise against torch in the tests).
    Used at ref:107, ref:471, ref:474, ref:477."""
    start, end = F32(start), F32(end)
    if n == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(n - 1))
    out = np.empty(n, dtype=F32)
    half = n // 2
    for i in range(n):
        if i < half:
            out[i] = F32(F64(start) + F64(step) * F64(i))       # fmaf(step, i, start): exact product, one rounding
        else:
            out[i] = F32(F64(end) - F64(step) * F64(n - i - 1))
    return out


def interp32(x, xp, yp):
    """Piecewise-linear y(x) through (xp, yp), xp ascending, the two outermost segments extended
    to infinity.  Restates interpolate_fn (ref:1253-1292), which finds the bracketing segment
    with a sort of [x, xp]; a binary search gives the same segment."""
    x = _f(x)
    K = xp.shape[0]
    idx = np.searchsorted(xp, x, side="left")
    i0 = np.where(idx == 0, 0, np.where(idx == K, K - 2, idx - 1))
    i1 = i0 + 1
    x0, x1, y0, y1 = xp[i0], xp[i1], yp[i0], yp[i1]
    return (y0 + (x - x0) * (y1 - y0) / (x1 - x0)).astype(F32)


# --------------------------------------------------------------------------------------------
# NoiseScheduleVP (ref:6-167)
# --------------------------------------------------------------------------------------------
class Schedule:
    def __init__(self, kind, log_alpha=None, beta_0=0.1, beta_1=20.0):
        self.kind = kind
        self.T = 1.0
        if kind == "discrete":
            self.log_alpha = np.ascontiguousarray(log_alpha, dtype=F32)
            self.total_N = int(self.log_alpha.shape[0])                      # ref:106
            self.t_arr = linspace32(0.0, 1.0, self.total_N + 1)[1:].copy()   # ref:107
        else:
            self.total_N = 1000                                              # ref:110
            self.beta_0, self.beta_1 = beta_0, beta_1
            if kind == "cosine":
                # older vendored revision (examples/score_sde_pytorch/dpm_solver.py:114-124 = "legacy:N" below)
                self.cosine_s = 0.008
                self.cosine_log_alpha_0 = math.log(math.cos(self.cosine_s / (1. + self.cosine_s) * math.pi / 2.))
                self.T = 0.9946

    # ---- constructors ---------------------------------------------------------------
    @staticmethod
    def from_betas(betas, clip=True):
        """ref:100  log_alphas = 0.5 * log(1 - betas).cumsum(0); torch's CPU cumsum accumulates
        fp32 inputs in double and rounds each prefix to fp32.  clip=False: the older vendored revision (legacy:106)."""
        b = np.asarray(betas)
        if b.dtype == np.float64:
            la = 0.5 * np.cumsum(np.log(1.0 - b))
        else:
            l = log32(F32(1.0) - b.astype(F32))
            la = F32(0.5) * np.cumsum(l.astype(F64)).astype(F32)
        return Schedule("discrete", (Schedule._clip(la) if clip else la).astype(F32))

    @staticmethod
    def from_alphas_cumprod(ac, clip=True):
        a = np.asarray(ac)                                                   # ref:103
        la = 0.5 * np.log(a) if a.dtype == np.float64 else F32(0.5) * log32(a.astype(F32))
        return Schedule("discrete", (Schedule._clip(la) if clip else la).astype(F32))

    @staticmethod
    def cosine():
        return Schedule("cosine")

    @staticmethod
    def linear(beta_0=0.1, beta_1=20.0):
        return Schedule("linear", beta_0=beta_0, beta_1=beta_1)

    @staticmethod
    def _clip(la, clipped_lambda=-5.1):
        """numerical_clip_alpha (ref:114-125): drop the tail whose half-logSNR is below -5.1."""
        if la.dtype == np.float64:
            ls = 0.5 * np.log(1.0 - np.exp(2.0 * la))
            cl = clipped_lambda
        else:
            ls = F32(0.5) * log32(F32(1.0) - exp32(F32(2.0) * la))
            cl = F32(clipped_lambda)
        lam = la - ls
        idx = int(np.searchsorted(lam[::-1], cl, side="left"))
        return la[:-idx] if idx > 0 else la

    # ---- marginals --------------------------------------------------------------------
    def log_alpha_t(self, t):
        """marginal_log_mean_coeff (ref:127-134)."""
        t = _f(t)
        if self.kind == "discrete":
            return interp32(t.reshape(-1), self.t_arr, self.log_alpha)
        if self.kind == "cosine":                                            # legacy:135-137
            a = (((t + F32(self.cosine_s)) / F32(1. + self.cosine_s)) * F32(math.pi)) / F32(2.0)
            return (log32(cos32(a)) - F32(self.cosine_log_alpha_0)).astype(F32)
        b0, b1 = self.beta_0, self.beta_1
        return (F32(-0.25) * (t * t) * F32(b1 - b0) - F32(0.5) * t * F32(b0)).astype(F32)

    def alpha(self, t):
        return exp32(self.log_alpha_t(t))                                    # ref:140

    def std(self, t):
        return sqrt32(F32(1.0) - exp32(F32(2.0) * self.log_alpha_t(t)))      # ref:146

    def lam(self, t):
        la = self.log_alpha_t(t)                                             # ref:152-154
        return (la - F32(0.5) * log32(F32(1.0) - exp32(F32(2.0) * la))).astype(F32)

    def inv_lam(self, lam):
        """inverse_lambda (ref:156-167)."""
        lam = _f(lam)
        if self.kind == "linear":
            b0, b1 = self.beta_0, self.beta_1
            tmp = F32(2.0 * (b1 - b0)) * logaddexp32(F32(-2.0) * lam, F32(0.0))
            delta = F32(b0 ** 2) + tmp
            return (tmp / (sqrt32(delta) + F32(b0)) / F32(b1 - b0)).astype(F32)
        if self.kind == "cosine":                                            # legacy:171-175
            la = F32(-0.5) * logaddexp32(F32(-2.0) * lam, F32(0.0))
            ac = arccos32(exp32(la + F32(self.cosine_log_alpha_0)))
            return ((((ac * F32(2.0)) * F32(1. + self.cosine_s)) / F32(math.pi)) - F32(self.cosine_s)).astype(F32)
        la = F32(-0.5) * logaddexp32(F32(0.0), F32(-2.0) * lam)
        return interp32(la.reshape(-1), self.log_alpha[::-1].copy(), self.t_arr[::-1].copy())


# --------------------------------------------------------------------------------------------
# time grids (ref:453-539)
# --------------------------------------------------------------------------------------------
def time_steps(sch, skip_type, t_T, t_0, N):
    if skip_type == "logSNR":
        lam_T = float(sch.lam(F32(t_T)).reshape(-1)[0])                      # ref:469-471
        lam_0 = float(sch.lam(F32(t_0)).reshape(-1)[0])
        return sch.inv_lam(linspace32(lam_T, lam_0, N + 1)).reshape(-1)
    if skip_type == "time_uniform":
        return linspace32(t_T, t_0, N + 1)                                   # ref:474
    if skip_type == "time_quadratic":
        g = linspace32(t_T ** 0.5, t_0 ** 0.5, N + 1)                         # ref:477
        return (g * g).astype(F32)
    raise ValueError("Unsupported skip_type {}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'".format(skip_type))


def singlestep_orders(steps, order):
    """ref:514-533"""
    if order == 3:
        K = steps // 3 + 1
        if steps % 3 == 0:
            return [3] * (K - 2) + [2, 1], K
        if steps % 3 == 1:
            return [3] * (K - 1) + [1], K
        return [3] * (K - 1) + [2], K
    if order == 2:
        if steps % 2 == 0:
            return [2] * (steps // 2), steps // 2
        return [2] * (steps // 2) + [1], steps // 2 + 1
    if order == 1:
        return [1] * steps, steps
    raise ValueError("'order' must be '1' or '2' or '3'.")


def singlestep_grid(sch, steps, order, skip_type, t_T, t_0):
    orders, K = singlestep_orders(steps, order)
    if skip_type == "logSNR":
        outer = time_steps(sch, skip_type, t_T, t_0, K)                      # ref:536
    else:
        full = time_steps(sch, skip_type, t_T, t_0, steps)                   # ref:538
        outer = full[np.cumsum([0] + orders)]
    return outer, orders


# --------------------------------------------------------------------------------------------
# model_wrapper (ref:170-334)
# --------------------------------------------------------------------------------------------
def _b(v, x):
    return np.asarray(v, dtype=F32).reshape((-1,) + (1,) * (x.ndim - 1))


def wrap_model(model, sch, model_type="noise", guidance_type="uncond", condition=None,
               unconditional_condition=None, guidance_scale=1.0, cond_grad_fn=None):
    """Returns model_fn(x, t_vec) -> noise, t_vec of shape (B,).  `cond_grad_fn(x, t_input, cond)` stands in
    for the autograd gradient of the classifier log-probability (ref:300-307)."""

    def t_input_of(tc):
        if sch.kind == "discrete":
            return ((tc - F32(1.0 / sch.total_N)) * F32(1000.0)).astype(F32)   # ref:278
        return tc

    def noise_pred(x, tc, cond=None):
        out = model(x, t_input_of(tc), cond) if cond is not None else model(x, t_input_of(tc))
        if model_type == "noise":
            return out
        if model_type == "x_start":                                           # ref:290-292
            return ((x - _b(sch.alpha(tc), x) * out) / _b(sch.std(tc), x)).astype(F32)
        if model_type == "v":                                                 # ref:293-295
            return (_b(sch.alpha(tc), x) * out + _b(sch.std(tc), x) * x).astype(F32)
        if model_type == "score":                                             # ref:296-298
            return (-_b(sch.std(tc), x) * out).astype(F32)
        raise AssertionError(model_type)

    def model_fn(x, tc):
        tc = _f(tc)
        if guidance_type == "uncond":
            return noise_pred(x, tc)
        if guidance_type == "classifier":                                     # ref:315-321
            g = cond_grad_fn(x, t_input_of(tc), condition)
            noise = noise_pred(x, tc)
            return (noise - F32(guidance_scale) * _b(sch.std(tc), x) * g).astype(F32)
        if guidance_type == "classifier-free":                                # ref:322-330
            if guidance_scale == 1.0 or unconditional_condition is None:
                return noise_pred(x, tc, cond=condition)
            x_in = np.concatenate([x, x])
            t_in = np.concatenate([tc, tc])
            c_in = np.concatenate([unconditional_condition, condition])
            both = noise_pred(x_in, t_in, cond=c_in)
            nu, nc = both[: x.shape[0]], both[x.shape[0]:]
            if both.dtype == np.float16 and model_type == "noise":
                # a half-precision noise network: the reference blends on the network's own tensors (ref :330) -- three
                # half operations, each computed in fp32 and rounded once (torch's opmath; the Python-float scale is an fp32
                # scalar there, NOT rounded to half).  The result stays a half tensor, like the reference's.
                F16 = np.float16
                d = (nc.astype(F32) - nu.astype(F32)).astype(F16)
                p = (F32(guidance_scale) * d.astype(F32)).astype(F16)
                return (nu.astype(F32) + p.astype(F32)).astype(F16)
            return (nu + F32(guidance_scale) * (nc - nu)).astype(F32)
        raise AssertionError(guidance_type)

    return model_fn


# --------------------------------------------------------------------------------------------
# dynamic thresholding (ref:416-425)
# --------------------------------------------------------------------------------------------
def quantile_rows32(a, q):
    """torch.quantile(a, q, dim=1) for fp32 `a` [B, n], linear interpolation.  The fractional rank is
    computed in fp32 (q is materialised as an fp32 tensor), which matters: for n = 12288, q = 0.995
    the fp32 rank is 12225.5654296875, not 12225.565.  The interpolation is ATen's lerp, which is one fused multiply-add
    on the CPU (vec::fmadd(w or w - 1, hi - lo, lo or hi), whole vectors and tails alike) and on the device (the compiler
    contracts Lerp.h's two-operation form): checked against torch.quantile over 10^5 random rows in
    tests/test_oracle_golden.py."""
    n = a.shape[1]
    s = np.sort(a, axis=1)
    rank = F32(q) * F32(n - 1)
    lo = int(np.floor(rank))
    hi = int(np.ceil(rank))
    w = F32(rank - F32(lo))
    lo_v, hi_v = s[:, lo], s[:, hi]
    with np.errstate(invalid="ignore"):
        d = hi_v - lo_v                                                       # (inf - inf = NaN, like the reference)
    if w < F32(0.5):                                                          # ATen lerp
        out = fma_rows(w, d, lo_v)
    else:
        out = fma_rows(F32(w - F32(1.0)), d, hi_v)
    out[np.isnan(a).any(axis=1)] = np.nan                                     # torch.quantile: a row holding a NaN gives NaN
    return out


def dynamic_threshold(x0, ratio=0.995, max_val=1.0):
    B = x0.shape[0]
    s = quantile_rows32(np.abs(x0).reshape(B, -1), ratio)
    s = np.maximum(s, F32(max_val))                                           # (NaN-propagating, like torch.maximum)
    sb = _b(s, x0)
    with np.errstate(invalid="ignore"):
        return (np.minimum(np.maximum(x0, -sb), sb) / sb).astype(F32)         # torch.clamp = min(max(x, lo), hi), NaNs kept


# --------------------------------------------------------------------------------------------
# DPM_Solver (ref:337-1245)
# --------------------------------------------------------------------------------------------
class Solver:
    def __init__(self, model_fn, sch, algorithm_type="dpmsolver++", correcting_x0_fn=None,
                 correcting_xt_fn=None, thresholding_max_val=1.0, dynamic_thresholding_ratio=0.995):
        assert algorithm_type in ("dpmsolver", "dpmsolver++")
        self.net = model_fn
        self.sch = sch
        self.pp = algorithm_type == "dpmsolver++"
        if correcting_x0_fn == "dynamic_thresholding":
            correcting_x0_fn = lambda x0, t: dynamic_threshold(x0, dynamic_thresholding_ratio, thresholding_max_val)
        self.cx0 = correcting_x0_fn
        self.cxt = correcting_xt_fn
        self.nfe = 0

    # ---- model value: eps, or x0 for dpmsolver++ (ref:427-451) --------------------------
    def noise_pred(self, x, t):
        self.nfe += 1
        tv = np.full((x.shape[0],), F32(np.asarray(t).reshape(-1)[0]), dtype=F32)   # t.expand(B), ref:404
        return self.net(x, tv)

    def data_pred(self, x, t):
        noise = self.noise_pred(x, t)
        a, s = self._sc(self.sch.alpha(t)), self._sc(self.sch.std(t))
        x0 = ((x - s * noise) / a).astype(F32)                                 # ref:439
        if self.cx0 is not None:
            x0 = self.cx0(x0, t)
        return x0

    def model_value(self, x, t):
        return self.data_pred(x, t) if self.pp else self.noise_pred(x, t)

    @staticmethod
    def _sc(v):
        return F32(np.asarray(v).reshape(-1)[0])

    def _marg(self, t):
        s = self.sch
        return self._sc(s.lam(t)), self._sc(s.log_alpha_t(t)), self._sc(s.std(t))

    # ---- order 1 (ref:547-592) ----------------------------------------------------------
    def first_update(self, x, s, t, model_s=None):
        lam_s, la_s, sig_s = self._marg(s)
        lam_t, la_t, sig_t = self._marg(t)
        h = F32(lam_t - lam_s)
        if model_s is None:
            model_s = self.model_value(x, s)
        if self.pp:
            phi_1 = self._sc(expm1_32(-h))
            x_t = (sig_t / sig_s) * x - (self._sc(exp32(la_t)) * phi_1) * model_s
        else:
            phi_1 = self._sc(expm1_32(h))
            x_t = self._sc(exp32(la_t - la_s)) * x - (sig_t * phi_1) * model_s
        return x_t.astype(F32), dict(model_s=model_s)

    # ---- singlestep order 2 (ref:594-673) -----------------------------------------------
    def ss2_update(self, x, s, t, r1=None, model_s=None, solver_type="dpmsolver"):
        if solver_type not in ("dpmsolver", "taylor"):
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        r1 = 0.5 if r1 is None else r1
        sch = self.sch
        lam_s, la_s, sig_s = self._marg(s)
        lam_t, la_t, sig_t = self._marg(t)
        h = F32(lam_t - lam_s)
        s1 = sch.inv_lam(F32(lam_s + r1 * h))
        _, la_s1, sig_s1 = self._marg(s1)
        a_s1, a_t = self._sc(exp32(la_s1)), self._sc(exp32(la_t))
        if model_s is None:
            model_s = self.model_value(x, s)
        if self.pp:
            phi_11 = self._sc(expm1_32(-r1 * h))
            phi_1 = self._sc(expm1_32(-h))
            x_s1 = ((sig_s1 / sig_s) * x - (a_s1 * phi_11) * model_s).astype(F32)
            model_s1 = self.model_value(x_s1, s1)
            if solver_type == "dpmsolver":
                x_t = (sig_t / sig_s) * x - (a_t * phi_1) * model_s \
                    - (0.5 / r1) * (a_t * phi_1) * (model_s1 - model_s)
            else:
                x_t = (sig_t / sig_s) * x - (a_t * phi_1) * model_s \
                    + (1.0 / r1) * (a_t * (phi_1 / h + F32(1.0))) * (model_s1 - model_s)
        else:
            phi_11 = self._sc(expm1_32(r1 * h))
            phi_1 = self._sc(expm1_32(h))
            x_s1 = (self._sc(exp32(la_s1 - la_s)) * x - (sig_s1 * phi_11) * model_s).astype(F32)
            model_s1 = self.model_value(x_s1, s1)
            e_t = self._sc(exp32(la_t - la_s))
            if solver_type == "dpmsolver":
                x_t = e_t * x - (sig_t * phi_1) * model_s - (0.5 / r1) * (sig_t * phi_1) * (model_s1 - model_s)
            else:
                x_t = e_t * x - (sig_t * phi_1) * model_s \
                    - (1.0 / r1) * (sig_t * (phi_1 / h - F32(1.0))) * (model_s1 - model_s)
        return x_t.astype(F32), dict(model_s=model_s, model_s1=model_s1)

    # ---- singlestep order 3 (ref:675-794) -----------------------------------------------
    def ss3_update(self, x, s, t, r1=None, r2=None, model_s=None, model_s1=None, solver_type="dpmsolver"):
        if solver_type not in ("dpmsolver", "taylor"):
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        r1 = 1.0 / 3.0 if r1 is None else r1
        r2 = 2.0 / 3.0 if r2 is None else r2
        sch = self.sch
        lam_s, la_s, sig_s = self._marg(s)
        lam_t, la_t, sig_t = self._marg(t)
        h = F32(lam_t - lam_s)
        s1 = sch.inv_lam(F32(lam_s + r1 * h))
        s2 = sch.inv_lam(F32(lam_s + r2 * h))
        _, la_s1, sig_s1 = self._marg(s1)
        _, la_s2, sig_s2 = self._marg(s2)
        a_s1, a_s2, a_t = (self._sc(exp32(v)) for v in (la_s1, la_s2, la_t))
        one, half = F32(1.0), F32(0.5)
        if self.pp:
            phi_11 = self._sc(expm1_32(-r1 * h))
            phi_12 = self._sc(expm1_32(-r2 * h))
            phi_1 = self._sc(expm1_32(-h))
            phi_22 = F32(self._sc(expm1_32(-r2 * h)) / (r2 * h) + one)
            phi_2 = F32(phi_1 / h + one)
            phi_3 = F32(phi_2 / h - half)
        else:
            phi_11 = self._sc(expm1_32(r1 * h))
            phi_12 = self._sc(expm1_32(r2 * h))
            phi_1 = self._sc(expm1_32(h))
            phi_22 = F32(self._sc(expm1_32(r2 * h)) / (r2 * h) - one)
            phi_2 = F32(phi_1 / h - one)
            phi_3 = F32(phi_2 / h - half)
        if model_s is None:
            model_s = self.model_value(x, s)
        if self.pp:
            if model_s1 is None:
                x_s1 = ((sig_s1 / sig_s) * x - (a_s1 * phi_11) * model_s).astype(F32)
                model_s1 = self.model_value(x_s1, s1)
            x_s2 = ((sig_s2 / sig_s) * x - (a_s2 * phi_12) * model_s
                    + r2 / r1 * (a_s2 * phi_22) * (model_s1 - model_s)).astype(F32)
            model_s2 = self.model_value(x_s2, s2)
            base = (sig_t / sig_s) * x - (a_t * phi_1) * model_s
            if solver_type == "dpmsolver":
                x_t = base + (1.0 / r2) * (a_t * phi_2) * (model_s2 - model_s)
            else:
                D1_0 = (1.0 / r1) * (model_s1 - model_s)
                D1_1 = (1.0 / r2) * (model_s2 - model_s)
                D1 = (r2 * D1_0 - r1 * D1_1) / (r2 - r1)
                D2 = 2.0 * (D1_1 - D1_0) / (r2 - r1)
                x_t = base + (a_t * phi_2) * D1 - (a_t * phi_3) * D2
        else:
            if model_s1 is None:
                x_s1 = (self._sc(exp32(la_s1 - la_s)) * x - (sig_s1 * phi_11) * model_s).astype(F32)
                model_s1 = self.model_value(x_s1, s1)
            x_s2 = (self._sc(exp32(la_s2 - la_s)) * x - (sig_s2 * phi_12) * model_s
                    - r2 / r1 * (sig_s2 * phi_22) * (model_s1 - model_s)).astype(F32)
            model_s2 = self.model_value(x_s2, s2)
            base = self._sc(exp32(la_t - la_s)) * x - (sig_t * phi_1) * model_s
            if solver_type == "dpmsolver":
                x_t = base - (1.0 / r2) * (sig_t * phi_2) * (model_s2 - model_s)
            else:
                D1_0 = (1.0 / r1) * (model_s1 - model_s)
                D1_1 = (1.0 / r2) * (model_s2 - model_s)
                D1 = (r2 * D1_0 - r1 * D1_1) / (r2 - r1)
                D2 = 2.0 * (D1_1 - D1_0) / (r2 - r1)
                x_t = base - (sig_t * phi_2) * D1 - (sig_t * phi_3) * D2
        return x_t.astype(F32), dict(model_s=model_s, model_s1=model_s1, model_s2=model_s2)

    # ---- multistep order 2 (ref:796-852) -- the north-star "2M" update -------------------
    def ms2_update(self, x, m_list, t_list, t, solver_type="dpmsolver"):
        if solver_type not in ("dpmsolver", "taylor"):
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        m1, m0 = m_list[-2], m_list[-1]
        lam_p1, _, _ = self._marg(t_list[-2])
        lam_p0, la_p0, sig_p0 = self._marg(t_list[-1])
        lam_t, la_t, sig_t = self._marg(t)
        a_t = self._sc(exp32(la_t))
        h_0 = F32(lam_p0 - lam_p1)
        h = F32(lam_t - lam_p0)
        r0 = F32(h_0 / h)
        D1_0 = (1.0 / r0) * (m0 - m1)
        if self.pp:
            phi_1 = self._sc(expm1_32(-h))
            if solver_type == "dpmsolver":
                x_t = (sig_t / sig_p0) * x - (a_t * phi_1) * m0 - 0.5 * (a_t * phi_1) * D1_0
            else:
                x_t = (sig_t / sig_p0) * x - (a_t * phi_1) * m0 + (a_t * (phi_1 / h + F32(1.0))) * D1_0
        else:
            phi_1 = self._sc(expm1_32(h))
            e_t = self._sc(exp32(la_t - la_p0))
            if solver_type == "dpmsolver":
                x_t = e_t * x - (sig_t * phi_1) * m0 - 0.5 * (sig_t * phi_1) * D1_0
            else:
                x_t = e_t * x - (sig_t * phi_1) * m0 - (sig_t * (phi_1 / h - F32(1.0))) * D1_0
        return x_t.astype(F32)

    # ---- multistep order 3 (ref:854-904) ------------------------------------------------
    def ms3_update(self, x, m_list, t_list, t, solver_type="dpmsolver"):
        m2, m1, m0 = m_list
        lam_p2, _, _ = self._marg(t_list[0])
        lam_p1, _, _ = self._marg(t_list[1])
        lam_p0, la_p0, sig_p0 = self._marg(t_list[2])
        lam_t, la_t, sig_t = self._marg(t)
        a_t = self._sc(exp32(la_t))
        h_1 = F32(lam_p1 - lam_p2)
        h_0 = F32(lam_p0 - lam_p1)
        h = F32(lam_t - lam_p0)
        r0, r1 = F32(h_0 / h), F32(h_1 / h)
        D1_0 = (1.0 / r0) * (m0 - m1)
        D1_1 = (1.0 / r1) * (m1 - m2)
        D1 = D1_0 + (r0 / (r0 + r1)) * (D1_0 - D1_1)
        D2 = (1.0 / (r0 + r1)) * (D1_0 - D1_1)
        if self.pp:
            phi_1 = self._sc(expm1_32(-h))
            phi_2 = F32(phi_1 / h + F32(1.0))
            phi_3 = F32(phi_2 / h - F32(0.5))
            x_t = (sig_t / sig_p0) * x - (a_t * phi_1) * m0 + (a_t * phi_2) * D1 - (a_t * phi_3) * D2
        else:
            phi_1 = self._sc(expm1_32(h))
            phi_2 = F32(phi_1 / h - F32(1.0))
            phi_3 = F32(phi_2 / h - F32(0.5))
            x_t = self._sc(exp32(la_t - la_p0)) * x - (sig_t * phi_1) * m0 - (sig_t * phi_2) * D1 - (sig_t * phi_3) * D2
        return x_t.astype(F32)

    def multistep_update(self, x, m_list, t_list, t, order, solver_type="dpmsolver"):
        if order == 1:
            return self.first_update(x, t_list[-1], t, model_s=m_list[-1])[0]
        if order == 2:
            return self.ms2_update(x, m_list, t_list, t, solver_type)
        if order == 3:
            return self.ms3_update(x, m_list, t_list, t, solver_type)
        raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))

    def singlestep_update(self, x, s, t, order, solver_type="dpmsolver", r1=None, r2=None):
        if order == 1:
            return self.first_update(x, s, t)[0]
        if order == 2:
            return self.ss2_update(x, s, t, r1=r1, solver_type=solver_type)[0]
        if order == 3:
            return self.ss3_update(x, s, t, r1=r1, r2=r2, solver_type=solver_type)[0]
        raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))

    # ---- adaptive DPM-Solver-12 / -23 (ref:956-1010) --------------------------------------
    def adaptive(self, x, order, t_T, t_0, h_init=0.05, atol=0.0078, rtol=0.05, theta=0.9, t_err=1e-5,
                 solver_type="dpmsolver"):
        sch = self.sch
        s = F32(t_T)
        lam_s = self._sc(sch.lam(s))
        lam_0 = self._sc(sch.lam(F32(t_0)))
        h = F32(h_init)
        x_prev = x
        nfe = 0
        if order == 2:
            lower = lambda x, s, t: self.first_update(x, s, t)
            higher = lambda x, s, t, **kw: self.ss2_update(x, s, t, r1=0.5, solver_type=solver_type, **kw)[0]
        elif order == 3:
            r1, r2 = 1.0 / 3.0, 2.0 / 3.0
            lower = lambda x, s, t: self.ss2_update(x, s, t, r1=r1, solver_type=solver_type)
            higher = lambda x, s, t, **kw: self.ss3_update(x, s, t, r1=r1, r2=r2, solver_type=solver_type, **kw)[0]
        else:
            raise ValueError("For adaptive step size solver, order must be 2 or 3, got {}".format(order))
        while abs(F32(s - F32(t_0))) > t_err:
            t = self._sc(sch.inv_lam(F32(lam_s + h)))
            x_lower, kw = lower(x, s, t)
            x_higher = higher(x, s, t, **kw)
            delta = np.maximum(F32(atol), F32(rtol) * np.maximum(np.abs(x_lower), np.abs(x_prev)))
            v = ((x_higher - x_lower) / delta).reshape(x.shape[0], -1)
            E = F32(np.sqrt(np.mean(np.square(v).astype(F64), axis=-1)).astype(F32).max())
            if E <= 1.0:
                x, s, x_prev = x_higher, t, x_lower
                lam_s = self._sc(sch.lam(s))
            h = min(F32(F32(theta) * h * F32(F64(E) ** (-1.0 / order))), F32(lam_0 - lam_s))
            nfe += order
        self.adaptive_nfe = nfe
        return x

    # ---- add_noise / inverse / sample (ref:1012-1245) -------------------------------------
    def add_noise(self, x, t, noise):
        t = _f(t).reshape(-1)
        a, s = self.sch.alpha(t), self.sch.std(t)
        xr = x.reshape((1,) + x.shape)
        xt = _b(a, xr) * xr + _b(s, xr) * noise
        return xt[0] if t.shape[0] == 1 else xt

    def inverse(self, x, steps=20, t_start=None, t_end=None, **kw):
        t_0 = 1.0 / self.sch.total_N if t_start is None else t_start
        t_T = self.sch.T if t_end is None else t_end
        return self.sample(x, steps=steps, t_start=t_0, t_end=t_T, **kw)

    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform",
               method="multistep", lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver",
               atol=0.0078, rtol=0.05, return_intermediate=False):
        sch = self.sch
        t_0 = 1.0 / sch.total_N if t_end is None else t_end
        t_T = sch.T if t_start is None else t_start
        assert t_0 > 0 and t_T > 0
        x = np.asarray(x)
        inter = []
        cxt = self.cxt
        step = 0

        def post(x, t, step):
            if cxt is not None:
                x = cxt(x, t, step)
            if return_intermediate:
                inter.append(x)
            return x

        if method == "adaptive":
            x = self.adaptive(x.astype(F32), order, t_T, t_0, atol=atol, rtol=rtol, solver_type=solver_type)
        elif method == "multistep":
            assert steps >= order
            ts = time_steps(sch, skip_type, t_T, t_0, steps)
            step = 0
            t = ts[0]
            t_list = [t]
            m_list = [self.model_value(x, t)]                                 # ref:1179 (sees the caller's dtype)
            x = post(x, t, step)
            for step in range(1, order):                                      # ref:1185-1193
                t = ts[step]
                x = self.multistep_update(x, m_list, t_list, t, step, solver_type)
                x = post(x, t, step)
                t_list.append(t)
                m_list.append(self.model_value(x, t))
            for step in range(order, steps + 1):                              # ref:1195-1213
                t = ts[step]
                so = min(order, steps + 1 - step) if (lower_order_final and steps < 10) else order
                x = self.multistep_update(x, m_list, t_list, t, so, solver_type)
                x = post(x, t, step)
                t_list = t_list[1:] + [t]
                m_list = m_list[1:] + [None]
                if step < steps:
                    m_list[-1] = self.model_value(x, t)
        elif method in ("singlestep", "singlestep_fixed"):
            if method == "singlestep":
                outer, orders = singlestep_grid(sch, steps, order, skip_type, t_T, t_0)
            else:
                K = steps // order
                orders = [order] * K
                outer = time_steps(sch, skip_type, t_T, t_0, K)
            step = -1
            for step, o in enumerate(orders):                                 # ref:1221-1232
                s, t = outer[step], outer[step + 1]
                inner = time_steps(sch, skip_type, float(s), float(t), o)
                lam_in = sch.lam(inner)
                hh = F32(lam_in[-1] - lam_in[0])
                r1 = None if o <= 1 else F32((lam_in[1] - lam_in[0]) / hh)
                r2 = None if o <= 2 else F32((lam_in[2] - lam_in[0]) / hh)
                x = self.singlestep_update(x, s, t, o, solver_type=solver_type, r1=r1, r2=r2)
                x = post(x, t, step)
        else:
            raise ValueError("Got wrong method {}".format(method))
        if denoise_to_zero:                                                   # ref:1235-1241
            t = F32(t_0)
            x = self.data_pred(x, t)
            x = post(x, t, step + 1)
        return (x, inter) if return_intermediate else x
