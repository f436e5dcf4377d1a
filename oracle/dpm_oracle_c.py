"""ctypes loader and driver of oracle/dpm_oracle_kernels.c  --  TEST INFRASTRUCTURE ONLY (see that file's header).

`K` exposes the C functions on numpy fp32 arrays; `Stepper` drives a DPM-Solver / DPM-Solver++ multistep trajectory
(ref :1171-1213) through them, with every scalar taken from the numpy oracle's schedule (oracle/dpm_oracle.py), so that
tests can hold C restatement == numpy restatement == golden fixtures, and bench.py's `cpu_baseline` leg can time a fused,
multi-threaded CPU port of the hot path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import dpm_oracle as O

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "dpm_oracle_kernels.c")
LIB = os.path.join(_HERE, "_build", "libdpm_oracle.so")
# -ffp-contract=off: the reference's tensor expressions round after every operation; x86-64-v3 (AVX2, no FMA use): the
# library is built in the build container and travels to the GPU box's host
CFLAGS = ["-O3", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-mavx2", "-fvisibility=hidden"]


def build(force=False):
    """compile the C restatement (gcc); returns the library path"""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc"] + CFLAGS + [SRC, "-o", LIB, "-lm"], check=True)
    return LIB


_lib = None
_fp = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImportError("%s not built: run __graft_entry__.build() (gcc)" % LIB)
        _lib = C.CDLL(LIB)
        _lib.dpmo_version.restype = C.c_int
        _lib.dpmo_max_threads.restype = C.c_int
        _lib.dpmo_dynamic_threshold.restype = C.c_int
        assert _lib.dpmo_version() >= 1
    return _lib


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


class K:
    """the C functions on contiguous fp32 numpy arrays (outputs are fresh arrays)"""

    @staticmethod
    def cfg_blend(nu, nc, scale):
        out = np.empty_like(nu)
        lib().dpmo_cfg_blend(_p(nu), _p(nc), C.c_float(scale), _p(out), C.c_int64(nu.size))
        return out

    @staticmethod
    def eps_to_x0(x, eps, alpha, sigma):
        out = np.empty_like(x)
        lib().dpmo_eps_to_x0(_p(x), _p(eps), C.c_float(alpha), C.c_float(sigma), _p(out), C.c_int64(x.size))
        return out

    @staticmethod
    def to_noise(kind, x, out_, alpha, sigma):
        noise = np.empty_like(x)
        n = C.c_int64(x.size)
        if kind == "x_start":
            lib().dpmo_xstart_to_noise(_p(x), _p(out_), C.c_float(alpha), C.c_float(sigma), _p(noise), n)
        elif kind == "v":
            lib().dpmo_v_to_noise(_p(x), _p(out_), C.c_float(alpha), C.c_float(sigma), _p(noise), n)
        elif kind == "score":
            lib().dpmo_score_to_noise(_p(out_), C.c_float(sigma), _p(noise), n)
        else:
            raise ValueError(kind)
        return noise

    @staticmethod
    def dynamic_threshold(x0, ratio=0.995, max_val=1.0):
        out = np.ascontiguousarray(x0, dtype=F32).copy()
        B = out.shape[0]
        s = np.empty(B, dtype=F32)
        rc = lib().dpmo_dynamic_threshold(_p(out), C.c_int64(B), C.c_int64(out.size // B), C.c_float(ratio), C.c_float(max_val), _p(s))
        assert rc == 0
        return out, s

    @staticmethod
    def update_first(x, m, cx, c0):
        out = np.empty_like(x)
        lib().dpmo_update_first(_p(x), _p(m), C.c_float(cx), C.c_float(c0), _p(out), C.c_int64(x.size))
        return out

    @staticmethod
    def update_ms2(x, m0, m1, cx, c0, k0, c1):
        out = np.empty_like(x)
        lib().dpmo_update_ms2(_p(x), _p(m0), _p(m1), C.c_float(cx), C.c_float(c0), C.c_float(k0), C.c_float(c1), _p(out),
                              C.c_int64(x.size))
        return out

    @staticmethod
    def update_ms3(x, m0, m1, m2, cx, c0, c1, c2, k0, k1, k2, k3, plus):
        out = np.empty_like(x)
        lib().dpmo_update_ms3(_p(x), _p(m0), _p(m1), _p(m2), *[C.c_float(v) for v in (cx, c0, c1, c2, k0, k1, k2, k3)], C.c_int(int(plus)),
                              _p(out), C.c_int64(x.size))
        return out


# ------------------------------------------------------------------------------------------------
# the scalars of the multistep updates, in the reference's fp32 operation order (from the numpy oracle's schedule)
# ------------------------------------------------------------------------------------------------
def _marg(sch, t):
    sc = O.Solver._sc
    return sc(sch.lam(t)), sc(sch.log_alpha_t(t)), sc(sch.std(t))


def coef_first(sch, pp, s, t):
    """(cx, c0) of x_t = cx * x - c0 * model_s (ref :553-590)"""
    sc = O.Solver._sc
    lam_s, la_s, sig_s = _marg(sch, s)
    lam_t, la_t, sig_t = _marg(sch, t)
    h = F32(lam_t - lam_s)
    if pp:
        return F32(sig_t / sig_s), F32(sc(O.exp32(la_t)) * sc(O.expm1_32(-h)))
    return sc(O.exp32(la_t - la_s)), F32(sig_t * sc(O.expm1_32(h)))


def coef_ms2(sch, pp, t_p1, t_p0, t, solver_type="dpmsolver"):
    """(cx, c0, k0, c1) of x_t = cx * x - c0 * m0 - c1 * (k0 * (m0 - m1)) (ref :805-851)"""
    sc = O.Solver._sc
    lam_p1, _, _ = _marg(sch, t_p1)
    lam_p0, la_p0, sig_p0 = _marg(sch, t_p0)
    lam_t, la_t, sig_t = _marg(sch, t)
    a_t = sc(O.exp32(la_t))
    h_0, h = F32(lam_p0 - lam_p1), F32(lam_t - lam_p0)
    r0 = F32(h_0 / h)
    k0 = F32(1.0 / r0)
    if pp:
        phi_1 = sc(O.expm1_32(-h))
        c0 = F32(a_t * phi_1)
        c1 = F32(0.5 * c0) if solver_type == "dpmsolver" else F32(-F32(a_t * F32(phi_1 / h + F32(1.0))))
        return F32(sig_t / sig_p0), c0, k0, c1
    phi_1 = sc(O.expm1_32(h))
    c0 = F32(sig_t * phi_1)
    c1 = F32(0.5 * c0) if solver_type == "dpmsolver" else F32(sig_t * F32(phi_1 / h - F32(1.0)))
    return sc(O.exp32(la_t - la_p0)), c0, k0, c1


def coef_ms3(sch, pp, t_p2, t_p1, t_p0, t):
    """(cx, c0, c1, c2, k0, k1, k2, k3, plus) of the third-order multistep update (ref :860-903)"""
    sc = O.Solver._sc
    lam_p2, _, _ = _marg(sch, t_p2)
    lam_p1, _, _ = _marg(sch, t_p1)
    lam_p0, la_p0, sig_p0 = _marg(sch, t_p0)
    lam_t, la_t, sig_t = _marg(sch, t)
    a_t = sc(O.exp32(la_t))
    h_1, h_0, h = F32(lam_p1 - lam_p2), F32(lam_p0 - lam_p1), F32(lam_t - lam_p0)
    r0, r1 = F32(h_0 / h), F32(h_1 / h)
    k0, k1 = F32(1.0 / r0), F32(1.0 / r1)
    k2, k3 = F32(r0 / F32(r0 + r1)), F32(1.0 / F32(r0 + r1))
    if pp:
        phi_1 = sc(O.expm1_32(-h))
        phi_2 = F32(phi_1 / h + F32(1.0))
        phi_3 = F32(phi_2 / h - F32(0.5))
        return F32(sig_t / sig_p0), F32(a_t * phi_1), F32(a_t * phi_2), F32(a_t * phi_3), k0, k1, k2, k3, True
    phi_1 = sc(O.expm1_32(h))
    phi_2 = F32(phi_1 / h - F32(1.0))
    phi_3 = F32(phi_2 / h - F32(0.5))
    return sc(O.exp32(la_t - la_p0)), F32(sig_t * phi_1), F32(sig_t * phi_2), F32(sig_t * phi_3), k0, k1, k2, k3, False


class Stepper:
    """DPM_Solver.sample(method='multistep', skip_type='time_uniform') of a noise-prediction network (ref :1171-1213), every
    tensor expression through the C kernels.  `fused=True` (DPM-Solver++ order 2 only) runs each stage as the ONE pass a GPU
    launch makes (dpmo_stage_2m): the CPU port timed by bench.py."""

    def __init__(self, sch, algorithm_type="dpmsolver++", thresholding=False, ratio=0.995, max_val=1.0):
        self.sch, self.pp = sch, algorithm_type == "dpmsolver++"
        self.thr, self.ratio, self.max_val = thresholding, ratio, max_val

    def model_value(self, net, x, t):
        tv = np.full((x.shape[0],), F32(t), dtype=F32)
        eps = np.ascontiguousarray(net(x, tv), dtype=F32)
        if not self.pp:
            return eps
        sc = O.Solver._sc
        x0 = K.eps_to_x0(x, eps, sc(self.sch.alpha(t)), sc(self.sch.std(t)))
        return K.dynamic_threshold(x0, self.ratio, self.max_val)[0] if self.thr else x0

    def sample(self, net, x, steps=20, order=2, lower_order_final=True, solver_type="dpmsolver", t_T=None, t_0=None):
        sch = self.sch
        t_T = sch.T if t_T is None else t_T
        t_0 = 1.0 / sch.total_N if t_0 is None else t_0
        ts = O.time_steps(sch, "time_uniform", t_T, t_0, steps)
        x = np.ascontiguousarray(x, dtype=F32)
        ms, tp = [self.model_value(net, x, ts[0])], [ts[0]]
        for step in range(1, steps + 1):
            t = ts[step]
            if step < order:
                so = step
            else:
                so = min(order, steps + 1 - step) if (lower_order_final and steps < 10) else order
            if so == 1:
                x = K.update_first(x, ms[-1], *coef_first(sch, self.pp, tp[-1], t))
            elif so == 2:
                x = K.update_ms2(x, ms[-1], ms[-2], *coef_ms2(sch, self.pp, tp[-2], tp[-1], t, solver_type))
            else:
                x = K.update_ms3(x, ms[-1], ms[-2], ms[-3], *coef_ms3(sch, self.pp, tp[-3], tp[-2], tp[-1], t))
            if step < steps:
                ms.append(self.model_value(net, x, t))
                tp.append(t)
                ms, tp = ms[-order:], tp[-order:]
        return x

    def sample_2m_fused(self, eps, x, steps=20, threads=None):
        """DPM-Solver++(2M), frozen noise prediction `eps`, steps >= 10: one dpmo_stage_2m pass per stage"""
        assert self.pp and not self.thr and steps >= 10
        L = lib()
        if threads:
            L.dpmo_set_threads(int(threads))
        sch, sc = self.sch, O.Solver._sc
        ts = O.time_steps(sch, "time_uniform", sch.T, 1.0 / sch.total_N, steps)
        xs = [np.ascontiguousarray(x, dtype=F32).copy(), np.empty_like(x, dtype=F32)]
        mb = [np.empty_like(xs[0]), np.empty_like(xs[0])]
        eps = np.ascontiguousarray(eps, dtype=F32)
        n = C.c_int64(xs[0].size)
        for step in range(1, steps + 1):
            s_, t = ts[step - 1], ts[step]
            a_e, s_e = sc(sch.alpha(s_)), sc(sch.std(s_))
            if step == 1:
                cx, c0 = coef_first(sch, True, s_, t)
                k0 = c1 = F32(0)
            else:
                cx, c0, k0, c1 = coef_ms2(sch, True, ts[step - 2], s_, t)
            xi, xo = xs[(step - 1) % 2], xs[step % 2]
            L.dpmo_stage_2m(_p(xi), _p(eps), _p(mb[step % 2]), C.c_float(a_e), C.c_float(s_e), C.c_float(cx), C.c_float(c0),
                            C.c_float(k0), C.c_float(c1), C.c_int(int(step == 1)), _p(xo),
                            _p(mb[(step + 1) % 2]) if step < steps else None, n)
        return xs[steps % 2]
