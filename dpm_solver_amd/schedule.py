"""NoiseScheduleVP -- host mirror of the reference class (dpm_solver_pytorch.py:6-167).

Same constructor, attributes and methods; the arithmetic lives in the C planner
(csrc/dpm_host.cpp), which reproduces the reference's fp32 operation order.  The marginal_* /
inverse_lambda methods are host functions (the hot loop never calls them: every scalar a sampling
run needs is precomputed into the plan); calling them on a device tensor copies it to the host.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


class NoiseScheduleVP:
    _SCHEDULES = ['discrete', 'linear']
    _clip = True            # numerical_clip_alpha at lambda = -5.1 (ref :114-125)

    def __init__(self, schedule='discrete', betas=None, alphas_cumprod=None, continuous_beta_0=0.1,
                 continuous_beta_1=20., dtype=torch.float32):
        if schedule not in self._SCHEDULES:
            raise ValueError("Unsupported noise schedule {}. The schedule needs to be {}".format(
                schedule, " or ".join("'%s'" % v for v in self._SCHEDULES)))
        self.schedule = schedule
        self.T = 1.
        self.dtype = dtype
        self._h = C.c_void_p()
        self._dev_tables = {}
        if schedule == 'discrete':
            src = betas if betas is not None else alphas_cumprod
            assert src is not None
            arr = src.detach().cpu().numpy() if torch.is_tensor(src) else np.asarray(src)
            arr = np.ascontiguousarray(arr.reshape(-1))
            f64 = arr.dtype == np.float64
            if not f64:
                arr = np.ascontiguousarray(arr, dtype=np.float32)
            name = "dpm_schedule_create_%s_%s" % ("betas" if betas is not None else "alphas_cumprod",
                                                  "f64" if f64 else "f32")
            ptr = arr.ctypes.data_as(C.POINTER(C.c_double if f64 else C.c_float))
            L.check(getattr(L.lib, name)(ptr, int(arr.shape[0]), 1 if self._clip else 0, C.byref(self._h)))
            K = C.c_int()
            if dtype == torch.float64:
                # ref :105-107: the tables keep the values they were computed in (double for double inputs, the fp32 values
                # converted otherwise); double-precision runs (a double state) evaluate every scalar on them in double
                L.check(L.lib.dpm_schedule_set_table_dtype(self._h, L.DTYPE_F64))
                la, ta = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
                L.check(L.lib.dpm_schedule_tables_f64(self._h, C.byref(la), C.byref(ta), C.byref(K)))
            else:
                la, ta = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
                L.check(L.lib.dpm_schedule_tables(self._h, C.byref(la), C.byref(ta), C.byref(K)))
            self.total_N = K.value
            self.log_alpha_array = torch.from_numpy(np.ctypeslib.as_array(la, (K.value,)).copy()).reshape((1, -1)).to(dtype=dtype)
            self.t_array = torch.from_numpy(np.ctypeslib.as_array(ta, (K.value,)).copy()).reshape((1, -1)).to(dtype=dtype)
        elif schedule == 'cosine':
            # older vendored revision only (LegacyNoiseScheduleVP): T = 1 has numerical issues, so T = 0.9946
            self.total_N = 1000
            self.beta_0 = continuous_beta_0
            self.beta_1 = continuous_beta_1
            self.cosine_s = 0.008
            self.T = 0.9946
            L.check(L.lib.dpm_schedule_create_cosine(C.byref(self._h)))
        else:
            self.total_N = 1000
            self.beta_0 = continuous_beta_0
            self.beta_1 = continuous_beta_1
            L.check(L.lib.dpm_schedule_create_linear(float(continuous_beta_0), float(continuous_beta_1), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                L.lib.dpm_schedule_destroy(h)
            except Exception:
                pass
            self._h = None

    def numerical_clip_alpha(self, log_alphas, clipped_lambda=-5.1):
        """Drop the trailing entries of a log-alpha table whose half-logSNR is below `clipped_lambda` (ref :114-125:
        the cosine schedules of i-DDPM / guided-diffusion / GLIDE are numerically unstable near t = T).  Returns
        `log_alphas[:K]`, a view like the reference's slice; the count comes from the C planner, in the table's own
        arithmetic type (the constructor applies the same clip inside `dpm_schedule_create_*`)."""
        if not torch.is_tensor(log_alphas):
            log_alphas = torch.as_tensor(log_alphas)
        arr = log_alphas.detach().cpu().reshape(-1).numpy()
        f64 = arr.dtype == np.float64
        arr = np.ascontiguousarray(arr, dtype=np.float64 if f64 else np.float32)
        keep = C.c_int()
        fn = L.lib.dpm_numerical_clip_len_f64 if f64 else L.lib.dpm_numerical_clip_len_f32
        L.check(fn(arr.ctypes.data_as(C.POINTER(C.c_double if f64 else C.c_float)), int(arr.shape[0]),
                   float(clipped_lambda), C.byref(keep)))
        idx = int(arr.shape[0]) - keep.value
        if idx > 0:
            log_alphas = log_alphas[:-idx]
        return log_alphas

    # ---- host evaluation ------------------------------------------------------------------
    def _eval_np(self, what, v):
        v = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(-1))
        out = np.empty_like(v)
        L.check(L.lib.dpm_schedule_eval(self._h, what, v.ctypes.data_as(C.POINTER(C.c_float)), int(v.shape[0]),
                                        out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def _eval_np64(self, what, v):
        v = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
        out = np.empty_like(v)
        L.check(L.lib.dpm_schedule_eval_f64(self._h, what, v.ctypes.data_as(C.POINTER(C.c_double)), int(v.shape[0]),
                                            out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def _double_scalars(self, t):
        """the reference's type promotion: a double time, or double tables ('discrete', dtype=float64), make the scalar a
        double (ref :127-134: interpolate_fn concatenates t with the tables)"""
        return t.dtype == torch.float64 or (self.schedule == 'discrete' and self.dtype == torch.float64)

    def _eval(self, what, t):
        if not torch.is_tensor(t):
            t = torch.as_tensor(t, dtype=torch.float32)
        if self._double_scalars(t):
            out = torch.from_numpy(self._eval_np64(what, t.detach().to(device="cpu", dtype=torch.float64).numpy()))
        else:
            out = torch.from_numpy(self._eval_np(what, t.detach().to(device="cpu", dtype=torch.float32).numpy()))
        # the reference flattens for 'discrete' (reshape((-1))) and keeps the shape for 'linear'
        if self.schedule != 'discrete':
            out = out.reshape(t.shape)
        return out.to(device=t.device)

    def device_alpha_sigma(self, t):
        """(alpha_t, sigma_t) of a DEVICE tensor of times, per element, computed on the device with torch operations and no
        host synchronisation: ref :127-146 -- piecewise-linear interpolation of log alpha on the tables (the segment found by
        torch.searchsorted instead of the reference's sort, same interpolation formula), exp, sqrt(1 - exp(2 log alpha)).  For
        callers of model_fn(x, t) with per-sample times (model_wrapper's callable, ref :282-330); the solver itself never
        needs it: every scalar of a sampling run is precomputed by the C planner."""
        tt = t.reshape(-1)
        if self.schedule == 'discrete':
            key = (str(t.device),)
            tab = self._dev_tables.get(key)
            if tab is None:
                tab = self._dev_tables[key] = (self.t_array.reshape(-1).to(t.device), self.log_alpha_array.reshape(-1).to(t.device))
            xp, yp = tab
            dt = torch.promote_types(tt.dtype, xp.dtype)
            x = tt.to(dt)
            xp = xp.to(dt)
            K = xp.shape[0]
            idx = torch.searchsorted(xp, x.contiguous(), right=False)          # #{xp < x}
            i0 = torch.where(idx == 0, torch.zeros_like(idx), torch.where(idx == K, torch.full_like(idx, K - 2), idx - 1))
            i1 = i0 + 1
            # (`end_y - start_y`, ref :1290, is an operation between two table entries: in the tables' own dtype also when
            # the times are doubles and the tables fp32)
            la = yp[i0] + (x - xp[i0]) * (yp[i1] - yp[i0]) / (xp[i1] - xp[i0])
        elif self.schedule == 'linear':
            la = -0.25 * tt ** 2 * (self.beta_1 - self.beta_0) - 0.5 * tt * self.beta_0
        elif self.schedule == 'cosine':
            # older vendored revision (examples/score_sde_pytorch/dpm_solver.py:135-138): log alpha_t = log cos((t + s) / (1 + s)
            # * pi / 2) - log cos(s / (1 + s) * pi / 2), the reference's own tensor expression
            import math
            s = self.cosine_s
            la = torch.log(torch.cos((tt + s) / (1. + s) * math.pi / 2.)) - math.log(math.cos(s / (1. + s) * math.pi / 2.))
        else:
            raise NotImplementedError("device_alpha_sigma: schedule %r" % self.schedule)
        return torch.exp(la), torch.sqrt(1. - torch.exp(2. * la))

    def marginal_log_mean_coeff(self, t):
        """log(alpha_t) of a continuous-time label t in [0, T]  (ref :127-134)."""
        return self._eval(L.EVAL_LOG_ALPHA, t)

    def marginal_alpha(self, t):
        """alpha_t  (ref :136-140)."""
        return self._eval(L.EVAL_ALPHA, t)

    def marginal_std(self, t):
        """sigma_t  (ref :142-146)."""
        return self._eval(L.EVAL_STD, t)

    def marginal_lambda(self, t):
        """lambda_t = log(alpha_t) - log(sigma_t)  (ref :148-154)."""
        return self._eval(L.EVAL_LAMBDA, t)

    def inverse_lambda(self, lamb):
        """t of a given half-logSNR lambda_t  (ref :156-167)."""
        if not torch.is_tensor(lamb):
            lamb = torch.as_tensor(lamb, dtype=torch.float32)
        if self._double_scalars(lamb):
            out = torch.from_numpy(self._eval_np64(L.EVAL_INV_LAMBDA, lamb.detach().to(device="cpu", dtype=torch.float64).numpy()))
        else:
            out = torch.from_numpy(self._eval_np(L.EVAL_INV_LAMBDA, lamb.detach().to(device="cpu", dtype=torch.float32).numpy()))
        if self.schedule != 'discrete':
            # (ref :158: logaddexp against a (1,)-shaped zero -- a 0-dim lambda comes back (1,)-shaped)
            out = out.reshape(lamb.shape if lamb.dim() else (1,))
        return out.to(device=lamb.device)


class LegacyNoiseScheduleVP(NoiseScheduleVP):
    """NoiseScheduleVP of the older revision the reference still vendors for the ScoreSDE example
    (examples/score_sde_pytorch/dpm_solver.py:6-176): it additionally offers the continuous-time 'cosine' schedule
    (:114-124, T = 0.9946) and does NOT clip discrete schedules (total_N = len(log_alphas), :106).  The solver classes
    of the two revisions are otherwise identical, so `DPM_Solver` serves both."""
    _SCHEDULES = ['discrete', 'linear', 'cosine']
    _clip = False
