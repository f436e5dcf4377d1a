"""Machinery of the public per-update methods (ref :427-451, :547-954), split out of solver.py in round 6: one model
evaluation as one launch (`eval_model`), 1-3 singlestep stages from a state (`exec_single`, `singlestep_stages`) and one
multistep update from given model values (`multistep`).  The functions take the DPM_Solver as `self`; solver.py binds them as
methods (`_eval_model`, `_exec_single`, `_singlestep_stages`, `_multistep`) under the reference-named methods that call them."""
import ctypes as C

import torch

from . import _device as DV
from . import _lib as L

def eval_model(self, x, t, to_x0):
    DV._require_gpu(x)
    mt, gd, sc = self._model_codes()
    st = L.Stage()
    st.h1_slot = st.h2_slot = st.m_slot = -1
    dbl, tf64 = self._double_call(x, t)
    c64 = None
    dev = x.device
    if dbl:
        c64 = L.StageF64()
        L.check(L.lib.dpm_coef_prologue_f64(self._h, self._td(t), int(tf64), mt, gd, sc, C.byref(st), C.byref(c64)))
        tdt = torch.float64 if tf64 else torch.float32
        te_t, ti_t = self._tt(c64.t_eval, dev, dtype=tdt), self._tt(c64.t_input, dev, dtype=tdt)
        tf = c64.t_eval
    else:
        tf = self._tf(t)
        L.check(L.lib.dpm_coef_prologue(self._h, tf, mt, gd, sc, C.byref(st)))
        te_t, ti_t = self._tt(st.t_eval, dev), self._tt(st.t_input, dev)
    st.form = L.FORM_DENOISE
    st.flags = L.F_TO_X0 if to_x0 else 0
    self._prep_stage(st)
    outs = self._network(x, te_t, ti_t)
    # Which operands the reference's expression really involves decides the result's dtype (ref :288-330, :433-442): the
    # schedule's scalars (doubles when `dbl`) and x enter through the x_start / v / score conversions, the classifier term
    # and eps -> x0 only.  A noise-prediction network asked for its noise comes back untouched -- in the NETWORK's dtype --
    # and its classifier-free blend `uncond + scale * (cond - uncond)` (a Python-float scale) stays there too.
    pure_noise = (not to_x0) and mt == L.MODEL["noise"] and gd != L.GUIDE["classifier"]
    if pure_noise and gd == L.GUIDE["uncond"]:
        return outs[0]
    if pure_noise:
        sd = outs[0].dtype if outs[0].dtype in DV._DT else torch.float32
    else:
        # (the scalars are doubles when `dbl`; whether the RESULT is double is torch's type promotion of the operands)
        sd = torch.float64 if self._out_double(x, (t,), (), evaluates=True) else self._call_sdtype(x, (t,), (), evaluates=True)
        if not dbl and to_x0 and sd in (torch.float16, torch.bfloat16) and outs[0].dtype is torch.float32:
            sd = torch.float32
    out, _ = self._run_stage(st, None, x, outs, None, None, sd, t if torch.is_tensor(t) else self._tt(tf, dev), want_m=False,
                             coef64=self._stage64(st, c64) if sd is torch.float64 else None)
    return out


def exec_single(self, stages, x, given, want, c64s=None, tf64=False, times=()):
    """Run 1-3 singlestep stages starting from state x.  `given[i]` = model value already known for
    stage i; `want` = return the model values.  Returns (x_t, [m_0, m_1, m_2]).  c64s: the stages' doubles
    (dpm_coef_singlestep_f64) when the call's scalars are doubles; tf64: the caller's time tensors are doubles."""
    DV._require_gpu(x)
    dev = x.device
    n_given = sum(1 for v in given.values() if v is not None)
    ev = n_given < len(stages)
    sd = (torch.float64 if self._out_double(x, times, tuple(given.values()), evaluates=ev)
          else self._call_sdtype(x, times, tuple(given.values()), evaluates=ev))
    mt, gd, sc = self._model_codes()
    k64 = lambda i, st_: (self._stage64(st_, c64s[i] if c64s is not None else None) if sd is torch.float64 else None)
    n = len(stages)
    ms = [given.get(i) for i in range(n)]
    tmp = None
    x_t = None
    for i, st in enumerate(stages):
        last = i == n - 1
        if not last and ms[i] is not None and ms[i + 1] is not None:
            continue                      # this stage's output would only feed an evaluation we already have
        h1 = ms[0] if st.h1_slot >= 0 else None
        h2 = ms[1] if st.h2_slot >= 0 else None
        if ms[i] is not None:
            out, _ = self._run_given(st, x, ms[i], h1, h2, sd, want_m=False, coef64=k64(i, st))
        else:
            xe = x if i == 0 else tmp
            if c64s is not None:
                t64 = bool(c64s[i].time_f64 & 1)
                L.check(L.lib.dpm_coef_prologue_f64(self._h, c64s[i].t_eval, int(t64), mt, gd, sc, C.byref(st), C.byref(c64s[i])))
                tdt = torch.float64 if t64 else torch.float32
                te_t, ti_t = self._tt(c64s[i].t_eval, dev, dtype=tdt), self._tt(c64s[i].t_input, dev, dtype=tdt)
            else:
                L.check(L.lib.dpm_coef_prologue(self._h, st.t_eval, mt, gd, sc, C.byref(st)))
                te_t, ti_t = self._tt(st.t_eval, dev), self._tt(st.t_input, dev)
            self._prep_stage(st)
            outs = self._network(xe, te_t, ti_t)
            need_m = want or (i == 0 and n > 1) or (i == 1 and n == 3 and stages[2].h2_slot >= 0)
            out, m = self._run_stage(st, x, None if i == 0 else xe, outs, h1, h2, sd, te_t, want_m=need_m, coef64=k64(i, st))
            ms[i] = m
        if last:
            x_t = out
        else:
            tmp = out
    return x_t, ms


def singlestep_stages(self, x, order, solver_code, s, t, r1, r2, mode):
    """(stages, their doubles or None, the time tensors are doubles) of a singlestep update s -> t"""
    dbl, tf64 = self._double_call(x, s, t, r1, r2)
    st = (L.Stage * order)()
    if not dbl:
        L.check(L.lib.dpm_coef_singlestep(self._h, self._algo, solver_code, order, self._tf(s), self._tf(t), r1 if not mode else
                                          self._tf(r1), r2 if not mode else self._tf(r2), mode, st))
        return [st[i] for i in range(order)], None, False
    c64 = (L.StageF64 * order)()
    L.check(L.lib.dpm_coef_singlestep_f64(self._h, self._algo, solver_code, order, self._td(s), self._td(t), int(tf64),
                                          self._td(r1), self._td(r2), mode, st, c64))
    return [st[i] for i in range(order)], [c64[i] for i in range(order)], tf64


def multistep(self, x, model_prev_list, t_prev_list, t, order, solver_type):
    DV._require_gpu(x)
    st = L.Stage()
    dbl, tf64 = self._double_call(x, t, *t_prev_list[-order:])
    c64 = None
    if dbl:
        tp = (C.c_double * order)(*[self._td(v) for v in t_prev_list[-order:]])
        c64 = L.StageF64()
        L.check(L.lib.dpm_coef_multistep_f64(self._h, self._algo, L.SOLVER[solver_type], order, tp, self._td(t), int(tf64),
                                             C.byref(st), C.byref(c64)))
    else:
        tp = (C.c_float * order)(*[self._tf(v) for v in t_prev_list[-order:]])
        L.check(L.lib.dpm_coef_multistep(self._h, self._algo, L.SOLVER[solver_type], order, tp, self._tf(t), C.byref(st)))
    h1 = model_prev_list[-2] if order >= 2 else None
    h2 = model_prev_list[-3] if order >= 3 else None
    tms, mds = (t,) + tuple(t_prev_list[-order:]), tuple(model_prev_list[-order:])
    sd = torch.float64 if self._out_double(x, tms, mds) else self._call_sdtype(x, tms, mds)
    x_t, _ = self._run_given(st, x, model_prev_list[-1], h1, h2, sd, want_m=False,
                             coef64=self._stage64(st, c64) if sd is torch.float64 else None)
    return x_t
