"""Frozen sampling plans and their per-device time tensors (split out of solver.py in round 6).

`_Plan` wraps a `dpm_plan` of the C planner (one stage record per network evaluation) and caches, per device and batch, the
time vectors handed to the network and the callbacks; `get_plan` is DPM_Solver's plan cache (`DPM_Solver._get_plan`).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

def _versions(tensors):
    """version counters of `tensors` (in-place writes bump them); None where they are not tracked -- tensors created under
    torch.inference_mode() -- which switches the detection of writes into the shared time vectors off (never an error)"""
    try:
        return tuple(t._version for t in tensors)
    except RuntimeError:
        return None


class _Plan:
    """A frozen `dpm_plan` plus the per-device time tensors handed to the network / callbacks."""

    def __init__(self, sched_handle, desc):
        self.handle = C.c_void_p()
        L.check(L.lib.dpm_plan_create(sched_handle, C.byref(desc), C.byref(self.handle)))
        n = L.lib.dpm_plan_num_stages(self.handle)
        self.slots = L.lib.dpm_plan_num_slots(self.handle)
        self.stages = []
        self.stages64 = None                 # double-precision plans: the dpm_stage_f64 twin of every stage
        for i in range(n):
            st = L.Stage()
            L.check(L.lib.dpm_plan_stage(self.handle, i, C.byref(st)))
            self.stages.append(st)
        if desc.precision:
            self.stages64 = []
            for i in range(n):
                s64 = L.StageF64()
                L.check(L.lib.dpm_plan_stage_f64(self.handle, i, C.byref(s64)))
                self.stages64.append(s64)
        self._dev = {}
        self._views = {}
        self.times_written = False
        # a singlestep update of order >= 2 is part of the plan (a stage evaluates the network on an intermediate state)
        self.has_inner_nodes = any(st.xe_src == L.SRC_TMP for st in self.stages)
        # static buffer roles per stage, as plan_run_impl (dpm_host.cpp) rotates them: indices into
        # [x_T, scratch 1, scratch 2, scratch 3] for the update's x, the state the network saw, and the output
        self.roles = []
        state, tmp = 0, -1
        for st in self.stages:
            xe = tmp if st.xe_src == L.SRC_TMP else state
            out = 1
            while out == state or out == xe:
                out += 1
            self.roles.append((state, xe, out))
            if st.emits_state:
                state, tmp = out, -1
            else:
                tmp = out

    def times(self, device):
        """(t_eval, t_input, t_out) as fp32 device vectors, one host-to-device copy per plan and device."""
        key = str(device)
        if key not in self._dev:
            arr = np.array([[s.t_eval for s in self.stages], [s.t_input for s in self.stages],
                            [s.t_out for s in self.stages]], dtype=np.float32)
            self._dev[key] = torch.from_numpy(arr).to(device)
        return self._dev[key]

    def times64(self, device):
        """double-precision plans: the same three rows in double"""
        key = ("f64", str(device))
        if key not in self._dev:
            arr = np.array([[s.t_eval for s in self.stages64], [s.t_input for s in self.stages64],
                            [s.t_out for s in self.stages64]], dtype=np.float64)
            self._dev[key] = torch.from_numpy(arr).to(device)
        return self._dev[key]

    def time_views(self, device, batch, cfg):
        """per stage: 0-dim t_eval / t_out, t_eval and t_input expanded to (batch,) and, under classifier-free
        guidance, t_input expanded to (2*batch,) -- views of times(), built once per (device, batch).

        The vectors are SHARED by every later call of the plan (the reference hands the network a fresh tensor per call,
        ref :404).  A network or callback that writes into its time argument in place (`t.mul_(1000)`) is detected through
        the tensors' version counters: the next call finds them changed, rebuilds the vectors from the host plan and sets
        `times_written` -- DPM_Solver then hands out clones (`fresh_time_tensors`).  Inside one trajectory every row is
        handed out once, so the trajectory during which the first write happens is still correct."""
        key = (str(device), int(batch), bool(cfg))
        hit = self._views.get(key)
        if hit is not None and hit["ver"] is not None and _versions(hit["base"]) != hit["ver"]:
            self._views.pop(key)
            self._dev.pop(str(device), None)
            self.times_written = True
            hit = None
        if hit is None:
            T = self.times(device)
            n = len(self.stages)
            # contiguous (batch,) vectors like the reference hands to the network (t.expand(B) of a fresh tensor,
            # torch.cat([t] * 2) under CFG) for models that need contiguous inputs.  Materialised once per (plan, batch)
            # -- two small kernels here, none per step.  `repeat` always copies: the vectors never alias times().
            te = T[0].reshape(n, 1).repeat(1, batch)
            ti = T[1].reshape(n, 1).repeat(1, 2 * batch if cfg else batch)
            if len(self._views) >= 8:                 # bounded: one entry per (device, batch, cfg)
                self._views.pop(next(iter(self._views)))
            hit = dict(t_eval=[T[0, i] for i in range(n)], t_out=[T[2, i] for i in range(n)],
                       t_eval_b=[te[i] for i in range(n)],
                       t_input_b=[ti[i, :batch] for i in range(n)],
                       t_input_2b=[ti[i] for i in range(n)] if cfg else None,
                       base=(T, te, ti))
            if self.stages64 is not None:
                # A double-precision run hands the network the time in the dtype the reference's tensor has there: the
                # grids torch.linspace builds are fp32 tensors also then (ref :472-477), whereas the singlestep solvers'
                # inner nodes and the logSNR grid come out of inverse_lambda on double tables (ref :156-167) as doubles
                # (dpm_stage_f64.time_f64, set by the planner).
                T64 = self.times64(device)
                te64 = T64[0].reshape(n, 1).repeat(1, batch)
                ti64 = T64[1].reshape(n, 1).repeat(1, 2 * batch if cfg else batch)
                for i, s64 in enumerate(self.stages64):
                    if s64.time_f64 & 1:
                        hit["t_eval"][i], hit["t_eval_b"][i] = T64[0, i], te64[i]
                        hit["t_input_b"][i] = ti64[i, :batch]
                        if cfg:
                            hit["t_input_2b"][i] = ti64[i]
                    if s64.time_f64 & 2:
                        hit["t_out"][i] = T64[2, i]
                hit["base"] = (T, te, ti, T64, te64, ti64)
            hit["ver"] = _versions(hit["base"])
            self._views[key] = hit
        return hit

    def written(self, V):
        """True when a network / callback wrote into the shared time tensors of `V` since they were built"""
        return V["ver"] is not None and _versions(V["base"]) != V["ver"]

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                L.lib.dpm_plan_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


class _Cloning:
    """list of cached tensors whose items are handed out as clones (DPM_Solver.fresh_time_tensors)"""

    def __init__(self, items):
        self._items = items

    def __getitem__(self, i):
        return self._items[i].clone()


def get_plan(self, precision=0, **kw):
    mt, gd, sc = self._model_codes()
    key = (tuple(sorted(kw.items())), mt, gd, sc, self._thresholding, float(self.dynamic_thresholding_ratio),
           float(self.thresholding_max_val), self.algorithm_type, int(precision))
    plan = self._plans.get(key)
    if plan is None:
        d = L.PlanDesc()
        d.algorithm_type = self._algo
        d.method = L.METHOD[kw["method"]]
        d.order = int(kw["order"])
        d.steps = int(kw["steps"])
        d.skip_type = L.SKIP[kw["skip_type"]]
        d.solver_type = L.SOLVER[kw["solver_type"]]
        d.lower_order_final = int(bool(kw["lower_order_final"]))
        d.denoise_to_zero = int(bool(kw["denoise_to_zero"]))
        d.model_type, d.guidance, d.guidance_scale = mt, gd, sc
        d.thresholding = int(self._thresholding)
        d.precision = int(precision)
        d.t_start, d.t_end = float(kw["t_T"]), float(kw["t_0"])
        d.thr_ratio = float(self.dynamic_thresholding_ratio)
        d.thr_max = float(self.thresholding_max_val)
        plan = _Plan(self._h, d)
        self._plans[key] = plan
    return plan
