"""The sampling loops (ref :1171-1237), split out of solver.py in round 6.  A plan is a list of stages -- one network
evaluation + one fused kernel each -- and three loops walk it:

  run_plan_fast    no Python callbacks, no intermediates: prebuilt launch records (launch_list._FastRun), per stage the opaque
                   network call, three pointer patches, one dpm_stage_launch
  run_plan_group   the same over several requests (sample_requests): per stage the network calls of all requests and ONE
                   dpm_stage_launch_multi
  run_plan         the general loop: correcting_xt_fn / callable correcting_x0_fn / return_intermediate / the MaskBlend epilogue

The functions take the DPM_Solver as `self`; solver.py binds them as methods (`_run_plan`, `_run_plan_fast`, `_run_plan_group`)."""
import ctypes as C

import torch

from . import _device as DV
from . import _lib as L
from .correctors import MaskBlend
from .launch_list import _FastRun, _bind_outputs
from .plan_cache import _Cloning

def run_plan_group(self, plan, xs, sd, cfg):
    """`_run_plan_fast` over several requests: a set of launch records per request (_FastRun), per stage the network
    calls of all requests and one dpm_stage_launch_multi over a contiguous array of their dpm_buffers."""
    device = xs[0].device
    stream, idx, capturing, other = DV._launch_ctx(device)
    R, shape = len(xs), xs[0].shape
    V = self._time_views(plan, device, shape[0], cfg)
    tb, ti, t2 = V["t_eval_b"], V["t_input_b"], V["t_input_2b"]
    if self.fresh_time_tensors:
        tb, ti, t2 = _Cloning(tb), _Cloning(ti), (_Cloning(t2) if cfg else None)
    wrapped, model_fn = self._wrapped, self._model_fn

    def net(x_t, i, x2=None):
        if wrapped is not None:
            return wrapped.raw_outputs(x_t, tb[i], ti[i], t2[i] if cfg else None, x_in2=x2)
        return model_fn(x_t, tb[i]), None, None
    first0 = net(xs[0], 0)                   # on the callers' x_T (ref :1179, :1222); decides the state dtype
    if not self.fresh_time_tensors and plan.written(V):
        # the network edits its time argument in place and the requests of a stage share one row: clones from here on
        self.fresh_time_tensors = True
        V = self._time_views(plan, device, shape[0], cfg)
        tb, ti, t2 = _Cloning(V["t_eval_b"]), _Cloning(V["t_input_b"]), (_Cloning(V["t_input_2b"]) if cfg else None)
    first = [first0] + [net(x, 0) for x in xs[1:]]
    sd = self._promoted(sd, first[0][0], plan)
    mf = DV._mf_of(first[0][0]) if first[0][0].shape == shape else None
    key = (id(plan), tuple(shape), sd, idx, stream, cfg, R, mf, bool(self.cluster_in_graph), int(self.thr_spin_limit))
    grp = None if capturing else self._fast_groups.get(key)
    if grp is None:
        runs = [_FastRun(self, plan, shape, sd, device, cfg, mf) for _ in range(R)]
        arrs = []
        for i in range(len(plan.stages)):
            a = (L.Buffers * R)()
            for r in range(R):
                C.memmove(C.byref(a, r * C.sizeof(L.Buffers)), C.byref(runs[r].bufs[i]), C.sizeof(L.Buffers))
            arrs.append(a)
        grp = (runs, arrs)
        if not capturing:
            if len(self._fast_groups) >= 4:
                self._fast_groups.pop(next(iter(self._fast_groups)))
            self._fast_groups[key] = grp
    runs, arrs = grp
    x0s = [DV._conv(x, sd, mf) for x in xs]
    outs = [DV._empty(shape, sd, device, mf) for _ in range(R)]
    last, roles = runs[0].last, plan.roles
    launch = DV._stage_launch_multi_raw
    for i, a in enumerate(arrs):
        xi, xei, _ = roles[i]
        keep = []
        for r in range(R):
            b, fr = a[r], runs[r]
            p0 = x0s[r].data_ptr()
            if xi == 0:
                b.x = p0
            if xei == 0:
                xe_t, x2 = x0s[r], None
                if xi != 0:
                    b.xe = p0
            else:
                xe_t, x2 = fr.xbuf[xei], (fr.xfull[xei] if cfg else None)
            if i == last:
                b.x_out = outs[r].data_ptr()
            e = first[r] if i == 0 else net(xe_t, i, x2)
            if sd is torch.float64:
                e = self._cfg_pre(e, sd)
            keep.append(_bind_outputs(b, e[0], e[1], e[2], sd, shape, mf))
        st_ref = runs[0].refs[i][0]
        if other:
            with torch.cuda.device(idx):
                rc = launch(st_ref, a, R, stream)
        else:
            rc = launch(st_ref, a, R, stream)
        if rc:
            L.check(rc)
    return [DV._in_layout_of(o, x) for o, x in zip(outs, xs)]


def run_plan_fast(self, plan, x, sd, cfg):
    """`_run_plan` without correctors / intermediates: prebuilt launch records (see _FastRun), the result in a
    fresh tensor.  Per stage: the opaque network call, three pointer patches, one dpm_stage_launch."""
    device = x.device
    stream, idx, capturing, other = DV._launch_ctx(device)
    B = x.shape[0]
    V = self._time_views(plan, device, B, cfg)
    tb, ti, t2 = V["t_eval_b"], V["t_input_b"], V["t_input_2b"]
    wrapped = self._wrapped
    model_fn = self._model_fn
    # the first evaluation is on the caller's x_T whatever the plan (ref :1179, :1222): run it before choosing
    # the buffers, its output dtype decides the state dtype (see _promoted)
    if self.fresh_time_tensors:
        tb, ti, t2 = _Cloning(tb), _Cloning(ti), (_Cloning(t2) if cfg else None)
    if wrapped is not None:
        first = wrapped.raw_outputs(x, tb[0], ti[0], t2[0] if cfg else None, x_in2=None)
    else:
        first = (model_fn(x, tb[0]), None, None)
    sd = self._promoted(sd, first[0], plan)
    # the network's layout is the run's (see DV._mf_of): an NHWC network gets NHWC states and its outputs are bound as
    # they are; x_T is brought there once and the result goes back to x_T's layout, like ATen would return it
    mf = DV._mf_of(first[0]) if first[0].shape == x.shape else None
    key = (id(plan), tuple(x.shape), sd, idx, stream, cfg, mf, bool(self.cluster_in_graph), int(self.thr_spin_limit))
    fr = None if capturing else self._fast.get(key)      # a captured graph bakes its buffers in: give it its own
    if fr is None:
        fr = _FastRun(self, plan, x.shape, sd, device, cfg, mf)
        if not capturing:
            if len(self._fast) >= 8:
                self._fast.pop(next(iter(self._fast)))
            self._fast[key] = fr
    x0 = DV._conv(x, sd, mf)
    p0 = x0.data_ptr()
    out = DV._empty(x.shape, sd, device, mf)
    bufs, refs, roles = fr.bufs, fr.refs, plan.roles
    bufs[fr.last].x_out = out.data_ptr()
    xbuf, xfull = fr.xbuf, fr.xfull
    launch = DV._stage_launch_raw
    for i, b in enumerate(bufs):
        xi, xei, _ = roles[i]
        if xi == 0:
            b.x = p0
        if xei == 0:
            xe_t, x2 = x0, None
            if xi != 0:
                b.xe = p0
        else:
            xe_t, x2 = xbuf[xei], (xfull[xei] if cfg else None)
        if i == 0:
            e0, e1, g = first
        elif wrapped is not None:
            e0, e1, g = wrapped.raw_outputs(xe_t, tb[i], ti[i], t2[i] if cfg else None, x_in2=x2)
        else:
            e0, e1, g = model_fn(xe_t, tb[i]), None, None
        if sd is torch.float64:
            e0, e1, g = self._cfg_pre((e0, e1, g), sd)
        keep = _bind_outputs(b, e0, e1, g, sd, x.shape, mf)
        if other:
            with torch.cuda.device(idx):
                rc = launch(refs[i][0], refs[i][1], stream)
        else:
            rc = launch(refs[i][0], refs[i][1], stream)
        if rc:
            L.check(rc)
    return out if (mf is None and x.is_contiguous()) else DV._in_layout_of(out, x)


def run_plan(self, plan, x, method, cxt, keep, intermediates):
    if not plan.stages:             # singlestep_fixed with steps < order: no update at all (ref :1218-1232), x comes back as it is
        return x if self._group is None else list(self._group)
    device = x.device
    sd = self._sdtype(x)
    cfg = self._wrapped is not None and self._wrapped.effective_guidance == "classifier-free"
    # denoise_to_zero evaluates the data prediction at a (1,)-shaped time (ref :1236: `torch.ones((1,)) * t_0`): on a
    # continuous schedule its alpha_t / sigma_t are then dimensioned fp32 tensors and the RESULT of a half-precision run
    # is fp32 (on a discrete schedule every run is fp32 anyway).  That last stage runs in fp32 in the general loop below.
    wide_last = (self._state_dtype is None and sd not in (torch.float32, torch.float64) and len(plan.stages) > 0
                 and plan.stages[-1].form == L.FORM_DENOISE and self.noise_schedule.schedule != 'discrete')
    if cxt is None and not keep and self._user_x0 is None and x.dim() > 0 and x.numel() > 0 and not wide_last:
        if self._group is not None:
            return self._run_plan_group(plan, self._group, sd, cfg)
        return self._run_plan_fast(plan, x, sd, cfg)
    if self._group is not None:          # (requests that need the general loop run one after the other)
        grp, self._group = self._group, None
        try:
            return [self._run_plan(plan, xg, method, cxt, keep, intermediates) for xg in grp]
        finally:
            self._group = grp
    V = self._time_views(plan, device, x.shape[0] if x.dim() > 0 else 1, cfg)
    if self.fresh_time_tensors:
        V = dict(V, t_eval_b=_Cloning(V["t_eval_b"]), t_input_b=_Cloning(V["t_input_b"]),
                 t_input_2b=_Cloning(V["t_input_2b"]) if cfg else None)
    blend = cxt if isinstance(cxt, MaskBlend) else None      # folded into the stage kernels' epilogue
    if blend is not None:
        cxt = None
    # classifier-free guidance evaluates the network on cat([x] * 2) (ref :326): let the stage kernel that
    # produces x write both halves of that buffer instead (not possible when an opaque corrector edits x after it)
    dup = cfg and cxt is None and x.dim() > 0
    n_st = len(plan.stages)
    state, state2 = x, None
    tmp, tmp2 = None, None
    hist = [None] * max(plan.slots, 1)
    for i, ps in enumerate(plan.stages):
        st = ps.copy()                # launches may edit flags
        from_tmp = st.xe_src == L.SRC_TMP
        xe = tmp if from_tmp else state
        outs = self._network(xe, None, None, x_in2=tmp2 if from_tmp else state2,
                             pre=(V["t_eval_b"][i], V["t_input_b"][i], V["t_input_2b"][i] if cfg else None))
        if i == 0:
            sd = self._promoted(sd, outs[0], plan)
        if i == 0 and method == 'multistep':
            # ref :1179-1183: the model sees the caller's x_T; the corrector and the list see it afterwards
            if cxt is not None:
                state = cxt(state, V["t_eval"][0], 0)
            elif blend is not None:
                state = blend.apply(state if state.dtype == sd else state.to(sd), ps.t_eval, 0)
            if keep:
                intermediates.append(state)
        h1 = hist[st.h1_slot] if st.h1_slot >= 0 else None
        h2 = hist[st.h2_slot] if st.h2_slot >= 0 else None
        ext = {}
        if dup and i + 1 < n_st:
            ext["dup"] = True
        if blend is not None and st.emits_state:
            ext["blend"] = blend.operands(x.shape, sd, device, ps.t_out, st.outer_step)
        c64 = None
        if sd is torch.float64:
            c64 = self._stage64(st, plan.stages64[i] if plan.stages64 is not None else None)
        if wide_last and i == n_st - 1:
            sd = torch.float32
        x_out, m_out = self._run_stage(st, state, xe, outs, h1, h2, sd, V["t_eval"][i], ext=ext or None, coef64=c64)
        if st.m_slot >= 0:
            hist[st.m_slot] = m_out
        if st.emits_state:
            if cxt is not None:
                t_cb = V["t_out"][i].reshape(1) if st.form == L.FORM_DENOISE else V["t_out"][i]
                x_out = cxt(x_out, t_cb, st.outer_step)
            if keep:
                intermediates.append(DV._in_layout_of(x_out, x))
            state, state2 = x_out, ext.get("x2")
            tmp, tmp2 = None, None
        else:
            tmp, tmp2 = x_out, ext.get("x2")
    return DV._in_layout_of(state, x) if x.dim() > 0 else state
