"""The adaptive step-size solver (ref :956-1010), split out of solver.py in round 6: the device-side controller
(`adaptive_device`: C ABI dpm_adaptive_*, no device -> host synchronisation decides anything), the reference's host loop
(`solve`), and the ONE predicate that chooses between them (`runs_on_device`), shared with capture() and auto_capture.
The functions take the DPM_Solver as `self`; solver.py binds them as methods."""
import ctypes as C

import numpy as np
import torch

from . import _device as DV
from . import _lib as L
from .launch_list import _bind_outputs

_F32 = np.float32

class _AdaptiveHandle:
    """a dpm_adaptive handle (device-resident controller state + schedule tables); created outside stream capture"""

    def __init__(self, sched_handle, desc):
        self.handle = C.c_void_p()
        L.check(L.lib.dpm_adaptive_create(sched_handle, C.byref(desc), C.byref(self.handle)))
        self.order = int(desc.order)

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                L.lib.dpm_adaptive_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


class _AdaptiveRun:
    """Device-side adaptive solver (dpm_adaptive_*): the states one run works on and the launch records of the
    3 (order 2) / 4 (order 3) stage launches of an iteration, built once per (configuration, shape, dtype, stream)."""

    def __init__(self, owner, shape, sd, device, cfg):
        self.owner = owner                  # keeps the handle alive
        self.handle = owner.handle
        self.order = owner.order
        B = int(shape[0])
        n = 1
        for d in shape:
            n *= int(d)
        self.n, self.B, self.cfg = n, B, cfg
        mk = lambda: torch.empty(shape, dtype=sd, device=device)
        self.x_prev, self.x_lower, self.x_higher, self.mid1, self.mid2, self.m_s, self.m_s1 = (mk() for _ in range(7))
        self.tv_len = max(2 * B if cfg else B, 1)      # B = 0: an empty shard still runs the controller
        self.tvec = torch.zeros((3, 2, self.tv_len), dtype=torch.float32, device=device)
        self.E = torch.zeros((1,), dtype=torch.float32, device=device)
        tm = []
        for i in range(5):
            st = L.Stage()
            L.check(L.lib.dpm_adaptive_stage_template(self.handle, i, C.byref(st)))
            tm.append(st)

        def given(st):       # the update of `st` with the model value already known (no prologue), cf. _run_given
            g = st.copy()
            g.flags = st.flags & L.F_BASE_HIST
            g.model_type, g.guidance = L.MODEL["noise"], L.GUIDE["uncond"]
            return g

        def rec(st, x, xe, h1, h2, out, m_out):
            st = st.copy()
            b = L.Buffers()
            b.n, b.batch, b.state_dtype, b.eps_dtype = n, max(B, 1), DV._DT[sd], DV._DT[sd]
            if xe is not None:
                b.xe = xe.data_ptr()
            if h1 is not None:
                b.h1 = h1.data_ptr()
            if h2 is not None:
                b.h2 = h2.data_ptr()
            b.x_out = out.data_ptr()
            if m_out is not None:
                st.flags |= L.F_STORE_M
                b.m_out = m_out.data_ptr()
            else:
                st.flags &= ~L.F_STORE_M
            return st, b

        taylor3 = self.order == 3 and tm[4].form == L.FORM_SS3T
        if self.order == 2:
            # eval (x, s) -> x_lower (first update) and m_s; x_s1 from m_s; eval (x_s1, s1) -> x_higher
            self.seq = [(0, True, None) + rec(tm[0], None, None, None, None, self.x_lower, self.m_s),
                        (2, False, self.m_s) + rec(given(tm[2]), None, None, None, None, self.mid1, None),
                        (3, True, self.mid1) + rec(tm[3], None, self.mid1, self.m_s, None, self.x_higher, None)]
        else:
            # eval (x, s) -> x_s1, m_s; eval (x_s1, s1) -> x_lower (singlestep-2), m_s1; x_s2 from m_s, m_s1;
            # eval (x_s2, s2) -> x_higher (singlestep-3)
            self.seq = [(0, True, None) + rec(tm[0], None, None, None, None, self.mid1, self.m_s),
                        (1, True, self.mid1) + rec(tm[1], None, self.mid1, self.m_s, None, self.x_lower, self.m_s1),
                        (3, False, self.m_s1) + rec(given(tm[3]), None, None, self.m_s, None, self.mid2, None),
                        (4, True, self.mid2) + rec(tm[4], None, self.mid2, self.m_s, self.m_s1 if taylor3 else None,
                                                   self.x_higher, None)]


def adaptive_device(self, x, order, t_T, t_0, h_init, atol, rtol, theta, t_err, solver_type):
    """dpm_solver_adaptive with the controller on the device (C ABI dpm_adaptive_*): no
    device -> host synchronisation decides anything.  The host enqueues one iteration ahead of the device
    (`adaptive_lookahead`): before iteration i it waits -- on an event, not on a tensor -- until the device has taken
    the decisions up to iteration i - 1 - lookahead and reads the host-mapped `done` word.  Iterations enqueued after
    the device reached t_0 are no-ops in the solver kernels (the network calls in them are the price of the
    look-ahead: at most `lookahead` iterations).  Under stream capture exactly `adaptive_max_iterations` are
    recorded."""
    device = x.device
    sd = self._sdtype(x)
    mt, gd, sc = self._model_codes()
    cfg = self._wrapped is not None and self._wrapped.effective_guidance == "classifier-free"
    stream, idx, capturing, other = DV._launch_ctx(device)
    key = ("adaptive", order, float(t_T), float(t_0), float(h_init), float(atol), float(rtol), float(theta), float(t_err),
           solver_type, mt, gd, sc, self.algorithm_type, tuple(x.shape), sd, idx, stream, capturing)
    ar = self._fast.get(key)
    if ar is None:
        hkey = key[:15] + (idx,)
        owner = self._adaptive_handles.get(hkey)
        if owner is None:
            if capturing:
                raise RuntimeError("adaptive solver under stream capture: run the same sample() call once eagerly first "
                                   "(the device-side controller allocates its state then; DPM_Solver.capture does that)")
            d = L.AdaptiveDesc()
            d.algorithm_type, d.solver_type, d.order = self._algo, L.SOLVER[solver_type], int(order)
            d.model_type, d.guidance, d.guidance_scale = mt, gd, sc
            d.t_start, d.t_end, d.h_init = float(t_T), float(t_0), float(h_init)
            d.atol, d.rtol, d.theta, d.t_err = float(atol), float(rtol), float(theta), float(t_err)
            if len(self._adaptive_handles) >= 8:     # bounded: one handle per (t range, tolerances, ...) combination
                self._adaptive_handles.pop(next(iter(self._adaptive_handles)))
            owner = self._adaptive_handles[hkey] = _AdaptiveHandle(self._h, d)
        ar = _AdaptiveRun(owner, x.shape, sd, device, cfg)
        if len(self._fast) >= 8:
            self._fast.pop(next(iter(self._fast)))
        self._fast[key] = ar
    B, n, h = ar.B, ar.n, ar.handle
    xs = torch.empty(x.shape, dtype=sd, device=device)
    xs.copy_(x)
    ar.x_prev.copy_(xs)
    px = xs.data_ptr()
    for _, _, _, st, b in ar.seq:
        b.x = px
    dcode = DV._DT[sd]
    tv, tvp, ep = ar.tvec, ar.tvec.data_ptr(), ar.E.data_ptr()
    begin = lambda: L.check(L.lib.dpm_adaptive_begin(h, px, ar.x_prev.data_ptr(), ar.x_lower.data_ptr(),
                                                    ar.x_higher.data_ptr(), n, dcode, ep, tvp, ar.tv_len, stream))
    ctx = torch.cuda.device(idx) if other else None
    if ctx is not None:
        ctx.__enter__()
    try:
        L.check(L.lib.dpm_adaptive_reset(h, stream))
        max_it = self.adaptive_max_iterations or (64 if capturing else 100000)
        look = min(max(int(self.adaptive_lookahead), 0), 24)
        events = []
        done = C.c_int(0)
        it = 0
        while it < max_it:
            if not capturing and it > look:
                j = it - 1 - look
                events[j].synchronize()                      # begin #j has run: its verdict is in the status ring
                events[j] = None
                if L.lib.dpm_adaptive_done_at(h, j):         # the same answer on every rank of a sharded run
                    break
            begin()
            if not capturing:
                ev = torch.cuda.Event()
                ev.record()
                events.append(ev)
            ev_i = 0
            for which, evaluate, src, st, b in (ar.seq if n else ()):   # an empty shard: controller + collectives only
                if evaluate:
                    xe_t = xs if src is None else src
                    te, ti = tv[ev_i, 0, :B], tv[ev_i, 1, :B]
                    if self._wrapped is not None:
                        outs = self._wrapped.raw_outputs(xe_t, te, ti, tv[ev_i, 1] if cfg else None, x_in2=None)
                    else:
                        outs = (self._model_fn(xe_t, te), None, None)
                    keep = _bind_outputs(b, outs[0], outs[1], outs[2], sd, x.shape)
                    ev_i += 1
                else:
                    b.e0 = src.data_ptr()
                L.check(L.lib.dpm_adaptive_stage_launch(h, which, C.byref(st), C.byref(b), stream))
            L.check(L.lib.dpm_adaptive_error(h, ar.x_lower.data_ptr(), ar.x_higher.data_ptr(), ar.x_prev.data_ptr(), B,
                                             n // max(B, 1), dcode, ep, stream))
            if self.error_reduce is not None:               # batch-sharded runs: MAX all-reduce over the ranks
                ar.E.copy_(self.error_reduce(ar.E[0]).reshape(1))
            it += 1
        begin()                                              # the decision on (and commit of) the last iteration
        if not capturing:
            torch.cuda.current_stream(device).synchronize()  # the one wait of the run: its end
            nfe = C.c_int(0)
            L.lib.dpm_adaptive_poll(h, C.byref(done), C.byref(nfe), None, None)
            if not done.value:
                raise RuntimeError("adaptive solver: t_end not reached within %d iterations" % max_it)
            print('adaptive solver nfe', nfe.value)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    return xs


def runs_on_device(self, x):
    """True when method='adaptive' on `x` takes the device-side controller, False when it takes the reference's host loop
    (one .item() per iteration) -- the ONE predicate dpm_solver_adaptive, capture() and auto_capture share: a host loop
    synchronises every iteration and can never be recorded into a graph.

    The device path's state dtype is _sdtype(x): fp32 for a 'discrete' schedule whatever x is (the reference's (1,)-shaped
    fp32 coefficients promote the first update), the explicit state_dtype when given.  The one case it cannot know before
    the first network output is a half-precision x on a 'linear' schedule (see _promoted): host loop.  So are dynamic
    thresholding, a callable correcting_x0_fn and double states.
    The choice must be the same on every rank of a batch-sharded run (error_reduce set): the two loops issue different
    numbers of all-reduces.  Every condition is rank-uniform; an EMPTY shard (batch < world) takes the device path too
    when sharded -- it runs the controller and the collectives, no stage launches."""
    half_unknown = self._state_dtype is None and x.dtype is not torch.float32 and self.noise_schedule.schedule != 'discrete'
    nonempty = x.numel() > 0 or (self.error_reduce is not None and x.dim() > 0)
    return bool(self.adaptive_on_device and x.is_cuda and x.dim() > 0 and nonempty and not self._thresholding
                and self._user_x0 is None and not half_unknown and self._sdtype(x) is not torch.float64)


def solve(self, x, order, t_T, t_0, h_init=0.05, atol=0.0078, rtol=0.05, theta=0.9, t_err=1e-5,
                        solver_type='dpmsolver'):
    DV._require_gpu(x)
    if order not in (2, 3):
        raise ValueError("For adaptive step size solver, order must be 2 or 3, got {}".format(order))
    if solver_type not in ['dpmsolver', 'taylor']:
        raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
    if self._adaptive_runs_on_device(x):
        return self._adaptive_device(x, order, t_T, t_0, h_init, atol, rtol, theta, t_err, solver_type)
    ns = self.noise_schedule
    # the reference's loop variables are tensors of x's dtype (`t_T * torch.ones((1,)).to(x)`, ref :958): with a double
    # state every scalar of the loop -- and of the updates it calls -- is a double, whatever the schedule's dtype
    dbl = self._sdtype(x) is torch.float64
    FT = np.float64 if dbl else _F32
    ev = ns._eval_np64 if dbl else ns._eval_np
    lam = lambda v: FT(ev(L.EVAL_LAMBDA, [v])[0])
    tm = (lambda v: torch.tensor(float(v), dtype=torch.float64)) if dbl else float     # the time argument of the updates
    s = FT(t_T)
    lambda_s = lam(s)
    lambda_0 = lam(FT(t_0))
    h = FT(h_init)
    x_prev = x
    nfe = 0
    if order == 2:
        r1 = 0.5
        lower_update = lambda x, s, t: self.dpm_solver_first_update(x, s, t, return_intermediate=True)
        higher_update = lambda x, s, t, **kw: self.singlestep_dpm_solver_second_update(
            x, s, t, r1=r1, solver_type=solver_type, **kw)
    elif order == 3:
        r1, r2 = 1. / 3., 2. / 3.
        lower_update = lambda x, s, t: self.singlestep_dpm_solver_second_update(
            x, s, t, r1=r1, return_intermediate=True, solver_type=solver_type)
        higher_update = lambda x, s, t, **kw: self.singlestep_dpm_solver_third_update(
            x, s, t, r1=r1, r2=r2, solver_type=solver_type, **kw)
    else:
        raise ValueError("For adaptive step size solver, order must be 2 or 3, got {}".format(order))
    while abs(FT(s - FT(t_0))) > t_err:
        t = FT(ev(L.EVAL_INV_LAMBDA, [FT(lambda_s + h)])[0])
        x_lower, lower_noise_kwargs = lower_update(x, tm(s), tm(t))
        x_higher = higher_update(x, tm(s), tm(t), **lower_noise_kwargs)
        E_dev = DV._adaptive_error(x_lower, x_higher, x_prev, atol, rtol)
        if self.error_reduce is not None:
            E_dev = self.error_reduce(E_dev)     # batch-sharded runs: MAX all-reduce over the ranks
        E = FT(E_dev.item())                     # the one host sync per iteration, as in the reference (ref :1002)
        if E != E:
            # a NaN estimate is never accepted and turns h into NaN: the reference's loop then spins forever (ref :1002-1008)
            raise FloatingPointError("adaptive solver: the error estimate is NaN at t = %r (the state or the network output "
                                     "is not finite); the reference's loop would not terminate here" % float(s))
        if E <= 1.:
            x = x_higher
            s = t
            x_prev = x_lower
            lambda_s = lam(s)
        # torch.float_power(E, -1 / order).float(): the power is an fp32 number also in a double-precision run (ref :1007)
        # (E == 0 -- identical estimates -- gives inf like torch.float_power does: the step is then capped by the range)
        with np.errstate(divide="ignore", over="ignore"):
            h = min(FT(FT(theta) * h * FT(_F32(np.float64(E) ** (-1. / order)))), FT(lambda_0 - lambda_s))
        nfe += order
    print('adaptive solver nfe', nfe)
    return x
