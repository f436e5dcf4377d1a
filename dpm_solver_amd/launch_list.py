"""Prebuilt launch records of a `sample()` call (split out of solver.py in round 6): `_FastRun` holds, per (plan, shape,
dtypes, device, stream), the scratch states, cached model values and one ready `dpm_stage` + `dpm_buffers` pair per stage;
`_bind_outputs` points a record at the fresh network outputs."""
import ctypes as C

import torch

from . import _device as DV
from . import _lib as L

def _bind_outputs(b, e0, e1, g, sd, shape, mf=None):
    """Point a launch record at the fresh network outputs (e0 / e1 / g): choose the eps dtype the kernels have
    ((fp32 state, any eps) and equal low-precision pairs), read channel slices of a wider output in place
    (eps_stride), convert / copy only when there is no kernel for the layout.  `mf`: the memory format the run's
    buffers are in (None: default-contiguous; channels_last when the network works in NHWC, see DV._mf_of) -- outputs in
    that format are bound as they are.  Returns the tensors to keep alive."""
    ed = e0.dtype
    if ed is not sd and (sd is not torch.float32 or ed not in DV._DT):
        ed = sd
    if g is not None and g.dtype is not ed and sd is torch.float32:
        # the classifier's gradient is an fp32 tensor next to a half-precision network output (ref :320-321: `noise - scale *
        # sigma_t * cond_grad` promotes to fp32): the kernel reads both in ONE dtype -- widen the output, never round the gradient
        ed = torch.float32
    stride = 0
    if mf is not None:
        dense = lambda t: t.is_contiguous(memory_format=mf)
        if not (e0.dtype is ed and dense(e0) and (e1 is None or (e1.dtype is ed and dense(e1)))
                and (g is None or (g.dtype is ed and dense(g)))):
            e0, e1, g = DV._conv(e0, ed, mf), DV._conv(e1, ed, mf), DV._conv(g, ed, mf)
    elif e0.dtype is ed and e0.is_contiguous() and (e1 is None or (e1.dtype is ed and e1.is_contiguous())) \
            and (g is None or (g.dtype is ed and g.is_contiguous())):
        pass
    elif e0.dtype is ed and not e0.is_contiguous() and DV._sample_strided(e0) and e0.shape == shape and (
            e1 is None or (e1.dtype is ed and DV._sample_strided(e1) and e1.stride(0) == e0.stride(0))):
        stride = int(e0.stride(0))      # channel slice of a wider output: read in place
        g = DV._conv(g, ed)
    else:
        e0, e1, g = DV._conv(e0, ed), DV._conv(e1, ed), DV._conv(g, ed)
    b.e0 = e0.data_ptr()
    b.e1 = e1.data_ptr() if e1 is not None else None
    b.g = g.data_ptr() if g is not None else None
    b.eps_dtype = DV._DT[ed]
    b.eps_stride = stride
    return e0, e1, g


class _FastRun:
    """Everything of a `sample()` call that does not change from call to call, built once per (plan, shape, dtypes,
    device, stream): the scratch states the stages ping-pong through, the cached model values, the thresholding
    workspace, and one ready `dpm_stage` + `dpm_buffers` pair per stage with every static pointer filled in.  A call then
    only patches the caller's x_T, the fresh output tensor and the network outputs into those structs and launches.
    Scratch buffers are internal (never handed out), so reusing them across calls on the same stream is safe; the
    result of a call is always a fresh tensor."""

    def __init__(self, solver, plan, shape, sd, device, dup, mf=None):
        B = int(shape[0])
        n = 1
        for d in shape:
            n *= int(d)
        self.shape, self.sd, self.dup, self.n, self.mf = tuple(shape), sd, dup, n, mf
        full = ((2 * B,) + tuple(shape[1:])) if dup else tuple(shape)
        # scratch states and cached model values in the run's memory format (DV._mf_of): the network is handed states in
        # the layout it answers in, the kernels see flat storage either way
        self.xfull = [None] + [DV._empty(full, sd, device, mf) for _ in range(3)]   # [2B,...] under CFG
        self.xbuf = [None] + [t[:B] for t in self.xfull[1:]]
        self.hist = [DV._empty(shape, sd, device, mf) for _ in range(plan.slots)]
        self.ws = None
        self.thr_hint = None
        nstg = len(plan.stages)
        self.stages, self.bufs, self.refs = [], [], []
        esz = torch.empty((), dtype=sd).element_size()
        self.last = nstg - 1
        for i, ps in enumerate(plan.stages):
            st = solver._prep_stage(ps.copy())
            b = L.Buffers()
            xi, xei, oi = plan.roles[i]
            if xi > 0:
                b.x = self.xbuf[xi].data_ptr()
            if xei != xi and xei > 0:
                b.xe = self.xbuf[xei].data_ptr()
            if i != self.last:
                b.x_out = self.xbuf[oi].data_ptr()
                if dup:
                    b.x_out2 = b.x_out + n * esz
            if st.h1_slot >= 0:
                b.h1 = self.hist[st.h1_slot].data_ptr()
            if st.h2_slot >= 0:
                b.h2 = self.hist[st.h2_slot].data_ptr()
            if st.flags & L.F_STORE_M:
                b.m_out = self.hist[st.m_slot].data_ptr()
            b.n, b.batch = n, max(B, 1)
            b.state_dtype = DV._DT[sd]
            if solver._opts_ptr() is not None:
                b.opts = solver._opts_ptr()
            if sd is torch.float64:
                self.coef64 = getattr(self, "coef64", [])
                self.coef64.append(solver._stage64(st, plan.stages64[i] if plan.stages64 is not None else None))
                b.coef64 = C.pointer(self.coef64[-1])
            if st.flags & L.F_THRESH:
                nb = L.lib.dpm_threshold_workspace_bytes(b.batch, n // b.batch)
                if nb:
                    if self.ws is None:      # zero-filled once; every launch leaves it zero-filled
                        self.ws = torch.zeros(nb, dtype=torch.uint8, device=device)
                        # per-sample state the clustered kernel carries from stage to stage (dpm_buffers.thr_hint): the
                        # previous thresholds, from which it predicts the next select bound; stage 0 resets it
                        self.thr_hint = torch.zeros(L.THR_HINT_WORDS * max(B, 1), dtype=torch.float32, device=device)
                    b.workspace = self.ws.data_ptr()
                    b.thr_hint = self.thr_hint.data_ptr()
            self.stages.append(st)
            self.bufs.append(b)
            self.refs.append((C.byref(st), C.byref(b)))
