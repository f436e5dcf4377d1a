"""DPM_Solver -- host mirror of the reference class (dpm_solver_pytorch.py:337-1245).

Same constructor, same `.sample()` / `.inverse()` / `.add_noise()` and the same public per-update
methods.  What differs is where the work happens:

  * every scalar of a sampling run is computed once by the C planner into a list of stages
    (`dpm_plan_create`); this class only walks that list;
  * per stage it calls the opaque network (PyTorch-ROCm, current stream) and then launches ONE fused
    HIP kernel through the C ABI (`dpm_stage_launch`) on the same stream;
  * there is no CPU path: tensors must live on an AMD GPU, and a missing library fails at import.

`ref :NNN` = line in the reference's dpm_solver_pytorch.py.
"""
import ctypes as C
import inspect

import numpy as np
import torch

from . import _lib as L
from .correctors import MaskBlend
from .wrapper import WrappedModel

_DT = {torch.float32: L.DTYPE_F32, torch.float16: L.DTYPE_F16, torch.bfloat16: L.DTYPE_BF16, torch.float64: L.DTYPE_F64}
_F32 = np.float32


def _require_gpu(x):
    if not torch.is_tensor(x) or not x.is_cuda:
        raise RuntimeError(
            "dpm_solver_amd runs on MI355X (gfx950) through its HIP library; got a %s tensor. There is no "
            "CPU fallback -- move the state and the model to the GPU." % (x.device if torch.is_tensor(x) else type(x)))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _sample_strided(t):
    """True when every sample of `t` is a contiguous block but consecutive samples are spaced wider apart: the
    channel slice out[:, :C] of a learned-variance network's [B,2C,H,W] output (runners/diffusion.py:596-603)."""
    return (not t.is_contiguous()) and t.dim() >= 2 and t.shape[0] >= 1 and t[0].is_contiguous() \
        and t.stride(0) > t[0].numel()


def _raw_stream(dev):
    """hipStream_t of torch's current stream on `dev` as an int (what dpm_* entry points take as void*)"""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx), idx


def _launch_ctx(dev):
    """(hipStream_t, device index, stream is capturing, device is not the current one) for launches on `dev`"""
    stream, idx = _raw_stream(dev)
    return stream, idx, torch.cuda.is_current_stream_capturing(), idx != torch.cuda.current_device()


# the C entry point the prebuilt launch records of _FastRun go through (a module attribute so that the CPU test suite
# can put its numpy double of the kernel behind the very same records)
_stage_launch_raw = L.lib.dpm_stage_launch
_stage_launch_multi_raw = L.lib.dpm_stage_launch_multi


def _mf_of(t):
    """The memory format of a dense tensor that is NOT laid out in the default order: torch.channels_last (4-D, NHWC) /
    torch.channels_last_3d (5-D); None = default-contiguous, or neither.  The stage kernels are elementwise over the flat
    storage and thresholding only needs every sample to be one contiguous block -- both hold for these formats -- so a
    trajectory whose network works in NHWC (MIOpen's preferred layout on gfx9) runs on the network's storage untouched,
    where the reference's ATen kernels would read the outputs strided (ref :439, :827-831 are layout-agnostic)."""
    if t.is_contiguous():
        return None
    d = t.dim()
    if d == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return torch.channels_last
    if d == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
        return torch.channels_last_3d
    return None


def _conv(t, dt, mf=None):
    """`t` as a dense tensor of dtype `dt` in memory format `mf` (None: default-contiguous); no copy when it already is one"""
    if t is None:
        return None
    if t.dtype != dt:
        t = t.to(dt)
    if mf is None:
        return t if t.is_contiguous() else t.contiguous()
    return t if t.is_contiguous(memory_format=mf) else t.contiguous(memory_format=mf)


def _empty(shape, dt, dev, mf=None):
    return torch.empty(shape, dtype=dt, device=dev, memory_format=mf if mf is not None else torch.contiguous_format)


def _in_layout_of(t, ref):
    """`t` in the memory format of `ref` (what ATen's elementwise kernels would have returned for an update whose first
    operand is `ref`); a no-op when it already is, or when `ref` is in neither of the two formats"""
    if ref.is_contiguous():
        return _conv(t, t.dtype)
    mf = _mf_of(ref)
    return t if mf is None else _conv(t, t.dtype, mf)


def _launch_stage(st, x, xe, e0, e1, g, h1, h2, state_dtype, want_m=None, ext=None, opts=None, coef64=None):
    """One `dpm_stage_launch` on the current stream.  Allocates x_out (and m_out when the stage stores
    its model value) through torch's caching allocator; returns (x_out, m_out).

    ext (optional dict): 'dup' -> write x_out twice into one [2B,...] buffer (returned as ext['x2'], x_out is its
    first half): the network input of classifier-free guidance; 'blend' -> (mask, period, a, b, alpha, sigma), the
    MaskBlend epilogue."""
    ref_t = x if x is not None else xe
    dev = ref_t.device
    sd = state_dtype
    # the network's layout decides the launch's: a channels_last output is consumed in place and every other operand is
    # brought to that order (no-ops from the second stage on, the states this function hands out are in it).  The mask
    # blend's operands are indexed with a flat period in the default order: such launches stay there.
    mf = _mf_of(e0) if (e0.shape == ref_t.shape and not (ext is not None and ext.get("blend") is not None)) else None
    x, xe, h1, h2 = _conv(x, sd, mf), _conv(xe, sd, mf), _conv(h1, sd, mf), _conv(h2, sd, mf)
    ed = e0.dtype
    if ed not in _DT or (sd != torch.float32 and ed != sd) or ed is torch.float64:
        ed = sd  # only (fp32 state, any eps), equal low-precision pairs and (double, double) have kernels
    eps_stride = 0
    if mf is None and e0.dtype == ed and not e0.is_contiguous() and _sample_strided(e0) and e0.shape == ref_t.shape and (
            e1 is None or (e1.dtype == ed and _sample_strided(e1) and e1.stride(0) == e0.stride(0))):
        eps_stride = int(e0.stride(0))          # read the slice in place: no .contiguous() copy
        g = _conv(g, ed)
    else:
        e0, e1, g = _conv(e0, ed, mf), _conv(e1, ed, mf), _conv(g, ed, mf)
    shape = ref_t.shape
    B = int(shape[0]) if len(shape) > 0 else 1
    b = L.Buffers()                               # zero-initialised
    x2 = None
    if ext is not None and ext.get("dup") and len(shape) > 0:
        x2 = _empty((2 * B,) + tuple(shape[1:]), sd, dev, mf)
        x_out = x2[:B]
        ext["x2"] = x2
        b.x_out2 = x2.data_ptr() + x_out.numel() * x_out.element_size()
    else:
        x_out = _empty(shape, sd, dev, mf)
    store = bool(st.flags & L.F_STORE_M) if want_m is None else want_m
    m_out = None
    if store:
        st.flags |= L.F_STORE_M
        m_out = _empty(shape, sd, dev, mf)
        b.m_out = m_out.data_ptr()
    else:
        st.flags &= ~L.F_STORE_M
    if x is not None:
        b.x = x.data_ptr()
    if xe is not None and (x is None or xe.data_ptr() != x.data_ptr()):
        b.xe = xe.data_ptr()
    b.e0 = e0.data_ptr()
    if e1 is not None:
        b.e1 = e1.data_ptr()
    if g is not None:
        b.g = g.data_ptr()
    if h1 is not None:
        b.h1 = h1.data_ptr()
    if h2 is not None:
        b.h2 = h2.data_ptr()
    b.x_out = x_out.data_ptr()
    b.n = ref_t.numel()
    b.batch = max(B, 1)
    b.state_dtype = _DT[sd]
    b.eps_dtype = _DT[ed]
    b.eps_stride = eps_stride
    if opts is not None:
        b.opts = opts
    if coef64 is not None and sd is torch.float64:
        b.coef64 = C.pointer(coef64)           # the stage's scalars in double (a double-precision plan)
    stream, idx = _raw_stream(dev)
    ws = None
    if st.flags & L.F_THRESH:
        nb = L.lib.dpm_threshold_workspace_bytes(b.batch, b.n // b.batch)
        if nb:
            ws = _cluster_workspace(dev, idx, stream, nb)
            b.workspace = ws.data_ptr()
    if ext is not None and ext.get("blend") is not None:
        mask, period, ba, bb, alpha, sigma = ext["blend"]
        st.flags |= L.F_BLEND
        st.blend_alpha, st.blend_sigma = alpha, sigma
        b.mask, b.blend_a, b.mask_period = mask.data_ptr(), ba.data_ptr(), period
        if bb is not None:
            b.blend_b = bb.data_ptr()
    if idx == torch.cuda.current_device():
        L.check(L.lib.dpm_stage_launch(C.byref(st), C.byref(b), stream))
    else:
        with torch.cuda.device(idx):
            L.check(L.lib.dpm_stage_launch(C.byref(st), C.byref(b), stream))
    return x_out, m_out


_WS_CACHE = {}


def _cluster_workspace(dev, idx, stream, nbytes):
    """Workspace of the clustered thresholding kernel (dpm_threshold_workspace_bytes): zero-filled ONCE here -- every
    launch leaves it zero-filled again -- and kept per (device, stream): launches that share one must be ordered.  Under
    stream capture the graph gets a workspace of its own (its address is baked in)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    key = (idx, stream)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        if len(_WS_CACHE) >= 16:
            _WS_CACHE.pop(next(iter(_WS_CACHE)))
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        _WS_CACHE[key] = ws
    return ws


def _add_noise(sched_handle, x, noise, t_host):
    """out[j] = alpha(t_j) * x + sigma(t_j) * noise[j] for the host times t_host (numpy): one kernel per time.  fp32 times:
    the schedule in fp32 (converted exactly when x is double); float64 times (x must be double): the schedule in double."""
    nt = int(t_host.shape[0])
    out = torch.empty((nt, *x.shape), dtype=x.dtype, device=x.device)
    stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    with torch.cuda.device(x.device):
        if t_host.dtype == np.float64:
            assert x.dtype is torch.float64
            L.check(L.lib.dpm_add_noise_launch_f64(sched_handle, t_host.ctypes.data_as(C.POINTER(C.c_double)), nt, _ptr(x),
                                                   _ptr(noise), _ptr(out), x.numel(), stream))
        else:
            L.check(L.lib.dpm_add_noise_launch(sched_handle, t_host.ctypes.data_as(C.POINTER(C.c_float)), nt, _ptr(x),
                                               _ptr(noise), _ptr(out), x.numel(), _DT[x.dtype], stream))
    return out


def _adaptive_error(x_lower, x_higher, x_prev, atol, rtol):
    """max over the batch of the adaptive solver's per-sample error norm (ref :999-1001) as a 0-dim device tensor:
    one kernel (per-sample RMS + atomic max), no host synchronisation here."""
    B = x_lower.shape[0]
    if B == 0:          # an empty shard of a batch-sharded run: contributes nothing to the batch maximum
        return torch.zeros((), dtype=torch.float32, device=x_lower.device)
    if x_lower.dtype is torch.float64:
        # double state (not a performance path): the reference's own tensor expression (ref :997-1001), on the device
        delta = torch.max(torch.ones_like(x_lower) * atol, rtol * torch.max(torch.abs(x_lower), torch.abs(x_prev.to(x_lower.dtype))))
        v = ((x_higher - x_lower) / delta).reshape((B, -1))
        return torch.sqrt(torch.square(v).mean(dim=-1)).max()
    per_sample = x_lower.numel() // max(B, 1)
    e_dev = torch.empty((B + 1,), dtype=torch.float32, device=x_lower.device)
    xp = x_prev if x_prev.dtype == x_lower.dtype else x_prev.to(x_lower.dtype)
    with torch.cuda.device(x_lower.device):
        L.check(L.lib.dpm_adaptive_error_launch(
            _ptr(x_lower.contiguous()), _ptr(x_higher.contiguous()), _ptr(xp.contiguous()), float(atol), float(rtol),
            _ptr(e_dev), B, per_sample, _DT[x_lower.dtype],
            C.c_void_p(torch.cuda.current_stream(x_lower.device).cuda_stream)))
    return e_dev[B]


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


def _bind_outputs(b, e0, e1, g, sd, shape, mf=None):
    """Point a launch record at the fresh network outputs (e0 / e1 / g): choose the eps dtype the kernels have
    ((fp32 state, any eps) and equal low-precision pairs), read channel slices of a wider output in place
    (eps_stride), convert / copy only when there is no kernel for the layout.  `mf`: the memory format the run's
    buffers are in (None: default-contiguous; channels_last when the network works in NHWC, see _mf_of) -- outputs in
    that format are bound as they are.  Returns the tensors to keep alive."""
    ed = e0.dtype
    if ed is not sd and (sd is not torch.float32 or ed not in _DT):
        ed = sd
    stride = 0
    if mf is not None:
        dense = lambda t: t.is_contiguous(memory_format=mf)
        if not (e0.dtype is ed and dense(e0) and (e1 is None or (e1.dtype is ed and dense(e1)))
                and (g is None or (g.dtype is ed and dense(g)))):
            e0, e1, g = _conv(e0, ed, mf), _conv(e1, ed, mf), _conv(g, ed, mf)
    elif e0.dtype is ed and e0.is_contiguous() and (e1 is None or (e1.dtype is ed and e1.is_contiguous())) \
            and (g is None or (g.dtype is ed and g.is_contiguous())):
        pass
    elif e0.dtype is ed and not e0.is_contiguous() and _sample_strided(e0) and e0.shape == shape and (
            e1 is None or (e1.dtype is ed and _sample_strided(e1) and e1.stride(0) == e0.stride(0))):
        stride = int(e0.stride(0))      # channel slice of a wider output: read in place
        g = _conv(g, ed)
    else:
        e0, e1, g = _conv(e0, ed), _conv(e1, ed), _conv(g, ed)
    b.e0 = e0.data_ptr()
    b.e1 = e1.data_ptr() if e1 is not None else None
    b.g = g.data_ptr() if g is not None else None
    b.eps_dtype = _DT[ed]
    b.eps_stride = stride
    return e0, e1, g


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
            b.n, b.batch, b.state_dtype, b.eps_dtype = n, max(B, 1), _DT[sd], _DT[sd]
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
        # scratch states and cached model values in the run's memory format (_mf_of): the network is handed states in
        # the layout it answers in, the kernels see flat storage either way
        self.xfull = [None] + [_empty(full, sd, device, mf) for _ in range(3)]   # [2B,...] under CFG
        self.xbuf = [None] + [t[:B] for t in self.xfull[1:]]
        self.hist = [_empty(shape, sd, device, mf) for _ in range(plan.slots)]
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
            b.state_dtype = _DT[sd]
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


class DPM_Solver:
    # Engine options (extensions; class-level defaults, settable per instance).
    # adaptive solver: controller on the device (no host synchronisation per iteration); False = the reference's host
    # loop (one .item() per iteration, ref :1002).  With the device controller the host enqueues `adaptive_lookahead`
    # iterations ahead of the device's decisions: up to that many iterations enqueued after t_end was reached still call
    # the network (their solver kernels are no-ops) -- the reported NFE and the result do not change, the number of
    # network calls can be (lookahead + 1) * order larger than the reference's.
    adaptive_on_device = True
    adaptive_lookahead = 1
    # per-call launch options of the C ABI (dpm_launch_opts), handed to every launch of this solver:
    #   cluster_in_graph  dynamic thresholding keeps its workgroup clusters under hipGraph capture also for samples that fit
    #                     one workgroup (default: one workgroup per sample there -- a replayed graph runs outside the
    #                     library's per-device chain of clustered launches)
    #   thr_spin_limit    polls before a wait between the workgroups of a thresholding cluster gives up and the workgroup
    #                     finishes its sample alone (0 = the library's default, 4096: milliseconds)
    cluster_in_graph = False
    thr_spin_limit = 0
    # auto_capture = N > 0 (opt-in): after N sample() calls with the same arguments, shape, dtype and stream, the call is
    # hipGraph-captured (DPM_Solver.capture) and later calls replay the graph -- one launch per trajectory instead of one per
    # kernel, which is what a launch-bound loop needs ([8,4,64,64]: 98 -> 52 us per trajectory with a frozen network).  Not
    # the default: a replay does not re-run the PYTHON side of the network (counters, hooks, host-side control flow,
    # conditioning tensors swapped for new objects), which the reference's eager loop does at every call -- the caller has
    # to know its network is a pure function of (x, t) on fixed tensors, as for torch.cuda.graph.  Calls with Python callbacks
    # (correcting_xt_fn, a callable correcting_x0_fn), return_intermediate or a host-side adaptive loop are never captured.
    auto_capture = 0

    def __init__(self, model_fn, noise_schedule, algorithm_type="dpmsolver++", correcting_x0_fn=None,
                 correcting_xt_fn=None, thresholding_max_val=1., dynamic_thresholding_ratio=0.995,
                 state_dtype=None):
        """Construct a DPM-Solver (signature of ref :338-347; `state_dtype` is an extension).

        state_dtype: dtype the solver keeps x and the cached model values in.  None follows the reference:
        with a 'discrete' schedule a half-precision x_T is promoted to fp32 by the first update (the
        reference's coefficients are (1,)-shaped fp32 tensors), with 'linear' it keeps x's dtype.
        """
        self.model = lambda x, t: model_fn(x, t.expand((x.shape[0])))
        self._model_fn = model_fn
        self._wrapped = model_fn if isinstance(model_fn, WrappedModel) else None
        self.noise_schedule = noise_schedule
        assert algorithm_type in ["dpmsolver", "dpmsolver++"]
        self.algorithm_type = algorithm_type
        self._thresholding = correcting_x0_fn == "dynamic_thresholding"
        if self._thresholding:
            self.correcting_x0_fn = self.dynamic_thresholding_fn
            self._user_x0 = None
        else:
            self.correcting_x0_fn = correcting_x0_fn
            self._user_x0 = correcting_x0_fn
        self._user_x0_nargs = None
        if self._user_x0 is not None:
            try:  # the older vendored revision calls correcting_x0_fn(x0) with one argument
                self._user_x0_nargs = len([p for p in inspect.signature(self._user_x0).parameters.values()
                                           if p.default is p.empty and p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)])
            except (TypeError, ValueError):
                self._user_x0_nargs = 2
        self.correcting_xt_fn = correcting_xt_fn
        self.dynamic_thresholding_ratio = dynamic_thresholding_ratio
        self.thresholding_max_val = thresholding_max_val
        self._state_dtype = state_dtype
        # The (batch,) time vectors handed to the network are built once per (plan, batch) and shared by every call
        # (the reference makes a fresh tensor per call, ref :404).  A network that writes into its time argument gets a
        # clone per call: fresh_time_tensors = True, switched on automatically when such a write is detected
        # (_Plan.time_views).
        self.fresh_time_tensors = False
        self._plans = {}
        self._auto = {}                      # auto_capture: (arguments, shape, ...) -> [calls seen, GraphedSample]
        self._fast = {}
        self._fast_groups = {}
        self._group = None                   # sample_requests: the requests advanced together by this call
        self._adaptive_handles = {}
        # adaptive solver: optional hook applied to the 0-dim batch-maximum error before the controller reads it
        self.error_reduce = None
        self.adaptive_max_iterations = None  # bound of the loop (required knowledge under hipGraph capture: default 64)

    # ------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------
    @property
    def _h(self):
        return self.noise_schedule._h

    @property
    def _algo(self):
        return L.ALGO[self.algorithm_type]

    def _opts_ptr(self):
        """ctypes pointer to this solver's dpm_launch_opts, or None when everything is at its default"""
        key = (bool(self.cluster_in_graph), int(self.thr_spin_limit))
        if key == (False, 0):
            return None
        hit = getattr(self, "_opts_cache", None)
        if hit is None or hit[0] != key:
            o = L.LaunchOpts()
            o.cluster_in_graph, o.thr_spin_limit = int(key[0]), key[1]
            hit = self._opts_cache = (key, o, C.pointer(o))
        return hit[2]

    def _model_codes(self):
        if self._wrapped is not None:
            w = self._wrapped
            return L.MODEL[w.model_type], L.GUIDE[w.effective_guidance], float(w.guidance_scale)
        return L.MODEL["noise"], L.GUIDE["uncond"], 1.0

    def _sdtype(self, x):
        if self._state_dtype is not None:
            return self._state_dtype
        if x.dtype not in _DT:
            raise NotImplementedError("dpm_solver_amd: state dtype %s is not supported (fp64 / fp32 / fp16 / bf16)" % x.dtype)
        # torch's type promotion between x and the reference's (1,)-shaped coefficient tensors (ref :573-576), which have the
        # dtype of the schedule's tables for 'discrete' and of the time tensor (fp32) otherwise
        if x.dtype is torch.float64:
            return torch.float64
        if self.noise_schedule.schedule == 'discrete':
            return torch.float64 if getattr(self.noise_schedule, "dtype", torch.float32) == torch.float64 else torch.float32
        return x.dtype

    def _precision(self, sd):
        """1: the scalars of the run are doubles -- a double state on a 'discrete' schedule declared dtype=float64 (every
        coefficient of the reference is then a double tensor); 0: fp32 scalars (meeting a double state they are converted
        exactly, like the reference's fp32 coefficient tensors are by type promotion)"""
        ns = self.noise_schedule
        return int(sd is torch.float64 and ns.schedule == 'discrete' and getattr(ns, "dtype", torch.float32) == torch.float64)

    @staticmethod
    def _tf(t):
        """time argument (tensor of one element, or float) -> fp32 host value"""
        if torch.is_tensor(t):
            return float(t.detach().reshape(-1)[0].float().item())
        return float(_F32(t))

    def _tt(self, value, device, shape1=False, dtype=torch.float32):
        t = torch.full((1,) if shape1 else (), float(value), dtype=dtype, device=device)
        return t

    @staticmethod
    def _td(t):
        """time argument -> host double (exact for fp32 / fp64 tensors and Python floats)"""
        if torch.is_tensor(t):
            return float(t.detach().reshape(-1)[0].double().item())
        return float(t)

    def _double_call(self, x, *times):
        """(scalars are doubles, the time tensors are doubles) for a public per-update call: torch's type promotion makes
        every scalar of the reference a double when a time tensor is one or the schedule's tables are (ref :127-134)"""
        tf64 = any(torch.is_tensor(t) and t.dtype is torch.float64 for t in times)
        ns = self.noise_schedule
        return bool(tf64 or (ns.schedule == 'discrete' and getattr(ns, "dtype", torch.float32) == torch.float64)), tf64

    def _time_views(self, plan, device, batch, cfg):
        """plan.time_views; a network that was caught writing into the shared time vectors gets clones from now on"""
        V = plan.time_views(device, batch, cfg)
        if plan.times_written and not self.fresh_time_tensors:
            self.fresh_time_tensors = True
        return V

    def _call_x0(self, x0, t):
        if self._user_x0_nargs == 1:
            return self._user_x0(x0)
        return self._user_x0(x0, t)

    def _network(self, x_eval, t_eval_t, t_input_t, x_in2=None, pre=None):
        """the opaque call: raw output(s) of the network at (x_eval, t).  x_in2: [2B,...] buffer both halves of
        which already hold x_eval (written by the previous stage kernel) -- the CFG network input.  pre: the
        time tensors already expanded to the batch (t_eval_b, t_input_b, t_input_2b)."""
        B = x_eval.shape[0]
        if pre is not None:
            te, ti, t2 = pre
        else:
            te, ti = t_eval_t.expand(B), t_input_t.expand(B)
            t2 = t_input_t.expand(2 * B) if (self._wrapped is not None and
                                             self._wrapped.effective_guidance == "classifier-free") else None
        if self._wrapped is not None:
            return self._wrapped.raw_outputs(x_eval, te, ti, t2, x_in2=x_in2)
        return self._model_fn(x_eval, te), None, None

    def _promoted(self, sd, e0, plan=None):
        """State dtype after the first network evaluation.  Without an explicit `state_dtype` the reference's type
        promotion applies: a half-precision state combined with a network output of another floating dtype (fp32
        eps next to an fp16 x on a 'linear' schedule, or fp16 next to bf16) continues in fp32 from the first update on
        (ref :573-576 are plain tensor expressions).  An explicit `state_dtype` keeps the state there and the output
        is converted to it.

        A half state on a continuous schedule also leaves half precision when a singlestep update of order >= 2 runs: its
        intermediate time comes out of inverse_lambda, which for 'linear' builds a (1,)-shaped tensor (ref :161: `torch.zeros((1,))`)
        -- a dimensioned fp32 coefficient, and the state is fp32 from that update on (measured against the reference:
        tests/test_differential_reference.py).  The first such update is the trajectory's first or second, so the run continues
        in fp32 from its first update; the one deviation is that the reference rounds the FIRST model value to half once."""
        if self._state_dtype is None and e0.dtype is torch.float64:
            return torch.float64                    # a double network output promotes every state
        if self._state_dtype is None and sd not in (torch.float32, torch.float64) and e0.dtype is not sd and e0.dtype in _DT:
            return torch.float32
        if (self._state_dtype is None and sd not in (torch.float32, torch.float64) and plan is not None and plan.has_inner_nodes
                and self.noise_schedule.schedule != 'discrete'):
            return torch.float32
        return sd

    def _cfg_pre(self, outs, sd):
        """Double state, classifier-free guidance, a noise-prediction network that answers in a narrower dtype: the reference
        blends `uncond + scale * (cond - uncond)` in the NETWORK's dtype (ref :326-330, a Python-float scale) before any double
        scalar touches the result.  The double kernel would blend in double, so the blend is taken here, in the reference's
        own expression, and handed on as both halves (e1 + s * (e0 - e1) with e0 == e1 is exact)."""
        e0, e1, g = outs
        if (sd is torch.float64 and e1 is not None and e0.dtype is not torch.float64 and self._wrapped is not None
                and self._wrapped.model_type == "noise"):
            e = e1 + self._wrapped.guidance_scale * (e0 - e1)
            return e, e, g
        return outs

    def _stage64(self, st, s64=None):
        """the dpm_stage_f64 of a launch on a double state: the double-precision plan's record, or -- fp32 scalars (an fp32
        schedule) -- the stage's floats converted exactly, as torch's type promotion converts the reference's fp32
        coefficient tensors; either way with the thresholding parameters as the doubles torch.quantile / torch.maximum make
        of the Python floats (ref :422-423)"""
        out = L.StageF64()
        if s64 is not None:
            C.memmove(C.byref(out), C.byref(s64), C.sizeof(L.StageF64))
        else:
            for f in ("t_eval", "t_input", "t_out", "alpha_e", "sigma_e", "cfg_scale", "cg_scale", "cx", "c0", "c1", "c2",
                      "blend_alpha", "blend_sigma"):
                setattr(out, f, float(getattr(st, f)))
            for j in range(5):
                out.k[j] = float(st.k[j])
        out.thr_ratio = float(self.dynamic_thresholding_ratio)
        out.thr_max = float(self.thresholding_max_val)
        return out

    def _prep_stage(self, st):
        """stage flags that depend on this solver's correctors"""
        if st.flags & L.F_TO_X0:
            if self._thresholding:
                st.flags |= L.F_THRESH
                st.thr_ratio = float(self.dynamic_thresholding_ratio)
                st.thr_max = float(self.thresholding_max_val)
        return st

    def _run_stage(self, st, x, xe, outs, h1, h2, sd, t_eval_t, want_m=None, ext=None, coef64=None):
        """outs = (e0, e1, g) fresh network outputs.  Handles a *callable* correcting_x0_fn by splitting the
        stage: prologue kernel -> user function (opaque torch) -> combination kernel."""
        e0, e1, g = outs if sd is not torch.float64 else self._cfg_pre(outs, sd)
        if self._user_x0 is not None and (st.flags & L.F_TO_X0):
            s1 = st.copy()
            s1.form = L.FORM_DENOISE
            s1.flags = L.F_TO_X0
            x0, _ = _launch_stage(s1, None, xe if xe is not None else x, e0, e1, g, None, None, sd, want_m=False,
                                  opts=self._opts_ptr(), coef64=coef64)
            x0 = self._call_x0(x0, t_eval_t)                                             # ref :440-441
            return self._run_given(st, x, x0, h1, h2, sd, want_m, ext=ext, coef64=coef64)
        return _launch_stage(st, x, xe, e0, e1, g, h1, h2, sd, want_m=want_m, ext=ext, opts=self._opts_ptr(), coef64=coef64)

    def _run_given(self, st, x, m, h1, h2, sd, want_m=None, ext=None, coef64=None):
        """the update of `st` with the model value already known (no prologue)"""
        s2 = st.copy()
        s2.flags = st.flags & L.F_BASE_HIST
        s2.model_type = L.MODEL["noise"]
        s2.guidance = L.GUIDE["uncond"]
        store = bool(st.flags & L.F_STORE_M) if want_m is None else want_m
        x_out, _ = _launch_stage(s2, x if x is not None else m, None, m, None, None, h1, h2, sd, want_m=False, ext=ext,
                                 opts=self._opts_ptr(), coef64=coef64)
        return x_out, (m if store else None)

    # ------------------------------------------------------------------------------------------
    # model evaluations (ref :416-451, :541-545)
    # ------------------------------------------------------------------------------------------
    def dynamic_thresholding_fn(self, x0, t=None):
        """The dynamic thresholding method (ref :416-425) as one kernel launch."""
        _require_gpu(x0)
        st = L.Stage()
        st.h1_slot = st.h2_slot = st.m_slot = -1
        st.form = L.FORM_DENOISE
        st.flags = L.F_THRESH
        st.thr_ratio = float(self.dynamic_thresholding_ratio)
        st.thr_max = float(self.thresholding_max_val)
        st.alpha_e, st.sigma_e, st.cfg_scale = 1.0, 0.0, 1.0
        sd = x0.dtype if x0.dtype in _DT else torch.float32
        out, _ = _launch_stage(st, None, x0, x0, None, None, None, None, sd, want_m=False, opts=self._opts_ptr(),
                               coef64=self._stage64(st) if sd is torch.float64 else None)
        return out

    def _eval_model(self, x, t, to_x0):
        _require_gpu(x)
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
            sd = outs[0].dtype if outs[0].dtype in _DT else torch.float32
        else:
            sd = torch.float64 if dbl else self._sdtype(x)
        out, _ = self._run_stage(st, None, x, outs, None, None, sd, t if torch.is_tensor(t) else self._tt(tf, dev), want_m=False,
                                 coef64=self._stage64(st, c64) if sd is torch.float64 else None)
        return out

    def noise_prediction_fn(self, x, t):
        """Return the noise prediction model (ref :427-431)."""
        return self._eval_model(x, t, to_x0=False)

    def data_prediction_fn(self, x, t):
        """Return the data prediction model, with corrector (ref :433-442)."""
        return self._eval_model(x, t, to_x0=True)

    def model_fn(self, x, t):
        """noise prediction for 'dpmsolver', data prediction for 'dpmsolver++' (ref :444-451)."""
        return self._eval_model(x, t, to_x0=self.algorithm_type == "dpmsolver++")

    def denoise_to_zero_fn(self, x, s):
        """ref :541-545"""
        return self.data_prediction_fn(x, s)

    # ------------------------------------------------------------------------------------------
    # time grids (ref :453-539)
    # ------------------------------------------------------------------------------------------
    def get_time_steps(self, skip_type, t_T, t_0, N, device):
        if skip_type not in L.SKIP:
            raise ValueError("Unsupported skip_type {}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'".format(skip_type))
        out = np.empty(N + 1, dtype=np.float32)
        L.check(L.lib.dpm_time_steps(self._h, L.SKIP[skip_type], float(t_T), float(t_0), int(N),
                                     out.ctypes.data_as(C.POINTER(C.c_float))))
        return torch.from_numpy(out).to(device)

    def get_orders_and_timesteps_for_singlestep_solver(self, steps, order, skip_type, t_T, t_0, device):
        if order not in (1, 2, 3):
            raise ValueError("'order' must be '1' or '2' or '3'.")
        if skip_type not in L.SKIP:
            raise ValueError("Unsupported skip_type {}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'".format(skip_type))
        outer = np.empty(steps + 2, dtype=np.float32)
        orders = (C.c_int * (steps + 1))()
        n = C.c_int()
        L.check(L.lib.dpm_singlestep_grid(self._h, int(steps), int(order), L.SKIP[skip_type], float(t_T), float(t_0),
                                          outer.ctypes.data_as(C.POINTER(C.c_float)), orders, C.byref(n)))
        return torch.from_numpy(outer[: n.value + 1].copy()).to(device), [int(orders[i]) for i in range(n.value)]

    # ------------------------------------------------------------------------------------------
    # public per-update methods (ref :547-954)
    # ------------------------------------------------------------------------------------------
    def _exec_single(self, stages, x, given, want, c64s=None, tf64=False):
        """Run 1-3 singlestep stages starting from state x.  `given[i]` = model value already known for
        stage i; `want` = return the model values.  Returns (x_t, [m_0, m_1, m_2]).  c64s: the stages' doubles
        (dpm_coef_singlestep_f64) when the call's scalars are doubles; tf64: the caller's time tensors are doubles."""
        _require_gpu(x)
        dev = x.device
        sd = torch.float64 if c64s is not None else self._sdtype(x)
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

    def _singlestep_stages(self, x, order, solver_code, s, t, r1, r2, mode):
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

    def dpm_solver_first_update(self, x, s, t, model_s=None, return_intermediate=False):
        """DPM-Solver-1 (equivalent to DDIM) from time `s` to time `t` (ref :547-592)."""
        stages, c64s, tf64 = self._singlestep_stages(x, 1, 0, s, t, 0., 0., 0)
        x_t, ms = self._exec_single(stages, x, {0: model_s} if model_s is not None else {}, return_intermediate, c64s, tf64)
        return (x_t, {'model_s': ms[0]}) if return_intermediate else x_t

    def singlestep_dpm_solver_second_update(self, x, s, t, r1=0.5, model_s=None, return_intermediate=False,
                                            solver_type='dpmsolver'):
        """Singlestep solver DPM-Solver-2 from time `s` to time `t` (ref :594-673)."""
        if solver_type not in ['dpmsolver', 'taylor']:
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        if r1 is None:
            r1 = 0.5
        mode = 1 if torch.is_tensor(r1) else 0
        stages, c64s, tf64 = self._singlestep_stages(x, 2, L.SOLVER[solver_type], s, t, r1 if mode else float(r1), 0., mode)
        x_t, ms = self._exec_single(stages, x, {0: model_s} if model_s is not None else {}, return_intermediate, c64s, tf64)
        return (x_t, {'model_s': ms[0], 'model_s1': ms[1]}) if return_intermediate else x_t

    def singlestep_dpm_solver_third_update(self, x, s, t, r1=1. / 3., r2=2. / 3., model_s=None, model_s1=None,
                                           return_intermediate=False, solver_type='dpmsolver'):
        """Singlestep solver DPM-Solver-3 from time `s` to time `t` (ref :675-794)."""
        if solver_type not in ['dpmsolver', 'taylor']:
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        if r1 is None:
            r1 = 1. / 3.
        if r2 is None:
            r2 = 2. / 3.
        mode = 1 if (torch.is_tensor(r1) or torch.is_tensor(r2)) else 0
        stages, c64s, tf64 = self._singlestep_stages(x, 3, L.SOLVER[solver_type], s, t, r1 if mode else float(r1),
                                                     r2 if mode else float(r2), mode)
        given = {}
        if model_s is not None:
            given[0] = model_s
        if model_s1 is not None:
            given[1] = model_s1
        if 1 in given and 0 not in given:
            # reference: model_s is evaluated at (x, s) even when model_s1 is supplied (ref :720-721)
            given[0] = self.model_fn(x, s)
        # the taylor combination reads model_s1 (h2): keep it even when not asked for
        x_t, ms = self._exec_single(stages, x, given, return_intermediate or solver_type == 'taylor', c64s, tf64)
        return (x_t, {'model_s': ms[0], 'model_s1': ms[1], 'model_s2': ms[2]}) if return_intermediate else x_t

    def _multistep(self, x, model_prev_list, t_prev_list, t, order, solver_type):
        _require_gpu(x)
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
        sd = torch.float64 if dbl else self._sdtype(x)
        x_t, _ = self._run_given(st, x, model_prev_list[-1], h1, h2, sd, want_m=False,
                                 coef64=self._stage64(st, c64) if sd is torch.float64 else None)
        return x_t

    def multistep_dpm_solver_second_update(self, x, model_prev_list, t_prev_list, t, solver_type="dpmsolver"):
        """Multistep solver DPM-Solver-2 from time `t_prev_list[-1]` to time `t` (ref :796-852)."""
        if solver_type not in ['dpmsolver', 'taylor']:
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        return self._multistep(x, model_prev_list, t_prev_list, t, 2, solver_type)

    def multistep_dpm_solver_third_update(self, x, model_prev_list, t_prev_list, t, solver_type='dpmsolver'):
        """Multistep solver DPM-Solver-3 from time `t_prev_list[-1]` to time `t` (ref :854-904)."""
        return self._multistep(x, model_prev_list, t_prev_list, t, 3, solver_type if solver_type in L.SOLVER else 'dpmsolver')

    def singlestep_dpm_solver_update(self, x, s, t, order, return_intermediate=False, solver_type='dpmsolver',
                                     r1=None, r2=None):
        """Singlestep DPM-Solver with the order `order` from time `s` to time `t` (ref :906-930)."""
        if order == 1:
            return self.dpm_solver_first_update(x, s, t, return_intermediate=return_intermediate)
        elif order == 2:
            return self.singlestep_dpm_solver_second_update(x, s, t, return_intermediate=return_intermediate,
                                                            solver_type=solver_type, r1=r1)
        elif order == 3:
            return self.singlestep_dpm_solver_third_update(x, s, t, return_intermediate=return_intermediate,
                                                           solver_type=solver_type, r1=r1, r2=r2)
        else:
            raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))

    def multistep_dpm_solver_update(self, x, model_prev_list, t_prev_list, t, order, solver_type='dpmsolver'):
        """Multistep DPM-Solver with the order `order` from time `t_prev_list[-1]` to time `t` (ref :932-954)."""
        if order == 1:
            return self.dpm_solver_first_update(x, t_prev_list[-1], t, model_s=model_prev_list[-1])
        elif order == 2:
            return self.multistep_dpm_solver_second_update(x, model_prev_list, t_prev_list, t, solver_type=solver_type)
        elif order == 3:
            return self.multistep_dpm_solver_third_update(x, model_prev_list, t_prev_list, t, solver_type=solver_type)
        else:
            raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))

    # ------------------------------------------------------------------------------------------
    # adaptive step size (ref :956-1010): the control loop is host logic, the work is stage kernels
    # ------------------------------------------------------------------------------------------
    def _adaptive_device(self, x, order, t_T, t_0, h_init, atol, rtol, theta, t_err, solver_type):
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
        stream, idx, capturing, other = _launch_ctx(device)
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
        dcode = _DT[sd]
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

    def _adaptive_runs_on_device(self, x):
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

    def dpm_solver_adaptive(self, x, order, t_T, t_0, h_init=0.05, atol=0.0078, rtol=0.05, theta=0.9, t_err=1e-5,
                            solver_type='dpmsolver'):
        _require_gpu(x)
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
            E_dev = _adaptive_error(x_lower, x_higher, x_prev, atol, rtol)
            if self.error_reduce is not None:
                E_dev = self.error_reduce(E_dev)     # batch-sharded runs: MAX all-reduce over the ranks
            E = FT(E_dev.item())                     # the one host sync per iteration, as in the reference (ref :1002)
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

    # ------------------------------------------------------------------------------------------
    # add_noise / inverse / sample (ref :1012-1245)
    # ------------------------------------------------------------------------------------------
    def add_noise(self, x, t, noise=None):
        """xt = alpha_t * x + sigma_t * noise for every t; returns (t_size, batch, *shape) (ref :1012-1030)."""
        _require_gpu(x)
        nt = int(t.reshape(-1).shape[0])
        if noise is None:
            noise = torch.randn((nt, *x.shape), device=x.device)
        if x.dtype not in _DT:
            raise NotImplementedError("add_noise: dtype %s" % x.dtype)
        dbl, _ = self._double_call(x, t)
        # alpha_t / sigma_t of the reference are (nt,)-shaped tensors (ref :1023): when they are doubles -- a double t, or tables
        # declared dtype=float64 -- or x / noise is, torch's type promotion makes the result float64.  Double scalars are
        # evaluated in double at the double times; fp32 scalars meeting a double tensor are converted exactly.
        wide = dbl or x.dtype is torch.float64 or noise.dtype is torch.float64
        xdt = torch.float64 if wide else x.dtype
        th = t.detach().to(device="cpu", dtype=torch.float64 if dbl else torch.float32).reshape(-1).numpy().copy()
        out = _add_noise(self._h, x.to(xdt).contiguous(), noise.to(xdt).contiguous(), th)
        return out.squeeze(0) if nt == 1 else out

    def inverse(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type='time_uniform',
                method='multistep', lower_order_final=True, denoise_to_zero=False, solver_type='dpmsolver',
                atol=0.0078, rtol=0.05, return_intermediate=False):
        """Inverse the sample `x` from time `t_start` to `t_end` by DPM-Solver (ref :1032-1045)."""
        t_0 = 1. / self.noise_schedule.total_N if t_start is None else t_start
        t_T = self.noise_schedule.T if t_end is None else t_end
        assert t_0 > 0 and t_T > 0, "Time range needs to be greater than 0. For discrete-time DPMs, it needs to be in [1 / N, 1], where N is the length of betas array"
        return self.sample(x, steps=steps, t_start=t_0, t_end=t_T, order=order, skip_type=skip_type,
                           method=method, lower_order_final=lower_order_final, denoise_to_zero=denoise_to_zero,
                           solver_type=solver_type, atol=atol, rtol=rtol, return_intermediate=return_intermediate)

    def _get_plan(self, precision=0, **kw):
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

    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type='time_uniform',
               method='multistep', lower_order_final=True, denoise_to_zero=False, solver_type='dpmsolver',
               atol=0.0078, rtol=0.05, return_intermediate=False):
        """Compute the sample at time `t_end` by DPM-Solver, given the initial `x` at time `t_start`
        (signature and semantics of ref :1047-1245; see that docstring for the argument meanings)."""
        t_0 = 1. / self.noise_schedule.total_N if t_end is None else t_end
        t_T = self.noise_schedule.T if t_start is None else t_start
        assert t_0 > 0 and t_T > 0, "Time range needs to be greater than 0. For discrete-time DPMs, it needs to be in [1 / N, 1], where N is the length of betas array"
        if return_intermediate:
            assert method in ['multistep', 'singlestep', 'singlestep_fixed'], "Cannot use adaptive solver when saving intermediate values"
        if self.correcting_xt_fn is not None:
            assert method in ['multistep', 'singlestep', 'singlestep_fixed'], "Cannot use adaptive solver when correcting_xt_fn is not None"
        _require_gpu(x)
        device = x.device
        intermediates = []
        cxt = self.correcting_xt_fn
        if self.auto_capture and self._group is None and not torch.cuda.is_current_stream_capturing():
            hit = self._auto_captured(x, dict(steps=steps, t_start=t_start, t_end=t_end, order=order, skip_type=skip_type,
                                              method=method, lower_order_final=lower_order_final,
                                              denoise_to_zero=denoise_to_zero, solver_type=solver_type, atol=atol, rtol=rtol),
                                      return_intermediate)
            if hit is not None:
                return hit
        with torch.no_grad():
            if method == 'adaptive':
                x = self.dpm_solver_adaptive(x, order=order, t_T=t_T, t_0=t_0, atol=atol, rtol=rtol, solver_type=solver_type)
                if denoise_to_zero:
                    x = self.denoise_to_zero_fn(x, self._tt(t_0, device, shape1=True))
            elif method in ['multistep', 'singlestep', 'singlestep_fixed']:
                if skip_type not in L.SKIP:
                    raise ValueError("Unsupported skip_type {}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'".format(skip_type))
                if solver_type not in L.SOLVER:
                    raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
                if method == 'multistep':
                    # the order is validated where the reference validates it -- when an update of that order is reached
                    # (ref :948-954): order=4 with steps <= 6 and lower_order_final never reaches one and runs (the planner
                    # raises the reference's ValueError otherwise)
                    assert steps >= order
                elif order not in (1, 2, 3):
                    raise ValueError("'order' must be '1' or '2' or '3'.")
                plan = self._get_plan(precision=self._precision(self._sdtype(x)), method=method, order=order, steps=steps,
                                      skip_type=skip_type, solver_type=solver_type,
                                      lower_order_final=lower_order_final, denoise_to_zero=denoise_to_zero,
                                      t_T=float(t_T), t_0=float(t_0))
                x = self._run_plan(plan, x, method, cxt, return_intermediate, intermediates)
            else:
                raise ValueError("Got wrong method {}".format(method))
        if return_intermediate:
            return x, intermediates
        else:
            return x

    def sample_requests(self, xs, **sample_kwargs):
        """(extension) `sample()` for several independent requests that are in flight together -- a server's batch of
        sampling jobs with their own states, e.g. `xs = [x_T_0, ..., x_T_31]`, all of one shape, dtype and device.
        Returns the list of results, identical to `[sample(x, **sample_kwargs) for x in xs]`.

        The requests are advanced stage by stage: the network is called once per request (each on its own state), then
        ONE fused kernel advances all of them (dpm_stage_launch_multi; 32 requests per launch).  A lone 42 MB launch
        whose inputs come from HBM -- they always do when a network ran in between -- spends a third of its time ramping
        up and draining; fused, that is paid once per stage instead of once per request (8.5 -> 6.6 us per
        `[256,4,64,64]` fp16 request-stage), and thresholded stages of small batches stop needing workgroup clusters
        (11.4 -> 1.4 us per request-stage for 32 requests of `[32,3,64,64]`).  Calls with correctors written in Python,
        `return_intermediate` or the adaptive method run the requests one after the other."""
        xs = list(xs)
        kw = sample_kwargs
        ok = (len(xs) > 1 and kw.get("method", "multistep") in ("multistep", "singlestep", "singlestep_fixed")
              and not kw.get("return_intermediate", False) and self.correcting_xt_fn is None and self._user_x0 is None
              and all(torch.is_tensor(x) and x.shape == xs[0].shape and x.dtype == xs[0].dtype and x.device == xs[0].device
                      for x in xs) and xs[0].dim() > 0 and xs[0].numel() > 0)
        if not ok:
            return [self.sample(x, **kw) for x in xs]
        self._group = xs
        try:
            return self.sample(xs[0], **kw)      # _run_plan picks the group up and returns the list of results
        finally:
            self._group = None

    def _run_plan_group(self, plan, xs, sd, cfg):
        """`_run_plan_fast` over several requests: a set of launch records per request (_FastRun), per stage the network
        calls of all requests and one dpm_stage_launch_multi over a contiguous array of their dpm_buffers."""
        device = xs[0].device
        stream, idx, capturing, other = _launch_ctx(device)
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
        mf = _mf_of(first[0][0]) if first[0][0].shape == shape else None
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
        x0s = [_conv(x, sd, mf) for x in xs]
        outs = [_empty(shape, sd, device, mf) for _ in range(R)]
        last, roles = runs[0].last, plan.roles
        launch = _stage_launch_multi_raw
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
        return [_in_layout_of(o, x) for o, x in zip(outs, xs)]

    def _auto_captured(self, x, kw, return_intermediate):
        """auto_capture: the replayed result of this call, or None when the call is not (yet) served by a graph"""
        if (return_intermediate or self.correcting_xt_fn is not None or self._user_x0 is not None or not x.is_cuda or x.dim() == 0
                or x.numel() == 0 or (kw["method"] == "adaptive" and not self._adaptive_runs_on_device(x))):
            return None                           # Python callbacks / a host-side adaptive loop: never captured
        dev = x.device
        # everything a replay bakes in: the call's arguments, the tensor's geometry, the stream -- and every solver / wrapper
        # setting the plan and the kernels depend on (the components of _get_plan's key + the state dtype): changing one of
        # them between calls must miss the cache, not replay the old settings
        w = self._wrapped
        key = (tuple(sorted((k, (float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else v)) for k, v in kw.items())),
               tuple(x.shape), x.dtype, x.stride(), dev.index, torch.cuda.current_stream(dev).cuda_stream,
               bool(self.cluster_in_graph), int(self.thr_spin_limit), self._model_codes(), self._thresholding,
               float(self.dynamic_thresholding_ratio), float(self.thresholding_max_val), self.algorithm_type, self._state_dtype,
               self._sdtype(x), bool(self.adaptive_on_device), int(self.adaptive_lookahead), self.adaptive_max_iterations,
               None if w is None else (id(w.condition), id(w.unconditional_condition), id(w.model), id(w.classifier_fn),
                                       float(w.classifier_scale) if hasattr(w, "classifier_scale") else None))
        ent = self._auto.get(key)
        if ent is None:
            if len(self._auto) >= 4:
                self._auto.pop(next(iter(self._auto)))
            ent = self._auto[key] = [0, None]
        if ent[1] is False:
            return None                           # a capture of this call failed once: it stays eager
        if ent[1] is None:
            ent[0] += 1
            if ent[0] <= int(self.auto_capture):
                return None                       # eager until the call has been seen auto_capture times
            saved, self.auto_capture = self.auto_capture, 0      # the capture's own warm-up runs go through sample()
            try:
                ent[1] = self.capture(x, **kw)
            except Exception:
                # a network that is not capturable (host synchronisation, data-dependent control flow): auto_capture is an
                # optimisation the caller opted into, not a contract -- the call is served eagerly, now and from now on
                ent[1] = False
                import warnings
                warnings.warn("dpm_solver_amd: auto_capture could not record this sample() call into a graph; it stays eager")
                return None
            finally:
                self.auto_capture = saved
        return ent[1](x).clone()                  # a graph's output buffer is overwritten by the next replay: hand out a copy

    def capture(self, x, warmup=2, **sample_kwargs):
        """hipGraph-capture `sample(x, **sample_kwargs)` for a fixed shape (extension).

        The step loop is launch-bound between network calls -- a stage kernel runs for microseconds -- so the whole
        trajectory (network calls included) is recorded once into a HIP graph and replayed with one launch:

            g = dpm_solver.capture(x_T, steps=20, order=2)     # warm-up runs + capture (on a side stream)
            out = g(x_T_new)                                    # copy into the static input, replay

        Requirements are those of torch.cuda.graph: the wrapped network must be capturable (no host
        synchronisation, no data-dependent control flow), shapes are frozen, and the returned tensors are static
        buffers that the next replay overwrites.  method='adaptive' is captured with its device-side controller: exactly
        `adaptive_max_iterations` (default 64) iterations are recorded, those after t_end is reached do nothing."""
        if sample_kwargs.get("method", "multistep") == "adaptive" and not self._adaptive_runs_on_device(x):
            raise NotImplementedError("the host-side adaptive control loop synchronises every iteration; it cannot be "
                                      "captured into a graph (the device-side controller can: adaptive_on_device = True, no "
                                      "dynamic thresholding / callable correcting_x0_fn, an fp32 or explicit half state)")
        return GraphedSample(self, x, warmup, sample_kwargs)

    def _run_plan_fast(self, plan, x, sd, cfg):
        """`_run_plan` without correctors / intermediates: prebuilt launch records (see _FastRun), the result in a
        fresh tensor.  Per stage: the opaque network call, three pointer patches, one dpm_stage_launch."""
        device = x.device
        stream, idx, capturing, other = _launch_ctx(device)
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
        # the network's layout is the run's (see _mf_of): an NHWC network gets NHWC states and its outputs are bound as
        # they are; x_T is brought there once and the result goes back to x_T's layout, like ATen would return it
        mf = _mf_of(first[0]) if first[0].shape == x.shape else None
        key = (id(plan), tuple(x.shape), sd, idx, stream, cfg, mf, bool(self.cluster_in_graph), int(self.thr_spin_limit))
        fr = None if capturing else self._fast.get(key)      # a captured graph bakes its buffers in: give it its own
        if fr is None:
            fr = _FastRun(self, plan, x.shape, sd, device, cfg, mf)
            if not capturing:
                if len(self._fast) >= 8:
                    self._fast.pop(next(iter(self._fast)))
                self._fast[key] = fr
        x0 = _conv(x, sd, mf)
        p0 = x0.data_ptr()
        out = _empty(x.shape, sd, device, mf)
        bufs, refs, roles = fr.bufs, fr.refs, plan.roles
        bufs[fr.last].x_out = out.data_ptr()
        xbuf, xfull = fr.xbuf, fr.xfull
        launch = _stage_launch_raw
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
        return out if (mf is None and x.is_contiguous()) else _in_layout_of(out, x)

    def _run_plan(self, plan, x, method, cxt, keep, intermediates):
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
                    intermediates.append(_in_layout_of(x_out, x))
                state, state2 = x_out, ext.get("x2")
                tmp, tmp2 = None, None
            else:
                tmp, tmp2 = x_out, ext.get("x2")
        return _in_layout_of(state, x) if x.dim() > 0 else state


class GraphedSample:
    """A captured `DPM_Solver.sample()` call (see DPM_Solver.capture)."""

    def __init__(self, solver, x, warmup, sample_kwargs):
        _require_gpu(x)
        self.solver = solver
        self.kwargs = dict(sample_kwargs)
        self.static_x = x.clone()
        dev = x.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                     # plans, time tensors, allocator pools: all warm before capture
            # at least TWO runs: a network that writes into its time argument is detected at the second call (version
            # counters, _Plan.time_views), and the rebuild of the shared time vectors it triggers -- a pageable host-to-device
            # copy and an allocation -- must not land inside the capture
            for _ in range(max(int(warmup), 2)):
                solver.sample(self.static_x, **self.kwargs)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # No garbage collection while the stream is capturing: a finaliser that frees device resources inside the capture
        # region -- a communicator of a destroyed process group, another graph, anything a cycle kept alive -- makes a call
        # that is illegal there, and the error surfaces inside a C++ destructor (the process aborts).  torch.cuda.graph
        # collects once before it begins; what becomes garbage during the capture waits until it is over.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                self.static_out = solver.sample(self.static_x, **self.kwargs)
        finally:
            if gc_was_on:
                gc.enable()

    def replay(self):
        self.graph.replay()
        return self.static_out

    def __call__(self, x):
        self.static_x.copy_(x)
        return self.replay()
