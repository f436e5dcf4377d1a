"""Device entry points and tensor-layout helpers of the host side (split out of solver.py in round 6).

Everything that touches the GPU goes through a name of THIS module -- `_launch_stage` (one dpm_stage_launch with freshly
allocated outputs), `_stage_launch_raw` / `_stage_launch_multi_raw` (the C entry points behind the prebuilt launch records),
`_add_noise`, `_adaptive_error`, `_launch_ctx`, `_require_gpu` -- and the other modules call them as `DV.<name>`, looked up at
call time.  That makes this module the ONE place where the CPU test suite puts its numpy double of the kernels
(tests/kernel_double.py) and the timing tools their event-bracketed launches; `dpm_solver_amd.solver` forwards reads and
writes of these names here, so `solver._stage_launch_raw = shim` keeps working.

`ref :NNN` = line in the reference's dpm_solver_pytorch.py.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

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
    if g is not None and g.dtype is not ed and sd is torch.float32:
        ed = torch.float32  # an fp32 classifier gradient next to a half network output: widen the output (see _bind_outputs)
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
    if x_higher.dtype != x_lower.dtype:
        # a half-precision state on a continuous schedule: the lower-order estimate is still half, the higher-order one already
        # fp32 (its inner node promotes it, ref :161) -- the kernel reads all three tensors in ONE dtype: widen (exact)
        x_lower, x_higher = x_lower.float(), x_higher.float()
    xp = x_prev if x_prev.dtype == x_lower.dtype else x_prev.to(x_lower.dtype)
    # dense copies in the default order where an operand is not (a channels_last network: the states are in its layout) -- held
    # by NAME until the launch is enqueued: a temporary dropped right after its data_ptr() was taken hands its block back to
    # the caching allocator, and the next copy may land in it (two operands aliasing: a zero estimate, every step accepted)
    xl_c, xh_c, xp_c = x_lower.contiguous(), x_higher.contiguous(), xp.contiguous()
    with torch.cuda.device(x_lower.device):
        L.check(L.lib.dpm_adaptive_error_launch(
            _ptr(xl_c), _ptr(xh_c), _ptr(xp_c), float(atol), float(rtol),
            _ptr(e_dev), B, per_sample, _DT[x_lower.dtype],
            C.c_void_p(torch.cuda.current_stream(x_lower.device).cuda_stream)))
    del xl_c, xh_c, xp_c
    return e_dev[B]
