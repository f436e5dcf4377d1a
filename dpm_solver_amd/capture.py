"""hipGraph capture of `sample()`, split out of solver.py in round 6: `GraphedSample` (DPM_Solver.capture) and
the opt-in `auto_capture` bookkeeping (`auto_captured`, bound as DPM_Solver._auto_captured)."""
import torch

from . import _device as DV

class GraphedSample:
    """A captured `DPM_Solver.sample()` call (see DPM_Solver.capture)."""

    def __init__(self, solver, x, warmup, sample_kwargs):
        DV._require_gpu(x)
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


def auto_captured(self, x, kw, return_intermediate):
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
