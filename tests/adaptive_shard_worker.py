"""Worker of test_gpu_extensions.py::test_adaptive_sharded_with_an_empty_shard_on_the_gpu: `world` gloo ranks share
cuda:0, the batch is smaller than the world, so the last rank owns an EMPTY shard.  Every rank must take the device-side
controller (dpm_adaptive_*) and issue the same number of MAX all-reduces (the empty rank used to take the
host loop and the counts differed -> hang)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, batch, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cases as C
        import dpm_solver_amd as D
        import dpm_solver_amd.solver as S
        from dpm_solver_amd import distributed as DD
        from engine_cases import make_schedule
        dev = "cuda:0"
        ns = make_schedule("vp_linear")
        mk = lambda: D.DPM_Solver(D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns), ns, algorithm_type="dpmsolver")
        rng = np.random.default_rng(8)
        x = torch.from_numpy(rng.standard_normal((batch, 3, 16, 16)).astype(np.float32)).to(dev)
        calls = {"device": 0, "reduce": 0}
        real = S.DPM_Solver._adaptive_device

        def counted(self, *a, **k):
            calls["device"] += 1
            return real(self, *a, **k)
        S.DPM_Solver._adaptive_device = counted
        real_ar = dist.all_reduce

        def counted_ar(*a, **k):
            calls["reduce"] += 1
            return real_ar(*a, **k)
        dist.all_reduce = counted_ar
        out = DD.sample_sharded(mk(), x, gather=False, method="adaptive", order=2, t_end=1e-3)   # gather=False: x is sharded below
        dist.all_reduce = real_ar
        S.DPM_Solver._adaptive_device = real
        lo, hi = DD.shard_bounds(batch, rank, world)
        q.put((rank, calls["device"], calls["reduce"], tuple(out.shape), None))
        # second pass: the shard really is this rank's slice
        sol = mk()
        shard = x[lo:hi]
        out = DD.sample_sharded(sol, shard, gather=False, method="adaptive", order=2, t_end=1e-3)
        torch.cuda.synchronize()
        full = mk().sample(x, method="adaptive", order=2, t_end=1e-3)
        ok = bool(torch.equal(out, full[lo:hi])) and out.shape[0] == hi - lo
        q.put((rank, -1, -1, tuple(out.shape), ok))
    finally:
        dist.destroy_process_group()
