"""N>1 path on CPU: world_size-2 gloo processes shard a batch, sample their shards (host logic + numpy kernel
double), all-gather, and must reproduce the unsharded result bit for bit (batch-shard invariance)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, batch, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cases as C
        import dpm_solver_amd.solver as S
        from dpm_solver_amd import distributed as DD
        from engine_cases import build_solver, sample_kwargs
        from kernel_double import launch_raw_double, launch_stage_double
        S._launch_stage = launch_stage_double
        S._stage_launch_raw = launch_raw_double
        S._launch_ctx = lambda dev: (None, 0, False, False)
        S._require_gpu = lambda x: None
        case = dict(C.E2E_BY_NAME["cfg_ms2"], shape=(batch, 4, 8, 8))
        x = torch.from_numpy(C.x_T_for(case))
        lo, hi = DD.shard_bounds(batch, rank, world)
        # per-sample conditioning is sharded the same way as the state
        scase = dict(case, shape=(hi - lo, 4, 8, 8))
        dpm = build_solver(scase, "cpu")
        kw = sample_kwargs(case, return_intermediate=False)
        out = DD.sample_sharded(dpm, x, **kw)
        full = build_solver(case, "cpu").sample(x, **kw)
        ok = bool(torch.equal(out, full)) and out.shape[0] == batch
        # adaptive step sizes couple the batch through max_b E_b (ref :1001): one MAX all-reduce per iteration keeps
        # every rank on the unsharded run's accept / reject sequence
        from kernel_double import adaptive_error_double
        import dpm_solver_amd as D
        from engine_cases import make_schedule
        S._adaptive_error = adaptive_error_double
        ns = make_schedule("vp_linear")
        xa = x * torch.linspace(0.5, 2.0, batch).reshape(-1, 1, 1, 1)          # samples with different error norms
        mk = lambda: D.DPM_Solver(D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns), ns, algorithm_type="dpmsolver")
        oa = DD.sample_sharded(mk(), xa, method="adaptive", order=2, t_end=1e-3)
        fa = mk().sample(xa, method="adaptive", order=2, t_end=1e-3)
        ok = ok and bool(torch.equal(oa, fa))
        assert DD.rank_seed(7) == 7 + rank
        # gather_samples on its own: ragged shards take the padding path, an EMPTY shard (batch < world) included
        for nb in (1, 3):
            lo2, hi2 = DD.shard_bounds(nb, rank, world)
            loc = torch.arange(nb * 6, dtype=torch.float32).reshape(nb, 2, 3)[lo2:hi2]
            got = DD.gather_samples(loc, batch=nb)
            ok = ok and bool(torch.equal(got, torch.arange(nb * 6, dtype=torch.float32).reshape(nb, 2, 3)))
        # a channels_last batch: the shards and the gathered result keep the layout, values as in the default layout
        xcl = x.contiguous(memory_format=torch.channels_last)
        ocl = DD.sample_sharded(build_solver(scase, "cpu"), xcl, **kw)
        ok = ok and bool(torch.equal(ocl, full)) and ocl.is_contiguous(memory_format=torch.channels_last) and ocl.shape == full.shape
        # a batch smaller than the world: the rank with the empty shard still joins every collective (adaptive: the
        # MAX all-reduce of every iteration; then the gather)
        x1 = xa[:1]
        o1 = DD.sample_sharded(mk(), x1, method="adaptive", order=2, t_end=1e-3)
        f1 = mk().sample(x1, method="adaptive", order=2, t_end=1e-3)
        ok = ok and bool(torch.equal(o1, f1)) and o1.shape[0] == 1
        o2 = DD.sample_sharded(build_solver(dict(case, shape=(hi - lo if False else (1 if rank == 0 else 0), 4, 8, 8)), "cpu"),
                               x[:1], **kw)
        ok = ok and bool(torch.equal(o2, full[:1]))
        q.put((rank, ok, tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4, 5])
def test_sharded_sampling_matches_unsharded(batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + batch
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res


def test_shard_bounds_cover_batch():
    from dpm_solver_amd.distributed import shard_bounds
    for b in [0, 1, 7, 8, 9, 256]:
        for w in [1, 2, 3, 8]:
            spans = [shard_bounds(b, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == b
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
