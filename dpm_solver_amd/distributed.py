"""Batch-sharded sampling across the GPUs of one node: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) / xGMI.

The solver path shards by independent samples: no operation of sample() couples two samples, so
each rank owns a contiguous slice of the batch, its own plan copy and network replica, and nothing crosses
ranks until the single all-gather of the finished samples.  (The reference itself never moves a tensor between
ranks: examples/ddpm_and_guided-diffusion/main.py:249-265 spawns one process per GPU with seed + rank and each
writes its own PNGs.)  The 'adaptive' method is the one exception -- its error norm takes a max over the batch
(dpm_solver_pytorch.py:1001): there one 4-byte MAX all-reduce per iteration keeps the sharded run identical to
the unsharded one.
"""
import torch
import torch.distributed as dist


def rank_seed(seed, rank=None):
    """Per-rank RNG convention of the reference harness: args.seed + rank (main.py:262-265)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return int(seed) + int(rank)


def shard_bounds(batch, rank, world):
    """[lo, hi) of rank's contiguous slice; the first `batch % world` ranks get one extra sample."""
    base, extra = divmod(int(batch), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x, rank=None, world=None, group=None):
    """This rank's slice of a batch-leading tensor (also use it for per-sample conditioning)."""
    if world is None:
        world = dist.get_world_size(group)
    if rank is None:
        rank = dist.get_rank(group)
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


def gather_samples(x_local, batch=None, group=None):
    """The one collective of the sharded path: all-gather the finished shards into the full batch on every rank.
    Equal shards use a single all_gather_into_tensor (one RCCL call; on the fully connected xGMI mesh each
    rank's shard goes to its 7 peers over 7 distinct links).  Ragged shards are padded to the largest one."""
    world = dist.get_world_size(group)
    if world == 1:
        return x_local
    # a channels_last shard (the solver returns x_T's layout) is gathered as it lies in memory: the batch is the outermost
    # dimension of both layouts, so the collective sees each sample's block unchanged -- viewed [N,H,W,C] it is a plain
    # contiguous tensor -- and the full batch comes back in channels_last as well (no layout copy before the all-gather).
    # Every rank must read the gathered buffer the same way: an EMPTY shard has no layout of its own, so ragged batches
    # exchange the flag together with the shard sizes.
    nd = x_local.dim()
    cl = False
    if nd in (4, 5) and x_local.numel() > 0 and not x_local.is_contiguous():
        cl = x_local.is_contiguous(memory_format=torch.channels_last if nd == 4 else torch.channels_last_3d)
    n_local = torch.tensor([x_local.shape[0], int(cl)], device=x_local.device, dtype=torch.int64)
    if batch is not None and batch % world == 0:
        sizes = [batch // world] * world
    else:
        all_n = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(all_n, n_local, group=group)
        sizes = [int(v[0].item()) for v in all_n]
        cl = any(int(v[1].item()) for v in all_n)
    mx = max(sizes)
    perm = (0,) + tuple(range(2, nd)) + (1,) if cl else None
    xl = (x_local.permute(perm) if perm is not None else x_local).contiguous()   # (no copy for a dense channels_last shard)
    if xl.shape[0] != mx:
        pad = torch.zeros((mx - xl.shape[0],) + tuple(xl.shape[1:]), dtype=xl.dtype, device=xl.device)
        xl = torch.cat([xl, pad])
    out = torch.empty((world * mx,) + tuple(xl.shape[1:]), dtype=xl.dtype, device=xl.device)
    dist.all_gather_into_tensor(out, xl, group=group)
    if not all(s == mx for s in sizes):
        out = torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)])
    if perm is not None:
        inv = (0, len(perm) - 1) + tuple(range(1, len(perm) - 1))
        out = out.permute(inv)
    return out


def sample_sharded(solver, x_T, group=None, gather=True, **sample_kwargs):
    """DPM_Solver.sample() on this rank's shard of x_T (a full-batch tensor present on every rank, or already a
    shard when `gather=False`), followed by the all-gather of the results."""
    adaptive = sample_kwargs.get("method", "multistep") == "adaptive"
    if sample_kwargs.get("return_intermediate"):
        raise NotImplementedError("return_intermediate is per-rank state; gather the shards yourself")
    full = x_T.shape[0]
    xs = shard_batch(x_T, group=group) if gather else x_T
    if adaptive:
        # the adaptive controller reads E = max over the WHOLE batch of the per-sample error norm (ref :1001): one
        # 4-byte MAX all-reduce per iteration makes every rank take the same accept / reject and step-size decisions
        # as the unsharded run (same number of iterations on every rank, so the collectives pair up)
        def reduce_max(e):
            if e.is_cuda and dist.get_backend(group) == "gloo":      # test rigs: gloo ranks sharing one GPU
                c = e.detach().cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
                return c.to(e.device)
            e = e.clone()
            dist.all_reduce(e, op=dist.ReduceOp.MAX, group=group)
            return e
        prev, solver.error_reduce = solver.error_reduce, reduce_max
        try:
            out = solver.sample(xs, **sample_kwargs)
        finally:
            solver.error_reduce = prev
    else:
        out = solver.sample(xs, **sample_kwargs)
    return gather_samples(out, batch=full, group=group) if gather else out
