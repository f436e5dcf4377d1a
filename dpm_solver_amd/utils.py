"""Module-level helpers the reference exports next to the classes (dpm_solver_pytorch.py:1253-1305)."""
import torch


def interpolate_fn(x, xp, yp):
    """Piecewise-linear y = f(x) through the keypoints (xp, yp); beyond the ends the outermost segments
    are extended.  x: [N, C], xp / yp: [C, K] -> [N, C]   (semantics of ref :1253-1292).

    The reference finds the bracketing segment by sorting [x, xp] for every query; a binary search
    (torch.searchsorted) selects the same segment.  Not on the sampling path: the engine's schedule
    lookups happen once per plan in the C planner."""
    N, K = x.shape[0], xp.shape[1]
    xq = x.t().contiguous()                                   # [C, N]
    idx = torch.searchsorted(xp.contiguous(), xq, right=False)  # #{xp < x}
    i0 = torch.where(idx == 0, torch.zeros_like(idx), torch.where(idx == K, torch.full_like(idx, K - 2), idx - 1))
    i1 = i0 + 1
    x0, x1 = torch.gather(xp, 1, i0), torch.gather(xp, 1, i1)
    y0, y1 = torch.gather(yp, 1, i0), torch.gather(yp, 1, i1)
    return (y0 + (xq - x0) * (y1 - y0) / (x1 - x0)).t()


def expand_dims(v, dims):
    """[N] -> [N, 1, ..., 1] with `dims` dimensions in total (ref :1295-1305)."""
    return v[(...,) + (None,) * (dims - 1)]
