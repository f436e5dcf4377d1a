#!/usr/bin/env python3
"""Drop-in use of the MI355X engine with the reference's own API (compare README.md:380-420 of LuChengTHU/dpm-solver).

    python examples/quickstart.py            # needs an MI355X (no CPU fallback)

A small convolutional network stands in for the UNet; everything else is the reference's calling convention:
NoiseScheduleVP -> model_wrapper (classifier-free guidance) -> DPM_Solver.sample().  The last part shows the two
opt-in accelerations: a fused DiffEdit-style mask blend and hipGraph capture of the whole trajectory.
"""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpm_solver_pytorch import NoiseScheduleVP, model_wrapper, DPM_Solver   # noqa: E402  (resolves to dpm_solver_amd)
from dpm_solver_amd import MaskBlend                                        # noqa: E402


class TinyEps(nn.Module):
    """eps_theta(x, t, cond): two convolutions, a time and a class embedding"""

    def __init__(self, ch=4, width=32, n_cls=10):
        super().__init__()
        self.inp, self.out = nn.Conv2d(ch, width, 3, padding=1), nn.Conv2d(width, ch, 3, padding=1)
        self.temb, self.cemb = nn.Linear(1, width), nn.Embedding(n_cls + 1, width)

    def forward(self, x, t, cond):
        h = self.inp(x) + (self.temb(t[:, None] / 1000.) + self.cemb(cond))[:, :, None, None]
        return self.out(torch.nn.functional.silu(h))


def main():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = TinyEps().to(dev).eval()
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2      # SD-v1 schedule
    ns = NoiseScheduleVP("discrete", betas=betas)
    B = 16
    cond = torch.randint(0, 10, (B,), device=dev)
    uncond = torch.full((B,), 10, device=dev)
    model_fn = model_wrapper(net, ns, model_type="noise", guidance_type="classifier-free", condition=cond,
                             unconditional_condition=uncond, guidance_scale=7.5)
    solver = DPM_Solver(model_fn, ns, algorithm_type="dpmsolver++")
    x_T = torch.randn(B, 4, 64, 64, device=dev)
    with torch.no_grad():
        x0 = solver.sample(x_T, steps=20, order=2, skip_type="time_uniform", method="multistep")
    print("sample:", tuple(x0.shape), x0.dtype, "finite:", bool(torch.isfinite(x0).all()))

    # hipGraph capture: the 20 network calls and 20 stage kernels become one graph launch
    graphed = solver.capture(x_T, steps=20, order=2)
    assert torch.equal(graphed(x_T), x0)
    for name, fn in (("eager", lambda: solver.sample(x_T, steps=20, order=2)), ("captured", lambda: graphed(x_T))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print("%-9s %.0f us per 20-step trajectory" % (name, (time.perf_counter() - t0) / 20 * 1e6))

    # several requests in flight (a server): per stage the network once per request, then one fused solver launch
    xs = [torch.randn(B, 4, 64, 64, device=dev) for _ in range(8)]
    with torch.no_grad():
        ys = solver.sample_requests(xs, steps=20, order=2)
        assert all(torch.equal(y, solver.sample(x, steps=20, order=2)) for x, y in zip(xs, ys))
    print("sample_requests: %d requests, identical to sample() of each" % len(ys))

    # a network in channels_last (NHWC, MIOpen's preferred layout): the solver runs on the network's own storage -- its
    # scratch states are allocated in that layout, nothing is copied per stage -- and returns x_T's layout
    net_cl = TinyEps().to(dev).eval()
    net_cl.load_state_dict(net.state_dict())
    net_cl = net_cl.to(memory_format=torch.channels_last)
    solver_cl = DPM_Solver(model_wrapper(net_cl, ns, model_type="noise", guidance_type="classifier-free", condition=cond,
                                         unconditional_condition=uncond, guidance_scale=7.5), ns, algorithm_type="dpmsolver++")
    with torch.no_grad():
        y_cl = solver_cl.sample(x_T.to(memory_format=torch.channels_last), steps=20, order=2)
    rel = ((y_cl - x0).abs().max() / x0.abs().max()).item()
    print("channels_last: result in channels_last = %s, differs from the NCHW run by %.2g of its scale (other MIOpen kernels)"
          % (y_cl.is_contiguous(memory_format=torch.channels_last), rel))

    # DiffEdit / inpainting: keep the masked-out region on the known image, noised to the current level
    mask = (torch.rand(64, 64, device=dev) > 0.5).float()
    known = torch.randn(B, 4, 64, 64, device=dev)
    edit = DPM_Solver(model_fn, ns, correcting_xt_fn=MaskBlend(ns, mask, x0=known, noise=torch.randn_like(known)))
    with torch.no_grad():
        y = edit.sample(x_T, steps=20, order=2)
    err = ((y - known) * (1 - mask)).abs().max().item()
    print("inpaint: max |y - known| outside the mask at t_end = %.3g (the known image, noised to t = 1e-3)" % err)


if __name__ == "__main__":
    main()
