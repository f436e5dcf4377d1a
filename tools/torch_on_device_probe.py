#!/usr/bin/env python3
"""What does torch ITSELF do on this GPU in the two places where the engine's bits were decided by a rounding detail in
round 6?  (i) half tensor * Python float (the classifier-free blend's `guidance_scale * (noise - noise_uncond)`, ref :330):
one rounding of the exact product (v_fma_mixlo_f16) or fp32 product, then conversion (the CPU kernels; torch's documented
opmath)?  (ii) torch.quantile's interpolation: one fused multiply-add (the CPU kernel) or two roundings?
Prints one JSON line; recorded under profiles/."""
import json

import numpy as np
import torch

g = np.random.default_rng(0)
d = torch.from_numpy(g.standard_normal(1 << 20).astype(np.float32)).half()
out = {}
for s in (7.3, 7.5, 2.5):
    cpu = d * s
    dev = (d.cuda() * s).cpu()
    # the exact product (35 bits fit a double) rounded once (numpy converts double -> half directly; torch goes through float)
    once = torch.from_numpy((d.double().numpy() * float(np.float32(s))).astype(np.float16))
    twice = (d.float() * np.float32(s)).half()                   # fp32 product, then conversion
    out["half*%g" % s] = dict(device_ne_cpu=int((dev != cpu).sum()), cpu_is_two_roundings=bool(torch.equal(cpu, twice)),
                             device_is_two_roundings=bool(torch.equal(dev, twice)), device_is_one_rounding=bool(torch.equal(dev, once)),
                             one_vs_two_differ=int((once != twice).sum()), n=d.numel())
x = torch.from_numpy(np.abs(g.standard_normal((20000, 65))).astype(np.float32))
ps = [0.3, 0.5, 0.77, 0.9, 0.97, 0.995]
ne = 0
for p in ps:
    ne += int((torch.quantile(x, p, dim=1) != torch.quantile(x.cuda(), p, dim=1).cpu()).sum())
out["quantile"] = dict(rows=x.shape[0] * len(ps), device_ne_cpu=ne)
out["device"] = torch.cuda.get_device_name(0)
out["torch"] = torch.__version__
print(json.dumps(out))
