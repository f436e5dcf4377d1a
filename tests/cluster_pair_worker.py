"""Worker of test_two_processes_run_clustered_thresholding_on_one_gpu (tests/test_gpu_parity.py): one of `world` processes
that run clustered dynamic-thresholding trajectories on cuda:0 concurrently.  Prints one JSON line.

    python cluster_pair_worker.py <rendezvous dir> <rank> <world>"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import dpm_solver_amd as D  # noqa: E402
from dpm_solver_amd import _lib as L  # noqa: E402
from engine_cases import make_schedule  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()


def main():
    rdv, rank, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dev = "cuda:0"
    ns = make_schedule("ddpm")
    shapes = [(32, 3, 64, 64), (3, 3, 160, 160)]                  # k = 6 clusters; large samples (k > 1 always)
    xs = [torch.from_numpy(np.random.default_rng(50 + i).standard_normal(s).astype(np.float32)).to(dev)
          for i, s in enumerate(shapes)]
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns, correcting_x0_fn="dynamic_thresholding")
    for s in shapes:
        assert L.lib.dpm_threshold_workspace_bytes(s[0], int(np.prod(s[1:]))) > 0
    want = [digest(dpm.sample(x, steps=10, order=2)) for x in xs]          # undisturbed (the peers are still starting up)
    want2 = [digest(dpm.sample(x, steps=10, order=2)) for x in xs]
    assert want == want2
    torch.cuda.synchronize()
    L.cluster_timeout_poll()
    # rendezvous: everybody is warm
    open(os.path.join(rdv, "ready_%d" % rank), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(rdv, "ready_%d" % r)) for r in range(world)):
        if time.time() - t0 > 300:
            print(json.dumps(dict(ok=False, why="rendezvous timed out")))
            return 1
        time.sleep(0.01)
    bad, n = [], 0
    t_start = time.time()
    while time.time() - t_start < 4.0:
        for i, x in enumerate(xs):
            out = dpm.sample(x, steps=10, order=2)
            if n % 8 == 0:                                             # a digest costs a synchronisation
                if digest(out) != want[i]:
                    bad.append((n, i))
            n += 1
    torch.cuda.synchronize()
    t_end = time.time()
    open(os.path.join(rdv, "done_%d_%f_%f" % (rank, t_start, t_end)), "w").close()
    time.sleep(0.3)
    spans = []
    for f in os.listdir(rdv):
        if f.startswith("done_"):
            _, r, a, b = f.split("_")
            spans.append((int(r), float(a), float(b)))
    others = [s for s in spans if s[0] != rank]
    overlapped = any(min(t_end, b) - max(t_start, a) > 1.0 for _, a, b in others) if others else False
    final = [digest(dpm.sample(x, steps=10, order=2)) for x in xs]
    print(json.dumps(dict(ok=not bad and final == want, bad=bad[:5], trajectories=n, overlapped=overlapped or world == 1,
                          timeouts_recovered=bool(L.cluster_timeout_poll()), rank=rank)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
