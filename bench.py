#!/usr/bin/env python3
"""Solver-update benchmark: DPM-Solver++(2M), 20 steps, [256,4,64,64] fp16, frozen model_fn (BASELINE.json
configs[1]) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

What is timed is the situation every real sampling loop is in: the inputs of a solver stage come from **HBM**, not
from the 256 MiB Infinity Cache, because a network ran since they were written (ref :1195-1213).  With a frozen network
that situation is produced by keeping R = 32 independent sampling requests of [256,4,64,64] in flight and advancing them
stage by stage (`dpm_plan_run_multi`): between two stages of one request the other 31 stream 1.3 GB through the chip.
The 32 requests of one stage are ONE fused launch (`dpm_stage_launch_multi` -> stage_kernel_multi, a pointer-table
kernel over 32 x 4.2 M elements), so a launch's ramp-up and drain are paid once per 1.3 GB.

A *step* is P = 30 (`--trajectories-per-step`) complete 20-stage trajectories of those R requests: 600 fused launches,
P x R x 256 = 245 760 samples, about 0.125 s -- so that the driver's `--steps 20` is a SUSTAINED 2.5 s measurement (clocks,
HBM thermals), not an 80 ms burst.  Every input is resident in HBM before the clock starts; the network outputs are
pre-staged in buffers distinct from x.  A timed region shorter than 2 s is not reported: the step count is
raised until the region is long enough, and `steps` in the JSON line is the number of steps actually timed
(`steps_requested` is what --steps asked for).  `sustained` carries the region's length and the throughput of its second
half over its first half (a throttling check).

Back-to-back fused launches are also the CONSERVATIVE regime: each launch finds the write-back Infinity Cache full of its
predecessor's dirty stores and pays for them; behind a real network (reads) the cache is clean, absorbs up to 256 MB of the
launch's stores and the same launch is 12-14 % shorter (tools/history/duty_cycle.py, profiles/r03_in_loop.md).  Every byte of the
timed region is paid for inside it.

One JSON line on rank 0:
  value            whole-job Msamples/s = N * K * R * 256 / wall, wall = barrier/sync-bracketed, max over ranks
  roofline         HBM roofline of the fused 2M stage kernel.  `achieved` = algorithmic bytes of the timed region
                   (98*n*s per request trajectory = 18 x 5 n s + 2 x 4 n s) / its GPU time by HIP events on the launch
                   stream, i.e. bytes per launch / average launch duration, dispatch gaps included -- the average
                   rocprofv3 --kernel-trace --stats reports for the kernel (profiles/).  Secondary entries:
                   `cache_resident` (ONE request, stages back to back: inputs in the Infinity Cache -- last round's
                   headline), `single_request_cold` (requests interleaved but one launch each), the no-arithmetic
                   ceilings of the same streams.
                   `in_network_loop`: the stage kernel inside a REAL torch network loop -- DPM_Solver.sample() on ONE
                   [256,4,64,64] request with a random-init torch network as model_fn (hundreds of MB of activations per
                   call, its last kernel writes eps microseconds before the stage kernel reads it): kernel-only durations
                   by events attached to each launch, and the wall time the 20 solver stages add to the 20 network calls.
  cpu_baseline     the reference itself ($DPM_REFERENCE_DIR or /root/reference: dpm_solver_pytorch.py, unmodified) on
                   this box's host cores when it is present, else the numpy oracle (a port); rank 0, N=1 only, bounded.
                   `reference_on_gpu_box`: the unmodified reference timed on an MI355X box's own host cores
                   (profiles/cpu_baseline_reference_gpubox.json, tools/cpu_baseline.py through gpurun).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# tests/test_bench_plumbing.py only: the N > 1 plumbing of this file (rank / world from the environment, process group,
# barrier + max-over-ranks timing, per_rank_wall_s, the final all-gather and its gather_ms, the JSON line) run by two gloo
# ranks on CPU with the timed region replaced by a sleep -- so that the multi-GPU fields cannot rot while no 8-GPU node is
# available.  Never set outside that test: a stubbed line says so ("stub": true) and carries no measurement.
STUB = os.environ.get("DPM_BENCH_STUB") == "1"

B, SHAPE, STEPS_SOLVER = 256, (4, 64, 64), 20
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
MIN_REGION_S = 2.0          # shorter timed regions are not reported (sustained clocks, not a burst)
_DT = {"fp16": torch.float16, "fp32": torch.float32, "bf16": torch.bfloat16}


def sd_alphas_cumprod():
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas).astype(np.float32)


def baseline_metric():
    """the metric string of BASELINE.json (value = its Msamples/s part; the HBM GB/s part is `roofline`)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "solver-update Msamples/sec + HBM GB/s, DPM-Solver++(2M) 20-step, Bx4x64x64"


def make_sets(n_sets, dtype, dev, seed, eps_dtype=None):
    """n_sets independent requests: x_T, frozen eps, 3 state scratch buffers, 2 history slots each."""
    from dpm_solver_amd import _lib as L
    code = {torch.float16: L.DTYPE_F16, torch.float32: L.DTYPE_F32, torch.bfloat16: L.DTYPE_BF16}
    eps_dtype = eps_dtype or dtype
    g = torch.Generator(device="cpu").manual_seed(seed)
    sets = []
    for _ in range(n_sets):
        x_T = torch.randn((B,) + SHAPE, generator=g).to(dev, dtype)
        eps = torch.randn((B,) + SHAPE, generator=g).to(dev, eps_dtype)
        xb = [x_T] + [torch.empty_like(x_T) for _ in range(3)]
        hb = [torch.empty_like(x_T) for _ in range(2)]
        rb = L.RunBuffers()
        for i in range(4):
            rb.xbuf[i] = xb[i].data_ptr()
        for i in range(2):
            rb.hist[i] = hb[i].data_ptr()
        rb.e0 = eps.data_ptr()
        rb.n, rb.batch = x_T.numel(), B
        rb.state_dtype, rb.eps_dtype = code[dtype], code[eps_dtype]
        sets.append(dict(rb=rb, x=xb, h=hb, eps=eps))
    return sets


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's own DPM_Solver.sample() (ref :1047-1245) when the checkout is present, else the port
# ---------------------------------------------------------------------------------------------------------------
def reference_dir():
    d = os.environ.get("DPM_REFERENCE_DIR", "/root/reference")
    return d if os.path.exists(os.path.join(d, "dpm_solver_pytorch.py")) else None


def _time_loop(fn, budget_s, min_runs=2):
    fn()                                                   # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s and n >= min_runs:
            return n, el


def cpu_baseline_reference(ref, ac, budget_s=8.0):
    """/root/reference/dpm_solver_pytorch.py, unmodified, on CPU tensors: 2M++ 20 steps, [256,4,64,64] fp32, frozen eps,
    at 1 thread and at all host cores."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_dpm_reference", os.path.join(ref, "dpm_solver_pytorch.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B,) + SHAPE, generator=g)
    eps = torch.randn((B,) + SHAPE, generator=g)
    ns = R.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(ac))
    sol = R.DPM_Solver(R.model_wrapper(lambda xx, t: eps, ns), ns, algorithm_type="dpmsolver++")
    run = lambda: sol.sample(x, steps=STEPS_SOLVER, order=2, skip_type="time_uniform", method="multistep")
    cores = os.cpu_count() or 1
    out = {}
    before = torch.get_num_threads()
    # 1 thread, then doubling thread counts up to all host cores: on a many-core host the reference's ~10 small ATen
    # kernels per step stop scaling long before the core count (and collapse when oversubscribed), so the best count is
    # searched, not assumed; a count that is more than twice as slow as the best so far ends the sweep
    counts = sorted({1, cores} | {c for c in (4, 8, 16, 32, 64, 128) if c < cores})
    per = max(budget_s / (len(counts) + 1), 1.0)
    try:
        for th in counts:
            torch.set_num_threads(th)
            n, el = _time_loop(run, per, min_runs=2 if th == 1 else 1)
            out[th] = dict(value=round(B * n / el / 1e6, 6), unit="Msamples/s", threads=th, trajectories=n,
                           seconds=round(el, 2), ms_per_trajectory=round(el / n * 1e3, 2))
            best_ms = min(v["ms_per_trajectory"] for v in out.values())
            if out[th]["ms_per_trajectory"] > 2.0 * best_ms and th != 1:
                break
    finally:
        torch.set_num_threads(before)
    best = max(out.values(), key=lambda v: v["value"])
    return dict(value=best["value"], unit="Msamples/s", cores=best["threads"], threads=best["threads"], host_cores=cores,
                kind="reference",
                sample="%d trajectories of [%d,4,64,64] fp32, DPM_Solver.sample(steps=20, order=2, multistep) of the "
                       "unmodified reference dpm_solver_pytorch.py on CPU tensors, frozen eps, torch %s, best of the thread "
                       "counts %s: %d threads of the host's %d cores, %.1f s"
                       % (best["trajectories"], B, torch.__version__, sorted(out), best["threads"], cores, best["seconds"]),
                single_thread=out[1], best=best, by_threads=[out[k] for k in sorted(out)])


def cpu_baseline_port(ac, budget_s=8.0):
    """numpy oracle (a port of the reference's algorithm) on the same workload: one thread, and the batch sharded over
    all host cores (numpy releases the GIL inside its loops)."""
    from oracle import dpm_oracle as O
    osch = O.Schedule.from_alphas_cumprod(ac)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B,) + SHAPE).astype(np.float32)
    eps = rng.standard_normal((B,) + SHAPE).astype(np.float32)

    def traj(lo, hi):
        e = eps[lo:hi]
        O.Solver(O.wrap_model(lambda xx, t: e, osch), osch, algorithm_type="dpmsolver++").sample(
            x[lo:hi], steps=STEPS_SOLVER, order=2)

    cores = min(os.cpu_count() or 1, 32)                 # >= 8 samples per thread: below that Python overhead dominates
    n1, el1 = _time_loop(lambda: traj(0, B), budget_s / 2, min_runs=1)

    def sharded():
        per = (B + cores - 1) // cores
        ths = [threading.Thread(target=traj, args=(i * per, min(B, (i + 1) * per))) for i in range(cores) if i * per < B]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    nc, elc = _time_loop(sharded, budget_s / 2, min_runs=1)
    v1, vc = B * n1 / el1 / 1e6, B * nc / elc / 1e6
    best_v, best_c = (vc, cores) if vc > v1 else (v1, 1)
    return dict(value=round(best_v, 6), unit="Msamples/s", cores=best_c, kind="port",
                sample="numpy oracle (oracle/dpm_oracle.py), [%d,4,64,64] fp32 2M++ 20 steps, frozen eps: %d trajectories "
                       "in %.1f s on 1 thread, %d in %.1f s with the batch sharded over %d threads"
                       % (B, n1, el1, nc, elc, cores),
                single_thread=dict(value=round(v1, 6), threads=1), all_cores=dict(value=round(vc, 6), threads=cores))


def cpu_baseline_port_c(ac, budget_s=8.0):
    """The plain-C restatement of the hot path (oracle/dpm_oracle_kernels.c, test infrastructure: the checker, here the timed
    CPU baseline) on the same workload: DPM-Solver++(2M), 20 steps, [256,4,64,64] fp32, frozen eps -- ONE fused pass per stage
    (dpmo_stage_2m: what a GPU launch does), OpenMP over the host cores, the thread count searched like the reference's."""
    from oracle import dpm_oracle as O
    from oracle import dpm_oracle_c as OC
    osch = O.Schedule.from_alphas_cumprod(ac)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B,) + SHAPE).astype(np.float32)
    eps = rng.standard_normal((B,) + SHAPE).astype(np.float32)
    st = OC.Stepper(osch)
    host = os.cpu_count() or 1
    counts = sorted({1, host} | {c for c in (4, 8, 16, 32, 64, 128) if c < host})
    per = max(budget_s / (len(counts) + 1), 0.5)
    out = {}
    for th in counts:
        run = lambda: st.sample_2m_fused(eps, x, STEPS_SOLVER, threads=th)
        n, el = _time_loop(run, per, min_runs=2)
        out[th] = dict(value=round(B * n / el / 1e6, 6), threads=th, trajectories=n, seconds=round(el, 2),
                       ms_per_trajectory=round(el / n * 1e3, 3))
        if out[th]["ms_per_trajectory"] > 2.0 * min(v["ms_per_trajectory"] for v in out.values()) and th != 1:
            break
    best = max(out.values(), key=lambda v: v["value"])
    n_el = B * int(np.prod(SHAPE))
    gbs = n_el * 4 * (18 * 5 + 2 * 4) / (best["ms_per_trajectory"] * 1e-3) / 1e9
    return dict(value=best["value"], unit="Msamples/s", cores=best["threads"], host_cores=host, kind="port",
                sample="plain-C restatement of the fused 2M++ stage (oracle/dpm_oracle_kernels.c: dpmo_stage_2m, gcc -O3 -mavx2 "
                       "-ffp-contract=off, OpenMP), [%d,4,64,64] fp32, 20 steps, frozen eps, one pass per stage: %d trajectories in "
                       "%.1f s on %d threads of the host's %d (best of the thread counts %s); bit-identical to the numpy oracle "
                       "(tests/test_oracle_c.py)" % (B, best["trajectories"], best["seconds"], best["threads"], host, sorted(out)),
                host_memory_gbs=round(gbs, 1), single_thread=out[1], best=best, by_threads=[out[k] for k in sorted(out)])


def cpu_baseline(ac):
    """`cpu_baseline` of the JSON line: ALWAYS what was timed in this run on this box's host cores.  Where the reference
    checkout is present ($DPM_REFERENCE_DIR or /root/reference) that is the unmodified reference (kind "reference"); on the
    GPU box it is not -- the reference is not part of this repository -- so the numpy oracle is timed (kind "port") and the
    reference's own figure measured earlier on an MI355X box's host cores rides along under `reference_committed`
    (profiles/cpu_baseline_reference_gpubox.json: tools/cpu_baseline.py through gpurun; not measured in this run).
    `cores` = the threads actually used, `host_cores` = the box."""
    ref = reference_dir()
    if ref:
        out = cpu_baseline_reference(ref, ac)
        out["measured_in_this_run"] = True
        out["source"] = os.path.join(ref, "dpm_solver_pytorch.py") + ", timed in this run"
        return out
    # two ports are timed: the numpy oracle (ten passes per stage, like the reference's ATen loop -- the figure the ratio
    # port_over_reference below refers to) and the plain-C restatement of the FUSED stage over all host cores -- what these CPU
    # cores can do on this path at best, and therefore the headline `value`
    npy = cpu_baseline_port(ac, budget_s=6.0)
    try:
        out = cpu_baseline_port_c(ac)
        out["numpy_port"] = npy
        out["numpy_port_value"], out["numpy_port_cores"] = npy["value"], npy["cores"]
    except Exception as e:                                   # no gcc-built library on this box: the numpy port alone
        out = npy
        out["c_port_error"] = "%s: %s" % (type(e).__name__, e)
        out["numpy_port_value"], out["numpy_port_cores"] = npy["value"], npy["cores"]
    out["measured_in_this_run"] = True
    out["host_cores"] = os.cpu_count() or 1
    p = os.path.join(ROOT, "profiles", "cpu_baseline_reference_gpubox.json")
    try:
        committed = json.load(open(p))
    except Exception:
        committed = None
    if committed and committed.get("kind") == "reference":
        committed["measured_in_this_run"] = False
        committed["source"] = ("profiles/cpu_baseline_reference_gpubox.json: the unmodified reference timed on an MI355X box's "
                               "host cores (tools/cpu_baseline.py through gpurun); committed, not re-measured in this run")
        out["reference_committed"] = committed
    # The ratio port / reference measured on ONE box in ONE run (tools/cpu_baseline.py times both; round 6: 1.27 at one thread each,
    # 0.81 at the best thread count of each) lets a reader scale this run's port figure; scalars, because the driver's record of
    # the line keeps those.  The reference's CPU rate itself varies 8 x between boxes (0.0016-0.0138 Msamples/s at 16 threads,
    # profiles/README.md): `reference_committed` above stays the figure MOST favourable to the CPU.
    try:
        pv = json.load(open(os.path.join(ROOT, "profiles", "r06_cpu_baseline_port_vs_reference.json")))
        ratio = pv["port_over_reference"]
        out["port_over_reference"] = ratio["best"]
        out["port_over_reference_single_thread"] = ratio["single_thread"]
        out["port_over_reference_source"] = ("profiles/r06_cpu_baseline_port_vs_reference.json: port and reference timed on the same "
                                             "MI355X box's host cores in one run (tools/cpu_baseline.py through gpurun); committed")
        out["port_over_reference_refers_to"] = "numpy_port_value (the numpy oracle); c_port_over_reference: the C port, same file"
        if isinstance(pv.get("c_port_over_reference"), dict):
            out["c_port_over_reference"] = pv["c_port_over_reference"]["best"]
        out["reference_estimate"] = round(out["numpy_port_value"] / ratio["best"], 6)
        out["reference_estimate_unit"] = "Msamples/s (this run's numpy-port figure / the committed same-box ratio; not a measurement)"
        if committed and committed.get("value"):
            out["reference_best_seen"] = committed["value"]          # the most favourable box of rounds 2-6 (committed)
            out["value_over_reference_best_seen"] = round(out["value"] / committed["value"], 4)
    except Exception:
        pass
    return out


# ---------------------------------------------------------------------------------------------------------------
# the stage kernel inside a real torch network loop (ref :1195-1213): secondary measurement `in_network_loop`
# ---------------------------------------------------------------------------------------------------------------
class LoopNet(torch.nn.Module):
    """Random-init stand-in for a denoising network, sized so that one call moves far more than the 256 MiB Infinity
    Cache: kind 'gemm' = per-pixel MLP 4 -> W -> W -> 4 with a sinusoidal time embedding (hipBLASLt GEMMs over
    [B*H*W, W] activations: 0.5 GB per layer at W = 256), kind 'conv' = 3x3 conv stack 4 -> W/2 -> W/2 -> 4 (MIOpen).
    The last kernel of a call writes eps [B,4,64,64] contiguously, as a UNet's conv_out does."""

    def __init__(self, kind="gemm", width=256, dtype=torch.float16, device="cuda", channels=4):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.kind, self.width = kind, width
        C4 = channels
        mk = lambda *shape, scale: torch.nn.Parameter((torch.randn(*shape, generator=g) * scale).to(device, dtype),
                                                      requires_grad=False)
        self.freqs = torch.exp(torch.linspace(0., -6., 32)).to(device)
        self.wt = mk(64, width if kind == "gemm" else width // 2, scale=0.1)
        if kind == "gemm":
            self.w1, self.w2, self.w3 = mk(C4, width, scale=0.5), mk(width, width, scale=width ** -0.5), mk(width, C4, scale=width ** -0.5)
        else:
            c = width // 2
            self.c1 = torch.nn.Conv2d(C4, c, 3, padding=1).to(device, dtype)
            self.c2 = torch.nn.Conv2d(c, c, 3, padding=1).to(device, dtype)
            self.c3 = torch.nn.Conv2d(c, C4, 3, padding=1).to(device, dtype)
        self.before_last = None        # hook called right before the call's last kernel is enqueued (prefetch experiment)

    def forward(self, x, t):
        B, C, H, W = x.shape
        a = t.float()[:, None] * self.freqs[None, :] * 1e-2
        temb = torch.cat([a.sin(), a.cos()], dim=1).to(x.dtype) @ self.wt             # [B, width]
        F = torch.nn.functional
        if self.kind == "gemm":
            h = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
            h = F.silu(h @ self.w1 + temb[:, None, :])
            h = F.silu(h @ self.w2)
            o = (h @ self.w3).reshape(B, H, W, C).permute(0, 3, 1, 2)
            if self.before_last is not None:
                self.before_last()                  # the last kernel of the call: the layout copy that writes eps
            return o.contiguous()
        h = F.silu(self.c1(x) + temb[:, :, None, None])
        h = F.silu(self.c2(h))
        if self.before_last is not None:
            self.before_last()
        return self.c3(h)


def cold_start(timeout=300):
    """`cold_start_ms` of the bench line: `import dpm_solver_amd` + the first `sample()` of a FRESH
    process -- tools/cold_start.py, one subprocess per scenario, the product library -- at `[8,4,64,64]` and with dynamic
    thresholding at `[32,3,64,64]`; the process's own torch import and HIP context are the caller's and listed apart."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "cold.json")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "cold_start.py"), "--repeat", "1", "--scenarios", "plain,thresholding",
               "--out", out]
        env = {k: v for k, v in os.environ.items() if k not in ("DPM_SOLVER_AMD_LIB", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
            if r.returncode != 0:
                return dict(error="tools/cold_start.py exited %d: %s" % (r.returncode, r.stderr[-400:]))
            rows = {q["scenario"]: q for q in json.load(open(out))["rows"]}
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
            return dict(error="%s: %s" % (type(e).__name__, e))
    p_, t_ = rows["plain"], rows["thresholding"]
    return dict(plain_8x4x64x64=p_["cold_start_ms"], thresholding_32x3x64x64=t_["cold_start_ms"],
                network_warm=dict(plain_8x4x64x64=p_.get("cold_start_network_warm_ms"), thresholding_32x3x64x64=t_.get("cold_start_network_warm_ms"),
                                  network_first_call_ms=p_.get("network_first_call_ms"),
                                  what="the same in a second fresh process whose stand-in network ran once before the first sample(): "
                                       "without torch loading the network's own kernels"),
                import_dpm_solver_amd_ms=p_["import_dpm_solver_amd_ms"], first_sample_ms=p_["first_sample_ms"],
                second_sample_ms=p_["second_sample_ms"], import_torch_and_context_ms=p_["import_torch_and_context_ms"],
                library_bytes=p_["library_bytes"], measured_in_this_run=True,
                how="fresh process per scenario (tools/cold_start.py): import dpm_solver_amd + first sample(), 2M++ 20 steps "
                    "(thresholding: 25 steps); torch import + HIP context apart")


def lab_secondary(dtype_name, eps_dtype_name, loop_net, requests, timeout=600):
    """The secondary measurements that need what the product library does not export -- event pairs attached to single
    launches inside a torch network loop (`in_network_loop`), the no-arithmetic kernels (`no_arithmetic_ceiling`,
    `lone_launch_floor`) -- run in a SUBPROCESS on the lab build of the same sources (tools/lab_secondary.py,
    DPM_SOLVER_AMD_LIB=tools/_variants/lab/libdpm_lab.so): this process, the one the headline is timed in, loads the product
    library only.  Returns the subprocess's JSON dict, or {"error": ...}."""
    lab = os.path.join(ROOT, "tools", "_variants", "lab", "libdpm_lab.so")
    if not os.path.exists(lab):
        return dict(error="no lab build (%s): run __graft_entry__.build()" % lab)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "lab_secondary.py"), "--dtype", dtype_name, "--loop-net", loop_net,
           "--requests", str(requests)]
    if eps_dtype_name:
        cmd += ["--eps-dtype", eps_dtype_name]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, DPM_SOLVER_AMD_LIB=lab), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=timeout)
        if r.returncode != 0:
            return dict(error="tools/lab_secondary.py exited %d: %s" % (r.returncode, r.stderr[-400:]))
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                           # a secondary must not cost the headline
        return dict(error="%s: %s" % (type(e).__name__, e))


# ---------------------------------------------------------------------------------------------------------------
# preflight: every rank proves its OWN device ordinal before anything is timed
# ---------------------------------------------------------------------------------------------------------------
def _torch_2m_double(ac, x, eps, steps):
    """DPM-Solver++(2M), time_uniform grid, 'discrete' schedule, frozen eps -- the textbook recurrence (ref :553-576,
    :805-852, :1171-1213) as a dozen torch-double lines on x's device.  The preflight's in-run checker: independent of the
    library, of its planner and of oracle/ (which only tests, smoke() and the cpu_baseline leg may touch)."""
    N = ac.shape[0]
    la_tab = 0.5 * torch.log(torch.from_numpy(ac.astype(np.float64)))
    t_tab = torch.linspace(0.0, 1.0, N + 1, dtype=torch.float64)[1:]

    def la(t):                                                    # piecewise-linear log(alpha_t) (ref :127-134)
        i = int(torch.searchsorted(t_tab, torch.tensor(t, dtype=torch.float64)).clamp(1, N - 1))
        w = (t - float(t_tab[i - 1])) / float(t_tab[i] - t_tab[i - 1])
        return float(la_tab[i - 1]) + w * float(la_tab[i] - la_tab[i - 1])
    alpha = lambda t: float(np.exp(la(t)))
    sigma = lambda t: float(np.sqrt(1.0 - np.exp(2.0 * la(t))))
    lam = lambda t: la(t) - np.log(sigma(t))
    ts = [float(np.float32(v)) for v in torch.linspace(1.0, 1.0 / N, steps + 1, dtype=torch.float32)]
    xd, ed = x.double(), eps.double()
    x0 = lambda xx, t: (xx - sigma(t) * ed) / alpha(t)            # eps -> x0 (ref :439)
    ms, tp = [x0(xd, ts[0])], [ts[0]]
    for k in range(1, steps + 1):
        t = ts[k]
        h = lam(t) - lam(tp[-1])
        xn = sigma(t) / sigma(tp[-1]) * xd - alpha(t) * np.expm1(-h) * ms[-1]
        if k > 1:                                                 # second-order term (ref :827-831)
            r0 = (lam(tp[-1]) - lam(tp[-2])) / h
            xn = xn - 0.5 * alpha(t) * np.expm1(-h) * ((ms[-1] - ms[-2]) / r0)
        xd = xn
        if k < steps:
            ms.append(x0(xd, t))
            tp.append(t)
    return xd


def preflight(D, L, dev, rank, world, dist, stub=False):
    """Three checks on THIS rank's device, before the timed region; returns a dict (and never raises: a failed check is
    reported per rank and fails the run after every rank has printed its line):
      smoke         one DPM-Solver++(2M) 20-step trajectory [4,4,64,64] fp32 through DPM_Solver.sample() against a torch-double
                    restatement computed on the same device (<= 1e-5 of the result's magnitude)
      thresholding  one CLUSTERED dynamic-thresholding launch (k > 1 workgroups per sample: [4,3,64,64]) -- the per-device
                    context (`device_context(dev)`) and the host-mapped fault word of an ordinal != 0 -- bit-exact against
                    torch.quantile / clamp on the same device
      collective    an all-gather of 512 KiB over the run's process group (RCCL on GPUs), contents checked"""
    t0 = time.perf_counter()
    res = dict(rank=rank, device=str(dev), ordinal=(None if stub else torch.cuda.current_device()))
    try:
        if stub:
            res["smoke"] = res["thresholding"] = "stubbed"
        else:
            ac = sd_alphas_cumprod()
            g = torch.Generator(device="cpu").manual_seed(77 + rank)
            x = torch.randn((4, 4, 64, 64), generator=g).to(dev)
            eps = torch.randn((4, 4, 64, 64), generator=g).to(dev)
            ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(ac))
            dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: eps, ns), ns, algorithm_type="dpmsolver++")
            got = dpm.sample(x, steps=20, order=2)
            want = _torch_2m_double(ac, x, eps, 20)
            err = float((got.double() - want).abs().max() / want.abs().max())
            assert got.device == x.device and err < 1e-5, "smoke rel-err %.3g on %s" % (err, dev)
            res["smoke"] = dict(ok=True, rel_err=err)
            x0 = torch.randn((4, 3, 64, 64), generator=g).to(dev) * 2.0
            assert L.lib.dpm_threshold_workspace_bytes(4, 3 * 64 * 64) > 0, "not a clustered launch"
            dthr = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx, ns), ns, correcting_x0_fn="dynamic_thresholding")
            gt = dthr.dynamic_thresholding_fn(x0)
            s_ = torch.quantile(torch.abs(x0).reshape(4, -1), 0.995, dim=1)
            s_ = torch.maximum(s_, torch.ones_like(s_)).reshape(4, 1, 1, 1)
            wt = torch.clamp(x0, -s_, s_) / s_
            torch.cuda.synchronize(dev)
            assert torch.equal(gt, wt), "clustered thresholding differs from torch.quantile on %s (max %.3g)" % (
                dev, float((gt - wt).abs().max()))
            res["thresholding"] = dict(ok=True, clustered=True, timeouts=int(L.lib.dpm_cluster_timeout_poll()))
        if dist is not None:
            nb = 512 * 1024 // 4
            mine = torch.full((nb,), float(rank + 1), dtype=torch.float32, device=dev)
            allv = torch.empty((world * nb,), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(allv, mine)
            if not stub:
                torch.cuda.synchronize(dev)
            want_g = torch.arange(1, world + 1, dtype=torch.float32, device=dev).repeat_interleave(nb)
            assert torch.equal(allv, want_g), "all-gather contents"
            res["collective"] = dict(ok=True, bytes_per_rank=nb * 4, backend=dist.get_backend(), ranks=dist.get_world_size())
        else:
            res["collective"] = "no process group (plain python launch)"
        res["ok"] = True
    except Exception as e:                                       # reported per rank; the caller fails the run
        res["ok"] = False
        res["error"] = "%s: %s" % (type(e).__name__, e)
    res["seconds"] = round(time.perf_counter() - t0, 3)
    print("[preflight] rank %d of %d on %s: %s (%.2f s)%s" % (rank, world, dev, "PASS" if res["ok"] else "FAIL", res["seconds"],
                                                              "" if res["ok"] else " -- " + res["error"]), file=sys.stderr, flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--trajectories-per-step", type=int, default=30,
                    help="a step = this many 20-stage trajectories of the requests in flight (30 x 32 x 256 samples, ~0.125 s): "
                         "the driver's --steps 20 is then a sustained 2.5 s region")
    ap.add_argument("--min-region-s", type=float, default=MIN_REGION_S,
                    help="shortest timed region that is reported (profiling runs under rocprofv3 pass a small value)")
    ap.add_argument("--loop-net", default="conv", choices=["gemm", "conv", "none"],
                    help="network of the in_network_loop secondary measurement (conv: MIOpen 3x3 conv stack, falls back to "
                         "gemm: hipBLASLt per-pixel MLP)")
    ap.add_argument("--requests", type=int, default=32,
                    help="independent [256,4,64,64] sampling requests in flight, advanced stage by stage (one fused "
                         "launch per stage); 32 x 42 MB per stage > the 256 MiB Infinity Cache")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32", "bf16"])
    ap.add_argument("--eps-dtype", default=None, choices=["fp16", "fp32", "bf16"],
                    help="dtype of the network output (default: the state dtype); fp16 with --dtype fp32 = SD under autocast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preflight", action="store_true",
                    help="run the per-rank preflight (smoke trajectory, clustered thresholding launch, all-gather -- each on the "
                         "rank's own device ordinal), print its JSON record and exit; it also runs, untimed, at the start of every "
                         "N > 1 / torch.distributed.run launch")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (the other BASELINE configurations)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (profiling runs)")
    args = ap.parse_args()
    dtype = _DT[args.dtype]
    eps_dtype = _DT[args.eps_dtype] if args.eps_dtype else dtype
    R = args.requests

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # under torch.distributed.run (RANK is set) the process group is initialised for any world size, so the collective
    # path of the N > 1 runs (barrier, MAX all-reduce of the time, final all-gather) is also exercised on one GPU
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the host driver of these boxes only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails
        # with `hipIpcGetMemHandle: invalid argument` (already exported on the GPU boxes; kept for any other launcher)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner on stdout when NCCL_DEBUG=VERSION (set on the GPU boxes): keep stdout for the
        # one JSON line by pointing fd 1 at stderr while the communicator is created
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if STUB:
                dist.init_process_group("gloo")
                dist.barrier()
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
                torch.cuda.set_device(local_rank)
                dist.barrier()
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if STUB:
        dev = torch.device("cpu")
        sync = lambda *a: None

        class Event:                                     # perf_counter stand-in for torch.cuda.Event
            def __init__(self, enable_timing=True):
                self.t = None

            def record(self, stream=None):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return (other.t - self.t) * 1e3
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
        Event = torch.cuda.Event

    import dpm_solver_amd as D
    from dpm_solver_amd import _lib as L
    assert STUB or not L.IS_LAB or os.environ.get("DPM_BENCH_ALLOW_LAB") == "1", \
        "bench.py times the PRODUCT library (unset DPM_SOLVER_AMD_LIB)"

    # ---- preflight: every rank on its own ordinal, before anything is timed; all ranks report, then a failure stops the run --
    pre = None
    if True:
        mine = preflight(D, L, dev, rank, world, dist, stub=STUB)
        allp = [mine]
        if dist is not None:
            allp = [None] * world
            dist.all_gather_object(allp, mine)
        pre = dict(ok=all(p["ok"] for p in allp), ranks=allp, seconds=max(p["seconds"] for p in allp))
        if args.preflight:
            if rank == 0:
                print(json.dumps(dict(preflight=pre, n_gpus=world, device_ordinals=[p["ordinal"] for p in allp])), flush=True)
            if dist is not None:
                dist.barrier()
                dist.destroy_process_group()
            sys.exit(0 if pre["ok"] else 1)
        assert pre["ok"], "preflight failed: %s" % [p.get("error") for p in allp if not p["ok"]]

    ac = sd_alphas_cumprod()
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(ac))
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, algorithm_type="dpmsolver++", state_dtype=dtype)
    plan = dpm._get_plan(method="multistep", order=2, steps=STEPS_SOLVER, skip_type="time_uniform",
                         solver_type="dpmsolver", lower_order_final=True, denoise_to_zero=False,
                         t_T=1.0, t_0=1.0 / ns.total_N)
    n_stages = len(plan.stages)
    sets = make_sets(R, dtype, dev, seed=1234 + rank, eps_dtype=eps_dtype)   # independent samples per rank (seed + rank)
    if STUB:
        stream, sptr = None, None
    else:
        stream = torch.cuda.Stream(device=dev)          # a real stream: hipGraph capture cannot use the null stream
        stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(stream)
        sptr = C.c_void_p(stream.cuda_stream)
    rbs = (L.RunBuffers * R)(*[s_["rb"] for s_ in sets])
    resm = (C.c_int * R)()

    P = max(1, args.trajectories_per_step)

    def trajectory():
        """one 20-stage trajectory of the R requests in flight: 20 fused launches"""
        if STUB:
            time.sleep(2e-4 * (1 + rank))               # ranks of different speed: value must use the slowest
            return
        L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, sptr, None, resm))

    def step():
        for _ in range(P):
            trajectory()

    def barrier():
        sync(dev)
        if dist is not None:
            dist.barrier()
        sync(dev)

    # ---- parity spot check outside the timed region: fused launches == the Python host loop (single launches) ---
    trajectory()
    sync(dev)
    for r in (() if STUB else sorted({0, R - 1})):
        s_ = sets[r]
        dchk = D.DPM_Solver(D.model_wrapper(lambda x, t, e=s_["eps"]: e, ns), ns, state_dtype=dtype)
        want = dchk.sample(s_["x"][0], steps=STEPS_SOLVER, order=2)
        torch.cuda.synchronize(dev)
        assert torch.equal(s_["x"][resm[r]], want), "fused multi-request launches and the Python loop disagree (request %d)" % r

    for _ in range(args.warmup):
        step()
    steps = args.steps
    while True:
        barrier()
        ev0, ev1, evm = (Event(enable_timing=True) for _ in range(3))
        t0 = time.perf_counter()
        ev0.record(stream)                              # HIP events on the launch stream, around the timed region
        for k in range(steps):
            if k == steps // 2:
                evm.record(stream)                      # midpoint: second half vs first half = throttling check
            step()
        ev1.record(stream)
        barrier()
        wall = time.perf_counter() - t0
        region_us = ev0.elapsed_time(ev1) * 1e3         # GPU time of the K steps = K * P * 20 fused launches
        h1_us, h2_us = ev0.elapsed_time(evm) * 1e3, evm.elapsed_time(ev1) * 1e3
        wall_min = wall
        if dist is not None:
            tw = torch.tensor([wall, -wall], dtype=torch.float64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)                 # max over ranks (and, negated, the min)
            wall, wall_min = float(tw[0].item()), -float(tw[1].item())
        if wall >= args.min_region_s:
            break
        steps = int(steps * max(2.0, 1.3 * args.min_region_s / max(wall, 1e-6))) + 1   # too short to report: time more steps

    # ---- roofline of the timed region ---------------------------------------------------------------------------
    n_el = B * int(np.prod(SHAPE))
    ssz = torch.empty((), dtype=dtype).element_size()
    esz = torch.empty((), dtype=eps_dtype).element_size()
    alg_bytes = n_el * (4 * ssz + esz)                   # steady-state 2M stage of ONE request: reads x, eps, m_prev; writes x', m
    traj_alg_bytes = n_el * (18 * (4 * ssz + esz) + (3 * ssz + esz) + (3 * ssz + esz))  # + first (no history) and last (no m)
    launches = steps * P * n_stages
    launch_us = region_us / launches
    achieved = traj_alg_bytes * R * steps * P / (region_us * 1e-6) / 1e9
    k1 = steps // 2
    sustained = dict(region_s=round(wall, 3), gpu_region_s=round(region_us * 1e-6, 3),
                     second_half_over_first_half=(round((h1_us / max(k1, 1)) / (h2_us / max(steps - k1, 1)), 4)
                                                  if 0 < k1 < steps else None),
                     note="throughput of the second half of the timed region / the first half (HIP events at the "
                          "midpoint); < 1 would mean clocks or HBM throttled while the region ran")
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per launch from rocprofv3 --pmc passes
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("fused_" + args.dtype, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    traffic_source=("profiles/traffic.json (rocprofv3 --pmc passes of this command, committed: "
                                    "FETCH_SIZE x 2 + WRITE_SIZE per fused launch; not measured in this run)"
                                    if traffic is not None else None),
                    kernel="stage_kernel_multi<%s,%s,FORM_TWO,GUIDE_NONE,SPEC_NOISE_X0> (2M steady state, %d requests per launch)"
                           % (args.dtype, args.eps_dtype or args.dtype, min(R, L.MULTI_MAX)),
                    mode="hbm_cold: %d requests of [%d,4,64,64] in flight, advanced stage by stage, one fused launch per "
                         "stage (%.0f MB per launch)" % (R, B, traj_alg_bytes * R / n_stages / 1e6),
                    launch_us=round(launch_us, 3), per_request_stage_us=round(launch_us / R, 4),
                    algorithmic_bytes_per_launch=round(traj_alg_bytes * R / n_stages),
                    how="algorithmic bytes of the timed region (98*n*s per request trajectory) / its GPU time by HIP events "
                        "on the launch stream; launch_us = that time / fused launches (dispatch gaps included, as in the "
                        "rocprofv3 kernel-trace average)",
                    host_wall=dict(achieved=round(traj_alg_bytes * R * steps * P / wall / 1e9, 1),
                                   frac=round(traj_alg_bytes * R * steps * P / wall / 1e9 / HBM_PEAK_GBS, 4)))

    if not args.no_secondary:
        # (1) kernel-only durations of the fused launches (start -> stop events of each launch)
        msb = (C.c_float * (R * n_stages))()
        ko = []
        for _ in range(3):
            L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, sptr, msb, resm))
            ko.append(np.frombuffer(msb, dtype=np.float32).reshape(R, n_stages)[:, 1:n_stages - 1].astype(np.float64).sum(axis=0))
        ko_us = float(np.mean(ko) * 1e3)                 # per fused steady-state launch
        roofline["kernel_only"] = dict(us=round(ko_us, 3), achieved=round(alg_bytes * R / ko_us / 1e3, 1),
                                       frac=round(alg_bytes * R / ko_us / 1e3 / HBM_PEAK_GBS, 4),
                                       how="steady-state fused launches, start->stop events of each launch")
        # (2) the same requests, one launch each (interleaved): what a single request's stage costs from HBM
        no_fuse = L.LaunchOpts()                         # dpm_launch_opts.no_fuse: a per-call option, no process-wide switch
        no_fuse.no_fuse = 1
        rbs[0].opts = C.pointer(no_fuse)
        cold = []
        for _ in range(3):
            L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, sptr, msb, resm))
            cold.append(np.frombuffer(msb, dtype=np.float32).reshape(R, n_stages)[:, 1:n_stages - 1].copy())
        rbs[0].opts = None
        cold = np.asarray(cold, dtype=np.float64) * 1e3
        cold_med = float(np.median(cold))
        stalled = cold > 50.0 * cold_med                 # see profiles/r02_stall.md
        cold_us = float(cold[~stalled].mean())
        roofline["single_request_cold"] = dict(
            kernel_us=round(cold_us, 3), median_us=round(cold_med, 3), max_us=round(float(cold.max()), 1),
            stalled_launches=int(stalled.sum()), launches=int(cold.size),
            achieved=round(alg_bytes / cold_us / 1e3, 1), frac=round(alg_bytes / cold_us / 1e3 / HBM_PEAK_GBS, 4),
            how="the %d requests advanced stage by stage with ONE launch per request (42 MB launches), kernel-only" % R)
        # (3) one request, stages back to back: its inputs sit in the Infinity Cache (last round's headline mode)
        res1 = C.c_int(-1)
        k2 = 200
        for i in range(8):
            L.check(L.lib.dpm_plan_run(plan.handle, C.byref(sets[i % min(8, R)]["rb"]), None, None, sptr, C.byref(res1)))
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(k2):
            L.check(L.lib.dpm_plan_run(plan.handle, C.byref(sets[i % min(8, R)]["rb"]), None, None, sptr, C.byref(res1)))
        e1.record(stream)
        torch.cuda.synchronize(dev)
        warm_us = e0.elapsed_time(e1) * 1e3
        buf = (C.c_float * n_stages)()
        ms1 = []
        for r in range(16):
            L.check(L.lib.dpm_plan_run_multi(plan.handle, C.byref(sets[r % min(8, R)]["rb"]), 1, sptr, buf, C.byref(res1)))
            ms1.append(np.frombuffer(buf, dtype=np.float32)[1:n_stages - 1].astype(np.float64).mean() * 1e3)
        k1_us = float(np.mean(ms1))
        roofline["cache_resident"] = dict(
            achieved=round(traj_alg_bytes * k2 / warm_us / 1e3, 1),
            frac=round(traj_alg_bytes * k2 / warm_us / 1e3 / HBM_PEAK_GBS, 4),
            launch_us=round(warm_us / (k2 * n_stages), 3), msamples_per_s=round(B * k2 / warm_us, 4),
            kernel_only_us=round(k1_us, 3), kernel_only_frac=round(alg_bytes / k1_us / 1e3 / HBM_PEAK_GBS, 4),
            how="ONE request, 20 launches back to back (dpm_plan_run, frozen outputs): x and m are re-read from the "
                "256 MiB Infinity Cache -- not the situation of a real sampling loop")
        # (5) SURVEY 8(d): what a plain device-to-device copy of the same footprint achieves on this box
        half = R * n_el * ssz * 5 // 2 // 4096 * 4096          # read + write = the fused launch's bytes
        src_c = torch.empty(half, dtype=torch.uint8, device=dev)
        dst_c = torch.empty(half, dtype=torch.uint8, device=dev)
        for _ in range(2):
            dst_c.copy_(src_c)
        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ce0.record(stream)
        for _ in range(6):
            dst_c.copy_(src_c)
        ce1.record(stream)
        torch.cuda.synchronize(dev)
        copy_us = ce0.elapsed_time(ce1) * 1e3 / 6
        del src_c, dst_c
        roofline["copy_ceiling"] = dict(
            pattern="torch copy_ (device to device), %d MB read + %d MB written" % (half // 10**6, half // 10**6),
            us=round(copy_us, 3), achieved=round(2 * half / copy_us / 1e3, 1),
            frac=round(2 * half / copy_us / 1e3 / HBM_PEAK_GBS, 4),
            stage_kernel_vs_copy=round(roofline["achieved"] / (2 * half / copy_us / 1e3), 3))

    # ---- the drop-in Python API on the same workload: DPM_Solver.sample() per request, frozen network ------------
    py_ms = py_req_ms = None
    if not args.no_secondary:
        s_ = sets[0]
        dpy = D.DPM_Solver(D.model_wrapper(lambda x, t, e=s_["eps"]: e, ns), ns, state_dtype=dtype)
        for _ in range(5):
            dpy.sample(s_["x"][0], steps=STEPS_SOLVER, order=2)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        kk = 50
        for i in range(kk):
            dpy.sample(sets[i % min(8, R)]["x"][0], steps=STEPS_SOLVER, order=2)
        torch.cuda.synchronize(dev)
        py_ms = (time.perf_counter() - t1) / kk * 1e3
        # ... and DPM_Solver.sample_requests() on the timed workload itself: the R requests advanced together from
        # Python, per stage R network calls (each request its own frozen output) and one fused launch
        calls = [0]

        def frozen(x, t):
            calls[0] += 1
            return sets[(calls[0] - 1) % R]["eps"]
        dpr = D.DPM_Solver(D.model_wrapper(frozen, ns), ns, state_dtype=dtype)
        xs = [s2["x"][0] for s2 in sets]
        for _ in range(2):
            calls[0] = 0
            dpr.sample_requests(xs, steps=STEPS_SOLVER, order=2)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        kk = 5
        for i in range(kk):
            calls[0] = 0
            dpr.sample_requests(xs, steps=STEPS_SOLVER, order=2)
        torch.cuda.synchronize(dev)
        py_req_ms = (time.perf_counter() - t1) / kk * 1e3
        del dpr

    # ---- the single end-of-sampling collective of the sharded path: all-gather of the final x (timed apart) ---
    gather_ms = None
    if dist is not None:
        final = sets[0]["x"][resm[0]]
        nb = final.shape[0]                              # concatenated along the batch: the form RCCL and gloo both take
        out = torch.empty((world * nb,) + tuple(final.shape[1:]), dtype=final.dtype, device=dev)
        dist.all_gather_into_tensor(out, final)                          # warm-up (communicator setup)
        barrier()
        tg = time.perf_counter()
        dist.all_gather_into_tensor(out, final)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert torch.equal(out[rank * nb:(rank + 1) * nb], final)
        if world > 1:                                    # seeds follow seed + rank: the shards must differ
            o = (rank + 1) % world
            assert not torch.equal(out[o * nb:(o + 1) * nb], final)

    # ---- secondaries on the LAB build, in a subprocess: the stage kernel inside a real torch network loop (one request,
    # the drop-in sample() call), the no-arithmetic ceilings, the floor of the lone launch ------------------------------
    lab_configs = None
    if not args.no_secondary and world == 1:
        torch.cuda.synchronize(dev)
        lab = lab_secondary(args.dtype, args.eps_dtype, args.loop_net, R)
        if "error" in lab:
            roofline["in_network_loop"] = dict(error=lab["error"])
        else:
            for k in ("in_network_loop", "no_arithmetic_ceiling", "lone_launch_floor"):
                if k in lab:
                    roofline[k] = lab[k]
            lab_configs = lab.get("configs_in_loop")
            nac = roofline.get("no_arithmetic_ceiling")
            if nac and "fused_size_us" in nac and "kernel_only" in roofline:
                nac["stage_kernel_vs_ceiling"] = round(nac["fused_size_us"] / roofline["kernel_only"]["us"], 3)
        inl = roofline.get("in_network_loop", {})
        if "error" not in inl:
            # the rocprofv3 kernel rows of the same loop, committed (bench.py cannot run under rocprofv3 inside itself): a
            # SEPARATE key -- `stage_kernel_us` / `frac` above are this run's own event measurement
            try:
                rows = json.load(open(os.path.join(ROOT, "profiles", "in_loop.json"))).get(args.dtype)
            except Exception:
                rows = None
            if rows:
                inl["rocprofv3_rows_committed"] = dict(rows, measured_in_this_run=False,
                                                       source="profiles/in_loop.json (tools/in_loop.py under rocprofv3 --kernel-trace)")

    # ---- the OTHER BASELINE configurations on this run's clock (product library, this process): tools/config_bench.py ----
    configs = None
    if world == 1 and not args.no_secondary and not args.no_configs and not STUB:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            import config_bench
            torch.cuda.synchronize(dev)
            tcb = time.perf_counter()
            configs = config_bench.run_all(dev)
            for k, v in (lab_configs or {}).items():
                if k in configs and isinstance(configs[k], dict):
                    configs[k]["in_network_loop"] = v
            configs["seconds"] = round(time.perf_counter() - tcb, 2)
        except Exception as e:
            configs = dict(error="%s: %s" % (type(e).__name__, e))

    if rank == 0:
        samples = world * steps * P * R * B
        line = {
            "metric": baseline_metric(),
            "value": round(samples / wall / 1e6, 4), "unit": "Msamples/s",
            "n_gpus": world, "steps": steps, "steps_requested": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / steps * 1e3, 5),
            "ms_per_trajectory": round(wall / steps / P * 1e3, 5),   # one 20-stage pass over the R requests (C loop)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",          # the arithmetic type of the path: fp32 whatever the storage type of the state is
            "storage_dtype": {"fp16": "f16", "fp32": "f32", "bf16": "bf16"}[args.dtype], "data": "synthetic",
            "config": {"workload": "DPM-Solver++ 2M, 20 steps, [256,4,64,64] %s state / %s network output per request (fp32 "
                                   "arithmetic), %d requests in flight per GPU advanced stage by stage (inputs of every "
                                   "stage come from HBM), frozen model_fn (eps pre-staged), SD-v1 scaled-linear schedule, "
                                   "time_uniform; a step = %d trajectories of the requests in flight"
                                   % (args.dtype, args.eps_dtype or args.dtype, R, P),
                       "batch_per_request": B, "requests_in_flight_per_gpu": R, "trajectories_per_step": P,
                       "samples_per_step_per_gpu": P * R * B, "solver_stages_per_step": P * n_stages,
                       "launch": "one fused launch per stage (dpm_plan_run_multi -> dpm_stage_launch_multi)",
                       "parallelism": "batch-sharded x%d, no data-path collective" % world},
            "msample_steps_per_s": round(samples * n_stages / wall / 1e6, 3),
            "sustained": sustained,
            "python_api_ms_per_trajectory": None if py_ms is None else round(py_ms, 4),
            # one trajectory of the R requests through DPM_Solver.sample_requests (compare with ms_per_trajectory, the C loop)
            "python_api_requests_ms_per_trajectory": None if py_req_ms is None else round(py_req_ms, 4),
            "roofline": roofline,
            "gather_ms": None if gather_ms is None else round(gather_ms, 4),
            "per_rank_wall_s": {"min": round(wall_min, 4), "max": round(wall, 4)},   # value uses the max
            # weak scaling: every GPU runs the N = 1 workload, so `value` / N is what one GPU of this run achieved -- the
            # number to hold against the N = 1 line of the same box (BENCH vs SCALE[N=1]; the fastest rank's own rate next to it)
            "value_per_gpu": round(samples / wall / 1e6 / world, 4),
            "fastest_rank_value_per_gpu": round(samples / world / wall_min / 1e6, 4),
            "launcher": "torch.distributed.run" if "RANK" in os.environ else "python",
            # what the collectives of this run really spanned: ranks of the process group (RCCL on GPUs) and the device
            # ordinal every rank computed on (from the preflight's all-gather; a plain python launch has no group)
            "rccl_ranks": (dist.get_world_size() if dist is not None else 0),
            "collective_backend": (dist.get_backend() if dist is not None else None),
            "device_ordinals": ([p["ordinal"] for p in pre["ranks"]] if pre is not None else [None if STUB else local_rank]),
            "preflight": pre,
        }

        if STUB:
            line["stub"] = True
            line["value"] = line["roofline"] = line["value_per_gpu"] = line["fastest_rank_value_per_gpu"] = None   # nothing was measured
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ac)
        if world == 1 and not args.no_secondary and not STUB:
            try:
                line["cold_start_ms"] = cold_start()
            except Exception as e:                           # a secondary never takes the primary line down
                line["cold_start_ms"] = dict(error="%s: %s" % (type(e).__name__, e))
        try:     # the unmodified reference on the same kind of GPU through PyTorch-ROCm eager (tools/gpu_reference.py; committed)
            gr = json.load(open(os.path.join(ROOT, "profiles", "gpu_reference.json")))
            line["reference_on_mi355x"] = dict(
                source="profiles/gpu_reference.json (tools/gpu_reference.py through gpurun; committed, not measured in this run)",
                what=gr["what"], rows=[{k: r[k] for k in ("case", "reference_on_gpu_ms", "engine_ms", "speedup",
                                                          "rel_err_vs_reference_on_gpu")} for r in gr["rows"]])
        except Exception:
            pass
        if isinstance(line.get("roofline"), dict):
            # The driver's record of this line keeps the SCALAR fields of `roofline` / `cpu_baseline` and the last 2000
            # characters of the line (BENCH_r05.json: nested objects are dropped from `parsed`): the figures a reader needs are
            # therefore repeated flat here, and the compact `configs` block is the line's LAST key.
            rf = line["roofline"]
            for k, sub, field in (("single_request_cold_us", "single_request_cold", "kernel_us"),
                                  ("single_request_cold_frac", "single_request_cold", "frac"),
                                  ("kernel_only_us", "kernel_only", "us"), ("kernel_only_frac", "kernel_only", "frac"),
                                  ("cache_resident_frac", "cache_resident", "frac"),
                                  ("in_network_loop_stage_kernel_us", "in_network_loop", "stage_kernel_us"),
                                  ("in_network_loop_frac", "in_network_loop", "frac"),
                                  ("in_network_loop_frac_of_floor", "in_network_loop", "frac_of_floor"),
                                  ("copy_ceiling_frac", "copy_ceiling", "frac")):
                if isinstance(rf.get(sub), dict) and field in rf[sub]:
                    rf[k] = rf[sub][field]
        if configs is not None:
            line["configs_detail"] = configs
            compact = {}
            for name, r in configs.items():
                if not isinstance(r, dict) or "case" not in r:
                    continue
                if "error" in r:
                    compact[name] = dict(error=r["error"][:80], measured_in_this_run=False)
                    continue
                c_ = dict(shape="x".join(str(v) for v in r["shape"]), stages=r["stages_per_trajectory"], us=r["us_per_stage"],
                          captured_us=r.get("captured", {}).get("us_per_stage"), algorithmic_bytes=r["algorithmic_bytes_per_stage"],
                          frac=r["frac"], captured_frac=r.get("captured", {}).get("frac"), x_floor=r.get("x_latency_bound_stage"),
                          measured_in_this_run=True)
                if "us_per_order3_step" in r:
                    c_["us_per_order3_step"] = r["us_per_order3_step"]
                il = r.get("in_network_loop")
                if isinstance(il, dict) and "stage_kernel_us" in il:
                    c_["in_loop_kernel_us"], c_["in_loop_frac"] = il["stage_kernel_us"], il["frac"]
                compact[name] = c_
                if isinstance(line.get("roofline"), dict):
                    line["roofline"]["%s_us_per_stage" % name] = r["us_per_stage"]
                    line["roofline"]["%s_frac" % name] = r["frac"]
                    if c_.get("in_loop_kernel_us") is not None:
                        line["roofline"]["%s_in_loop_kernel_us" % name] = c_["in_loop_kernel_us"]
            compact["how"] = ("us = HIP-event time of K eager DPM_Solver.sample() trajectories / stages, frozen model_fn, product "
                              "library, this run; captured = hipGraph replay; x_floor = us / cfg1's us; in_loop = kernel-only "
                              "behind a conv network (lab subprocess); detail: configs_detail")
            line["configs"] = compact                            # LAST key: inside the tail the driver keeps
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
