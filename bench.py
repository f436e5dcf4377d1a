#!/usr/bin/env python3
"""Solver-update benchmark: DPM-Solver++(2M), 20 steps, [256,4,64,64] fp16, frozen model_fn (BASELINE.json
configs[1]) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one complete 20-stage sampling trajectory of one batch of 256 latents: the 20 launches of the fused
stage kernel recorded by the C ABI (dpm_graph_create = dpm_plan_run under hipGraph capture) and replayed with one
dpm_graph_launch per trajectory (--mode eager: 20 individual launches through dpm_plan_run), the network frozen
(its output pre-staged in a buffer distinct from x), every input resident in HBM before the clock starts.  Steps cycle through
`--sets` independent buffer sets (default 8 x 64 MiB = 512 MiB) so that consecutive trajectories cannot live in
the 256 MiB Infinity Cache: in real use a UNet runs between two solver stages and evicts it anyway.

One JSON line on rank 0:
  value              whole-job Msamples/s = N * K * 256 / wall, wall = barrier/sync-bracketed, max over ranks
  roofline           HBM roofline of the stage kernel (2M steady state: reads x, eps, m_prev; writes x_next, m =
                     5 * n * sizeof(dtype) algorithmic bytes; first / last stage 4 * n * sizeof).  `achieved` = algorithmic
                     bytes of the timed region / its GPU time measured with HIP events on the launch stream, i.e. bytes
                     per launch / average launch duration with the dispatch gaps of the pipelined trajectory included --
                     the average rocprofv3 --kernel-trace --stats reports for the kernel (profiles/).  `kernel_only`
                     = the same for the kernel's own start -> stop time (hipExtLaunchKernelGGL events per launch).
  cpu_baseline       the numpy oracle (oracle/dpm_oracle.py, a port of the reference algorithm) timed on one host
                     core, rank 0, N=1 only, on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, SHAPE, STEPS_SOLVER = 256, (4, 64, 64), 20
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured float4-copy ceiling is ~6290


def sd_alphas_cumprod():
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas).astype(np.float32)


def baseline_metric():
    """the metric string of BASELINE.json (value = its Msamples/s part; the HBM GB/s part is `roofline`)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "solver-update Msamples/sec + HBM GB/s, DPM-Solver++(2M) 20-step, Bx4x64x64"


def make_sets(n_sets, dtype, dev, seed):
    """n_sets independent buffer sets: x_T, frozen eps, 3 state scratch buffers, 2 history slots."""
    from dpm_solver_amd import _lib as L
    g = torch.Generator(device="cpu").manual_seed(seed)
    sets = []
    for _ in range(n_sets):
        x_T = torch.randn((B,) + SHAPE, generator=g).to(dev, dtype)
        eps = torch.randn((B,) + SHAPE, generator=g).to(dev, dtype)
        xb = [x_T] + [torch.empty_like(x_T) for _ in range(3)]
        hb = [torch.empty_like(x_T) for _ in range(2)]
        rb = L.RunBuffers()
        for i in range(4):
            rb.xbuf[i] = xb[i].data_ptr()
        for i in range(2):
            rb.hist[i] = hb[i].data_ptr()
        rb.e0 = eps.data_ptr()
        rb.n, rb.batch = x_T.numel(), B
        rb.state_dtype = rb.eps_dtype = {torch.float16: L.DTYPE_F16, torch.float32: L.DTYPE_F32,
                                         torch.bfloat16: L.DTYPE_BF16}[dtype]
        sets.append(dict(rb=rb, x=xb, h=hb, eps=eps))
    return sets


def cpu_baseline(ac, budget_s=12.0):
    """numpy oracle (port of the reference's algorithm, one thread) on the same workload, bounded sample."""
    from oracle import dpm_oracle as O
    osch = O.Schedule.from_alphas_cumprod(ac)
    rng = np.random.default_rng(0)
    bs = B
    x = rng.standard_normal((bs,) + SHAPE).astype(np.float32)
    eps = rng.standard_normal((bs,) + SHAPE).astype(np.float32)
    O.Solver(O.wrap_model(lambda xx, t: eps[:4], osch), osch).sample(x[:4], steps=STEPS_SOLVER, order=2)  # warm-up
    sol = O.Solver(O.wrap_model(lambda xx, t: eps, osch), osch, algorithm_type="dpmsolver++")
    t0 = time.perf_counter()
    n = 0
    while True:
        sol.sample(x, steps=STEPS_SOLVER, order=2)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return dict(value=bs * n / el / 1e6, unit="Msamples/s", cores=1, kind="port",
                sample="%d trajectories of [%d,4,64,64] fp32 (numpy oracle, frozen eps), %.1f s" % (n, bs, el))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sets", type=int, default=8, help="independent buffer sets cycled through (cache defeat)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32", "bf16"])
    ap.add_argument("--mode", default="eager", choices=["graph", "eager"],
                    help="eager: 20 launches per trajectory through dpm_plan_run (default, measured fastest); graph: one hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    dtype = {"fp16": torch.float16, "fp32": torch.float32, "bf16": torch.bfloat16}[args.dtype]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # under torch.distributed.run (RANK is set) the process group is initialised for any world size, so the collective
    # path of the N > 1 runs (barrier, MAX all-reduce of the time, final all-gather) is also exercised on one GPU
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when NCCL_DEBUG=VERSION (set on the GPU boxes): keep stdout for the
        # one JSON line by pointing fd 1 at stderr while the communicator is created
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm
            torch.cuda.set_device(local_rank)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import dpm_solver_amd as D
    from dpm_solver_amd import _lib as L

    ac = sd_alphas_cumprod()
    ns = D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(ac))
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns, algorithm_type="dpmsolver++", state_dtype=dtype)
    plan = dpm._get_plan(method="multistep", order=2, steps=STEPS_SOLVER, skip_type="time_uniform",
                         solver_type="dpmsolver", lower_order_final=True, denoise_to_zero=False,
                         t_T=1.0, t_0=1.0 / ns.total_N)
    n_stages = len(plan.stages)
    sets = make_sets(args.sets, dtype, dev, seed=1234 + rank)          # independent samples per rank (seed + rank)
    stream = torch.cuda.Stream(device=dev)              # a real stream: hipGraph capture cannot use the null stream
    stream.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(stream)
    sptr = C.c_void_p(stream.cuda_stream)
    res = C.c_int(-1)

    def eager(i):
        L.check(L.lib.dpm_plan_run(plan.handle, C.byref(sets[i % len(sets)]["rb"]), None, None, sptr, C.byref(res)))

    graphs = []
    for s_ in sets:                                     # one captured trajectory per buffer set
        g = C.c_void_p()
        L.check(L.lib.dpm_graph_create(plan.handle, C.byref(s_["rb"]), None, None, sptr, C.byref(g)))
        assert L.lib.dpm_graph_num_nodes(g) == n_stages
        graphs.append(g)

    def replay(i):
        g = graphs[i % len(graphs)]
        L.check(L.lib.dpm_graph_launch(g, sptr))
        res.value = L.lib.dpm_graph_result(g)

    trajectory = replay if args.mode == "graph" else eager

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- parity spot check outside the timed region: native loop == Python host loop (same kernels) ---------
    s0 = sets[0]
    dchk = D.DPM_Solver(D.model_wrapper(lambda x, t: s0["eps"], ns), ns, state_dtype=dtype)
    want = dchk.sample(s0["x"][0], steps=STEPS_SOLVER, order=2)
    trajectory(0)
    torch.cuda.synchronize(dev)
    assert torch.equal(s0["x"][res.value], want), "native loop / graph replay and Python loop disagree"

    for i in range(args.warmup):
        trajectory(i)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)                                  # HIP events on the launch stream, around the timed region
    for i in range(args.steps):
        trajectory(i)
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    region_us = ev0.elapsed_time(ev1) * 1e3             # GPU time of the K trajectories = K * 20 stage launches
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    # the other launch mode, outside the contract's timed region (same K), for the record
    other = eager if args.mode == "graph" else replay
    for i in range(min(args.warmup, 8)):
        other(i)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for i in range(args.steps):
        other(i)
    torch.cuda.synchronize(dev)
    other_ms = (time.perf_counter() - t1) / args.steps * 1e3

    # ---- roofline.  Two durations of the stage kernel are reported:
    #   launch_us  = GPU time of the timed region (HIP events on the launch stream) / number of launches: the average
    #                launch duration in the pipelined trajectory, dispatch gap to the next dependent launch included.
    #                This is what rocprofv3 --kernel-trace --stats reports as the kernel's average (its timestamps of
    #                consecutive launches abut), and what `roofline.achieved` uses: the region's algorithmic bytes
    #                (98 N s per trajectory = 18 x 5N + 2 x 4N) / region time.
    #   kernel_only = start -> stop of each launch by hipExtLaunchKernelGGL events (dpm_plan_run_timed), steady-state
    #                2M stages only, 5 N s bytes: the kernel without the dispatch gap.
    n_el = B * int(np.prod(SHAPE))
    esz = torch.empty((), dtype=dtype).element_size()
    reps = max(2 * len(sets), 16)
    ms = np.zeros((reps, n_stages), dtype=np.float64)
    buf = (C.c_float * n_stages)()
    for r in range(reps):
        L.check(L.lib.dpm_plan_run_timed(plan.handle, C.byref(sets[r % len(sets)]["rb"]), sptr, buf, C.byref(res)))
        ms[r] = np.frombuffer(buf, dtype=np.float32)
    steady = ms[:, 1:n_stages - 1]                       # stages 1..18: the 5-stream 2M kernel
    k_us = float(steady.mean() * 1e3)
    alg_bytes = 5 * n_el * esz
    kernel_only = alg_bytes / (k_us * 1e-6) / 1e9
    traj_alg_bytes = (18 * 5 + 2 * 4) * n_el * esz       # 98 N elements per 20-step trajectory
    launch_us = region_us / (args.steps * n_stages)
    achieved = traj_alg_bytes * args.steps / (region_us * 1e-6) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per launch from rocprofv3 --pmc passes
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.dtype, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # HBM-cold variant of the same kernel: many requests are advanced stage by stage (dpm_plan_run_multi), so between
    # two stages of one request the other requests' traffic has flushed the 256 MiB Infinity Cache -- what happens in
    # real use, where a UNet runs between two solver stages.
    # 32 requests: > 1 GB of other traffic between two uses of a buffer.  (With 8, part of a request's buffers survives
    # in the cache and the figure is ~6 % too good.)
    cold_sets = sets + make_sets(max(0, 32 - len(sets)), dtype, dev, seed=4321 + rank)
    nreq = len(cold_sets)
    rbs = (L.RunBuffers * nreq)(*[s_["rb"] for s_ in cold_sets])
    resm = (C.c_int * nreq)()
    msb = (C.c_float * (nreq * n_stages))()
    cold = []
    for r in range(3):
        L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, nreq, sptr, msb, resm))
        cold.append(np.frombuffer(msb, dtype=np.float32).reshape(nreq, n_stages)[:, 1:n_stages - 1].copy())
    cold = np.asarray(cold, dtype=np.float64) * 1e3
    cold_med_us, cold_max_us = float(np.median(cold)), float(np.max(cold))
    # On some boxes one launch in a few thousand of this loop shows a 50-75 ms gap between its start and stop events
    # (seen with both store policies' kernels present on the box and never in the timed region): such launches are
    # counted, not averaged
    stalled = cold > 50.0 * cold_med_us
    cold_us = float(cold[~stalled].mean())
    # what the memory system sustains for this pattern and size with no arithmetic at all (3 read + 2 write streams)
    cal = {}
    msv = C.c_float()
    for mode in ("warm", "cold"):
        ts = []
        for it in range(24):
            s_ = sets[0] if mode == "warm" else cold_sets[it % nreq]
            # same cache policy as the stage kernel in that situation: default when warm, streaming loads when cold
            L.check(L.lib.dpm_calib_launch(1, 256, 8, L.lib.dpm_tuning_get(L.TUNE_NONTEMPORAL) if L.lib.dpm_tuning_get(L.TUNE_NONTEMPORAL) >= 0
                                           else (0 if mode == "warm" else 5),
                                           s_["x"][1].data_ptr(), s_["x"][2].data_ptr(), s_["x"][3].data_ptr(),
                                           s_["h"][0].data_ptr(), s_["h"][1].data_ptr(), n_el * esz, sptr, C.byref(msv)))
            if it >= 8:
                ts.append(msv.value)
        cal[mode] = float(np.mean(ts) * 1e3)
    roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    kernel="stage_kernel<%s,%s,FORM_TWO,GUIDE_NONE> (2M steady state)" % (args.dtype, args.dtype),
                    launch_us=round(launch_us, 3),
                    algorithmic_bytes_per_launch=round(traj_alg_bytes / n_stages),
                    how="algorithmic bytes of the timed region (98*N*s per 20-launch trajectory) / its GPU time by HIP "
                        "events on the launch stream; launch_us = that time / launches (dispatch gaps included, as in "
                        "the rocprofv3 kernel-trace average)",
                    kernel_only=dict(us=round(k_us, 3), us_min=round(float(steady.min() * 1e3), 3),
                                     achieved=round(kernel_only, 1), frac=round(kernel_only / HBM_PEAK_GBS, 4),
                                     algorithmic_bytes_per_launch=alg_bytes,
                                     how="steady-state 2M launches, start->stop events of each launch"),
                    first_last_stage_us=[round(float(ms[:, 0].mean() * 1e3), 3), round(float(ms[:, -1].mean() * 1e3), 3)],
                    trajectory_kernel_sum_us=round(float(ms.sum(axis=1).mean() * 1e3), 2),
                    host_wall=dict(achieved=round(traj_alg_bytes * args.steps / wall / 1e9, 1),
                                   frac=round(traj_alg_bytes * args.steps / wall / 1e9 / HBM_PEAK_GBS, 4)),
                    hbm_cold=dict(kernel_us=round(cold_us, 3), median_us=round(cold_med_us, 3), max_us=round(cold_max_us, 1),
                                  stalled_launches=int(stalled.sum()), launches=int(cold.size),
                                  achieved=round(alg_bytes / cold_us / 1e3, 1),
                                  frac=round(alg_bytes / cold_us / 1e3 / HBM_PEAK_GBS, 4),
                                  how="%d requests advanced stage by stage (dpm_plan_run_multi)" % nreq),
                    no_arithmetic_ceiling=dict(pattern="3 read + 2 write streams, same bytes, 256-thread blocks",
                                               warm_us=round(cal["warm"], 3), cold_us=round(cal["cold"], 3),
                                               frac_of_ceiling_warm=round(cal["warm"] / k_us, 3),
                                               frac_of_ceiling_cold=round(cal["cold"] / cold_us, 3)))

    # ---- the single end-of-sampling collective of the sharded path: all-gather of the final x (timed apart) ---
    gather_ms = None
    if dist is not None:
        final = sets[(args.steps - 1) % len(sets)]["x"][res.value]
        out = torch.empty((world,) + tuple(final.shape), dtype=final.dtype, device=dev)
        dist.all_gather_into_tensor(out, final)                          # warm-up (communicator setup)
        barrier()
        tg = time.perf_counter()
        dist.all_gather_into_tensor(out, final)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert torch.equal(out[rank], final)

    if rank == 0:
        samples = world * args.steps * B
        line = {
            "metric": baseline_metric(),
            "value": round(samples / wall / 1e6, 4), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",          # the arithmetic type of the path: fp32 whatever the storage type of the state is
            "storage_dtype": {"fp16": "f16", "fp32": "f32", "bf16": "bf16"}[args.dtype], "data": "synthetic",
            "config": {"workload": "DPM-Solver++ 2M, 20 steps, [256,4,64,64] %s state and network output per GPU (fp32 "
                                   "arithmetic), frozen model_fn (eps pre-staged), SD-v1 scaled-linear schedule, "
                                   "time_uniform" % args.dtype,
                       "batch_per_gpu": B, "solver_stages_per_step": n_stages, "buffer_sets": len(sets),
                       "launch": "hipGraph replay (dpm_graph_launch)" if args.mode == "graph" else "eager (dpm_plan_run)",
                       "parallelism": "batch-sharded x%d, no data-path collective" % world},
            "msample_steps_per_s": round(samples * n_stages / wall / 1e6, 3),
            ("eager_ms_per_step" if args.mode == "graph" else "graph_ms_per_step"): round(other_ms, 5),
            "roofline": roofline,
            "gather_ms": None if gather_ms is None else round(gather_ms, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ac)
        print(json.dumps(line), flush=True)
    for g in graphs:
        L.lib.dpm_graph_destroy(g)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
