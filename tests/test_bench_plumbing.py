"""bench.py's N > 1 plumbing under two gloo ranks on CPU: the driver launches
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
on an 8-GPU node that no round has had so far.  With DPM_BENCH_STUB=1 the same file runs its rank / world handling, the
process group, the barrier-bracketed timed region with the MAX over ranks, `per_rank_wall_s`, the final all-gather and
`gather_ms`, and prints its JSON line -- with the timed region replaced by a sleep (slower on rank 1) and every measurement
field nulled ("stub": true)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 8, 1])       # 8: the node the driver scales to
def test_bench_multi_rank_plumbing_with_a_stubbed_timed_region(world):
    env = dict(os.environ, DPM_BENCH_STUB="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "1",
           "--requests", "2", "--trajectories-per-step", "3", "--min-region-s", "0.002", "--no-secondary", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["stub"] is True and line["value"] is None and line["roofline"] is None
    assert line["n_gpus"] == world and line["steps"] >= 4 and line["steps_requested"] == 4 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    cfg = line["config"]
    assert cfg["requests_in_flight_per_gpu"] == 2 and cfg["trajectories_per_step"] == 3
    assert cfg["parallelism"].startswith("batch-sharded x%d" % world) and "model" not in cfg
    w = line["per_rank_wall_s"]
    assert 0 < w["min"] <= w["max"]
    # the reported time is the slowest rank's: ms_per_step * steps == per_rank_wall_s.max
    assert abs(line["ms_per_step"] * line["steps"] / 1e3 - w["max"]) < 2e-3 * max(1.0, w["max"])
    # rank r sleeps (1 + r) x 0.2 ms per trajectory and the region ends with a barrier: the reported time covers the SLOWEST
    # rank's work (every rank waits for it, so min and max agree to the barrier's skew)
    assert w["max"] >= line["steps"] * 3 * 2e-4 * world, (w, line["steps"])
    assert line["gather_ms"] is not None and line["gather_ms"] >= 0     # the single end-of-sampling collective, timed apart
    assert "cpu_baseline" not in line                                   # rank 0 at N = 1 only, and not with --no-cpu-baseline
    # the preflight ran on every rank before the timed region and its all-gather spanned the whole group
    pre = line["preflight"]
    assert pre["ok"] is True and [p["rank"] for p in pre["ranks"]] == list(range(world))
    assert all(p["collective"]["ok"] and p["collective"]["ranks"] == world and p["collective"]["bytes_per_rank"] == 512 * 1024
               for p in pre["ranks"])
    assert line["rccl_ranks"] == world and line["collective_backend"] == "gloo" and len(line["device_ordinals"]) == world
    assert r.stderr.count("[preflight] rank") == world and "FAIL" not in r.stderr


def test_preflight_only_mode_prints_its_record_and_exits():
    """`bench.py --gpus N --preflight`: the per-rank checks and nothing else -- under two gloo ranks
    the device checks are stubbed, the plumbing (per-rank lines on stderr, the gathered record, the exit code) is real"""
    env = dict(os.environ, DPM_BENCH_STUB="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["preflight"]["ok"] is True and len(rec["preflight"]["ranks"]) == 2
    assert "metric" not in rec and "value" not in rec                   # nothing was timed


def test_cpu_baseline_record_is_what_this_run_timed():
    """`cpu_baseline` is always what was timed in this run on this box (a headline key must be able to
    regress with the code under test): where the reference is absent (the GPU box) that is the numpy port, and the
    reference's own figure from the committed MI355X-box measurement rides along as `reference_committed`, marked as not
    measured in this run."""
    sys.path.insert(0, ROOT)
    import bench
    old = os.environ.get("DPM_REFERENCE_DIR")
    os.environ["DPM_REFERENCE_DIR"] = "/nonexistent"
    port_fn, cport_fn = bench.cpu_baseline_port, bench.cpu_baseline_port_c
    bench.cpu_baseline_port = lambda ac, budget_s=8.0: dict(value=0.0016, unit="Msamples/s", cores=1, kind="port", sample="stub")
    bench.cpu_baseline_port_c = lambda ac, budget_s=8.0: dict(value=0.05, unit="Msamples/s", cores=4, kind="port", sample="stub C")
    try:
        out = bench.cpu_baseline(bench.sd_alphas_cumprod())
    finally:
        bench.cpu_baseline_port, bench.cpu_baseline_port_c = port_fn, cport_fn
        if old is None:
            os.environ.pop("DPM_REFERENCE_DIR")
        else:
            os.environ["DPM_REFERENCE_DIR"] = old
    # the headline CPU figure is the fused plain-C port over the host cores; the numpy oracle's figure rides along
    assert out["kind"] == "port" and out["measured_in_this_run"] is True and out["value"] == 0.05 and out["cores"] == 4
    assert out["numpy_port_value"] == 0.0016 and out["numpy_port"]["sample"] == "stub"
    assert out["unit"] == "Msamples/s" and out["host_cores"] >= out["cores"] >= 1
    ref = out["reference_committed"]
    assert ref["kind"] == "reference" and ref["measured_in_this_run"] is False
    assert ref["source"].startswith("profiles/cpu_baseline_reference_gpubox.json")
    assert ref["cores"] == ref["threads"] == ref["best"]["threads"] and ref["value"] == ref["best"]["value"]
    # the committed same-box ratio port / reference rides along as scalars, with its source, and the
    # live port figure scaled by it is labelled an estimate
    assert 0.3 < out["port_over_reference"] < 3.0 and 0.3 < out["port_over_reference_single_thread"] < 3.0
    assert out["port_over_reference_source"].startswith("profiles/r06_cpu_baseline_port_vs_reference.json")
    assert abs(out["reference_estimate"] - out["numpy_port_value"] / out["port_over_reference"]) < 1e-6 and "not a measurement" in out["reference_estimate_unit"]
    assert out["reference_best_seen"] == ref["value"]


def test_secondaries_that_need_a_gpu_report_an_error_instead_of_raising():
    """`cold_start_ms` and the lab secondaries of the bench line run in subprocesses; without a GPU (here) each comes back
    as a dict with an `error` text -- a secondary never takes the primary line down"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the error path: no GPU")
    sys.path.insert(0, ROOT)
    import bench
    out = bench.cold_start(timeout=240)
    assert isinstance(out, dict) and "error" in out and "cold_start.py" in out["error"]
    lab = bench.lab_secondary("fp16", "fp16", "conv", 2, timeout=240)
    assert isinstance(lab, dict) and "error" in lab


def test_preflight_checker_agrees_with_the_oracle():
    """bench.py's preflight checks each rank's smoke trajectory against a torch-double restatement of DPM-Solver++(2M) it
    carries itself (bench.py may not touch oracle/ outside its cpu_baseline leg); HERE that restatement is held against the
    oracle, so the checker is itself checked"""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import dpm_oracle as O
    ac = bench.sd_alphas_cumprod()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 4, 8, 8)).astype(np.float32)
    e = rng.standard_normal((2, 4, 8, 8)).astype(np.float32)
    osch = O.Schedule.from_alphas_cumprod(ac)
    want = O.Solver(O.wrap_model(lambda xx, t: e, osch), osch).sample(x, steps=20, order=2)
    got = bench._torch_2m_double(ac, torch.from_numpy(x), torch.from_numpy(e), 20).numpy()
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-6


def test_c_port_cpu_baseline_runs_here():
    """the plain-C fused port that bench.py times as `cpu_baseline` (oracle/dpm_oracle_kernels.c): a short run on this machine's cores"""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import dpm_oracle_c as OC
    OC.build()
    out = bench.cpu_baseline_port_c(bench.sd_alphas_cumprod(), budget_s=2.0)
    assert out["kind"] == "port" and out["value"] > 0 and 1 <= out["cores"] <= out["host_cores"]
    assert out["single_thread"]["threads"] == 1 and "dpm_oracle_kernels.c" in out["sample"]
