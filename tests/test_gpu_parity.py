"""GPU parity tests: the HIP path (through the C ABI) against the numpy oracle, the reference goldens and
size-independent properties at BASELINE.json's full sizes.  Run on an MI355X:  pytest -m gpu
"""
import ctypes as C_
import os

import numpy as np
import pytest
import torch

import cases as C
import dpm_solver_amd as D
import dpm_solver_amd.solver as S
from conftest import rel_err
from dpm_solver_amd import _lib as L
from engine_cases import build_solver, make_schedule, run_case, sample_kwargs, tt
from kernel_double import install_cpu_double, launch_stage_double
from oracle import dpm_oracle as O
import test_oracle_golden as TO

pytestmark = pytest.mark.gpu
F32 = np.float32
TOL = 1e-5          # north-star tolerance: max|a-b| / max|b|, fp32 state
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need a GPU; run with -m 'not gpu' elsewhere"
    yield
    torch.cuda.synchronize()


def test_library_and_device():
    n_cu, lds = C_.c_int(), C_.c_int()
    arch = C_.create_string_buffer(64)
    L.check(L.lib.dpm_device_info(C_.byref(n_cu), C_.byref(lds), arch, 64))
    assert arch.value.decode().startswith("gfx950"), arch.value
    assert n_cu.value >= 64 and lds.value >= 64 * 1024
    print("device:", arch.value.decode(), n_cu.value, "CUs", lds.value, "B LDS")


# ------------------------------------------------------------------------------------------------
# end to end: HIP engine vs goldens from the real reference, vs oracle, vs the numpy kernel double
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [c["name"] for c in C.E2E_CASES])
def test_e2e_vs_reference_goldens_and_oracle(golden, name, monkeypatch):
    case = C.E2E_BY_NAME[name]
    trace = []
    xf, inter = run_case(case, DEV, trace)
    g = lambda k: golden.get("e2e", "e2e/%s/%s" % (name, k))
    assert xf.is_cuda and xf.dtype == torch.float32
    assert len(inter) == int(g("n_intermediates"))
    got = xf.cpu().numpy()
    e_ref = rel_err(got, g("final"))
    assert e_ref < TOL, ("vs reference golden", e_ref)
    xo, _ = TO.run_oracle_case(case)
    e_or = rel_err(got, xo)
    assert e_or < TOL, ("vs oracle", e_or)
    if case["intermediates"]:
        ri = g("intermediates")
        for i, v in enumerate(inter):
            assert rel_err(v.float().cpu().numpy(), ri[i]) < TOL, i
    np.testing.assert_array_equal(np.array([b for b, _ in trace]), g("trace_b"))
    # same host logic with the numpy kernel double on CPU: the device arithmetic is bit-identical
    install_cpu_double(monkeypatch, S, D)
    xd, _ = run_case(case, "cpu")
    np.testing.assert_array_equal(got, xd.numpy())


# ------------------------------------------------------------------------------------------------
# per-update methods (ref :547-954)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sname", ["sd", "vp_linear"])
@pytest.mark.parametrize("algo", ["dpmsolver++", "dpmsolver"])
def test_public_update_methods(golden, sname, algo):
    g = lambda k: torch.from_numpy(golden.get("updates", k)).to(DEV)
    r = lambda k: golden.get("updates", k)
    x, m = g("upd/x"), [g("upd/m%d" % i) for i in range(3)]
    t = [torch.tensor([v], device=DEV) for v in golden.get("updates", "upd/t")]
    ns = make_schedule(sname)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, tv: C.model_half(xx, tv), ns), ns, algorithm_type=algo)
    pre = "upd/%s/%s/" % (sname, algo)
    tol = 3e-6
    assert rel_err(dpm.dpm_solver_first_update(x, t[2], t[3], model_s=m[2]).cpu().numpy(), r(pre + "first")) < tol
    for st in ["dpmsolver", "taylor"]:
        got = dpm.multistep_dpm_solver_second_update(x, [m[1], m[2]], [t[1], t[2]], t[3], solver_type=st)
        assert rel_err(got.cpu().numpy(), r(pre + "ms2/" + st)) < tol
        got = dpm.multistep_dpm_solver_third_update(x, m, t[:3], t[3], solver_type=st)
        assert rel_err(got.cpu().numpy(), r(pre + "ms3/" + st)) < tol
        for (r1, r2, tag) in [(None, None, "def"), (0.3, 0.75, "cust")]:
            xt, im = dpm.singlestep_dpm_solver_second_update(x, t[2], t[3], r1=r1, return_intermediate=True, solver_type=st)
            assert rel_err(xt.cpu().numpy(), r(pre + "ss2/%s/%s/x_t" % (st, tag))) < tol
            xt, im = dpm.singlestep_dpm_solver_third_update(x, t[2], t[3], r1=r1, r2=r2, return_intermediate=True,
                                                            solver_type=st)
            assert rel_err(xt.cpu().numpy(), r(pre + "ss3/%s/%s/x_t" % (st, tag))) < tol
            assert rel_err(im["model_s2"].cpu().numpy(), r(pre + "ss3/%s/%s/model_s2" % (st, tag))) < tol


# ------------------------------------------------------------------------------------------------
# dynamic thresholding: exact order statistics (ref :416-425)
# ------------------------------------------------------------------------------------------------
def test_dynamic_thresholding_bit_exact(golden):
    ns = make_schedule("ddpm")
    for tag in "abcde":
        g = lambda k: golden.get("quantile", "quant/%s/%s" % (tag, k))
        p, mv = g("p_mv")
        dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding",
                           thresholding_max_val=float(mv), dynamic_thresholding_ratio=float(p))
        y = dpm.dynamic_thresholding_fn(torch.from_numpy(g("x0")).to(DEV), None)
        np.testing.assert_array_equal(y.cpu().numpy(), g("y"))          # vs torch.quantile in the reference
    # ties / plateaus / all-equal / negative zeros / n = 1, 2: against the oracle
    rng = np.random.default_rng(0)
    for shape, p in [((3, 1, 1, 1), 0.995), ((3, 1, 1, 2), 0.5), ((4, 1, 7, 9), 0.3), ((2, 3, 37, 41), 0.995),
                     ((2, 1, 100, 100), 0.999), ((5, 3, 64, 64), 0.0), ((2, 3, 96, 96), 0.75)]:
        x0 = (rng.standard_normal(shape) * 2.0).astype(F32)
        x0.reshape(shape[0], -1)[0, ::3] = np.float32(1.25)               # plateaus
        x0.reshape(shape[0], -1)[-1, :] = np.float32(-0.0)
        dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=p)
        y = dpm.dynamic_thresholding_fn(torch.from_numpy(x0).to(DEV), None)
        np.testing.assert_array_equal(y.cpu().numpy(), O.dynamic_threshold(x0, p, 1.0))


def test_dynamic_thresholding_large_samples(golden):
    """Samples beyond the LDS-resident path (config-3 sized rows, 3x256x256) take the multi-workgroup selection:
    still exact order statistics."""
    ns = make_schedule("ddpm")
    g = lambda k: golden.get("quantile", "quant/f/%s" % k)
    rng = np.random.default_rng(3)
    for shape in [(4, 3, 64, 64), (3, 3, 8, 8), (2, 1, 5, 7), (2, 3, 16, 16), (2, 2, 2, 2)]:
        rng.standard_normal(shape)                                   # replay the generator of make_golden.py
    x0 = (rng.standard_normal((2, 3, 256, 256)) * 2.5).astype(F32)
    dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding")
    y = dpm.dynamic_thresholding_fn(torch.from_numpy(x0).to(DEV), None).cpu().numpy()
    np.testing.assert_array_equal(y.reshape(2, -1)[:, :256], g("y_head"))        # vs torch.quantile in the reference
    np.testing.assert_array_equal(y, O.dynamic_threshold(x0, 0.995, 1.0))
    for shape, p in [((3, 1, 300, 200), 0.5), ((2, 3, 128, 128), 0.999), ((5, 1, 41000, 1), 0.25), ((1, 3, 512, 512), 1.0),
                     ((2, 1, 1, 50000), 0.0)]:
        x0 = (rng.standard_normal(shape) * 2.0).astype(F32)
        x0.reshape(shape[0], -1)[0, ::3] = np.float32(1.25)
        x0.reshape(shape[0], -1)[-1, 100:] = np.float32(-0.75)
        dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=p)
        y = dpm.dynamic_thresholding_fn(torch.from_numpy(x0).to(DEV), None)
        np.testing.assert_array_equal(y.cpu().numpy(), O.dynamic_threshold(x0, p, 1.0))


def test_dynamic_thresholding_topk_front_end():
    """Ratios close to 1 take the top-K front end of the select (per-thread maxima -> digit bound -> candidates);
    plateaus at the top overflow its candidate list and must fall back to the full histograms.  One workgroup per
    sample, clusters (small batches, large samples), ragged chunks, scalar (unaligned) rows: all exact."""
    ns = make_schedule("ddpm")
    rng = np.random.default_rng(11)
    cases = [((1024, 3, 64, 64), 0.995, None), ((40, 3, 64, 64), 0.995, 0.30), ((40, 3, 64, 64), 0.995, 0.55),
             ((3, 3, 64, 64), 0.999, 0.10), ((3, 3, 64, 64), 1.0, None), ((2, 3, 256, 256), 0.995, 0.002),
             ((2, 3, 256, 256), 0.995, 0.10), ((600, 1, 61, 67), 0.99, 0.05), ((600, 1, 50, 50), 0.98, None),
             ((2, 1, 333, 1001), 0.9995, 0.001), ((700, 1, 1, 97), 0.97, 0.2), ((5, 3, 64, 64), 0.76, None),
             # K = n - floor(p (n - 1)) around the front end's limit of 128, one workgroup per sample and clusters
             ((520, 3, 64, 64), 0.98967, None), ((520, 3, 64, 64), 0.98962, None), ((520, 3, 64, 64), 0.9895, 0.004),
             ((4, 3, 64, 64), 0.98967, None), ((4, 3, 64, 64), 0.9895, None), ((520, 1, 32, 64), 0.9380, None),
             ((520, 1, 32, 64), 0.9370, None)]
    for shape, p, top in cases:
        x0 = (rng.standard_normal(shape) * 2.0).astype(F32)
        rows = x0.reshape(shape[0], -1)
        if top is not None:          # a plateau holding the largest values of every second row
            n = rows.shape[1]
            idx = rng.permutation(n)[:max(1, int(top * n))]
            rows[::2, idx] = np.float32(9.5) * np.where(rng.random(idx.size) < 0.5, -1, 1).astype(F32)
        dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=p)
        y = dpm.dynamic_thresholding_fn(torch.from_numpy(x0).to(DEV), None)
        np.testing.assert_array_equal(y.cpu().numpy(), O.dynamic_threshold(x0, p, 1.0), err_msg=str((shape, p, top)))


def test_thresholding_one_element_per_lane_layout_sweeps_its_own_elements():
    """Round 6, found by tools/fuzz_gpu_thresh.py: samples whose length is not a multiple of 4 (or whose storage is not
    16-byte aligned) are walked one element per lane; when a wavefront's threads held more than four candidates each
    (K = n - floor(p (n - 1)) just under T / 4 = 128) the sweep that lists them read the wavefront's 16-byte rows -- other
    wavefronts' elements in this layout -- and returned the order statistic a few ranks off (39 of 390 samples at
    [130, 4095], p = 0.97).  One workgroup per sample and clusters, against torch.quantile itself (ref :420)."""
    ns = make_schedule("ddpm")
    for (B, per, offset) in [(130, 4095, 0), (130, 4092, 1), (130, 3071, 0), (130, 11210, 0), (130, 2049, 0), (3, 49153, 0),
                             (5, 98305, 1)]:
        for K in (60, 100, 110, 120, 127, 128, 129):
            p = (per - K + 0.3) / (per - 1.0)
            x0 = torch.from_numpy((np.random.default_rng(per + K).standard_normal((B, per))).astype(F32))
            flat = torch.empty(B * per + 8, dtype=torch.float32, device=DEV)
            xg = flat[offset:offset + B * per].reshape(B, per)
            xg.copy_(x0)
            dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=p,
                               thresholding_max_val=0.5)
            got = dpm.dynamic_thresholding_fn(xg, None).cpu()
            s = torch.maximum(torch.quantile(x0.abs(), p, dim=1), torch.tensor(0.5))[:, None]
            assert torch.equal(got, torch.clamp(x0, -s, s) / s), (B, per, offset, K)
            np.testing.assert_array_equal(got.numpy(), O.dynamic_threshold(x0.numpy(), p, 0.5), err_msg=str((B, per, offset, K)))


def test_fuzz_slice_of_thresholding_against_torch_quantile(monkeypatch, capsys):
    """500 random cases of tools/fuzz_gpu_thresh.py: dynamic_thresholding_fn on the GPU against the reference's own three
    lines (torch.quantile on the CPU, maximum, clamp / divide) -- sample sizes around the kernels' boundaries, any ratio,
    tie-heavy values, outliers, fp32 and fp64 -- bit for bit (recorded: profiles/r06_fuzz_gpu_thresh.json)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gpu_thresh as FT
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu_thresh.py", "--cases", "500", "--seed", "3"])
    n_bad = FT.main()
    out = capsys.readouterr().out
    assert n_bad == 0, out[-3000:]
    assert '"cases": 500' in out


@pytest.mark.lab
def test_cluster_single_exchange_route_and_its_fallback():
    """Clusters (k workgroups per sample) first try to settle a sample with one exchange of per-chunk candidates
    (cluster_select_once).  That is exact by construction or declared failed inside the kernel, in which case the
    cluster takes the general route: (i) samples whose largest values all sit in ONE chunk must still be exact,
    (ii) switching the one-hop route off gives bit-identical results, (iii) the workspace is all zero after every launch
    (it is zero-filled once, never per launch)."""
    ns = make_schedule("ddpm")
    rng = np.random.default_rng(23)
    cases = [((32, 3, 64, 64), 0.995), ((6, 3, 64, 64), 0.999), ((4, 3, 256, 256), 0.995), ((3, 3, 200, 160), 0.99),
             ((2, 3, 256, 256), 0.9995), ((40, 1, 96, 96), 0.995)]
    for shape, p in cases:
        for mode in ("random", "one_chunk", "two_chunks", "ties"):
            x0 = (rng.standard_normal(shape) * 1.5).astype(F32)
            rows = x0.reshape(shape[0], -1)
            n = rows.shape[1]
            K = max(2, int(n - np.floor(np.float32(p) * np.float32(n - 1))))
            if mode == "one_chunk":          # the K largest values of every sample in its first 2K elements
                rows[:, :2 * K] = (np.abs(rng.standard_normal((shape[0], 2 * K))) + 8.0).astype(F32)
            elif mode == "two_chunks":       # half of them at the very end of the sample
                rows[:, :K] = (np.abs(rng.standard_normal((shape[0], K))) + 8.0).astype(F32)
                rows[:, -K:] = -(np.abs(rng.standard_normal((shape[0], K))) + 8.0).astype(F32)
            elif mode == "ties":             # the wanted order statistics are a long run of equal values
                rows[:, rng.permutation(n)[:3 * K]] = np.float32(7.25)
            want = O.dynamic_threshold(x0, p, 1.0)
            dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=p)
            xg = torch.from_numpy(x0).to(DEV)
            y = dpm.dynamic_thresholding_fn(xg, None)
            np.testing.assert_array_equal(y.cpu().numpy(), want, err_msg=str((shape, p, mode)))
            L.lib.dpm_tuning_set(L.TUNE_CLUSTER_ONE_HOP, 0)
            try:
                y2 = dpm.dynamic_thresholding_fn(xg, None)
            finally:
                L.lib.dpm_tuning_set(L.TUNE_CLUSTER_ONE_HOP, 1)
            assert torch.equal(y, y2), (shape, p, mode)
            torch.cuda.synchronize()
            for ws in S._WS_CACHE.values():
                assert not bool(ws.any()), ("workspace not left zero-filled", shape, p, mode)


def test_thresholded_sampling_random_sweep():
    """seeded random sweep of sample() with dynamic thresholding -- batch / sample sizes (one workgroup per sample,
    clusters, ragged and unaligned rows), ratio (top-K front end and full histograms), max_val, order, steps --
    against the oracle: bit-identical.  DPM_THR_SWEEP=<count> extends the run (round 2's kernel: 20 000 configurations,
    no difference)"""
    rng = np.random.default_rng(7)
    done = 0
    total = int(os.environ.get("DPM_THR_SWEEP", "250"))
    # DPM_THR_SWEEP_FAULT=1|2|3 runs the sweep with forced cluster faults (every wait gives up at once / workgroup 1 of
    # every cluster out of the protocol, marked or not; short timeout), DPM_THR_SWEEP_ONE_HOP=0 on the general route only:
    # the in-kernel recovery must reproduce the oracle's bits on every configuration
    fault = int(os.environ.get("DPM_THR_SWEEP_FAULT", "0"))
    one_hop = int(os.environ.get("DPM_THR_SWEEP_ONE_HOP", "1"))
    if not fault and one_hop == 1:          # the product library's own behaviour: no knob touched
        _thr_sweep(rng, total)
        return
    L.require_lab("a forced-fault / general-route sweep")
    if fault:
        L.check(L.lib.dpm_tuning_set(L.TUNE_THR_DEBUG_FAULT, fault))
        L.check(L.lib.dpm_tuning_set(L.TUNE_THR_SPIN_LIMIT, 32))
    L.check(L.lib.dpm_tuning_set(L.TUNE_CLUSTER_ONE_HOP, one_hop))
    try:
        _thr_sweep(rng, total)
    finally:
        L.lib.dpm_tuning_set(L.TUNE_CLUSTER_ONE_HOP, 1)
        if fault:
            L.lib.dpm_tuning_set(L.TUNE_THR_DEBUG_FAULT, 0)
            L.lib.dpm_tuning_set(L.TUNE_THR_SPIN_LIMIT, 4096)


def _thr_sweep(rng, total):
    done = 0
    while done < total:
        B = int(rng.choice([1, 2, 3, 5, 8, 17, 40, 130, 600]))
        Cc, H, W = int(rng.integers(1, 4)), int(rng.choice([4, 7, 16, 31, 32, 64, 96])), int(rng.choice([4, 9, 16, 32, 64, 128]))
        if done % 40 == 39:                  # clusters of two that walk several samples (600 workgroups > the chip holds)
            B, Cc, H, W = 300, 1, 128, 128
        if B * Cc * H * W > 3e6 and B != 300:
            continue
        p, mv = float(rng.choice([0.5, 0.9, 0.95, 0.99, 0.995, 0.999, 1.0])), float(rng.choice([0.5, 1.0, 2.0]))
        sname = str(rng.choice(["ddpm", "sd"]))
        order = int(rng.integers(1, 4))
        # (DPM_THR_SWEEP_STEPS raises the trajectory length: more stages that select with a predicted bound)
        steps = int(rng.integers(order, int(os.environ.get("DPM_THR_SWEEP_STEPS", "8"))))
        x = (rng.standard_normal((B, Cc, H, W)) * float(rng.choice([0.3, 1.0, 2.0]))).astype(F32)
        scale = F32(rng.choice([0.5, 0.9, 1.3]))
        ns, osch = make_schedule(sname), TO.make_schedule(sname)
        dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * float(scale), ns), ns, correcting_x0_fn="dynamic_thresholding",
                           thresholding_max_val=mv, dynamic_thresholding_ratio=p)
        got = dpm.sample(torch.from_numpy(x).to(DEV), steps=steps, order=order).cpu().numpy()
        sol = O.Solver(O.wrap_model(lambda xx, t: (xx * scale).astype(F32), osch), osch,
                       correcting_x0_fn="dynamic_thresholding", thresholding_max_val=mv, dynamic_thresholding_ratio=p)
        np.testing.assert_array_equal(got, sol.sample(x, steps=steps, order=order), err_msg=str((B, Cc, H, W, p, mv, sname, steps, order)))
        #  the workspace of a clustered shape is left all zero by every launch -- also under the forced
        # faults of DPM_THR_SWEEP_FAULT with the predicted route (thr_hint) active, where a workgroup that gave up may have
        # rewritten the hint words a late peer predicts from
        for fr in dpm._fast.values():
            if getattr(fr, "ws", None) is not None:
                assert not bool(fr.ws.any()), ("workspace not left zero-filled", B, Cc, H, W, p, mv, sname, steps, order)
        for ws in S._WS_CACHE.values():
            assert not bool(ws.any()), ("shared workspace not left zero-filled", B, Cc, H, W, p, mv, sname, steps, order)
        done += 1


def _unaligned(t):
    """the same values as a view with a 4-byte storage offset: every launch that reads it takes the one-element-per-lane /
    catch-all kernels (16-byte accesses are not legal)"""
    buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
    buf[1:] = t.reshape(-1)
    v = buf[1:].reshape(t.shape)
    assert v.data_ptr() % 16 != 0
    return v


@pytest.mark.parametrize("guided", [False, True])
@pytest.mark.parametrize("mt", ["v", "x_start", "score"])
def test_thresholding_with_other_parameterisations_specialised_kernel_equals_the_catch_all(mt, guided):
    """Dynamic thresholding behind an x_start / v / score network takes the specialised thresholding kernel with the prologue
    chosen per sample at run time (HOT 3; round 4: 14-19 % per stage against the catch-all) -- unguided and under
    classifier-free guidance, forms LIN1 / TWO / MS3, one workgroup per sample and clusters.  Same bits as the catch-all
    kernel (forced by a network output that is not 16-byte aligned), and the oracle's values."""
    ns, osch = make_schedule("ddpm"), TO.make_schedule("ddpm")
    rng = np.random.default_rng(77)
    for shape in [(32, 3, 64, 64), (600, 3, 16, 16), (3, 3, 128, 128)]:
        B = shape[0]
        x = (rng.standard_normal(shape) * 1.5).astype(F32)
        cnp = (0.5 + np.arange(B, dtype=F32) / B).astype(F32)
        c = torch.from_numpy(cnp).to(DEV)

        def solver(unaligned):
            wrap = _unaligned if unaligned else (lambda o: o)
            if guided:
                net = lambda xx, t, cc: wrap(xx * (0.4 + 0.1 * cc.reshape(-1, 1, 1, 1)))
                fn = D.model_wrapper(net, ns, model_type=mt, guidance_type="classifier-free", condition=c,
                                     unconditional_condition=torch.zeros_like(c), guidance_scale=2.5)
            else:
                fn = D.model_wrapper(lambda xx, t: wrap(xx * 0.5), ns, model_type=mt)
            return D.DPM_Solver(fn, ns, correcting_x0_fn="dynamic_thresholding")
        if guided:
            ofn = O.wrap_model(lambda xx, t, cc: (xx * (F32(0.4) + F32(0.1) * cc.reshape(-1, 1, 1, 1))).astype(F32), osch, model_type=mt,
                               guidance_type="classifier-free", condition=cnp, unconditional_condition=np.zeros_like(cnp),
                               guidance_scale=2.5)
        else:
            ofn = O.wrap_model(lambda xx, t: (xx * F32(0.5)).astype(F32), osch, model_type=mt)
        osol = O.Solver(ofn, osch, correcting_x0_fn="dynamic_thresholding")
        for order in (1, 2, 3):
            xt = torch.from_numpy(x).to(DEV)
            got = solver(False).sample(xt, steps=6, order=order)
            ref = solver(True).sample(xt, steps=6, order=order)
            assert torch.equal(got, ref), (mt, guided, shape, order)
            want = osol.sample(x, steps=6, order=order)
            assert rel_err(got.cpu().numpy(), want) < TOL, (mt, guided, shape, order)


def test_cfg3_sized_thresholded_sampling():
    """[4,3,256,256] pixel-space 2M++ with dynamic thresholding and CFG: the large-sample path inside sample()."""
    case = dict(C.E2E_BY_NAME["cfg5_thresh"], shape=(4, 3, 256, 256), steps=10, model="cond",
                guidance_type="classifier-free", guidance_scale=3.0)
    xo, _ = TO.run_oracle_case(case)
    xf, _ = run_case(case, DEV)
    assert rel_err(xf.cpu().numpy(), xo) < TOL


# ------------------------------------------------------------------------------------------------
# add_noise, model evaluation methods, adaptive
# ------------------------------------------------------------------------------------------------
def test_add_noise(golden):
    for sname in ["sd", "vp_linear"]:
        dpm = D.DPM_Solver(lambda x, t: x, make_schedule(sname))
        for tag in ["one", "three"]:
            g = lambda k: golden.get("add_noise", "addnoise/%s/%s/%s" % (sname, tag, k))
            y = dpm.add_noise(tt(g("x"), DEV), tt(g("t"), DEV), noise=tt(g("noise"), DEV))
            assert y.shape == g("y").shape
            assert rel_err(y.cpu().numpy(), g("y")) < 2e-6


def test_model_evaluation_methods():
    for name in ["mt_v", "cfg_ms2", "clsg_ms2", "cfg5_thresh_small", "mt_xstart_noise"]:
        case = C.E2E_BY_NAME[name]
        dpm = build_solver(case, DEV)
        osol = TO.build_oracle_solver(case)
        xn = C.x_T_for(case)
        x = tt(xn, DEV)
        t = torch.tensor([0.6172], device=DEV)
        want_eps = osol.noise_pred(xn, F32(0.6172))
        want_x0 = osol.data_pred(xn, F32(0.6172))
        assert rel_err(dpm.noise_prediction_fn(x, t).cpu().numpy(), want_eps) < 2e-6
        assert rel_err(dpm.data_prediction_fn(x, t).cpu().numpy(), want_x0) < 2e-6
        assert rel_err(dpm._wrapped(x, t.expand(x.shape[0])).cpu().numpy(), want_eps) < 2e-6


@pytest.mark.parametrize("name,sname,order,algo", [("a12", "vp_linear", 2, "dpmsolver"), ("a23", "vp_linear", 3, "dpmsolver"),
                                                   ("a23pp", "sd", 3, "dpmsolver++")])
def test_adaptive(golden, capsys, name, sname, order, algo):
    ns = make_schedule(sname)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns), ns, algorithm_type=algo)
    g = lambda k: golden.get("adaptive", "adaptive/%s/%s" % (name, k))
    xf = dpm.sample(tt(g("x"), DEV), method="adaptive", order=order, t_end=1e-3)
    out = capsys.readouterr().out
    assert out.strip() == "adaptive solver nfe %d" % int(g("nfe"))      # same accept/reject sequence as the reference
    assert rel_err(xf.cpu().numpy(), g("final")) < TOL                  # north-star bound (measured: <= 3.2e-7)


@pytest.mark.parametrize("name,sname,order,algo", [("a12", "vp_linear", 2, "dpmsolver"), ("a23", "vp_linear", 3, "dpmsolver"),
                                                   ("a23pp", "sd", 3, "dpmsolver++")])
def test_adaptive_controller_on_the_device(golden, capsys, monkeypatch, name, sname, order, algo):
    """the adaptive solver's accept / reject and step-size controller runs on the device (dpm_adaptive_*).
    (i) nothing in the run reads a tensor back (`.item()` / `.cpu()` / `.tolist()` raise while it runs); (ii) same NFE and
    result as the reference goldens; (iii) agrees with the host-side control loop (the reference's structure);
    (iv) half-precision states work (the host loop had no bf16 error norm)."""
    ns = make_schedule(sname)
    g = lambda k: golden.get("adaptive", "adaptive/%s/%s" % (name, k))
    mk = lambda **kw: D.DPM_Solver(D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns), ns, algorithm_type=algo, **kw)
    x = tt(g("x"), DEV)
    dpm = mk()
    assert dpm.adaptive_on_device
    dpm.sample(x, method="adaptive", order=order, t_end=1e-3)            # warm-up: handle, buffers
    capsys.readouterr()

    def forbidden(*a, **k):
        raise AssertionError("device -> host read inside the adaptive loop")
    with monkeypatch.context() as m:
        for meth in ("item", "cpu", "tolist", "numpy"):
            m.setattr(torch.Tensor, meth, forbidden)
        xf = dpm.sample(x, method="adaptive", order=order, t_end=1e-3)
    out = capsys.readouterr().out
    assert out.strip() == "adaptive solver nfe %d" % int(g("nfe"))
    assert rel_err(xf.cpu().numpy(), g("final")) < TOL
    host = mk()
    host.adaptive_on_device = False
    xh = host.sample(x, method="adaptive", order=order, t_end=1e-3)
    assert capsys.readouterr().out.strip() == "adaptive solver nfe %d" % int(g("nfe"))
    assert rel_err(xf.cpu().numpy(), xh.cpu().numpy()) < 2e-6
    for sdt in (torch.float16, torch.bfloat16):
        xs = mk(state_dtype=sdt).sample(x.to(sdt), method="adaptive", order=order, t_end=1e-3)
        assert xs.dtype == sdt and torch.isfinite(xs.float()).all()
        assert rel_err(xs.float().cpu().numpy(), g("final")) < (2e-2 if sdt == torch.float16 else 1.5e-1)
    capsys.readouterr()


def test_adaptive_with_guidance_and_other_parameterisations(capsys):
    """the device-side controller fills the prologue scalars too (alpha, sigma at every evaluation time, the classifier
    term's scale): classifier-free guidance, v-prediction and x_start-prediction against the host-side loop"""
    ns = make_schedule("sd")
    rng = np.random.default_rng(41)
    x = torch.from_numpy(rng.standard_normal((3, 4, 16, 16)).astype(F32)).to(DEV)
    c = torch.tensor([0.5, 1.0, 1.5], device=DEV)
    cases = [dict(model_type="v"), dict(model_type="x_start"),
             dict(guidance_type="classifier-free", condition=c, unconditional_condition=torch.zeros_like(c), guidance_scale=2.5)]
    for kw in cases:
        net = (lambda xx, t, cc: xx * (0.4 + 0.1 * cc.reshape(-1, 1, 1, 1))) if "condition" in kw else (lambda xx, t: xx * 0.5)
        for algo in ("dpmsolver++", "dpmsolver"):
            for order in (2, 3):
                res = []
                for on_dev in (True, False):
                    dpm = D.DPM_Solver(D.model_wrapper(net, ns, **kw), ns, algorithm_type=algo)
                    dpm.adaptive_on_device = on_dev
                    res.append(dpm.sample(x, method="adaptive", order=order, t_end=5e-3, solver_type="taylor" if order == 3 else "dpmsolver"))
                o = capsys.readouterr().out.strip().splitlines()
                assert o[0] == o[1], (kw.keys(), algo, order, o)                # same NFE
                assert rel_err(res[0].cpu().numpy(), res[1].cpu().numpy()) < 5e-6, (list(kw), algo, order)


def test_adaptive_captured_into_a_graph(capsys):
    """DPM_Solver.capture accepts method='adaptive' now: a fixed number of iterations is recorded, those after the device
    reached t_end are no-ops"""
    ns = make_schedule("vp_linear")
    rng = np.random.default_rng(42)
    x = torch.from_numpy(rng.standard_normal((2, 3, 16, 16)).astype(F32)).to(DEV)
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns), ns, algorithm_type="dpmsolver")
    want = dpm.sample(x, method="adaptive", order=2, t_end=1e-3)
    nfe = int(capsys.readouterr().out.strip().split()[-1])
    dpm.adaptive_max_iterations = nfe // 2 + 6
    g = dpm.capture(x, method="adaptive", order=2, t_end=1e-3)
    capsys.readouterr()
    for _ in range(2):
        assert torch.equal(g(x), want)
    x2 = x * 1.5
    dpm.adaptive_max_iterations = None
    assert torch.equal(g(x2), dpm.sample(x2, method="adaptive", order=2, t_end=1e-3))


def test_callbacks(golden):
    case = C.E2E_BY_NAME["cfg1_small"]
    ns = make_schedule("sd")
    x = tt(C.x_T_for(case), DEV)
    mask = torch.from_numpy(golden.get("callbacks", "cb/mask")).to(DEV)
    cxt = lambda xt, t, step: xt * mask + (1.0 - mask) * (0.25 * step)
    cx0 = lambda x0, t: torch.clamp(x0, -1.5, 1.5)
    fn = D.model_wrapper(lambda xx, t: C.model_half(xx, t), ns)
    for tag, kw in [("xt", dict(correcting_xt_fn=cxt)), ("x0", dict(correcting_x0_fn=cx0)),
                    ("both", dict(correcting_xt_fn=cxt, correcting_x0_fn=cx0))]:
        for method, order, steps in [("multistep", 2, 8), ("singlestep", 3, 8)]:
            dpm = D.DPM_Solver(fn, ns, **kw)
            xf, inter = dpm.sample(x, steps=steps, order=order, method=method, denoise_to_zero=True, return_intermediate=True)
            pre = "cb/%s/%s/" % (tag, method)
            assert rel_err(xf.cpu().numpy(), golden.get("callbacks", pre + "final")) < TOL
            ri = golden.get("callbacks", pre + "intermediates")
            for i, v in enumerate(inter):
                assert rel_err(v.cpu().numpy(), ri[i]) < TOL


# ------------------------------------------------------------------------------------------------
# dtypes, ragged sizes, unaligned views, empty batch
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sdt,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_low_precision_state(sdt, tol):
    """fp16 / bf16 state (BASELINE cfg2's bandwidth mode): fp32 math, one rounding per stored tensor.  The
    reference itself cannot hold a half-precision state with a discrete schedule (it promotes), so the
    comparison is against the fp32 oracle with the half-precision tolerance stated here."""
    case = dict(C.E2E_BY_NAME["cfg1_small"], shape=(4, 4, 64, 64))
    xo, _ = TO.run_oracle_case(case)
    dpm = build_solver(case, DEV, state_dtype=sdt)
    x = tt(C.x_T_for(case), DEV).to(sdt)
    xf = dpm.sample(x, **sample_kwargs(case, False))
    assert xf.dtype == sdt
    assert rel_err(xf.float().cpu().numpy(), xo) < tol


@pytest.mark.parametrize("edt", [torch.float16, torch.bfloat16])
def test_half_eps_fp32_state(edt):
    """SD under autocast: the UNet returns fp16 eps, the solver state stays fp32."""
    case = dict(C.E2E_BY_NAME["cfg1_small"], shape=(2, 4, 32, 32), model="half")
    ns = make_schedule("sd")
    net = lambda x, t: (x * 0.5).to(edt)
    onet = lambda x, t: torch.from_numpy(x * F32(0.5)).to(edt).float().numpy()      # same rounding of eps
    dpm = D.DPM_Solver(D.model_wrapper(net, ns), ns)
    osch = TO.make_schedule("sd")
    xo = O.Solver(O.wrap_model(onet, osch), osch).sample(C.x_T_for(case), steps=20)
    xf = dpm.sample(tt(C.x_T_for(case), DEV), steps=20)
    assert xf.dtype == torch.float32
    assert rel_err(xf.cpu().numpy(), xo) < TOL


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (3, 1, 1, 7), (2, 3, 5, 7), (1, 4, 64, 64), (5, 3, 33, 31)])
def test_ragged_sizes_and_unaligned_views(shape):
    case = dict(C.E2E_BY_NAME["ms3"], shape=shape, model="tdep")
    xo, _ = TO.run_oracle_case(case)
    xf, _ = run_case(case, DEV)
    assert rel_err(xf.cpu().numpy(), xo) < TOL
    # state handed over as a view with a 4-byte storage offset: the launcher must take the scalar kernel
    dpm = build_solver(case, DEV)
    n = int(np.prod(shape))
    buf = torch.zeros(n + 1, device=DEV)
    buf[1:] = tt(C.x_T_for(case), DEV).reshape(-1)
    xv = buf[1:].reshape(shape)
    assert xv.data_ptr() % 16 != 0
    xf2 = dpm.sample(xv, **sample_kwargs(case, False))
    np.testing.assert_array_equal(xf2.cpu().numpy(), xf.cpu().numpy())


@pytest.mark.lab
def test_workgroup_size_of_the_streaming_kernel_does_not_change_a_bit():
    """The streaming stage kernel runs with 256 or 512 threads per workgroup (every 256-lane group takes tiles of its own; by
    default 512 when that leaves two workgroups per CU -- [128,4,64,64] requests and larger).  The lanes do the same work on the
    same tiles: every workgroup size forced through DPM_TUNE_BLOCK_THREADS gives the bits of the 256-thread launch -- fp32 / fp16 / bf16 states, half outputs next to an fp32 state (split layout + lane exchange),
    ragged tiles, classifier-free guidance with the duplicate store, a mask blend, singlestep forms, a BASELINE-sized
    request where the default picks 512."""
    ns = make_schedule("sd")
    g = torch.Generator().manual_seed(77)
    cond = torch.ones(1, device=DEV)
    xb, known, noise = (torch.randn((4, 4, 64, 64), generator=g).to(DEV) for _ in range(3))
    mask = (torch.rand(64, 64, generator=g) > 0.5).float().to(DEV)

    def runs():
        out = []
        for shape, sdt, edt in [((8, 4, 64, 64), torch.float32, None), ((8, 4, 64, 64), torch.float16, None),
                                ((8, 4, 64, 64), torch.bfloat16, None), ((8, 4, 64, 64), torch.float32, torch.float16),
                                ((3, 3, 33, 35), torch.float32, None), ((5, 4, 40, 40), torch.float16, None),
                                ((256, 4, 64, 64), torch.float16, None)]:
            x = torch.randn(shape, generator=torch.Generator().manual_seed(5)).to(DEV, sdt)
            net = lambda xx, t: (xx.float() * 0.5).to(edt or xx.dtype)
            kw = {} if sdt is torch.float32 else {"state_dtype": sdt}
            dpm = D.DPM_Solver(D.model_wrapper(net, ns), ns, **kw)
            out.append(dpm.sample(x, steps=6, order=2))
            out.append(dpm.sample(x, steps=6, order=3, method="singlestep"))
            if shape[0] <= 8:
                c = cond.expand(shape[0])
                cfg = D.DPM_Solver(D.model_wrapper(lambda xx, t, cc: (xx.float() * (0.3 + 0.2 * cc.reshape(-1, 1, 1, 1))).to(edt or xx.dtype),
                                                   ns, guidance_type="classifier-free", condition=c,
                                                   unconditional_condition=c * 0, guidance_scale=7.5), ns, **kw)
                out.append(cfg.sample(x, steps=6, order=2))
        edit = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns,
                            correcting_xt_fn=D.MaskBlend(ns, mask, x0=known, noise=noise))
        out.append(edit.sample(xb, steps=6, order=2))
        # requests in flight: the fused multi-request kernel takes the same two workgroup sizes
        for sdt in (torch.float16, torch.float32):
            fl = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns, **({} if sdt is torch.float32 else {"state_dtype": sdt}))
            out += fl.sample_requests([(xb * s).to(sdt) for s in (1.0, 0.5, 2.0, 1.5, 0.25)], steps=6, order=2)
        return out
    with _Tuned(block_threads=256):
        want = runs()
    for bt in (512, 0):
        with _Tuned(block_threads=bt):
            got = runs()
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), (bt, i)


def test_empty_batch():
    ns = make_schedule("sd")
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns)
    out = dpm.sample(torch.zeros(0, 4, 8, 8, device=DEV), steps=5)
    assert out.shape == (0, 4, 8, 8)


def test_cpu_tensor_rejected():
    ns = make_schedule("sd")
    dpm = D.DPM_Solver(D.model_wrapper(lambda x, t: x, ns), ns)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dpm.sample(torch.zeros(2, 4, 8, 8), steps=5)


# ------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties
# ------------------------------------------------------------------------------------------------
def _sd_solver(model=lambda x, t: x, **kw):
    ns = make_schedule("sd")
    return D.DPM_Solver(D.model_wrapper(model, ns), ns, **kw)


@pytest.mark.parametrize("sdt", [torch.float16, torch.float32])
def test_cfg2_full_size_shard_invariance_and_linearity(sdt):
    """[256,4,64,64], DPM-Solver++(2M), 20 steps.  (i) batch-shard invariance: sampling the whole batch equals
    sampling its shards (what the 8-GPU run relies on), bit for bit; (ii) with a linear frozen network the
    solver is linear in x_T: scaling by 2 is exact in binary floating point; (iii) a slice matches the oracle."""
    rng = np.random.default_rng(5)
    xn = rng.standard_normal((256, 4, 64, 64)).astype(F32)
    x = torch.from_numpy(xn).to(DEV).to(sdt)
    dpm = _sd_solver(state_dtype=sdt)
    full = dpm.sample(x, steps=20)
    assert full.dtype == sdt
    parts = torch.cat([dpm.sample(x[i:i + 64], steps=20) for i in range(0, 256, 64)])
    assert torch.equal(full, parts)
    twice = dpm.sample(x * 2, steps=20)
    if sdt == torch.float32:
        assert torch.equal(twice, full * 2)
    else:   # fp16 subnormals (|v| < 6.1e-5) round differently after doubling; the first stage divides by
        # alpha_T = 0.068, so a 2^-24 difference can grow to a few 1e-6 absolute by the end
        assert float((twice.float() - 2 * full.float()).abs().max()) <= 4e-6
    osch = TO.make_schedule("sd")
    xs = x[:2].float().cpu().numpy()
    xo = O.Solver(O.wrap_model(lambda a, t: a, osch), osch).sample(xs, steps=20)
    assert rel_err(full[:2].float().cpu().numpy(), xo) < (TOL if sdt == torch.float32 else 4e-3)


def test_cfg3_full_size_shard_invariance():
    """[64,3,256,256] fp32, DPM-Solver-3 singlestep, 15 NFE, CFG 7.5 (network batch 128)."""
    case = dict(C.E2E_BY_NAME["cfg3_dpmsolver"], shape=(64, 3, 256, 256))
    rng = np.random.default_rng(6)
    x = torch.from_numpy(rng.standard_normal(case["shape"]).astype(F32)).to(DEV)
    dpm = build_solver(case, DEV)
    full = dpm.sample(x, **sample_kwargs(case, False))
    case8 = dict(case, shape=(8, 3, 256, 256))
    dpm8 = build_solver(case8, DEV)
    part = dpm8.sample(x[8:16], **sample_kwargs(case, False))
    assert torch.equal(full[8:16], part)
    small = dict(case, shape=(2, 3, 256, 256))
    xo, _ = TO.build_oracle_solver(small).sample(x[:2].cpu().numpy(), **sample_kwargs(case, True))
    assert rel_err(full[:2].cpu().numpy(), xo) < TOL


def test_cfg5_full_size_thresholding():
    """[32,3,64,64] pixel space, 2M++ with dynamic thresholding, 25 steps: every sample equals the oracle's."""
    case = dict(C.E2E_BY_NAME["cfg5_thresh"], shape=(32, 3, 64, 64))
    xo, _ = TO.run_oracle_case(case)
    xf, _ = run_case(case, DEV)
    assert rel_err(xf.cpu().numpy(), xo) < TOL
    assert float(xf.abs().max()) <= 1.5


# ------------------------------------------------------------------------------------------------
# half-precision states at BASELINE sizes: BIT-EQUAL to the numpy double of the kernel driven by the same plan.
# The double (tests/kernel_double.py) does the stage arithmetic in fp32 and rounds every stored tensor once to the
# state dtype, like the kernel; the reference cannot hold a half state with a discrete schedule, so this -- not a
# loose tolerance against an fp32 run -- is what pins the fp16 / bf16 configurations (bench.py's headline included).
# ------------------------------------------------------------------------------------------------
def _double_on_cpu(monkeypatch, fn):
    with monkeypatch.context() as m:
        install_cpu_double(m, S, D)
        return fn()


@pytest.mark.parametrize("sdt", [torch.float16, torch.bfloat16])
def test_cfg2_full_size_half_state_bit_equal_to_double(sdt, monkeypatch):
    """[256,4,64,64], DPM-Solver++(2M), 20 steps, model_fn = x (BASELINE configs[1]), fp16 / bf16 state"""
    rng = np.random.default_rng(15)
    xc = torch.from_numpy(rng.standard_normal((256, 4, 64, 64)).astype(F32)).to(sdt)
    got = _sd_solver(state_dtype=sdt).sample(xc.to(DEV), steps=20)
    assert got.dtype == sdt
    want = _double_on_cpu(monkeypatch, lambda: _sd_solver(state_dtype=sdt).sample(xc, steps=20))
    assert want.dtype == sdt and not want.is_cuda
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))
    # and a network whose output is NOT the state: frozen eps in the state dtype (what bench.py times)
    eps = torch.from_numpy(rng.standard_normal((256, 4, 64, 64)).astype(F32)).to(sdt)
    epsd = eps.to(DEV)
    got = _sd_solver(lambda x, t: epsd, state_dtype=sdt).sample(xc.to(DEV), steps=20)
    want = _double_on_cpu(monkeypatch, lambda: _sd_solver(lambda x, t: eps, state_dtype=sdt).sample(xc, steps=20))
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("sdt", [torch.float16, torch.bfloat16])
def test_cfg5_full_size_half_state_bit_equal_to_double(sdt, monkeypatch):
    """[32,3,64,64], 2M++ with dynamic thresholding, 25 steps (BASELINE configs[4]), fp16 / bf16 state: the threshold is
    selected on the fp32 x0 values, every stored tensor is rounded once"""
    case = dict(C.E2E_BY_NAME["cfg5_thresh"], shape=(32, 3, 64, 64))
    xc = torch.from_numpy(C.x_T_for(case)).to(sdt)
    kw = sample_kwargs(case, False)
    got = build_solver(case, DEV, state_dtype=sdt).sample(xc.to(DEV), **kw)
    want = _double_on_cpu(monkeypatch, lambda: build_solver(case, "cpu", state_dtype=sdt).sample(xc, **kw))
    assert got.dtype == sdt
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("sdt", [torch.float16, torch.bfloat16])
def test_cfg3_full_size_half_state_bit_equal_to_double(sdt, monkeypatch):
    """[64,3,256,256], DPM-Solver-3 singlestep, 15 NFE, CFG 7.5 (BASELINE configs[2]) with a half state: the full-size
    run on the GPU, the double on an 8-sample shard of it (the path is shard-invariant, see above)"""
    case = dict(C.E2E_BY_NAME["cfg3_dpmsolver"], shape=(64, 3, 256, 256))
    rng = np.random.default_rng(16)
    xc = torch.from_numpy(rng.standard_normal(case["shape"]).astype(F32)).to(sdt)
    kw = sample_kwargs(case, False)
    got = build_solver(case, DEV, state_dtype=sdt).sample(xc.to(DEV), **kw)
    small = dict(case, shape=(8, 3, 256, 256))
    want = _double_on_cpu(monkeypatch, lambda: build_solver(small, "cpu", state_dtype=sdt).sample(xc[16:24], **kw))
    assert torch.equal(got[16:24].cpu().view(torch.int16), want.view(torch.int16))


# ------------------------------------------------------------------------------------------------
# the native sample loop of the C ABI (dpm_plan_run) == the Python loop
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(8, 4, 64, 64), (3, 3, 37, 41), (5, 4, 64, 65)])      # whole tiles / ragged / 2.5 tiles
@pytest.mark.parametrize("method,order,steps", [("multistep", 2, 20), ("multistep", 3, 12), ("singlestep", 3, 14)])
def test_plan_run_native_loop_matches_python_loop(method, order, steps, shape):
    ns = make_schedule("sd")
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV)
    eps = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV)       # frozen network output
    dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: eps, ns), ns)
    want = dpm.sample(x, steps=steps, order=order, method=method, solver_type="taylor")
    plan = dpm._get_plan(method=method, order=order, steps=steps, skip_type="time_uniform", solver_type="taylor",
                         lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    xb = [x.clone(), torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)]
    hb = [torch.empty_like(x) for _ in range(3)]
    rb = L.RunBuffers()
    for i in range(4):
        rb.xbuf[i] = xb[i].data_ptr()
    for i in range(3):
        rb.hist[i] = hb[i].data_ptr()
    rb.e0 = eps.data_ptr()
    rb.n, rb.batch, rb.state_dtype, rb.eps_dtype = x.numel(), shape[0], L.DTYPE_F32, L.DTYPE_F32
    res = C_.c_int(-1)
    L.check(L.lib.dpm_plan_run(plan.handle, C_.byref(rb), None, None,
                               C_.c_void_p(torch.cuda.current_stream().cuda_stream), C_.byref(res)))
    torch.cuda.synchronize()
    assert torch.equal(xb[res.value], want)
    assert torch.equal(xb[0], x)                                   # the caller's x_T is never written
    # with durations (dpm_plan_run_multi over ONE request): same result, one kernel-only duration per stage
    ms = (C_.c_float * len(plan.stages))()
    L.check(L.lib.dpm_plan_run_multi(plan.handle, C_.byref(rb), 1, C_.c_void_p(torch.cuda.current_stream().cuda_stream),
                                     ms, C_.byref(res)))
    assert torch.equal(xb[res.value], want)
    # one launch in several thousand shows a 50-85 ms start -> stop interval (profiles/r02_stall.md): judge the typical one
    assert all(v > 0.0 for v in ms) and 0.0 < float(np.median(list(ms))) < 5.0, list(ms)


@pytest.mark.parametrize("sdt", [torch.float16, torch.bfloat16])
def test_half_precision_stores_round_to_nearest_even(sdt):
    """the packed conversions on the store path (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32) against torch's rounding, ties and
    subnormals included: one first-order stage x_out = cx * x, computed in fp32 and rounded once"""
    rng = np.random.default_rng(12)
    n = 1 << 16
    base = torch.from_numpy(rng.standard_normal(n).astype(F32)).to(DEV)
    tiny = torch.from_numpy((rng.standard_normal(4096) * 1e-6).astype(F32)).to(DEV)          # fp16 subnormal range
    x = torch.cat([base, tiny, torch.tensor([0.0, -0.0, 1.0, 65000.0, -3e-8], device=DEV)]).to(sdt)
    x = torch.cat([x, x[: (-x.numel()) % 8]]).reshape(1, -1)                                  # whole 8-element groups
    for cx in (1.0 + 2.0 ** -9, 1.0 + 2.0 ** -11, 0.3333333432674408, 1.5):                  # 1 + 2^-9 / 2^-11: ties
        st = L.Stage()
        st.h1_slot = st.h2_slot = st.m_slot = -1
        st.form, st.flags, st.model_type, st.guidance = L.FORM_LIN1, 0, L.MODEL["noise"], L.GUIDE["uncond"]
        st.cx, st.c0, st.alpha_e, st.sigma_e, st.cfg_scale = cx, 0.0, 1.0, 0.0, 1.0
        out, _ = S._launch_stage(st, x, None, torch.zeros_like(x), None, None, None, None, sdt, want_m=False)
        want = (x.float() * np.float32(cx)).to(sdt)
        assert torch.equal(out.view(torch.int16), want.view(torch.int16)), cx


# ------------------------------------------------------------------------------------------------
# clustered thresholding: the select bound predicted from the previous stages (dpm_buffers.thr_hint)
# ------------------------------------------------------------------------------------------------
def _routes_per_stage(dpm, x, monkeypatch, **kw):
    """sample() with the route word of the hint buffer (1 predicted, 2 prediction rejected, 3 single exchange, 4 general)
    read back after every stage launch"""
    routes = []
    real = S._stage_launch_raw

    def spy(st, b, stream):
        rc = real(st, b, stream)
        torch.cuda.synchronize()
        fr = [v for v in dpm._fast.values() if getattr(v, "thr_hint", None) is not None]
        if fr:
            routes.append(fr[-1].thr_hint.view(-1, L.THR_HINT_WORDS)[:, 2].cpu().numpy().copy())
        return rc
    with monkeypatch.context() as m:
        m.setattr(S, "_stage_launch_raw", spy)
        out = dpm.sample(x, **kw)
    torch.cuda.synchronize()
    return out, routes


@pytest.mark.lab
@pytest.mark.parametrize("shape", [(4, 3, 64, 64), (32, 3, 64, 64), (3, 3, 80, 80)])
def test_threshold_bound_prediction_is_taken_and_exact(shape, monkeypatch):
    """A smooth trajectory: from the third thresholded stage on the clusters select with the PREDICTED bound (route 1);
    the results equal the run with the prediction switched off, bit for bit, and the reference-style quantile."""
    ns = make_schedule("ddpm")
    rng = np.random.default_rng(31)
    x = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV)
    mk = lambda: D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ns), ns, correcting_x0_fn="dynamic_thresholding")
    assert L.lib.dpm_threshold_workspace_bytes(shape[0], int(np.prod(shape[1:]))) > 0        # a clustered shape
    got, routes = _routes_per_stage(mk(), x, monkeypatch, steps=12, order=2)
    assert len(routes) == 12
    assert np.all(routes[0] >= 3) and np.all(routes[1] >= 3)                  # no history yet
    taken = np.mean([np.mean(r == 1) for r in routes[2:]])
    assert taken >= 0.8, (taken, [r.tolist() for r in routes])
    L.lib.dpm_tuning_set(L.TUNE_THR_PREDICT, 0)
    try:
        want, routes0 = _routes_per_stage(mk(), x, monkeypatch, steps=12, order=2)
    finally:
        L.lib.dpm_tuning_set(L.TUNE_THR_PREDICT, 1)
    assert all(np.all(r >= 3) for r in routes0)
    assert torch.equal(got, want)
    if shape[0] * int(np.prod(shape[1:])) <= 1 << 18:      # and the numpy double of the kernel (torch.quantile semantics)
        ncpu = make_schedule("ddpm")
        dbl = _double_on_cpu(monkeypatch, lambda: D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.5, ncpu), ncpu,
                                                               correcting_x0_fn="dynamic_thresholding").sample(x.cpu(), steps=12, order=2))
        assert torch.equal(got.cpu(), dbl)
    again = mk()
    assert torch.equal(again.sample(x, steps=12, order=2), want) and torch.equal(again.sample(x, steps=12, order=2), want)


@pytest.mark.lab
def test_threshold_bound_misprediction_is_detected():
    """A network whose output scale jumps from stage to stage: predictions that are too high (union smaller than K) or
    too low (slot overflow / oversized union) are rejected by every workgroup alike and the searched bound takes over --
    identical results with the prediction on and off, per-sample scales included."""
    shape = (6, 3, 64, 64)
    ns = make_schedule("ddpm")
    rng = np.random.default_rng(32)
    x = torch.from_numpy(rng.standard_normal(shape).astype(F32)).to(DEV)
    scales = [1.0, 0.9, 0.8, 3.0, 0.2, 0.21, 0.22, 5.0, 5.1, 0.01, 0.5, 0.5, 0.55, 40.0, 0.5]
    per_sample = torch.tensor([1.0, 0.3, 2.0, 1.0, 7.0, 0.05], device=DEV).reshape(-1, 1, 1, 1)

    def run():
        calls = [0]

        def net(xx, t):
            calls[0] += 1
            return xx * (scales[(calls[0] - 1) % len(scales)] * (per_sample if calls[0] % 3 == 0 else 1.0))
        return D.DPM_Solver(D.model_wrapper(net, ns), ns, correcting_x0_fn="dynamic_thresholding").sample(x, steps=15, order=2)
    got = run()
    L.lib.dpm_tuning_set(L.TUNE_THR_PREDICT, 0)
    try:
        want = run()
    finally:
        L.lib.dpm_tuning_set(L.TUNE_THR_PREDICT, 1)
    assert torch.isfinite(got).all() and torch.equal(got, want)


# ------------------------------------------------------------------------------------------------
# round 4: a cluster wait that times out is recovered from inside the kernel
# ------------------------------------------------------------------------------------------------
class _Tuned:
    """dpm_tuning_set for the duration of a `with` block"""

    def __init__(self, **knobs):
        self.knobs = {getattr(L, "TUNE_" + k.upper()): v for k, v in knobs.items()}

    def __enter__(self):
        self.old = {k: L.lib.dpm_tuning_get(k) for k in self.knobs}
        for k, v in self.knobs.items():
            L.check(L.lib.dpm_tuning_set(k, v))

    def __exit__(self, *exc):
        for k, v in self.old.items():
            L.lib.dpm_tuning_set(k, v)


def _thr_solver(ns, model=lambda xx, t: xx * 0.5, **kw):
    return D.DPM_Solver(D.model_wrapper(model, ns, **kw), ns, correcting_x0_fn="dynamic_thresholding")


def _non_finite_rows(shape, p, seed):
    """every row of a batch with one kind of non-finite content: nothing / one NaN / one inf / infs around the wanted rank /
    -infs / a NaN and infs; returns the fp32 tensor"""
    g = np.random.default_rng(seed)
    B, n = shape[0], int(np.prod(shape[1:]))
    x = g.standard_normal((B, n)).astype(F32)
    K = max(1, int(n - np.floor(F32(p) * F32(n - 1))))
    for b in range(B):
        kind, idx = b % 6, g.permutation(n)
        if kind in (1, 5):
            x[b, idx[0]] = np.nan
        if kind == 2:
            x[b, idx[0]] = np.inf
        if kind == 3:
            x[b, idx[:K + 1]] = np.inf
        if kind == 4:
            x[b, idx[:K]] = -np.inf
        if kind == 5:
            x[b, idx[1:K + 1]] = np.inf
    return torch.from_numpy(x.reshape(shape))


def _thr_reference(x0, p, max_val):
    """ref :416-425 on the CPU"""
    pq = torch.quantile(x0.abs().reshape(x0.shape[0], -1), p, dim=1)
    s = torch.maximum(pq, max_val * torch.ones_like(pq))[(...,) + (None,) * (x0.dim() - 1)]
    return torch.clamp(x0, -s, s) / s


@pytest.mark.parametrize("shape", [(12, 7), (12, 513), (13, 4095), (40, 3, 64, 64), (6, 3, 64, 64), (12, 12289), (6, 3, 256, 256),
                                   (1030, 3, 64, 64)])
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_thresholding_of_non_finite_samples_is_the_reference_lines(shape, dt):
    """Round 6 (rounds 1-5 clamped a NaN element to +-1 and never looked for one): torch.quantile gives NaN for a sample that
    holds a NaN anywhere, so the reference's whole sample is NaN; infinite values go through its three lines (an infinite
    quantile divides finite elements to 0, inf / inf and inf - inf between the order statistics are NaN).  One workgroup per
    sample, clusters on the single-exchange and the general route, ragged rows, full histograms (p = 0.5): bit for bit."""
    if dt is torch.float64 and int(np.prod(shape)) > 3_000_000:
        pytest.skip("size covered in fp32")
    ns = make_schedule("ddpm")
    for p in (0.5, 0.97, 0.995, 1.0):
        x0 = _non_finite_rows(shape, p, seed=len(shape) + shape[0]).to(dt)
        dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=p)
        got = dpm.dynamic_thresholding_fn(x0.to(DEV), None).cpu()
        want = _thr_reference(x0, p, 1.0)
        assert bool(((got == want) | (got.isnan() & want.isnan())).all()), (shape, p)
        assert bool(want.reshape(shape[0], -1)[1].isnan().all())          # (row 1 holds one NaN: the whole row)
        if dt is torch.float32:
            np.testing.assert_array_equal(got.numpy(), O.dynamic_threshold(x0.numpy(), p, 1.0))


@pytest.mark.lab
@pytest.mark.parametrize("mode,one_hop", [(1, 1), (2, 1), (3, 0)])
def test_non_finite_samples_when_a_cluster_wait_times_out(mode, one_hop):
    """the recovery path (solo_select) and the peers that go on without the lost workgroup look for the NaN like the healthy
    routes do"""
    ns = make_schedule("ddpm")
    for shape in ((6, 3, 64, 64), (3, 3, 160, 160)):
        for p in (0.5, 0.995):
            x0 = _non_finite_rows(shape, p, seed=7)
            dpm = D.DPM_Solver(lambda x, t: x, ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=p)
            with _Tuned(thr_debug_fault=mode, thr_spin_limit=48, cluster_one_hop=one_hop):
                got = dpm.dynamic_thresholding_fn(x0.to(DEV), None).cpu()
                torch.cuda.synchronize()
            want = _thr_reference(x0, p, 1.0)
            assert bool(((got == want) | (got.isnan() & want.isnan())).all()), (shape, p, mode)
    L.cluster_timeout_poll()


@pytest.mark.lab
@pytest.mark.parametrize("shape", [(32, 3, 64, 64), (5, 3, 64, 64), (2, 3, 160, 160), (300, 1, 128, 128)])
@pytest.mark.parametrize("mode,one_hop", [(1, 1), (2, 1), (3, 1), (1, 0), (2, 0), (3, 0)])
def test_cluster_wait_timeout_is_recovered_inside_the_kernel(shape, mode, one_hop):
    """Forced faults -- mode 1: every wait on a peer gives up at its first unsuccessful poll; modes 2 / 3: workgroup 1 of
    every cluster takes no part in it from the start, with / without marking its samples, so its peers see the mark or run
    into the (shortened) timeout -- on the single-exchange route and on the general route (merged histograms).  A
    workgroup whose wait timed out leaves the cluster protocol and computes the order statistics of its samples alone from
    global memory: the trajectory's bits do not change, nothing raises, the workspace is left zero-filled, and only the
    diagnostic word says that it happened.  [300,1,128,128]: 600 workgroups in clusters of two, more than the chip holds at
    once -- clusters walk several samples, a workgroup that left the protocol on one sample stays out on the next ones."""
    ns = make_schedule("ddpm")
    x = torch.from_numpy(np.random.default_rng(41).standard_normal(shape).astype(F32)).to(DEV)
    assert L.lib.dpm_threshold_workspace_bytes(shape[0], int(np.prod(shape[1:]))) > 0        # a clustered shape
    torch.cuda.synchronize()
    L.cluster_timeout_poll()                                                                  # clear
    want = _thr_solver(ns).sample(x, steps=8, order=2)
    torch.cuda.synchronize()
    assert not L.cluster_timeout_poll()                                                       # a healthy run reports nothing
    dpm = _thr_solver(ns)
    with _Tuned(thr_debug_fault=mode, thr_spin_limit=48, cluster_one_hop=one_hop):
        got = dpm.sample(x, steps=8, order=2)
        torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert L.cluster_timeout_poll()
    for fr in dpm._fast.values():
        if getattr(fr, "ws", None) is not None:
            assert not bool(fr.ws.any()), "the workspace must be all zero between launches"
    # and the same launch records keep working afterwards (nothing to re-zero, nothing evicted)
    assert torch.equal(dpm.sample(x, steps=8, order=2), want)
    torch.cuda.synchronize()
    assert not L.cluster_timeout_poll()


@pytest.mark.lab
@pytest.mark.parametrize("mode,one_hop,spin", [(1, 0, 0), (3, 0, 32), (2, 0, 32), (3, 1, 32), (3, 0, 512)])
def test_staggered_timeouts_in_a_twelve_workgroup_cluster(mode, one_hop, spin):
    """The case the forced-fault sweep (DPM_THR_SWEEP_FAULT) caught in the first version of the recovery: [40, 3, 64, 128]
    = clusters of 12 workgroups whose waits run out at different moments (a short limit).  Workgroups that had given up
    used to run on through the exchange with half-merged histograms and discard the outcome: a neighbouring chunk came
    out wrong once in a few launches and workspace words stayed dirty.  Now they leave the protocol.  Every intermediate
    state of several repetitions equals the undisturbed run; the workspace is all zero after every trajectory."""
    ns = make_schedule("sd")
    shape = (40, 3, 64, 128)
    x = torch.from_numpy(np.random.default_rng(123).standard_normal(shape).astype(F32)).to(DEV)
    mk = lambda: D.DPM_Solver(D.model_wrapper(lambda xx, t: xx * 0.9, ns), ns, correcting_x0_fn="dynamic_thresholding",
                              thresholding_max_val=0.5, dynamic_thresholding_ratio=0.995)
    _, want = mk().sample(x, steps=10, order=2, return_intermediate=True)
    import dpm_solver_amd.solver as S
    for rep in range(4):
        with _Tuned(thr_debug_fault=mode, thr_spin_limit=spin, cluster_one_hop=one_hop):
            _, got = mk().sample(x, steps=10, order=2, return_intermediate=True)
            torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), (rep, i)
        for ws in S._WS_CACHE.values():
            assert not bool(ws.any()), "the workspace must be all zero between launches"
    assert L.cluster_timeout_poll()


@pytest.mark.lab
def test_cluster_wait_timeout_with_cfg_half_state_and_requests_in_flight():
    """the recovery path recomputes x0 with the launch's own prologue: classifier-free guidance, fp16 state, fp32 state with
    fp16 outputs; and a fused multi-request launch of clustered shapes (a workspace per request)"""
    ns = make_schedule("ddpm")
    shape = (4, 3, 64, 64)
    g = torch.Generator().manual_seed(43)
    cond = torch.ones(shape[0], device=DEV)
    for sdt, edt in ((torch.float32, None), (torch.float16, None), (torch.float32, torch.float16)):
        x = torch.randn(shape, generator=g).to(DEV, sdt)
        net = lambda xx, t, c: (xx.float() * (0.4 + 0.2 * c.reshape(-1, 1, 1, 1))).to(edt or xx.dtype)
        mk = lambda: D.DPM_Solver(D.model_wrapper(net, ns, guidance_type="classifier-free", condition=cond,
                                                  unconditional_condition=cond * 0, guidance_scale=3.0), ns,
                                  correcting_x0_fn="dynamic_thresholding", **({"state_dtype": sdt} if sdt is not torch.float32 else {}))
        want = mk().sample(x, steps=6, order=2)
        wants = [mk().sample(x * s, steps=6, order=2) for s in (1.0, 0.5, 2.0)]
        with _Tuned(thr_debug_fault=2, thr_spin_limit=48):
            assert torch.equal(mk().sample(x, steps=6, order=2), want)
            for got, w in zip(mk().sample_requests([x, x * 0.5, x * 2.0], steps=6, order=2), wants):
                assert torch.equal(got, w)
        torch.cuda.synchronize()
    assert L.cluster_timeout_poll()


def test_two_processes_run_clustered_thresholding_on_one_gpu():
    """Two PROCESSES, each running clustered thresholding trajectories on cuda:0 at the same time: the library can only
    chain the clustered launches of its own process, so the two grids may hold the chip against each other.  Both must
    finish (bounded waits + recovery inside the kernel) with the bits of an undisturbed run."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as tmp:
        procs = [subprocess.Popen([sys.executable, os.path.join(here, "cluster_pair_worker.py"), tmp, str(r), "2"],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
        outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    res = [json.loads(so.strip().splitlines()[-1]) for so, _ in outs]
    for r in res:
        assert r["ok"], r
        assert r["trajectories"] >= 20 and r["overlapped"], r


# ------------------------------------------------------------------------------------------------
# round 5: a double-precision state through the C ABI (dpm_f64.hip) and the LDS-DMA variant of the lone-launch kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ns_dtype", [torch.float64, torch.float32])
def test_double_precision_state_on_the_gpu(ns_dtype, monkeypatch):
    """sample(x.double()) (ref :14, :105-107): DPM_DTYPE_F64 launches -- multistep, singlestep (double inner-node times),
    logSNR, classifier-free guidance with the duplicate store, v-prediction, dynamic thresholding (63-bit radix select),
    denoise_to_zero, the general loop with intermediates -- against the numpy double of the same stage arithmetic driven by
    the same plan: the GPU computes IEEE double without contraction, so the states agree to the last bits (1e-14 of the
    scale); tests/test_differential_reference.py pins that double to the reference (1e-12 / 1e-6)."""
    betas = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64)
    rng = np.random.default_rng(64)
    xc = torch.from_numpy(rng.standard_normal((6, 3, 16, 16)))
    cond = torch.arange(1, 7, dtype=torch.float64) * 0.5
    net = lambda xx, t: xx * 0.5 * torch.cos(t.to(xx.dtype) * 1e-3).reshape(-1, 1, 1, 1) + 0.1
    netc = lambda xx, t, c: net(xx, t) * (1.0 + 0.1 * c.to(xx.dtype).reshape(-1, 1, 1, 1))

    def solver(dev, kind):
        ns = D.NoiseScheduleVP("discrete", betas=betas, dtype=ns_dtype)
        if kind == "cfg":
            fn = D.model_wrapper(netc, ns, guidance_type="classifier-free", condition=cond.to(dev),
                                 unconditional_condition=torch.zeros(6, dtype=torch.float64, device=dev), guidance_scale=3.0)
            return D.DPM_Solver(fn, ns)
        if kind == "v":
            return D.DPM_Solver(D.model_wrapper(net, ns, model_type="v"), ns, algorithm_type="dpmsolver")
        if kind == "thr":
            return D.DPM_Solver(D.model_wrapper(net, ns), ns, correcting_x0_fn="dynamic_thresholding", dynamic_thresholding_ratio=0.93)
        return D.DPM_Solver(D.model_wrapper(net, ns), ns)
    cases = [("plain", dict(steps=12, order=2)), ("plain", dict(steps=9, order=3, method="singlestep")),
             ("plain", dict(steps=8, order=3, skip_type="logSNR", denoise_to_zero=True)), ("cfg", dict(steps=10, order=2)),
             ("v", dict(steps=9, order=3, method="singlestep", solver_type="taylor")), ("thr", dict(steps=10, order=2)),
             ("thr", dict(steps=7, order=2, method="singlestep", return_intermediate=True))]
    for kind, kw in cases:
        got = solver(DEV, kind).sample(xc.to(DEV), **kw)
        want = _double_on_cpu(monkeypatch, lambda: solver("cpu", kind).sample(xc, **kw))
        if kw.get("return_intermediate"):
            assert len(got[1]) == len(want[1])
            for a, b in zip(got[1], want[1]):
                assert a.dtype == torch.float64 and float((a.cpu() - b).abs().max()) <= 1e-14 * float(b.abs().max())
            got, want = got[0], want[0]
        assert got.dtype == torch.float64 and got.is_cuda
        assert float((got.cpu() - want).abs().max()) <= 1e-14 * float(want.abs().max()), (kind, kw)
    # more elements than one pass of the grid covers (several elements per lane in flight, the last group ragged)
    xb = torch.from_numpy(rng.standard_normal((83, 4, 63, 65)))
    got = solver(DEV, "plain").sample(xb.to(DEV), steps=4, order=3)
    want = _double_on_cpu(monkeypatch, lambda: solver("cpu", "plain").sample(xb, steps=4, order=3))
    assert float((got.cpu() - want).abs().max()) <= 1e-14 * float(want.abs().max())
    # several double requests in flight: no fused double kernel, dpm_stage_launch_multi launches them one by one -- same results
    for kind in ("plain", "thr"):
        reqs = [torch.from_numpy(rng.standard_normal((6, 3, 16, 16))).to(DEV) for _ in range(3)]
        outs = solver(DEV, kind).sample_requests(reqs, steps=6, order=2)
        for xr, o in zip(reqs, outs):
            assert o.dtype == torch.float64 and torch.equal(o, solver(DEV, kind).sample(xr, steps=6, order=2))
    # add_noise and the stand-alone thresholding call in double
    dpm = solver(DEV, "thr")
    x0 = (xc * 2.0).to(DEV)
    y = dpm.dynamic_thresholding_fn(x0, None)
    rows = x0.abs().reshape(6, -1)
    s = torch.maximum(torch.quantile(rows, 0.93, dim=1), torch.ones(6, dtype=torch.float64, device=DEV)).reshape(-1, 1, 1, 1)
    assert torch.equal(y, torch.clamp(x0, -s, s) / s)                       # torch.quantile's own double semantics, bit for bit
    tt_ = torch.tensor([0.3, 0.9])
    noise = torch.from_numpy(rng.standard_normal((2, 6, 3, 16, 16))).to(DEV)
    an = dpm.add_noise(x0, tt_, noise=noise)
    ns = dpm.noise_schedule
    a, sg = ns.marginal_alpha(tt_).double().to(DEV), ns.marginal_std(tt_).double().to(DEV)
    want_an = a.reshape(2, 1, 1, 1, 1) * x0 + sg.reshape(2, 1, 1, 1, 1) * noise
    assert an.dtype == torch.float64 and float((an - want_an).abs().max()) <= 1e-15 * float(want_an.abs().max())


@pytest.mark.parametrize("shape", [(5, 1, 7, 9), (3, 1, 32, 32), (4, 3, 64, 64), (2, 3, 100, 101)])
def test_double_thresholding_select_against_torch_quantile(shape):
    """the double-precision thresholding kernel's select (csrc/dpm_f64.hip: histogram passes over the 63-bit patterns until
    the selected bin fits an LDS list, rank counting there) against torch.quantile in double, bit for bit: samples smaller
    than the list (no histogram pass at all), odd sizes, heavy ties (more copies of one value than the list holds: every bit
    settled by histograms), constant samples, ratios whose rank is integral / fractional / the last element"""
    ns = D.NoiseScheduleVP("discrete", betas=torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64), dtype=torch.float64)
    g = torch.Generator(device=DEV).manual_seed(5)
    base = torch.randn(shape, dtype=torch.float64, device=DEV, generator=g) * 1.7
    B = shape[0]
    variants = {"random": base, "ties": torch.round(base * 2.0) / 2.0, "constant": torch.full_like(base, 1.25),
                "two_values": torch.where(base > 0.3, torch.full_like(base, 3.0), torch.full_like(base, 0.5)),
                "tiny": base * 1e-200, "with_zeros": torch.where(base.abs() < 1.0, torch.zeros_like(base), base)}
    for name, x0 in variants.items():
        for ratio in (0.5, 0.93, 0.995, 1.0, 0.0, 1.0 / 3.0):
            for mv in (1.0, 1e-250):
                dpm = D.DPM_Solver(D.model_wrapper(lambda xx, t: xx, ns), ns, correcting_x0_fn="dynamic_thresholding",
                                   dynamic_thresholding_ratio=ratio, thresholding_max_val=mv)
                y = dpm.dynamic_thresholding_fn(x0, None)
                s = torch.maximum(torch.quantile(x0.abs().reshape(B, -1), ratio, dim=1),
                                  mv * torch.ones(B, dtype=torch.float64, device=DEV)).reshape(-1, 1, 1, 1)
                assert torch.equal(y, torch.clamp(x0, -s, s) / s), (name, ratio, mv, shape)


@pytest.mark.lab
def test_lds_dma_variant_of_the_lone_launch_kernels_is_bit_identical():
    """The lone-launch north-star kernels (2-byte state and network output, unguided noise prediction, dpmsolver++ first /
    second order, inputs from HBM) read their three streams by LDS-DMA (stage_kernel_dma, round 5) -- the same elements per
    lane, the same arithmetic: forcing the register path (DPM_TUNE_LDS_DMA = 0) must give the same bits, for whole and
    ragged tile counts, both workgroup sizes, a grid that loops (capped) and one that does not."""
    ns = make_schedule("sd")
    for sdt in (torch.float16, torch.bfloat16):
        for shape in ((256, 4, 64, 64), (8, 4, 64, 64), (3, 4, 40, 40), (1, 1, 8, 8), (700, 4, 64, 64)):
            g = torch.Generator().manual_seed(sum(shape))
            x = torch.randn(shape, generator=g).to(DEV, sdt)
            eps = torch.randn(shape, generator=g).to(DEV, sdt)
            mk = lambda: D.DPM_Solver(D.model_wrapper(lambda xx, t: eps, ns), ns, state_dtype=sdt)
            outs = {}
            for dma in (1, 0):
                for bt in (0, 256, 512):
                    with _Tuned(lds_dma=dma, block_threads=bt):
                        outs[(dma, bt)] = mk().sample(x, steps=7, order=2)
            ref = outs[(0, 256)]
            for k, v in outs.items():
                assert torch.equal(v.view(torch.int16), ref.view(torch.int16)), (sdt, shape, k)


@pytest.mark.lab
@pytest.mark.parametrize("shape", [(32, 3, 64, 64), (300, 1, 128, 128), (8, 3, 256, 256), (1024, 3, 16, 16)])
def test_staggered_cluster_start_experiment_keeps_the_bits(shape):
    """DPM_TUNE_THR_STAGGER (lab build, profiles/r05_thresholding.md): clusters started out of phase -- an offset that stays
    far below the wait limit changes nothing; one near it (a late cluster's peers are equally late, clusters never wait for
    each other) does not either; workspace left zero-filled"""
    ns = make_schedule("ddpm")
    x = torch.from_numpy(np.random.default_rng(44).standard_normal(shape).astype(F32)).to(DEV)
    want = _thr_solver(ns).sample(x, steps=6, order=2)
    torch.cuda.synchronize()
    L.cluster_timeout_poll()                     # clear what earlier (fault-injecting) tests may have left
    for value in ((2 << 16) | 20, (3 << 16) | 70, (4 << 16) | 400):
        dpm = _thr_solver(ns)
        with _Tuned(thr_stagger=value):
            got = dpm.sample(x, steps=6, order=2)
            torch.cuda.synchronize()
        assert torch.equal(got, want), (shape, value)
        for fr in dpm._fast.values():
            if getattr(fr, "ws", None) is not None:
                assert not bool(fr.ws.any()), "the workspace must be all zero between launches"
    assert not L.cluster_timeout_poll()


@pytest.mark.lab
@pytest.mark.parametrize("shape", [(32, 3, 64, 64), (5, 3, 64, 64), (2, 3, 160, 160), (300, 1, 128, 128), (40, 3, 64, 128)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_elected_reducer_experiment_keeps_the_bits(shape, mode):
    """DPM_TUNE_THR_ELECT (lab build): workgroup 0 of a cluster reads the k slots, selects on the
    union and publishes the verdict, its peers wait for three words -- k slot reads per sample instead of k^2.  Same bits
    as the default protocol on the predicted and the searched route, with and without forced faults (a reducer that is
    out of the protocol publishes nothing: its peers time out and select alone), workspace left zero-filled."""
    ns = make_schedule("ddpm")
    x = torch.from_numpy(np.random.default_rng(43).standard_normal(shape).astype(F32)).to(DEV)
    want, wi = _thr_solver(ns).sample(x, steps=8, order=2, return_intermediate=True)
    for predict in (1, 0):
        dpm = _thr_solver(ns)
        knobs = dict(thr_elect=1, thr_predict=predict)
        if mode:
            knobs.update(thr_debug_fault=mode, thr_spin_limit=48)
        with _Tuned(**knobs):
            got = dpm.sample(x, steps=8, order=2)
            got2, gi = dpm.sample(x, steps=8, order=2, return_intermediate=True)
            torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(got2, want), (shape, mode, predict)
        assert all(torch.equal(a, b) for a, b in zip(gi, wi))
        for fr in dpm._fast.values():
            if getattr(fr, "ws", None) is not None:
                assert not bool(fr.ws.any()), "the workspace must be all zero between launches"
        for ws in S._WS_CACHE.values():
            assert not bool(ws.any()), "the workspace must be all zero between launches"
    L.cluster_timeout_poll()
