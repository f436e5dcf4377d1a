"""GPU tests of the fused multi-request path (dpm_stage_launch_multi / dpm_plan_run_multi): R independent requests
advanced by ONE launch per stage must end bit-identical to the same requests run one by one (dpm_plan_run), for every
state / network-output dtype pair, for plans that mix fused and unfused stages (singlestep mid-stages, thresholding),
ragged request counts (R > DPM_MULTI_MAX, R = 1) and unaligned buffers (fallback).  Run on an MI355X:  pytest -m gpu
"""
import ctypes as C_

import numpy as np
import pytest
import torch

import dpm_solver_amd as D
from dpm_solver_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_CODE = {torch.float16: L.DTYPE_F16, torch.float32: L.DTYPE_F32, torch.bfloat16: L.DTYPE_BF16}


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need a GPU; run with -m 'not gpu' elsewhere"
    yield
    torch.cuda.synchronize()


def sd_schedule():
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    return D.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(np.cumprod(1.0 - betas).astype(np.float32)))


def make_requests(n_req, shape, sd, ed, seed, cfg=False, offset=0, dup=False):
    g = torch.Generator().manual_seed(seed)
    reqs = []
    for _ in range(n_req):
        def buf(dt, src=None):
            # a state buffer under `dup` holds the state twice ([2B, ...]: the CFG network input); the view is its first half
            t = torch.empty(int(np.prod(shape)) * (2 if dup and dt is sd else 1) + offset, dtype=dt, device=DEV)
            v = t[offset:offset + int(np.prod(shape))].view(shape)
            if src is not None:
                v.copy_(src)
                if dup and dt is sd:
                    t[offset + int(np.prod(shape)):].view(shape).copy_(src)
            v.full = t
            return v
        x_T = buf(sd, torch.randn(shape, generator=g))
        e0 = buf(ed, torch.randn(shape, generator=g))
        e1 = buf(ed, torch.randn(shape, generator=g)) if cfg else None
        xb = [x_T] + [buf(sd) for _ in range(3)]
        hb = [buf(sd) for _ in range(3)]
        rb = L.RunBuffers()
        for i in range(4):
            rb.xbuf[i] = xb[i].data_ptr()
        for i in range(3):
            rb.hist[i] = hb[i].data_ptr()
        rb.e0 = e0.data_ptr()
        if cfg:
            rb.e1 = e1.data_ptr()
        rb.n, rb.batch = x_T.numel(), shape[0]
        rb.state_dtype, rb.eps_dtype = _CODE[sd], _CODE[ed]
        rb.dup_state = 1 if dup else 0
        reqs.append(dict(rb=rb, x=xb, h=hb, e0=e0, e1=e1))
    return reqs


def plan_for(ns, sd, **kw):
    mt = kw.pop("model_type", "noise")
    model = D.model_wrapper(lambda x, t: x, ns, model_type=mt) if not kw.pop("cfg", False) else D.model_wrapper(
        lambda x, t, c: x, ns, model_type=mt, guidance_type="classifier-free", condition=torch.zeros(1),
        unconditional_condition=torch.zeros(1),
        guidance_scale=kw.pop("scale", 3.0))
    dpm = D.DPM_Solver(model, ns, algorithm_type=kw.pop("algorithm_type", "dpmsolver++"), state_dtype=sd,
                       correcting_x0_fn=kw.pop("correcting_x0_fn", None))
    args = dict(method="multistep", order=2, steps=8, skip_type="time_uniform", solver_type="dpmsolver",
                lower_order_final=True, denoise_to_zero=False, t_T=1.0, t_0=1.0 / ns.total_N)
    args.update(kw)
    return dpm, dpm._get_plan(**args)


def run_both(plan, reqs, own_workspaces=False):
    """final states of every request: fused multi-request run vs one dpm_plan_run per request"""
    stream = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = len(reqs)
    rbs = (L.RunBuffers * n)(*[r["rb"] for r in reqs])
    res = (C_.c_int * n)()
    ws = []
    nb = L.lib.dpm_threshold_workspace_bytes(reqs[0]["rb"].batch, reqs[0]["rb"].n // reqs[0]["rb"].batch)
    if nb:
        # one workspace per request (the thresholded stages of all requests can then share a launch), or one for all
        ws = [torch.zeros(nb, dtype=torch.uint8, device=DEV) for _ in range(n if own_workspaces else 1)]
        for i in range(n):
            rbs[i].workspace = ws[i % len(ws)].data_ptr()
    L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, n, stream, None, res))
    torch.cuda.synchronize()
    fused = [reqs[i]["x"][res[i]].clone() for i in range(n)]
    single = []
    r1 = C_.c_int(-1)
    for i in range(n):
        for b in reqs[i]["x"][1:] + reqs[i]["h"]:
            b.fill_(float("nan"))
        L.check(L.lib.dpm_plan_run(plan.handle, C_.byref(rbs[i]), None, None, stream, C_.byref(r1)))
        torch.cuda.synchronize()
        single.append(reqs[i]["x"][r1.value].clone())
    for w in ws:
        assert not bool(w.any()), "a thresholding workspace was not left zero-filled"
    return fused, single


@pytest.mark.parametrize("sd,ed", [(torch.float16, torch.float16), (torch.float32, torch.float32),
                                   (torch.float32, torch.float16), (torch.float32, torch.bfloat16),
                                   (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("order", [1, 2, 3])
def test_fused_equals_single_multistep(sd, ed, order):
    ns = sd_schedule()
    _, plan = plan_for(ns, sd, order=order, steps=7)
    reqs = make_requests(5, (6, 4, 24, 24), sd, ed, seed=order)   # 13824 elements: partial last tile
    fused, single = run_both(plan, reqs)
    for a, b in zip(fused, single):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b)


@pytest.mark.parametrize("algo", ["dpmsolver", "dpmsolver++"])
def test_fused_equals_single_cfg(algo):
    ns = sd_schedule()
    _, plan = plan_for(ns, torch.float32, cfg=True, algorithm_type=algo, order=2, steps=6)
    reqs = make_requests(4, (3, 4, 32, 32), torch.float32, torch.float16, seed=3, cfg=True)
    fused, single = run_both(plan, reqs)
    for a, b in zip(fused, single):
        assert torch.equal(a, b)


@pytest.mark.parametrize("sd,ed", [(torch.float32, torch.float16), (torch.float16, torch.float16), (torch.float32, torch.float32)])
def test_fused_cfg_with_duplicate_state_store(sd, ed):
    """classifier-free guidance with the [2B, ...] network input written by the stage kernel (dup_state): the fused
    launch carries the second store too; both halves of every state buffer equal the single-request run's"""
    ns = sd_schedule()
    _, plan = plan_for(ns, sd, cfg=True, order=2, steps=6)
    reqs = make_requests(4, (3, 4, 32, 32), sd, ed, seed=9, cfg=True, dup=True)
    stream = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = len(reqs)
    rbs = (L.RunBuffers * n)(*[r["rb"] for r in reqs])
    res = (C_.c_int * n)()
    L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, n, stream, None, res))
    torch.cuda.synchronize()
    fused = [[b.full.clone() for b in reqs[i]["x"]] for i in range(n)]
    r1 = C_.c_int(-1)
    half = reqs[0]["x"][0].numel()
    for i in range(n):
        for b in reqs[i]["x"][1:]:
            b.full.fill_(float("nan"))
        L.check(L.lib.dpm_plan_run(plan.handle, C_.byref(rbs[i]), None, None, stream, C_.byref(r1)))
        torch.cuda.synchronize()
        assert r1.value == res[i]
        for j, b in enumerate(reqs[i]["x"]):
            if j == r1.value:     # the final state feeds no network call: only its first half is written
                assert torch.isfinite(b.full[:half].float()).all()
                assert torch.equal(fused[i][j][:half], b.full[:half])
            elif j > 0 and not bool(torch.isnan(b.full.float()).all()):   # intermediate states (a multistep plan ping-pongs
                                                                          # between two of the three): both halves written
                assert torch.isfinite(b.full.float()).all() and torch.equal(b.full[:half], b.full[half:])
                assert torch.equal(fused[i][j], b.full)


@pytest.mark.parametrize("mt", ["v", "x_start", "score"])
@pytest.mark.parametrize("cfg", [False, True])
def test_fused_equals_single_other_parameterisations(mt, cfg):
    """x_start / v / score networks run the fused launch with the general prologue"""
    ns = sd_schedule()
    _, plan = plan_for(ns, torch.float16, cfg=cfg, model_type=mt, order=2, steps=6)
    reqs = make_requests(3, (4, 4, 32, 32), torch.float16, torch.float16, seed=8, cfg=cfg)
    fused, single = run_both(plan, reqs)
    for a, b in zip(fused, single):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b)


@pytest.mark.parametrize("n_req", [1, 2, 33, 70])
def test_request_counts_beyond_one_launch(n_req):
    ns = sd_schedule()
    _, plan = plan_for(ns, torch.float16, order=2, steps=5)
    reqs = make_requests(n_req, (2, 4, 16, 16), torch.float16, torch.float16, seed=n_req)
    fused, single = run_both(plan, reqs)
    for a, b in zip(fused, single):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kw", [dict(method="singlestep", order=3, steps=9),
                                dict(method="singlestep", order=2, steps=6, solver_type="taylor"),
                                dict(correcting_x0_fn="dynamic_thresholding", order=2, steps=5),
                                dict(denoise_to_zero=True, order=2, steps=5)])
def test_plans_with_unfused_stages(kw):
    """stages outside the fused family (xe != x mid-stages, thresholding, the denoise stage) run request by request
    inside the same dpm_plan_run_multi"""
    ns = sd_schedule()
    _, plan = plan_for(ns, torch.float32, **kw)
    reqs = make_requests(3, (4, 3, 16, 16), torch.float32, torch.float32, seed=11)
    fused, single = run_both(plan, reqs)
    for a, b in zip(fused, single):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape,n_req", [((2, 3, 64, 64), 3), ((32, 3, 64, 64), 4), ((1, 3, 128, 128), 5), ((4, 3, 16, 16), 7),
                                         ((8, 3, 64, 64), 40)])
@pytest.mark.parametrize("sd,ed,cfg", [(torch.float32, torch.float32, False), (torch.float32, torch.float16, True),
                                       (torch.float16, torch.float16, False)])
def test_thresholded_stages_share_a_launch(shape, n_req, sd, ed, cfg):
    """dynamic thresholding over several requests: ONE launch per stage over all requests' samples (smaller or no
    clusters), bit-identical to the requests run one by one; with a single shared workspace the clustered shapes fall
    back to one launch per request"""
    ns = sd_schedule()
    _, plan = plan_for(ns, sd, correcting_x0_fn="dynamic_thresholding", order=2, steps=5, cfg=cfg)
    for own in (True, False):
        reqs = make_requests(n_req, shape, sd, ed, seed=n_req + shape[0], cfg=cfg)
        fused, single = run_both(plan, reqs, own_workspaces=own)
        for a, b in zip(fused, single):
            assert torch.isfinite(a.float()).all()
            assert torch.equal(a, b)


def test_thresholded_fused_launch_count():
    """the thresholded stages of R requests with their own workspaces really are one launch each"""
    ns = sd_schedule()
    _, plan = plan_for(ns, torch.float32, correcting_x0_fn="dynamic_thresholding", order=2, steps=5)
    reqs = make_requests(6, (8, 3, 64, 64), torch.float32, torch.float32, seed=4)
    n = len(reqs)
    rbs = (L.RunBuffers * n)(*[r["rb"] for r in reqs])
    nb = L.lib.dpm_threshold_workspace_bytes(8, 3 * 64 * 64)
    assert nb > 0
    ws = [torch.zeros(nb, dtype=torch.uint8, device=DEV) for _ in range(n)]
    for i in range(n):
        rbs[i].workspace = ws[i].data_ptr()
    res = (C_.c_int * n)()
    stream = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = (C_.c_float * (n * len(plan.stages)))()
    L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, n, stream, ms, res))
    torch.cuda.synchronize()
    per_stage = np.array(list(ms)).reshape(n, len(plan.stages))
    assert (per_stage > 0).all()
    # a fused launch's duration is spread evenly over its requests: equal times across requests, stage by stage
    assert np.allclose(per_stage, per_stage[0:1], rtol=0, atol=0)


def test_unaligned_and_ragged_fall_back():
    ns = sd_schedule()
    _, plan = plan_for(ns, torch.float32, order=2, steps=5)
    for shape, off in (((2, 3, 5, 7), 0), ((2, 4, 16, 16), 1)):          # n % 8 != 0; pointers off by one element
        reqs = make_requests(3, shape, torch.float32, torch.float32, seed=5, offset=off)
        fused, single = run_both(plan, reqs)
        for a, b in zip(fused, single):
            assert torch.equal(a, b)


def test_stage_launch_multi_direct_and_mismatched_requests():
    """dpm_stage_launch_multi on hand-built buffers; requests of different sizes are launched one by one"""
    ns = sd_schedule()
    _, plan = plan_for(ns, torch.float32, order=2, steps=5)
    st = plan.stages[2].copy()
    assert st.form == L.FORM_TWO
    stream = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    sizes = [4096, 4096, 2048]
    bs = (L.Buffers * 3)()
    keep = []
    for i, n in enumerate(sizes):
        t = [torch.randn(n, device=DEV) for _ in range(3)] + [torch.empty(n, device=DEV) for _ in range(2)]
        keep.append(t)
        bs[i].x, bs[i].e0, bs[i].h1, bs[i].x_out, bs[i].m_out = [v.data_ptr() for v in t]
        bs[i].n, bs[i].batch = n, 1
    for group in (bs, (L.Buffers * 2)(bs[0], bs[1])):
        L.check(L.lib.dpm_stage_launch_multi(C_.byref(st), group, len(group), stream))
        torch.cuda.synchronize()
        for i in range(len(group)):
            got = (keep[i][3].clone(), keep[i][4].clone())
            keep[i][3].zero_()
            keep[i][4].zero_()
            L.check(L.lib.dpm_stage_launch(C_.byref(st), C_.byref(bs[i]), stream))
            torch.cuda.synchronize()
            assert torch.equal(got[0], keep[i][3]) and torch.equal(got[1], keep[i][4])
    assert L.lib.dpm_stage_launch_multi(C_.byref(st), bs, 0, stream) == L.ERR_ARG


def test_fuse_switch_off_gives_identical_results():
    """dpm_launch_opts.no_fuse (a per-call option: no process-wide switch) launches request by request"""
    ns = sd_schedule()
    _, plan = plan_for(ns, torch.float16, order=2, steps=6)
    reqs = make_requests(4, (8, 4, 32, 32), torch.float16, torch.float16, seed=2)
    fused, _ = run_both(plan, reqs)
    opts = L.LaunchOpts()
    opts.no_fuse = 1
    for r in reqs:
        r["rb"].opts = C_.pointer(opts)
    try:
        unfused, _ = run_both(plan, reqs)
    finally:
        for r in reqs:
            r["rb"].opts = None
    for a, b in zip(fused, unfused):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kw", [dict(steps=8, order=2), dict(steps=9, order=3, method="singlestep"),
                                dict(steps=5, order=2, thr=True), dict(steps=6, order=2, cfg=True),
                                dict(steps=6, order=3, half=True), dict(steps=5, order=2, model_type="v", half=True),
                                dict(steps=6, order=3, skip_type="logSNR", denoise_to_zero=True),
                                dict(steps=7, order=2, method="singlestep", solver_type="taylor", thr=True)])
def test_sample_requests_python_api(kw, monkeypatch):
    """DPM_Solver.sample_requests: the requests' results equal sample() of each request bit for bit, and every stage is
    ONE dpm_stage_launch_multi call"""
    import dpm_solver_amd.solver as S
    kw = dict(kw)
    ns = sd_schedule()
    thr, cfg, half = kw.pop("thr", False), kw.pop("cfg", False), kw.pop("half", False)
    mt = kw.pop("model_type", "noise")
    if cfg:
        c = torch.ones(8, device=DEV)
        fn = D.model_wrapper(lambda x, t, cc: torch.tanh(x * 0.7) * (0.5 + 0.1 * cc.reshape(-1, 1, 1, 1)[:x.shape[0]]).to(x.dtype),
                             ns, model_type=mt, guidance_type="classifier-free", guidance_scale=3.0, condition=c,
                             unconditional_condition=c * 0)
    else:
        fn = D.model_wrapper(lambda x, t: torch.tanh(x * 0.7), ns, model_type=mt)
    dpm = D.DPM_Solver(fn, ns, correcting_x0_fn="dynamic_thresholding" if thr else None,
                       state_dtype=torch.float16 if half else None)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(DEV) for _ in range(6)]
    if half:
        xs = [x.half() for x in xs]
    want = [dpm.sample(x, **kw) for x in xs]
    calls = []
    real = S._stage_launch_multi_raw

    def spy(st, bs, n_req, stream):
        calls.append(int(n_req))
        return real(st, bs, n_req, stream)
    monkeypatch.setattr(S, "_stage_launch_multi_raw", spy)
    got = dpm.sample_requests(xs, **kw)
    torch.cuda.synchronize()
    plan_stages = len(calls)
    assert plan_stages >= kw["steps"] and all(c == len(xs) for c in calls)
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and torch.isfinite(a.float()).all() and torch.equal(a, b)
    again = dpm.sample_requests(xs, **kw)
    for a, b in zip(again, want):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------
# the fused kernel at the benchmark's own size (bench.py's timed workload), against the numpy double of the kernel
# (half states: bit for bit) and against the oracle (fp32: the north-star 1e-5)
# ------------------------------------------------------------------------------------------------
def _oracle_2m(ac, x, eps, steps=20):
    """oracle/dpm_oracle.py on a slice: DPM-Solver++(2M), frozen eps (ref :796-852 through ref :1195-1213)"""
    from oracle import dpm_oracle as O
    osch = O.Schedule.from_alphas_cumprod(ac)
    return O.Solver(O.wrap_model(lambda xx, t: eps, osch), osch, algorithm_type="dpmsolver++").sample(x, steps=steps, order=2)


def _sd_ac():
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas).astype(np.float32)


@pytest.mark.parametrize("sd", [torch.float16, torch.float32])
def test_fused_launch_at_bench_size_against_double_and_oracle(sd, monkeypatch):
    """dpm_plan_run_multi on 32 x [256,4,64,64] (what bench.py times: 20 fused launches of 32 requests, 1.3 GB each in
    fp16): requests 0, 15 and 31 -- fp16: bit-equal to tests/kernel_double.py driven by the same plan; fp32: <= 1e-5 of
    the tensor's scale against the oracle on 8-sample slices"""
    import dpm_solver_amd.solver as S
    from kernel_double import install_cpu_double
    shape, R = (256, 4, 64, 64), 32
    ns = sd_schedule()
    _, plan = plan_for(ns, sd, steps=20)
    reqs = make_requests(R, shape, sd, sd, seed=77)
    stream = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    rbs = (L.RunBuffers * R)(*[r["rb"] for r in reqs])
    res = (C_.c_int * R)()
    L.check(L.lib.dpm_plan_run_multi(plan.handle, rbs, R, stream, None, res))
    torch.cuda.synchronize()
    for r in (0, 15, 31):
        got = reqs[r]["x"][res[r]].cpu()
        x_T, eps = reqs[r]["x"][0].cpu(), reqs[r]["e0"].cpu()
        assert torch.isfinite(got.float()).all()
        if sd is torch.float16:
            with monkeypatch.context() as m:
                install_cpu_double(m, S, D)
                dbl = D.DPM_Solver(D.model_wrapper(lambda x, t: eps, ns), ns, state_dtype=sd)
                want = dbl.sample(x_T, steps=20, order=2)
            assert want.dtype == sd and torch.equal(got.view(torch.int16), want.view(torch.int16)), r
        else:
            for lo in (0, 124, 248):
                want = _oracle_2m(_sd_ac(), x_T[lo:lo + 8].numpy(), eps[lo:lo + 8].numpy())
                g = got[lo:lo + 8].numpy().astype(np.float64)
                err = float(np.abs(g - want).max() / np.abs(want).max())
                assert err <= 1e-5, (r, lo, err)
    # and the Python-loop result of one request (single launches), bit for bit
    chk = D.DPM_Solver(D.model_wrapper(lambda x, t: reqs[31]["e0"], ns), ns, state_dtype=sd)
    assert torch.equal(chk.sample(reqs[31]["x"][0], steps=20, order=2), reqs[31]["x"][res[31]])


@pytest.mark.parametrize("sd,ed", [(torch.float16, torch.float16), (torch.float32, torch.float16)])
def test_sample_requests_cfg_duplicate_store_at_sd_size(sd, ed, monkeypatch):
    """DPM_Solver.sample_requests, classifier-free guidance 7.5, 16 requests of [64,4,64,64] (an SD batch per request): the
    fused CFG kernel with the duplicate store of the [2B,...] network input.  fp16 state: bit-equal to the kernel double;
    fp32 state with fp16 network outputs (SD under autocast): <= 1e-5 against the oracle on slices.  Every stage is one
    fused launch and every launch but the last carries x_out2."""
    import dpm_solver_amd.solver as S
    from kernel_double import install_cpu_double
    from oracle import dpm_oracle as O
    shape, R, scale = (64, 4, 64, 64), 16, 7.5
    ns = sd_schedule()

    def net(lib):
        # a conditional network: output depends on x, t and the condition; returned in the network-output dtype `ed`
        def f(x, t, c):
            cc = c.reshape((-1,) + (1,) * (x.ndim - 1))
            out = x * (0.6 + 0.1 * cc)
            return out.to(ed) if lib is torch else out
        return f

    def solver(dev):
        cond = torch.ones(shape[0], device=dev)
        fn = D.model_wrapper(net(torch), ns, guidance_type="classifier-free", guidance_scale=scale, condition=cond,
                             unconditional_condition=cond * 0)
        return D.DPM_Solver(fn, ns, state_dtype=sd)

    g = torch.Generator().manual_seed(9)
    xs_cpu = [torch.randn(shape, generator=g).to(sd) for _ in range(R)]
    xs = [x.to(DEV) for x in xs_cpu]
    dup = []
    real = S._stage_launch_multi_raw

    def spy(st, bs, n_req, stream):
        dup.append((int(n_req), all(bool(bs[r].x_out2) for r in range(n_req))))
        return real(st, bs, n_req, stream)
    monkeypatch.setattr(S, "_stage_launch_multi_raw", spy)
    got = solver(DEV).sample_requests(xs, steps=20, order=2)
    torch.cuda.synchronize()
    assert len(dup) == 20 and all(n == R for n, _ in dup)
    assert all(d for _, d in dup[:-1]) and not dup[-1][1]          # the last stage feeds no network call
    monkeypatch.setattr(S, "_stage_launch_multi_raw", real)
    for r in (0, 7, 15):
        out = got[r].cpu()
        assert out.dtype == sd and torch.isfinite(out.float()).all()
        if sd is torch.float16:
            with monkeypatch.context() as m:
                install_cpu_double(m, S, D)
                want = solver("cpu").sample(xs_cpu[r], steps=20, order=2)
            assert torch.equal(out.view(torch.int16), want.view(torch.int16)), r
        else:
            osch = O.Schedule.from_alphas_cumprod(_sd_ac())
            for lo in (0, 56):
                c8 = np.ones(8, dtype=np.float32)
                # the oracle's network answers in fp16 like the torch network does (.to(ed)): its classifier-free blend is then the
                # reference's half arithmetic (ref :326-330), which the stage kernel reproduces operation by operation
                onet = lambda x, t, c: (x * (np.float32(0.6) + np.float32(0.1) * c.reshape(-1, 1, 1, 1))).astype(np.float16)
                ofn = O.wrap_model(onet, osch, guidance_type="classifier-free", guidance_scale=scale, condition=c8,
                                   unconditional_condition=c8 * 0)
                want = O.Solver(ofn, osch, algorithm_type="dpmsolver++").sample(xs_cpu[r][lo:lo + 8].numpy(), steps=20, order=2)
                d = np.abs(out[lo:lo + 8].numpy().astype(np.float64) - want) / np.abs(want).max()
                # element by element; a network that rounds its output to fp16 may turn an fp32-ulp difference between two
                # implementations into a half-ulp flip (x the guidance scale) of single elements -- none seen in 2M so far
                assert float((d > 1e-5).mean()) <= 1e-3 and float(d.max()) <= 4 * 2.0 ** -10 * scale, (r, lo, float(d.max()))


@pytest.mark.parametrize("sd,ed", [(torch.float16, torch.float16), (torch.float32, torch.float16)])
def test_a_large_single_launch_takes_the_fused_shape_and_keeps_its_bits(sd, ed):
    """Round 6 (Tuning::big_tiles, profiles/r06_big_single.md): a stage launch of >= 16384 tiles -- plain sample() on one large
    tensor -- is handed to the fused multi-request kernel as a group of one (uncapped grid, XCD-contiguous tiles).  Same
    arithmetic, same bits: the trajectory of a [2304,4,64,64] tensor (18432 tiles) equals the trajectories of its nine
    [256,4,64,64] slices, each far below the threshold, bit for bit -- second and third order, unguided and under CFG."""
    ns = sd_schedule()
    g = torch.Generator(device=DEV).manual_seed(11)
    B = 2304
    x = torch.randn((B, 4, 64, 64), device=DEV, generator=g).to(sd)
    eps = torch.randn((2 * B, 4, 64, 64), device=DEV, generator=g).to(ed)
    assert x.numel() // 2048 >= 16384
    kw_state = {} if sd is torch.float32 else dict(state_dtype=sd)
    for cfg in (False, True):
        for order in (2, 3):
            def solver(lo, hi):
                if cfg:
                    e = torch.cat([eps[lo:hi], eps[B + lo:B + hi]])
                    cond = torch.ones(hi - lo, device=DEV)
                    fn = D.model_wrapper(lambda xx, t, c: e, ns, guidance_type="classifier-free", condition=cond,
                                         unconditional_condition=cond * 0, guidance_scale=3.0)
                else:
                    e = eps[lo:hi]
                    fn = D.model_wrapper(lambda xx, t: e, ns)
                return D.DPM_Solver(fn, ns, algorithm_type="dpmsolver++", **kw_state)
            big = solver(0, B).sample(x, steps=6, order=order)
            for lo in range(0, B, 256):
                small = solver(lo, lo + 256).sample(x[lo:lo + 256], steps=6, order=order)
                assert torch.equal(big[lo:lo + 256], small), (cfg, order, lo)
