"""Shared definitions for the golden fixtures and the parity tests.

This module is imported by
  * tests/golden/make_golden.py  (runs the REAL reference, in the build container only),
  * the oracle tests (numpy restatement vs the goldens, CPU),
  * the GPU parity tests (HIP engine vs oracle and vs goldens).

It deliberately imports neither the reference nor the engine.  Every frozen "network"
below is built from single IEEE-754 fp32 operations (mul/add by constants), so the torch
version (fed to the reference and to the HIP engine) and the numpy version (fed to the
oracle) produce bit-identical outputs on identical inputs.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# Noise-schedule inputs (what a caller hands to NoiseScheduleVP).  Pure numpy, float64
# construction then one cast to float32 -- mirrors how the example apps build them:
#   SD scaled-linear : examples/stable-diffusion/ldm/modules/diffusionmodules/util.py:22-25
#   DDPM linear      : examples/ddpm_and_guided-diffusion/runners/diffusion.py:95-98
#   iDDPM cosine     : examples/ddpm_and_guided-diffusion/runners/diffusion.py:62-78,99-103
# --------------------------------------------------------------------------------------
def schedule_inputs(name):
    """Return dict(kind='discrete', betas=... | alphas_cumprod=...) or dict(kind='linear', ...)."""
    if name == "sd":
        betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
        ac = np.cumprod(1.0 - betas, axis=0)
        return dict(kind="discrete", alphas_cumprod=ac.astype(F32))
    if name == "ddpm":
        betas = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
        return dict(kind="discrete", betas=betas.astype(F32))
    if name in ("cosine4000", "cosine1000"):
        n = 4000 if name == "cosine4000" else 1000
        ab = lambda t: np.cos((t + 0.008) / 1.008 * np.pi / 2) ** 2
        betas = np.array([min(1 - ab((i + 1) / n) / ab(i / n), 0.999) for i in range(n)])
        return dict(kind="discrete", betas=betas.astype(F32))
    if name == "vp_linear":
        return dict(kind="linear", beta_0=0.1, beta_1=20.0)
    raise KeyError(name)


SCHEDULE_NAMES = ["sd", "ddpm", "cosine4000", "cosine1000", "vp_linear"]


# --------------------------------------------------------------------------------------
# Frozen networks.  Signature follows the reference's model contract
#   model(x, t_input[, cond]) -> output            (dpm_solver_pytorch.py:209-233)
# `lib` is the array namespace: torch for the reference / engine, numpy for the oracle.
# --------------------------------------------------------------------------------------
def _bshape(v, x):
    return v.reshape((-1,) + (1,) * (x.ndim - 1))


def model_ident(x, t, cond=None):
    return x


def model_half(x, t, cond=None):
    return x * 0.5


def model_tdep(x, t, cond=None):
    # exercises the t_input plumbing (dpm_solver_pytorch.py:271-280): every op is one fp32 op
    return x * _bshape(t * F32(0.0005).item() + 0.25, x)


def model_cond(x, t, cond):
    # BASELINE cfg3: x * (1 + 0.1 c), c = 0 (uncond) / 1 (cond), one value per sample
    return x * _bshape(cond * F32(0.1).item() + 1.0, x)


MODELS = dict(ident=model_ident, half=model_half, tdep=model_tdep, cond=model_cond)


def classifier_logp_torch(x, t_input, cond):
    """log p(cond | x_t) stand-in for classifier guidance (dpm_solver_pytorch.py:300-307).
    Quadratic, so its gradient w.r.t. x is exactly  -(x - 0.5 c)  in fp32."""
    d = x - _bshape(cond, x) * 0.5
    return -(d * d).reshape(x.shape[0], -1).sum(dim=1) * 0.5


def classifier_grad_numpy(x, t_input, cond):
    # analytic gradient of classifier_logp_torch; autograd yields -(d*1.0) ... see test notes
    d = x - _bshape(cond, x) * F32(0.5)
    return -d


# --------------------------------------------------------------------------------------
# End-to-end cases.  Keys mirror DPM_Solver(...) / .sample(...) kwargs
# (dpm_solver_pytorch.py:338-347, :1047-1050).
# --------------------------------------------------------------------------------------
def _case(name, schedule, shape, **kw):
    d = dict(
        name=name, schedule=schedule, shape=tuple(shape), x_dtype="float32", seed=0,
        algorithm_type="dpmsolver++", method="multistep", order=2, steps=20,
        skip_type="time_uniform", solver_type="dpmsolver", lower_order_final=True,
        denoise_to_zero=False, t_start=None, t_end=None,
        model="ident", model_type="noise", guidance_type="uncond", guidance_scale=1.0,
        thresholding=False, intermediates=True, call="sample",
    )
    d.update(kw)
    return d


E2E_CASES = [
    # --- the five BASELINE.json configs (reduced batch where the full one is only bytes) ---
    _case("cfg1_full", "sd", (8, 4, 64, 64), intermediates=False),
    _case("cfg1_small", "sd", (2, 4, 8, 8)),
    _case("cfg2_fp16", "sd", (2, 4, 64, 64), x_dtype="float16", intermediates=False),
    _case("cfg3_dpmsolver", "ddpm", (2, 3, 16, 16), algorithm_type="dpmsolver", method="singlestep",
          order=3, steps=15, model="cond", guidance_type="classifier-free", guidance_scale=7.5),
    _case("cfg3_pp", "ddpm", (2, 3, 16, 16), algorithm_type="dpmsolver++", method="singlestep",
          order=3, steps=15, model="cond", guidance_type="classifier-free", guidance_scale=7.5),
    _case("cfg5_thresh", "ddpm", (4, 3, 64, 64), steps=25, model="half", thresholding=True,
          intermediates=False),
    _case("cfg5_thresh_small", "ddpm", (3, 3, 8, 8), steps=25, model="half", thresholding=True),
    # --- multistep variants ---
    _case("ms1", "sd", (2, 4, 8, 8), order=1, steps=10, model="tdep"),
    _case("ms2_taylor", "sd", (2, 4, 8, 8), solver_type="taylor", steps=12, model="tdep"),
    _case("ms2_noise", "sd", (2, 4, 8, 8), algorithm_type="dpmsolver", steps=12, model="tdep"),
    _case("ms2_noise_taylor", "ddpm", (2, 3, 8, 8), algorithm_type="dpmsolver", solver_type="taylor",
          steps=12, model="half"),
    _case("ms3", "sd", (2, 4, 8, 8), order=3, steps=12, model="tdep"),
    _case("ms3_noise", "ddpm", (2, 3, 8, 8), order=3, steps=15, algorithm_type="dpmsolver", model="half"),
    _case("ms3_lof", "sd", (2, 4, 8, 8), order=3, steps=6, model="tdep"),            # steps<10 -> lower order tail
    _case("ms2_lof", "sd", (2, 4, 8, 8), order=2, steps=5, model="half"),
    _case("ms3_nolof", "sd", (2, 4, 8, 8), order=3, steps=6, model="tdep", lower_order_final=False),
    _case("ms2_dz", "sd", (2, 4, 8, 8), steps=8, denoise_to_zero=True, model="half"),
    _case("ms2_noise_dz", "ddpm", (2, 3, 8, 8), steps=8, denoise_to_zero=True, model="half",
          algorithm_type="dpmsolver"),
    _case("ms2_trange", "sd", (2, 4, 8, 8), steps=9, t_start=0.8, t_end=0.05, model="tdep"),
    _case("ms2_cosine", "cosine4000", (2, 3, 8, 8), steps=15, model="tdep"),
    _case("ms2_cosine1000", "cosine1000", (2, 3, 8, 8), steps=10, model="half"),
    # --- skip types / continuous schedule ---
    _case("ms2_logsnr_vp", "vp_linear", (2, 3, 8, 8), steps=10, skip_type="logSNR", model="half"),
    _case("ms2_logsnr_sd", "sd", (2, 4, 8, 8), steps=10, skip_type="logSNR", model="tdep"),
    _case("ms2_quad", "ddpm", (2, 3, 8, 8), steps=10, skip_type="time_quadratic", model="half"),
    _case("ms2_vp", "vp_linear", (2, 3, 8, 8), steps=12, model="tdep", t_end=1e-3),
    _case("ms3_noise_vp", "vp_linear", (2, 3, 8, 8), order=3, steps=12, algorithm_type="dpmsolver",
          model="half", t_end=1e-4),
    # --- singlestep variants ---
    _case("ss1", "sd", (2, 4, 8, 8), method="singlestep", order=1, steps=6, model="tdep"),
    _case("ss2_even", "sd", (2, 4, 8, 8), method="singlestep", order=2, steps=8, model="tdep"),
    _case("ss2_odd_taylor", "sd", (2, 4, 8, 8), method="singlestep", order=2, steps=7, model="tdep",
          solver_type="taylor"),
    _case("ss2_noise", "ddpm", (2, 3, 8, 8), method="singlestep", order=2, steps=8, model="half",
          algorithm_type="dpmsolver"),
    _case("ss2_noise_taylor", "ddpm", (2, 3, 8, 8), method="singlestep", order=2, steps=8, model="half",
          algorithm_type="dpmsolver", solver_type="taylor"),
    _case("ss3_mod0", "sd", (2, 4, 8, 8), method="singlestep", order=3, steps=12, model="tdep"),
    _case("ss3_mod1", "sd", (2, 4, 8, 8), method="singlestep", order=3, steps=13, model="tdep"),
    _case("ss3_mod2_taylor", "sd", (2, 4, 8, 8), method="singlestep", order=3, steps=14, model="tdep",
          solver_type="taylor"),
    _case("ss3_noise_taylor", "ddpm", (2, 3, 8, 8), method="singlestep", order=3, steps=12, model="half",
          algorithm_type="dpmsolver", solver_type="taylor"),
    _case("ss3_logsnr_vp", "vp_linear", (2, 3, 8, 8), method="singlestep", order=3, steps=10,
          skip_type="logSNR", model="half", algorithm_type="dpmsolver"),
    _case("ss3_fixed", "sd", (2, 4, 8, 8), method="singlestep_fixed", order=3, steps=10, model="tdep"),
    _case("ss2_fixed_dz", "ddpm", (2, 3, 8, 8), method="singlestep_fixed", order=2, steps=8, model="half",
          denoise_to_zero=True),
    # --- model parameterisations (dpm_solver_pytorch.py:288-298) ---
    _case("mt_xstart", "sd", (2, 4, 8, 8), steps=10, model="half", model_type="x_start"),
    _case("mt_v", "sd", (2, 4, 8, 8), steps=10, model="half", model_type="v"),
    _case("mt_score", "sd", (2, 4, 8, 8), steps=10, model="half", model_type="score"),
    _case("mt_v_vp_ss3", "vp_linear", (2, 3, 8, 8), method="singlestep", order=3, steps=9, model="half",
          model_type="v", t_end=1e-3),
    _case("mt_xstart_noise", "ddpm", (2, 3, 8, 8), steps=10, model="half", model_type="x_start",
          algorithm_type="dpmsolver"),
    # --- guidance ---
    _case("cfg_ms2", "sd", (2, 4, 8, 8), steps=10, model="cond", guidance_type="classifier-free",
          guidance_scale=7.5),
    _case("cfg_scale1", "sd", (2, 4, 8, 8), steps=10, model="cond", guidance_type="classifier-free",
          guidance_scale=1.0),
    _case("cfg_v", "sd", (2, 4, 8, 8), steps=10, model="cond", guidance_type="classifier-free",
          guidance_scale=3.0, model_type="v"),
    _case("clsg_ms2", "ddpm", (2, 3, 8, 8), steps=10, model="half", guidance_type="classifier",
          guidance_scale=2.0),
    _case("clsg_thresh", "ddpm", (2, 3, 16, 16), steps=10, model="half", guidance_type="classifier",
          guidance_scale=2.0, thresholding=True),
    # --- thresholding with other solvers ---
    _case("thresh_ss3", "ddpm", (2, 3, 16, 16), method="singlestep", order=3, steps=9, model="half",
          thresholding=True),
    _case("thresh_dz", "ddpm", (2, 3, 16, 16), steps=8, model="half", thresholding=True,
          denoise_to_zero=True),
    # --- inverse() (dpm_solver_pytorch.py:1032-1045) ---
    _case("inverse_ms2", "sd", (2, 4, 8, 8), steps=10, model="tdep", call="inverse"),
]

E2E_BY_NAME = {c["name"]: c for c in E2E_CASES}


def x_T_for(case):
    """Seeded initial state.  numpy Generator so it does not depend on the torch build."""
    rng = np.random.default_rng(case["seed"] + 1234)
    x = rng.standard_normal(case["shape"]).astype(F32)
    if case["x_dtype"] == "float16":
        x = x.astype(np.float16)
    return x


def cond_for(case):
    """(condition, unconditional_condition) per-sample scalars for guided cases."""
    b = case["shape"][0]
    return np.ones((b,), dtype=F32), np.zeros((b,), dtype=F32)


# --------------------------------------------------------------------------------------
# Stable-Diffusion adapter (examples/stable-diffusion/ldm/models/diffusion/dpm_solver/sampler.py):
# a stand-in for the latent-diffusion model object.  The adapter touches only `alphas_cumprod`,
# `betas.device`, `device` and `apply_model(x, t, c)`.
# --------------------------------------------------------------------------------------
SAMPLER_SHAPE = (2, 4, 8, 8)


def sampler_inputs():
    rng = np.random.default_rng(21)
    sh = SAMPLER_SHAPE
    return dict(x_T=rng.standard_normal(sh).astype(F32), x0=rng.standard_normal(sh).astype(F32),
                noise=rng.standard_normal(sh).astype(F32), mask=rng.random(sh[2:]).astype(F32),
                cond=np.array([1.0, 2.0], dtype=F32), uncond=np.zeros(2, dtype=F32))


class FakeLatentDiffusion:
    """`lib` = torch; tensors live on `device`"""

    def __init__(self, torch, device):
        si = schedule_inputs("sd")
        self.alphas_cumprod = torch.from_numpy(si["alphas_cumprod"]).to(device)
        self.betas = torch.zeros(1, device=device)
        self.device = torch.device(device)
        self.calls = []

    def apply_model(self, x, t, c):
        self.calls.append((tuple(x.shape), float(t.reshape(-1)[0])))
        return x * _bshape(c * F32(0.1).item() + 0.5, x)


# --------------------------------------------------------------------------------------
# guided-diffusion runner wiring (examples/ddpm_and_guided-diffusion/runners/diffusion.py:594-640): stand-ins for the
# 6-channel (learned variance) UNet and the noisy classifier.  Weights are fixed numbers, the classifier is linear in
# pooled features, so torch on CPU (golden) and on the GPU agree to rounding.
# --------------------------------------------------------------------------------------
GD_SHAPE = (4, 3, 8, 8)
GD_CLASSES = 5


def gd_inputs():
    rng = np.random.default_rng(31)
    return dict(x=rng.standard_normal(GD_SHAPE).astype(F32), y=np.array([0, 3, 1, 4], dtype=np.int64),
                w=(rng.standard_normal((GD_CLASSES, GD_SHAPE[1])) * 0.3).astype(F32),
                junk=rng.standard_normal((GD_SHAPE[0], 3) + GD_SHAPE[2:]).astype(F32))


def gd_network(torch, junk):
    """[B,3,H,W] -> [B,6,H,W]: mean half depends on x, t and the label; variance half is unrelated data"""
    def net(x, t, y=None):
        scale = (t * F32(0.0005).item() + 0.25) + (y.to(x.dtype) * F32(0.05).item() if y is not None else 0.0)
        return torch.cat([x * scale.reshape(-1, 1, 1, 1), junk.to(x.device)[: x.shape[0]]], dim=1)
    return net


def gd_classifier(torch, w):
    def clf(x, t):
        feat = x.mean(dim=(2, 3))                                   # [B, C]
        return feat @ w.to(x.device).t() + (t * F32(0.001).item()).reshape(-1, 1)
    return clf


GD_RUNS = [
    ("pp_clf_thresh_denoise", dict(sample_type="dpmsolver++", use_clf=True, thresholding=True, denoise=True, scale=2.0)),
    ("pp_uncond", dict(sample_type="dpmsolver++", use_clf=False, thresholding=False, denoise=False, scale=1.0)),
    ("eps_clf_ss3", dict(sample_type="dpmsolver", use_clf=True, thresholding=False, denoise=False, scale=1.5,
                         method="singlestep", order=3, timesteps=9)),
]


# --------------------------------------------------------------------------------------
# ScoreSDE example (examples/score_sde_pytorch/sampling.py:505-555): stand-in score model and runs
# --------------------------------------------------------------------------------------
SCORE_SDE_SHAPE = (4, 3, 8, 8)
SCORE_SDE_RUNS = [
    ("default", dict()),                                                          # singlestep-3, logSNR, 10 steps, dpmsolver
    ("denoise", dict(denoise=True, steps=12)),
    # (thresholding=True cannot be a golden: the vendored older revision calls correcting_x0_fn(x0) with one argument and
    #  its own dynamic_thresholding_fn(x0, t) then raises TypeError, examples/score_sde_pytorch/dpm_solver.py:449)
    ("pp_ms2", dict(algorithm_type="dpmsolver++", method="multistep", order=2, skip_type="time_uniform", steps=15)),
    ("adaptive", dict(method="adaptive", order=2)),
]


def score_sde_model(torch_mod):
    """a small deterministic 'score model': model(x, labels) with labels = t * 999 (models/utils.py:148)"""
    class M(torch_mod.nn.Module):
        def forward(self, x, labels):
            return x * (labels * 2e-4 + 0.35).reshape(-1, 1, 1, 1)
    return M()
