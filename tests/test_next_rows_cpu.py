"""Rows N2/N4 of SURVEY.md 8f on the CPU: scheduler maths (pinned by golden G4 from the reference's own
scheduler.py) and Lightning-checkpoint key rewriting (infer.py:46-69)."""
import os

import numpy as np
import torch

from mvd_amd.checkpoint import load_lightning_checkpoint, remap_lightning_state_dict
from mvd_amd.scheduler import DDPMScheduler, ShiftSNRScheduler, SNR_to_betas, compute_snr
from oracle import scheduler as OS


def test_shift_snr_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_shift_snr.npz"))
    base = DDPMScheduler()      # SD scaled-linear betas
    np.testing.assert_array_equal(base.betas.numpy(), g["betas_in"])
    np.testing.assert_allclose(compute_snr(torch.arange(1000), base).numpy(), g["snr"], rtol=1e-6)
    for scale in (6.0, 2.0):
        s = ShiftSNRScheduler.from_scheduler(base, "interpolated", shift_scale=scale, scheduler_class=DDPMScheduler)
        np.testing.assert_allclose(s.betas.numpy(), g[f"interpolated_{scale}"], rtol=0, atol=1e-9)
        d = ShiftSNRScheduler.from_scheduler(base, "default", shift_scale=scale, scheduler_class=DDPMScheduler)
        np.testing.assert_allclose(d.betas.numpy(), g[f"default_{scale}"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(SNR_to_betas(torch.from_numpy(g["snr"])).numpy(), g["betas_in"], atol=1e-6)   # fp32 round trip


def test_ddpm_step_coefficients_match_oracle_step():
    s = ShiftSNRScheduler.from_scheduler(DDPMScheduler(), "interpolated", shift_scale=6.0, scheduler_class=DDPMScheduler)
    s.set_timesteps(20)
    assert s.timesteps.tolist()[:3] == [950, 900, 850] and s.timesteps.tolist() == OS.leading_timesteps(1000, 20).tolist()
    g = torch.Generator().manual_seed(0)
    x, mo, nz = (torch.randn(2, 4, 8, 8, generator=g) for _ in range(3))
    for t in (950, 500, 50, 0):
        c0, c1, c2, c3, sig = s.step_coefficients(t)
        got = c2 * (c0 * mo + c1 * x) + c3 * x + sig * nz
        want = OS.ddpm_step(mo, t, x, s.alphas_cumprod, 1000, 20, "v_prediction", nz)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    e = DDPMScheduler(prediction_type="epsilon")
    e.set_timesteps(50)
    c0, c1, c2, c3, sig = e.step_coefficients(980)
    want = OS.ddpm_step(mo, 980, x, e.alphas_cumprod, 1000, 50, "epsilon", nz)
    torch.testing.assert_close(c2 * (c0 * mo + c1 * x) + c3 * x + sig * nz, want, rtol=1e-4, atol=1e-5)


def test_lightning_checkpoint_key_rewrite(tmp_path):
    from mvd_amd.config import UNetConfig
    from mvd_amd.mvd_unet import MultiViewUNet
    from oracle import mvd as OM, sd21_unet as OU
    params = OM.init_mvd_params(OU.UNetConfig.tiny(), 3, cam_dim=96, cam_hidden=48)
    ckpt = {}
    for k, v in params.items():          # what training.py saves: "unet." prefix, encoder keys WITHOUT ".unet."
        k2 = k.replace("image_encoder.unet.", "image_encoder.", 1)
        ckpt["unet." + k2] = v
    ckpt["vae.some.weight"] = torch.zeros(1)
    ckpt["unet.not_a_real_key"] = torch.zeros(1)
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": ckpt, "hyper_parameters": {}}, path)
    fixed = remap_lightning_state_dict(ckpt)
    assert "image_encoder.unet.conv_in.weight" in fixed and "vae.some.weight" not in fixed
    model = MultiViewUNet(None, unet_config=UNetConfig.tiny(), init="empty", cam_output_dim=96, cam_hidden_dim=48)
    missing, unexpected = load_lightning_checkpoint(model, str(path))
    assert missing == [] and unexpected == ["not_a_real_key"]
    torch.testing.assert_close(model.state_dict()["image_encoder.unet.conv_in.weight"], params["image_encoder.unet.conv_in.weight"])


def test_prepare_latents_generator_list_follows_diffusers_randn_tensor():
    """pipeline.py:22 takes ``generator: Union[torch.Generator, List[torch.Generator]]`` and hands it to the base class's
    ``prepare_latents`` (:87-95): one generator per latent row, a list of another length raises ValueError, a one-element list
    is that generator.  Host logic only (no engine)."""
    import pytest
    from types import SimpleNamespace
    from mvd_amd.pipeline import MVDPipeline
    pipe = MVDPipeline(unet=SimpleNamespace(), scheduler=SimpleNamespace(init_noise_sigma=2.0))
    gens = [torch.Generator().manual_seed(s) for s in (11, 12, 13)]
    got = pipe.prepare_latents(3, 4, 64, 48, torch.float32, torch.device("cpu"), gens)
    assert got.shape == (3, 4, 8, 6)
    for i, s in enumerate((11, 12, 13)):
        want = 2.0 * torch.randn(1, 4, 8, 6, generator=torch.Generator().manual_seed(s))
        assert torch.equal(got[i:i + 1], want)
    with pytest.raises(ValueError, match="list of generators of length 2"):
        pipe.prepare_latents(3, 4, 64, 48, torch.float32, torch.device("cpu"), gens[:2])
    one = pipe.prepare_latents(1, 4, 64, 48, torch.float32, torch.device("cpu"), [torch.Generator().manual_seed(5)])
    assert torch.equal(one, 2.0 * torch.randn(1, 4, 8, 6, generator=torch.Generator().manual_seed(5)))
    two = pipe.prepare_latents(2, 4, 64, 48, torch.float32, torch.device("cpu"), torch.Generator().manual_seed(5))
    assert torch.equal(two, 2.0 * torch.randn(2, 4, 8, 6, generator=torch.Generator().manual_seed(5)))
    # caller-provided latents are only scaled (prepare_latents' latents= branch)
    lat = torch.ones(1, 4, 8, 6)
    assert torch.equal(pipe.prepare_latents(1, 4, 64, 48, torch.float32, torch.device("cpu"), None, latents=lat), 2.0 * lat)


def test_ddpm_step_is_the_gaussian_posterior_from_first_principles():
    """``oracle.scheduler.ddpm_step`` restates diffusers' ``DDPMScheduler.step`` (pipeline.py:161; diffusers is absent, so there is
    no fixture of it).  What CAN pin it is the published algorithm itself (Ho et al. 2020, eq. 6-7; v-prediction: Salimans & Ho
    2022): with x_t = sqrt(a_t) x0 + sqrt(1 - a_t) eps and the TRUE eps / v as the model output, the step must (i) recover x0
    exactly and (ii) return the mean of the Gaussian posterior q(x_{t-k} | x_t, x0), derived here independently by conditioning
    x_{t-k} ~ N(sqrt(a_prev) x0, (1 - a_prev) I), x_t | x_{t-k} ~ N(sqrt(a_t / a_prev) x_{t-k}, (1 - a_t / a_prev) I) (precision-
    weighted form, no shared algebra with the restatement), and (iii) add noise with exactly the posterior's variance ("fixed_small");
    the last step (t - k < 0 -> a_prev = 1) returns x0 itself.  fp64, strided schedule as ``set_timesteps`` makes it."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2          # SD's scaled-linear betas
    acp = torch.cumprod(1.0 - betas, 0)
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    eps = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    nz = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    for n_steps in (20, 50, 1000):
        k = 1000 // n_steps
        for t in (999 // k * k if n_steps < 1000 else 999, 500 // k * k, k, 0):
            a_t = acp[t]
            a_prev = acp[t - k] if t - k >= 0 else torch.tensor(1.0, dtype=torch.float64)
            x_t = a_t.sqrt() * x0 + (1 - a_t).sqrt() * eps
            v = a_t.sqrt() * eps - (1 - a_t).sqrt() * x0
            # the posterior by Gaussian conditioning (precisions add, means are precision-weighted)
            a_step = a_t / a_prev
            if t - k >= 0:
                prec_prior, prec_like = 1.0 / (1 - a_prev), a_step / (1 - a_step)
                var_post = 1.0 / (prec_prior + prec_like)
                mean_post = var_post * (prec_prior * a_prev.sqrt() * x0 + prec_like * x_t / a_step.sqrt())
            else:                                        # a_prev = 1: x_{t-k} IS x0
                var_post, mean_post = torch.tensor(0.0, dtype=torch.float64), x0
            for kind, out in (("epsilon", eps), ("v_prediction", v)):
                mean = OS.ddpm_step(out, t, x_t, acp, 1000, n_steps, kind, torch.zeros_like(nz))
                assert torch.allclose(mean, mean_post, rtol=0, atol=1e-11), (n_steps, t, kind, (mean - mean_post).abs().max())
                noisy = OS.ddpm_step(out, t, x_t, acp, 1000, n_steps, kind, nz)
                want_std = var_post.clamp_min(1e-20).sqrt() if t > 0 else torch.tensor(0.0, dtype=torch.float64)
                assert torch.allclose(noisy - mean, want_std * nz, rtol=0, atol=1e-11), (n_steps, t, kind)
