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
