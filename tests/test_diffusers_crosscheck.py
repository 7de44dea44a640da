"""Optional cross-check of the UNPINNED oracles against diffusers itself.

``oracle/sd21_unet.py`` and ``oracle/vae.py`` restate diffusers 0.32.2 (`uv.lock:722-723` of the reference), which is
neither vendored nor installed in the build image: these tests SKIP there.  Wherever ``diffusers`` is importable they
build the diffusers module from the same config with random weights, load that state dict into the oracle (the key names
are diffusers' own) and compare the two fp32 CPU forwards -- the only thing that can pin those oracles outside the image.
Tolerance 1e-4 * max|ref| (same arithmetic, different summation order).
"""
import pytest
import torch

diffusers = pytest.importorskip("diffusers", reason="diffusers is not installed in this image (parity of the UNet / VAE oracles stays unpinned)")


def _close(a, b, tol=1e-4):
    err = (a - b).abs().max().item()
    assert err <= tol * b.abs().max().item() + 1e-6, (err, b.abs().max().item())


def test_unet_oracle_matches_diffusers():
    from oracle import sd21_unet as OU
    cfg = OU.UNetConfig.tiny()
    n = cfg.num_levels
    ref = diffusers.UNet2DConditionModel(
        sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
        down_block_types=("CrossAttnDownBlock2D",) * (n - 1) + ("DownBlock2D",),
        up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * (n - 1),
        block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
        cross_attention_dim=cfg.cross_attention_dim, attention_head_dim=cfg.num_heads,
        norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps, use_linear_projection=True,
        dual_cross_attention=False, only_cross_attention=False, upcast_attention=False).eval()
    sd = {k: v.detach().float() for k, v in ref.state_dict().items()}
    assert set(sd) == set(OU.param_shapes(cfg)), "state-dict key schema differs from diffusers"
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, cfg.in_channels, 16, 16, generator=g)
    text = torch.randn(2, 7, cfg.cross_attention_dim, generator=g)
    t = torch.tensor([10, 500])
    with torch.no_grad():
        want = ref(x, t, encoder_hidden_states=text).sample
        got = OU.unet_forward(sd, cfg, x, t, text)
    _close(got, want)


def test_vae_oracle_matches_diffusers():
    from oracle import vae as OV
    cfg = OV.VAEConfig.tiny()
    n = len(cfg.block_out_channels)
    ref = diffusers.AutoencoderKL(
        in_channels=cfg.in_channels, out_channels=cfg.in_channels, latent_channels=cfg.latent_channels,
        down_block_types=("DownEncoderBlock2D",) * n, up_block_types=("UpDecoderBlock2D",) * n,
        block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
        norm_num_groups=cfg.norm_num_groups, sample_size=32).eval()
    sd = {k: v.detach().float() for k, v in ref.state_dict().items()}
    assert set(sd) == set(OV.param_shapes(cfg)), "state-dict key schema differs from diffusers"
    g = torch.Generator().manual_seed(1)
    img = torch.randn(1, 3, 32, 32, generator=g)
    with torch.no_grad():
        post = ref.encode(img).latent_dist
        moments = OV.encode_moments(sd, cfg, img)
        _close(moments[:, : cfg.latent_channels], post.mean)
        z = post.mean
        _close(OV.decode(sd, cfg, z), ref.decode(z).sample)
