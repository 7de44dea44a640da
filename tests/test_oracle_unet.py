"""Guard rails for the (diffusers-boundary, parity-unpinned) UNet restatement."""
import math

import torch

from oracle import mvd as M
from oracle import sd21_unet as U


def test_sd21_param_count_identity():
    shapes = U.param_shapes(U.UNetConfig.sd21())
    assert sum(math.prod(s) for s in shapes.values()) == 865_910_724
    assert len(shapes) == 686


def test_sd21_key_schema_samples():
    s = U.param_shapes(U.UNetConfig.sd21())
    assert s["down_blocks.1.resnets.0.conv_shortcut.weight"] == (640, 320, 1, 1)
    assert s["up_blocks.1.resnets.2.conv1.weight"] == (1280, 1920, 3, 3)
    assert s["up_blocks.3.resnets.0.norm1.weight"] == (960,)
    assert s["down_blocks.0.attentions.1.transformer_blocks.0.attn2.to_k.weight"] == (320, 1024)
    assert s["mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight"] == (10240, 1280)
    assert "down_blocks.3.attentions.0.norm.weight" not in s
    assert "up_blocks.0.attentions.0.norm.weight" not in s
    assert "down_blocks.3.downsamplers.0.conv.weight" not in s
    assert "up_blocks.3.upsamplers.0.conv.weight" not in s


def test_tiny_forward_and_features():
    cfg = U.UNetConfig.tiny()
    p = U.init_params(cfg, 0)
    x = torch.randn(2, 4, 16, 16)
    text = torch.randn(2, 7, cfg.cross_attention_dim)
    cap = {}
    y = U.unet_forward(p, cfg, x, torch.tensor(500), text, capture=cap)
    assert y.shape == (2, 4, 16, 16) and torch.isfinite(y).all()
    assert list(cap) == U.feature_names(cfg) and len(cap) == 16
    assert cap["down_block_0_attn_0"].shape == (2, 64, 16, 16)
    assert cap["mid_block_attn_0"].shape == (2, 128, 2, 2)
    # batch independence of the plain UNet
    y0 = U.unet_forward(p, cfg, x[:1], torch.tensor(500), text[:1])
    torch.testing.assert_close(y0, y[:1], rtol=1e-4, atol=1e-4)


def _mv_inputs(cfg, B, g):
    x = torch.randn(B, 4, 16, 16, generator=g)
    text = torch.randn(B, 5, cfg.cross_attention_dim, generator=g)
    src = torch.eye(4).repeat(B, 1, 1)
    tgt = torch.eye(4).repeat(B, 1, 1)
    tgt[:, :3, 3] = torch.randn(B, 3, generator=g)
    lat = 0.18215 * torch.randn(B, 4, 16, 16, generator=g)
    return x, text, src, tgt, lat


def test_multiview_forward_branches():
    cfg = U.UNetConfig.tiny()
    params = M.init_mvd_params(cfg, 0, cam_dim=96, cam_hidden=48)
    g = torch.Generator().manual_seed(0)
    x, text, src, tgt, lat = _mv_inputs(cfg, 2, g)
    proj = M.draw_fourier_projection(96, g)
    t = torch.tensor(321)
    base = M.multiview_unet_forward(params, cfg, x, t, text)
    plain = U.unet_forward(M._sub(params, "base_unet."), cfg, x, t, text)
    assert torch.equal(base, plain)                       # both branches off == plain UNet
    feats = {}
    full = M.multiview_unet_forward(params, cfg, x, t, text, src, tgt, lat, fourier_proj=proj,
                                    features_out=feats)
    assert len(feats) == 16 and not torch.allclose(full, base)
    cam_only = M.multiview_unet_forward(params, cfg, x, t, text, src, tgt, None, fourier_proj=proj)
    img_only = M.multiview_unet_forward(params, cfg, x, t, text, None, None, lat)
    assert not torch.allclose(cam_only, base) and not torch.allclose(img_only, base)
    # switches override inputs (mvd_unet.py:241, 269)
    off = M.multiview_unet_forward(params, cfg, x, t, text, src, tgt, lat, fourier_proj=proj,
                                   use_camera_conditioning=False, use_image_conditioning=False)
    assert torch.equal(off, base)


def test_multiview_cfg_batch_mismatch_q4():
    """CFG: sample batch 2B, text B (repeated), source latents B -> adapter K/V re-chunked."""
    cfg = U.UNetConfig.tiny()
    params = M.init_mvd_params(cfg, 0, cam_dim=96, cam_hidden=48)
    g = torch.Generator().manual_seed(1)
    x, text, src, tgt, lat = _mv_inputs(cfg, 1, g)
    y = M.multiview_unet_forward(params, cfg, torch.cat([x, x]), torch.tensor(10), torch.cat([text, text]),
                                 None, None, lat)
    assert y.shape == (2, 4, 16, 16) and torch.isfinite(y).all()
    assert not torch.allclose(y[0], y[1])   # rows see different halves of the reference tokens
