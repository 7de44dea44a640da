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


def test_oracle_attention_equals_torch_multihead_attention():
    """An implementation the repo did not write: ``torch.nn.MultiheadAttention`` (Vaswani et al. multi-head attention, separate
    q / k / v projections because the text width differs from the channel width) loaded with the same weights must reproduce
    ``oracle.sd21_unet.attention`` -- the contiguous-by-head channel split, the 1/sqrt(d) scale and the biased out-projection of
    diffusers' ``Attention`` + ``AttnProcessor2_0`` (SURVEY 8a layer table) -- for self-attention (ctx = h) and text cross-attention."""
    import torch
    from oracle import sd21_unet as OU
    g = torch.Generator().manual_seed(5)
    for C, heads, xdim, n_ctx in ((320, 5, 320, None), (640, 10, 1024, 77), (128, 2, 96, 7)):
        key = "blk.attn"
        p = {f"{key}.to_q.weight": torch.randn(C, C, generator=g) / C ** 0.5,
             f"{key}.to_k.weight": torch.randn(C, xdim, generator=g) / xdim ** 0.5,
             f"{key}.to_v.weight": torch.randn(C, xdim, generator=g) / xdim ** 0.5,
             f"{key}.to_out.0.weight": torch.randn(C, C, generator=g) / C ** 0.5,
             f"{key}.to_out.0.bias": torch.randn(C, generator=g)}
        h = torch.randn(2, 48, C, generator=g)
        ctx = h if n_ctx is None else torch.randn(2, n_ctx, xdim, generator=g)
        mha = torch.nn.MultiheadAttention(C, heads, bias=True, kdim=xdim, vdim=xdim, batch_first=True)
        with torch.no_grad():
            if xdim == C:                                   # same widths: torch keeps one stacked in-projection
                mha.in_proj_weight.copy_(torch.cat([p[f"{key}.to_q.weight"], p[f"{key}.to_k.weight"], p[f"{key}.to_v.weight"]], 0))
            else:
                mha.q_proj_weight.copy_(p[f"{key}.to_q.weight"])
                mha.k_proj_weight.copy_(p[f"{key}.to_k.weight"])
                mha.v_proj_weight.copy_(p[f"{key}.to_v.weight"])
            mha.in_proj_bias.zero_()                        # to_q / to_k / to_v carry no bias
            mha.out_proj.weight.copy_(p[f"{key}.to_out.0.weight"])
            mha.out_proj.bias.copy_(p[f"{key}.to_out.0.bias"])
            want, _ = mha(h, ctx, ctx, need_weights=False)
            got = OU.attention(p, key, h, ctx, heads)
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (C, heads, (got - want).abs().max().item())
