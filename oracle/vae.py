"""Oracle: CPU fp32 restatement of diffusers-0.32.2 ``AutoencoderKL`` (the SD-2.1 VAE).  TEST INFRASTRUCTURE ONLY.

The reference reaches the VAE only through diffusers' ``StableDiffusionPipeline`` (``vae.encode(x).latent_dist.sample()``,
/root/reference/src/models/pipeline.py:115-116; ``vae.decode(z).sample``, :171-176).  diffusers is neither vendored nor
installed and the reference holds no fixtures at this boundary: PARITY UNPINNED (like oracle/sd21_unet.py).  Guard rail:
the parameter count of the SD-2.1 config reproduces the published 83,653,863 (tests/test_oracle_vae.py).

Published algorithm restated (diffusers/models/autoencoders/{autoencoder_kl,vae}.py, unet_2d_blocks.py, resnet.py,
attention_processor.py at 0.32.2), state-dict keys as diffusers names them:

* ``Encoder``: conv_in -> DownEncoderBlock2D x4 (R resnets; all but the last followed by ``Downsample2D(padding=0)`` =
  ``F.pad(x, (0,1,0,1))`` + 3x3 stride-2 conv) -> UNetMidBlock2D (resnet, attention, resnet) -> GroupNorm -> SiLU -> conv_out
  (2 x latent channels); then ``quant_conv`` 1x1.
* ``Decoder``: ``post_quant_conv`` 1x1 -> conv_in -> UNetMidBlock2D -> UpDecoderBlock2D x4 (R+1 resnets; all but the last
  followed by nearest-2x + 3x3 conv) -> GroupNorm -> SiLU -> conv_out.
* ``ResnetBlock2D`` (temb_channels=None): GN -> SiLU -> conv1 -> GN -> SiLU -> conv2, + (1x1 ``conv_shortcut(x)`` if the channel
  count changes else x); eps 1e-6, 32 groups.
* mid ``Attention``: 1 head of C channels, ``group_norm`` (32 groups, eps 1e-6) on the (B, C, HW) view, to_q/to_k/to_v/to_out.0
  with bias, softmax(q k^T / sqrt(C)) v, residual connection, rescale_output_factor 1.
* ``DiagonalGaussianDistribution``: mean, logvar = chunk(moments, 2, dim=1); logvar clamped to [-30, 20];
  sample = mean + exp(0.5 logvar) * noise.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215

    @staticmethod
    def sd21() -> "VAEConfig":
        return VAEConfig()

    @staticmethod
    def tiny() -> "VAEConfig":
        return VAEConfig(block_out_channels=(64, 64, 128), layers_per_block=1)


def _conv(out: Dict, k: str, co: int, ci: int, ks: int):
    out[f"{k}.weight"] = (co, ci, ks, ks)
    out[f"{k}.bias"] = (co,)


def _norm(out: Dict, k: str, c: int):
    out[f"{k}.weight"] = (c,)
    out[f"{k}.bias"] = (c,)


def _resnet_shapes(out: Dict, k: str, ci: int, co: int):
    _norm(out, f"{k}.norm1", ci); _conv(out, f"{k}.conv1", co, ci, 3)
    _norm(out, f"{k}.norm2", co); _conv(out, f"{k}.conv2", co, co, 3)
    if ci != co:
        _conv(out, f"{k}.conv_shortcut", co, ci, 1)


def _mid_shapes(out: Dict, k: str, c: int):
    _resnet_shapes(out, f"{k}.resnets.0", c, c)
    _norm(out, f"{k}.attentions.0.group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{k}.attentions.0.{n}.weight"] = (c, c)
        out[f"{k}.attentions.0.{n}.bias"] = (c,)
    _resnet_shapes(out, f"{k}.resnets.1", c, c)


def param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    out: Dict[str, Tuple[int, ...]] = {}
    ch = cfg.block_out_channels
    n = len(ch)
    _conv(out, "encoder.conv_in", ch[0], cfg.in_channels, 3)
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _resnet_shapes(out, f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        prev = c
        if i + 1 < n:
            _conv(out, f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
    _mid_shapes(out, "encoder.mid_block", ch[-1])
    _norm(out, "encoder.conv_norm_out", ch[-1])
    _conv(out, "encoder.conv_out", 2 * cfg.latent_channels, ch[-1], 3)
    _conv(out, "quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    _conv(out, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    _conv(out, "decoder.conv_in", ch[-1], cfg.latent_channels, 3)
    _mid_shapes(out, "decoder.mid_block", ch[-1])
    rev = list(reversed(ch))
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            _resnet_shapes(out, f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        prev = c
        if i + 1 < n:
            _conv(out, f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    _norm(out, "decoder.conv_norm_out", ch[0])
    _conv(out, "decoder.conv_out", cfg.in_channels, ch[0], 3)
    return out


def init_params(cfg: VAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded variance-preserving synthetic weights (std 1/sqrt(fan_in), norm scales 1 +- 0.1) under the diffusers key names."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k, s in param_shapes(cfg).items():
        if len(s) >= 2:
            fan_in = int(torch.tensor(s[1:]).prod())
            p[k] = torch.randn(s, generator=g) / math.sqrt(fan_in)
        elif k.endswith("weight"):
            p[k] = 1.0 + 0.1 * torch.randn(s, generator=g)
        else:
            p[k] = 0.1 * torch.randn(s, generator=g)
    return p


def _gn(p, k, x, cfg):
    return F.group_norm(x, cfg.norm_num_groups, p[f"{k}.weight"], p[f"{k}.bias"], cfg.norm_eps)


def _resnet(p, k, x, cfg):
    h = F.conv2d(F.silu(_gn(p, f"{k}.norm1", x, cfg)), p[f"{k}.conv1.weight"], p[f"{k}.conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(p, f"{k}.norm2", h, cfg)), p[f"{k}.conv2.weight"], p[f"{k}.conv2.bias"], padding=1)
    if f"{k}.conv_shortcut.weight" in p:
        x = F.conv2d(x, p[f"{k}.conv_shortcut.weight"], p[f"{k}.conv_shortcut.bias"])
    return x + h


def _attention(p, k, x, cfg):
    B, C, H, W = x.shape
    h = F.group_norm(x.view(B, C, H * W), cfg.norm_num_groups, p[f"{k}.group_norm.weight"], p[f"{k}.group_norm.bias"], cfg.norm_eps)
    h = h.transpose(1, 2)                                                   # (B, HW, C)
    q = F.linear(h, p[f"{k}.to_q.weight"], p[f"{k}.to_q.bias"])
    kk = F.linear(h, p[f"{k}.to_k.weight"], p[f"{k}.to_k.bias"])
    v = F.linear(h, p[f"{k}.to_v.weight"], p[f"{k}.to_v.bias"])
    a = torch.softmax(q @ kk.transpose(1, 2) / math.sqrt(C), dim=-1) @ v      # one head of C channels
    a = F.linear(a, p[f"{k}.to_out.0.weight"], p[f"{k}.to_out.0.bias"])
    return x + a.transpose(1, 2).reshape(B, C, H, W)


def _mid(p, k, x, cfg):
    x = _resnet(p, f"{k}.resnets.0", x, cfg)
    x = _attention(p, f"{k}.attentions.0", x, cfg)
    return _resnet(p, f"{k}.resnets.1", x, cfg)


def encode_moments(p: Dict[str, torch.Tensor], cfg: VAEConfig, image: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.encode(x): the (mean | logvar) moments the latent distribution is built from."""
    n = len(cfg.block_out_channels)
    x = F.conv2d(image, p["encoder.conv_in.weight"], p["encoder.conv_in.bias"], padding=1)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            x = _resnet(p, f"encoder.down_blocks.{i}.resnets.{j}", x, cfg)
        if i + 1 < n:
            k = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), p[f"{k}.weight"], p[f"{k}.bias"], stride=2)
    x = _mid(p, "encoder.mid_block", x, cfg)
    x = F.silu(_gn(p, "encoder.conv_norm_out", x, cfg))
    x = F.conv2d(x, p["encoder.conv_out.weight"], p["encoder.conv_out.bias"], padding=1)
    return F.conv2d(x, p["quant_conv.weight"], p["quant_conv.bias"])


def sample_latents(moments: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    mean, logvar = moments.chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise


def decode(p: Dict[str, torch.Tensor], cfg: VAEConfig, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode(z).sample"""
    n = len(cfg.block_out_channels)
    x = F.conv2d(z, p["post_quant_conv.weight"], p["post_quant_conv.bias"])
    x = F.conv2d(x, p["decoder.conv_in.weight"], p["decoder.conv_in.bias"], padding=1)
    x = _mid(p, "decoder.mid_block", x, cfg)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = _resnet(p, f"decoder.up_blocks.{i}.resnets.{j}", x, cfg)
        if i + 1 < n:
            k = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            x = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), p[f"{k}.weight"], p[f"{k}.bias"], padding=1)
    x = F.silu(_gn(p, "decoder.conv_norm_out", x, cfg))
    return F.conv2d(x, p["decoder.conv_out.weight"], p["decoder.conv_out.bias"], padding=1)
