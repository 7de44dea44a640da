"""Oracle: diffusers-0.32.2 ``UNet2DConditionModel.forward`` (SD-2.1 layout), CPU fp32.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED at the
diffusers boundary: diffusers (pinned 0.32.2, /root/reference/uv.lock:722-723)
is a third-party dependency absent from /root/reference, and the reference holds
no tests for it.  This file restates its published algorithm for the SD-2.1
config used at /root/reference/src/models/mvd_unet.py:46-52 and
/root/reference/src/models/image_encoder.py:18-22; the layer table it follows is
SURVEY.md section 8a.  Guard rails: parameter count == 865,910,724 and the
state-dict key schema (tests/test_oracle_unet.py).

The forward is functional over a flat ``{diffusers_key: tensor}`` state dict so
the same dict can be fed to the HIP engine.  Two extension points mirror how the
reference re-enters diffusers:

* ``attn_hook(name, kind, hidden_states, attn_out)`` -- called after every
  ``Attention`` (kind "self"/"cross"), returns the (possibly adapter-augmented)
  attention output.  This is the attention-processor protocol of
  /root/reference/src/models/attention.py:48-59.
* ``block_hook(name, tensor)`` -- called on each down/mid/up block output, i.e.
  the ``nn.Module`` forward hooks of /root/reference/src/models/mvd_unet.py:354-380.
* ``capture`` dict -- filled with every Transformer2DModel output (NCHW), i.e.
  the 16 hooks of /root/reference/src/models/image_encoder.py:36-84.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    """Subset of the diffusers UNet2DConditionModel config that SD-2.1 exercises."""

    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # diffusers calls this ``attention_head_dim`` but for SD2.x it is the head COUNT.
    num_heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 96
    # down: CrossAttn x (n-1) + plain Down; up: plain Up + CrossAttn x (n-1)
    time_embed_mult: int = 4

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * self.time_embed_mult

    @property
    def num_levels(self) -> int:
        return len(self.block_out_channels)

    def down_has_attn(self, i: int) -> bool:
        return i < self.num_levels - 1

    def up_has_attn(self, i: int) -> bool:
        return i > 0

    @staticmethod
    def sd21() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def tiny() -> "UNetConfig":
        """Reduced config with the same topology (head_dim stays 64)."""
        return UNetConfig(
            block_out_channels=(64, 128, 128, 128),
            num_heads=(1, 2, 2, 2),
            cross_attention_dim=128,
            sample_size=16,
        )


# ----------------------------------------------------------------------------
# parameter schema
# ----------------------------------------------------------------------------

def _resnet_shapes(p: str, cin: int, cout: int, temb: int, out: Dict[str, Tuple[int, ...]]):
    out[f"{p}.norm1.weight"] = (cin,)
    out[f"{p}.norm1.bias"] = (cin,)
    out[f"{p}.conv1.weight"] = (cout, cin, 3, 3)
    out[f"{p}.conv1.bias"] = (cout,)
    out[f"{p}.time_emb_proj.weight"] = (cout, temb)
    out[f"{p}.time_emb_proj.bias"] = (cout,)
    out[f"{p}.norm2.weight"] = (cout,)
    out[f"{p}.norm2.bias"] = (cout,)
    out[f"{p}.conv2.weight"] = (cout, cout, 3, 3)
    out[f"{p}.conv2.bias"] = (cout,)
    if cin != cout:
        out[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[f"{p}.conv_shortcut.bias"] = (cout,)


def _transformer_shapes(p: str, c: int, xdim: int, out: Dict[str, Tuple[int, ...]]):
    out[f"{p}.norm.weight"] = (c,)
    out[f"{p}.norm.bias"] = (c,)
    out[f"{p}.proj_in.weight"] = (c, c)
    out[f"{p}.proj_in.bias"] = (c,)
    b = f"{p}.transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        out[f"{b}.{n}.weight"] = (c,)
        out[f"{b}.{n}.bias"] = (c,)
    for a, kdim in (("attn1", c), ("attn2", xdim)):
        out[f"{b}.{a}.to_q.weight"] = (c, c)
        out[f"{b}.{a}.to_k.weight"] = (c, kdim)
        out[f"{b}.{a}.to_v.weight"] = (c, kdim)
        out[f"{b}.{a}.to_out.0.weight"] = (c, c)
        out[f"{b}.{a}.to_out.0.bias"] = (c,)
    out[f"{b}.ff.net.0.proj.weight"] = (8 * c, c)
    out[f"{b}.ff.net.0.proj.bias"] = (8 * c,)
    out[f"{b}.ff.net.2.weight"] = (c, 4 * c)
    out[f"{b}.ff.net.2.bias"] = (c,)
    out[f"{p}.proj_out.weight"] = (c, c)
    out[f"{p}.proj_out.bias"] = (c,)


def up_block_resnet_channels(cfg: UNetConfig) -> List[List[Tuple[int, int, int]]]:
    """Per up block, per resnet: (hidden_in, skip_in, out) channel counts.

    diffusers get_up_block wiring: reversed block_out_channels; resnet j of block i
    takes cat([hidden, skip]) with skip = in_channels of the mirrored down block for
    the last resnet, else the block's out channels.
    """
    rev = list(reversed(cfg.block_out_channels))
    n = cfg.num_levels
    res = []
    prev_out = rev[0]
    for i in range(n):
        out_c = rev[i]
        in_c = rev[min(i + 1, n - 1)]
        blocks = []
        for j in range(cfg.layers_per_block + 1):
            skip = in_c if j == cfg.layers_per_block else out_c
            hid = prev_out if j == 0 else out_c
            blocks.append((hid, skip, out_c))
        res.append(blocks)
        prev_out = out_c
    return res


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """diffusers state-dict keys -> shapes (insertion order = module order)."""
    s: Dict[str, Tuple[int, ...]] = {}
    c0 = cfg.block_out_channels[0]
    temb = cfg.time_embed_dim
    s["conv_in.weight"] = (c0, cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (c0,)
    s["time_embedding.linear_1.weight"] = (temb, c0)
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    prev = c0
    for i, c in enumerate(cfg.block_out_channels):
        for j in range(cfg.layers_per_block):
            _resnet_shapes(f"down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c, temb, s)
            if cfg.down_has_attn(i):
                _transformer_shapes(f"down_blocks.{i}.attentions.{j}", c, cfg.cross_attention_dim, s)
        if i < cfg.num_levels - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
        prev = c
    cm = cfg.block_out_channels[-1]
    _resnet_shapes("mid_block.resnets.0", cm, cm, temb, s)
    _transformer_shapes("mid_block.attentions.0", cm, cfg.cross_attention_dim, s)
    _resnet_shapes("mid_block.resnets.1", cm, cm, temb, s)
    for i, blocks in enumerate(up_block_resnet_channels(cfg)):
        for j, (hid, skip, out_c) in enumerate(blocks):
            _resnet_shapes(f"up_blocks.{i}.resnets.{j}", hid + skip, out_c, temb, s)
            if cfg.up_has_attn(i):
                _transformer_shapes(f"up_blocks.{i}.attentions.{j}", out_c, cfg.cross_attention_dim, s)
        if i < cfg.num_levels - 1:
            oc = blocks[0][2]
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (oc, oc, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (oc,)
    s["conv_norm_out.weight"] = (c0,)
    s["conv_norm_out.bias"] = (c0,)
    s["conv_out.weight"] = (cfg.out_channels, c0, 3, 3)
    s["conv_out.bias"] = (cfg.out_channels,)
    return s


def init_params(cfg: UNetConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (no pretrained SD-2.1 weights exist offline).

    Variance-preserving init (std = 1/sqrt(fan_in)) so that every branch
    contributes O(1) to the residual stream -- a N(0, 0.02) init would let the
    skip path hide errors in attention / FF branches.  Norm scales ~ 1 + 0.1 N,
    biases ~ 0.1 N so the affine terms are exercised.
    """
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if ".norm" in name or name.startswith("conv_norm_out"):
            if name.endswith(".weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                t = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        out[name] = t.to(dtype)
    return out


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------

def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    freqs = torch.exp(exponent)
    ang = timesteps[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)  # [cos | sin]


def _gn(x, p, key, groups, eps):
    return F.group_norm(x, groups, p[f"{key}.weight"], p[f"{key}.bias"], eps)


def resnet_block(p, key, x, temb_act, groups, eps):
    h = F.silu(_gn(x, p, f"{key}.norm1", groups, eps))
    h = F.conv2d(h, p[f"{key}.conv1.weight"], p[f"{key}.conv1.bias"], padding=1)
    t = F.linear(temb_act, p[f"{key}.time_emb_proj.weight"], p[f"{key}.time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.silu(_gn(h, p, f"{key}.norm2", groups, eps))
    h = F.conv2d(h, p[f"{key}.conv2.weight"], p[f"{key}.conv2.bias"], padding=1)
    if f"{key}.conv_shortcut.weight" in p:
        x = F.conv2d(x, p[f"{key}.conv_shortcut.weight"], p[f"{key}.conv_shortcut.bias"])
    return x + h


def attention(p, key, h, ctx, heads):
    """diffusers ``AttnProcessor2_0``: to_q/k/v (no bias), SDPA scale 1/sqrt(d), to_out[0]."""
    B, N, C = h.shape
    d = C // heads
    q = F.linear(h, p[f"{key}.to_q.weight"]).view(B, N, heads, d).transpose(1, 2)
    k = F.linear(ctx, p[f"{key}.to_k.weight"]).view(B, -1, heads, d).transpose(1, 2)
    v = F.linear(ctx, p[f"{key}.to_v.weight"]).view(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, p[f"{key}.to_out.0.weight"], p[f"{key}.to_out.0.bias"])


AttnHook = Callable[[str, str, torch.Tensor, torch.Tensor], torch.Tensor]


def transformer_2d(p, key, x, text, heads, groups, attn_hook: Optional[AttnHook], hook_name: str):
    B, C, H, W = x.shape
    res = x
    h = _gn(x, p, f"{key}.norm", groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = F.linear(h, p[f"{key}.proj_in.weight"], p[f"{key}.proj_in.bias"])
    b = f"{key}.transformer_blocks.0"
    n = F.layer_norm(h, (C,), p[f"{b}.norm1.weight"], p[f"{b}.norm1.bias"], 1e-5)
    a = attention(p, f"{b}.attn1", n, n, heads)
    if attn_hook is not None:
        a = attn_hook(hook_name, "self", n, a)
    h = h + a
    n = F.layer_norm(h, (C,), p[f"{b}.norm2.weight"], p[f"{b}.norm2.bias"], 1e-5)
    a = attention(p, f"{b}.attn2", n, text, heads)
    if attn_hook is not None:
        a = attn_hook(hook_name, "cross", n, a)
    h = h + a
    n = F.layer_norm(h, (C,), p[f"{b}.norm3.weight"], p[f"{b}.norm3.bias"], 1e-5)
    f = F.linear(n, p[f"{b}.ff.net.0.proj.weight"], p[f"{b}.ff.net.0.proj.bias"])
    val, gate = f.chunk(2, dim=-1)
    f = val * F.gelu(gate)
    f = F.linear(f, p[f"{b}.ff.net.2.weight"], p[f"{b}.ff.net.2.bias"])
    h = h + f
    h = F.linear(h, p[f"{key}.proj_out.weight"], p[f"{key}.proj_out.bias"])
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + res


def unet_forward(
    p: Dict[str, torch.Tensor],
    cfg: UNetConfig,
    sample: torch.Tensor,
    timestep: torch.Tensor,
    text: torch.Tensor,
    attn_hook: Optional[AttnHook] = None,
    block_hook: Optional[Callable[[str, torch.Tensor], torch.Tensor]] = None,
    capture: Optional[Dict[str, torch.Tensor]] = None,
) -> torch.Tensor:
    """One UNet forward.  ``sample`` (B,Cin,H,W), ``text`` (B,L,xdim), ``timestep`` 0-d or (B,)."""
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    B = sample.shape[0]
    t = torch.as_tensor(timestep)
    if t.ndim == 0:
        t = t[None]
    t = t.expand(B)
    temb = timestep_embedding(t, cfg.block_out_channels[0]).to(sample.dtype)
    temb = F.linear(temb, p["time_embedding.linear_1.weight"], p["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), p["time_embedding.linear_2.weight"], p["time_embedding.linear_2.bias"])
    temb_act = F.silu(temb)

    def cap(name, x):
        if capture is not None:
            capture[name] = x
        return x

    def bh(name, x):
        return block_hook(name, x) if block_hook is not None else x

    h = F.conv2d(sample, p["conv_in.weight"], p["conv_in.bias"], padding=1)
    skips = [h]
    for i in range(cfg.num_levels):
        for j in range(cfg.layers_per_block):
            h = resnet_block(p, f"down_blocks.{i}.resnets.{j}", h, temb_act, G, eps)
            if cfg.down_has_attn(i):
                name = f"down_block_{i}_attn_{j}"
                h = cap(name, transformer_2d(p, f"down_blocks.{i}.attentions.{j}", h, text,
                                             cfg.num_heads[i], G, attn_hook, name))
            skips.append(h)
        if i < cfg.num_levels - 1:
            h = F.conv2d(h, p[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                         p[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            skips.append(h)
        # forward hook modulates the returned hidden_states only, never the skip tuple (Q6)
        h = bh(f"down_{i}", h)

    h = resnet_block(p, "mid_block.resnets.0", h, temb_act, G, eps)
    h = cap("mid_block_attn_0", transformer_2d(p, "mid_block.attentions.0", h, text,
                                               cfg.num_heads[-1], G, attn_hook, "mid_block_attn_0"))
    h = resnet_block(p, "mid_block.resnets.1", h, temb_act, G, eps)
    h = bh("mid_0", h)

    rev_heads = list(reversed(cfg.num_heads))
    for i in range(cfg.num_levels):
        for j in range(cfg.layers_per_block + 1):
            skip = skips.pop()
            h = torch.cat([h, skip], dim=1)
            h = resnet_block(p, f"up_blocks.{i}.resnets.{j}", h, temb_act, G, eps)
            if cfg.up_has_attn(i):
                name = f"up_block_{i}_attn_{j}"
                h = cap(name, transformer_2d(p, f"up_blocks.{i}.attentions.{j}", h, text,
                                             rev_heads[i], G, attn_hook, name))
        if i < cfg.num_levels - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, p[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                         p[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        h = bh(f"up_{i}", h)

    h = F.silu(_gn(h, p, "conv_norm_out", G, eps))
    return F.conv2d(h, p["conv_out.weight"], p["conv_out.bias"], padding=1)


def feature_names(cfg: UNetConfig) -> List[str]:
    """Names (and order) of the Transformer2DModel outputs the ImageEncoder hooks capture."""
    names = []
    for i in range(cfg.num_levels):
        if cfg.down_has_attn(i):
            names += [f"down_block_{i}_attn_{j}" for j in range(cfg.layers_per_block)]
    names.append("mid_block_attn_0")
    for i in range(cfg.num_levels):
        if cfg.up_has_attn(i):
            names += [f"up_block_{i}_attn_{j}" for j in range(cfg.layers_per_block + 1)]
    return names
