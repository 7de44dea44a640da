"""Parameter containers mirroring diffusers' ``UNet2DConditionModel`` module tree (SD-2.x layout).

These modules exist to own parameters under the exact diffusers state-dict key names
(so Lightning checkpoints of the reference load with ``load_state_dict``), to expose the
attributes the reference pokes at (``attn.to_q.in_features``, ``attn.heads``,
``attn.processor``, ``block.attentions[j].transformer_blocks``) and nothing else: they have
NO torch forward -- the arithmetic lives in libmvd_hip.so.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch.nn as nn

from .attention import AttnProcessor2_0HIP
from .config import UNetConfig


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is a parameter container; the forward runs in the MVD HIP engine")



class Attention(_NoForward):
    def __init__(self, query_dim, cross_dim, heads, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.processor = AttnProcessor2_0HIP()     # diffusers' default processor, on the HIP kernels


class GEGLU(_NoForward):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)


class FeedForward(_NoForward):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])


class BasicTransformerBlock(_NoForward):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)


class Transformer2DModel(_NoForward):
    def __init__(self, dim, heads, cross_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, cross_dim)])
        self.proj_out = nn.Linear(dim, dim)


class ResnetBlock2D(_NoForward):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)


class _Sampler(_NoForward):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=1)


class _Block(_NoForward):
    pass


class TimestepEmbedding(_NoForward):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)


class UNet2DConditionParams(_NoForward):
    """Module tree with diffusers key names for the config in ``cfg``."""

    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        self.config = SimpleNamespace(
            sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
            block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block,
            attention_head_dim=tuple(cfg.num_heads), cross_attention_dim=cfg.cross_attention_dim,
            norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps)
        G, eps, X, T = cfg.norm_num_groups, cfg.norm_eps, cfg.cross_attention_dim, cfg.time_embed_dim
        c0 = cfg.block_out_channels[0]
        self.conv_in = nn.Conv2d(cfg.in_channels, c0, 3, padding=1)
        self.time_embedding = TimestepEmbedding(c0, T)
        res = {k: (ci, co) for k, ci, co in cfg.resnets()}
        tr = {k: (c, h) for k, _f, c, h in cfg.transformers()}
        n = cfg.num_levels
        self.down_blocks = nn.ModuleList()
        for i in range(n):
            b = _Block()
            b.resnets = nn.ModuleList([ResnetBlock2D(*res[f"down_blocks.{i}.resnets.{j}"], T, G, eps)
                                       for j in range(cfg.layers_per_block)])
            if cfg.down_has_attn(i):
                b.attentions = nn.ModuleList([Transformer2DModel(*tr[f"down_blocks.{i}.attentions.{j}"], X, G)
                                              for j in range(cfg.layers_per_block)])
            if i < n - 1:
                b.downsamplers = nn.ModuleList([_Sampler(cfg.block_out_channels[i], 2)])
            self.down_blocks.append(b)
        m = _Block()
        m.resnets = nn.ModuleList([ResnetBlock2D(*res["mid_block.resnets.0"], T, G, eps),
                                   ResnetBlock2D(*res["mid_block.resnets.1"], T, G, eps)])
        m.attentions = nn.ModuleList([Transformer2DModel(*tr["mid_block.attentions.0"], X, G)])
        self.mid_block = m
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(cfg.block_out_channels))
        for i in range(n):
            b = _Block()
            b.resnets = nn.ModuleList([ResnetBlock2D(*res[f"up_blocks.{i}.resnets.{j}"], T, G, eps)
                                       for j in range(cfg.layers_per_block + 1)])
            if cfg.up_has_attn(i):
                b.attentions = nn.ModuleList([Transformer2DModel(*tr[f"up_blocks.{i}.attentions.{j}"], X, G)
                                              for j in range(cfg.layers_per_block + 1)])
            if i < n - 1:
                b.upsamplers = nn.ModuleList([_Sampler(rev[i], 1)])
            self.up_blocks.append(b)
        self.conv_norm_out = nn.GroupNorm(G, c0, eps=eps)
        self.conv_out = nn.Conv2d(c0, cfg.out_channels, 3, padding=1)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def enable_gradient_checkpointing(self):   # accepted for API compatibility; inference-only engine
        pass
