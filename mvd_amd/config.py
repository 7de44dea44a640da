"""Host-side config of the SD-2.1 UNet + MVD wrapper (diffusers names; see include/mvd_hip.h)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_heads: Tuple[int, ...] = (5, 10, 20, 20)     # diffusers "attention_head_dim" (= head count for SD2.x)
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 96

    @property
    def num_levels(self) -> int:
        return len(self.block_out_channels)

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.block_out_channels[0]

    def down_has_attn(self, i: int) -> bool:
        return i < self.num_levels - 1

    def up_has_attn(self, i: int) -> bool:
        return i > 0

    @staticmethod
    def sd21() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def tiny() -> "UNetConfig":
        return UNetConfig(block_out_channels=(64, 128, 128, 128), num_heads=(1, 2, 2, 2),
                          cross_attention_dim=128, sample_size=16)

    # ---- structure enumeration shared by the packer and the parameter trees
    def resnets(self):
        """[(diffusers key, cin, cout)] in module order (= order of the fused time_emb_proj GEMM)."""
        out = []
        prev = self.block_out_channels[0]
        for i, c in enumerate(self.block_out_channels):
            for j in range(self.layers_per_block):
                out.append((f"down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c))
            prev = c
        cm = self.block_out_channels[-1]
        out += [("mid_block.resnets.0", cm, cm), ("mid_block.resnets.1", cm, cm)]
        rev = list(reversed(self.block_out_channels))
        n = self.num_levels
        prev_out = rev[0]
        for i in range(n):
            oc, ic = rev[i], rev[min(i + 1, n - 1)]
            for j in range(self.layers_per_block + 1):
                skip = ic if j == self.layers_per_block else oc
                hid = prev_out if j == 0 else oc
                out.append((f"up_blocks.{i}.resnets.{j}", hid + skip, oc))
            prev_out = oc
        return out

    def resnet_input_split(self):
        """{diffusers key: (c0, c1)}: the two source tensors of a resnet's input -- up-block resnets read cat([hidden, skip]) (the
        engine never materialises it: GroupNorm and the 1x1 shortcut read both sources), every other resnet has c1 = 0."""
        out = {}
        n = self.num_levels
        rev = list(reversed(self.block_out_channels))
        for key, cin, _cout in self.resnets():
            if key.startswith("up_blocks."):
                i, j = int(key.split(".")[1]), int(key.split(".")[3])
                skip = rev[min(i + 1, n - 1)] if j == self.layers_per_block else rev[i]
                out[key] = (cin - skip, skip)
            else:
                out[key] = (cin, 0)
        return out

    def transformers(self):
        """[(diffusers key, feature name, channels, heads)] in module order (= ImageEncoder hook order)."""
        out = []
        for i, c in enumerate(self.block_out_channels):
            if self.down_has_attn(i):
                for j in range(self.layers_per_block):
                    out.append((f"down_blocks.{i}.attentions.{j}", f"down_block_{i}_attn_{j}", c, self.num_heads[i]))
        out.append(("mid_block.attentions.0", "mid_block_attn_0", self.block_out_channels[-1], self.num_heads[-1]))
        rev_c = list(reversed(self.block_out_channels))
        rev_h = list(reversed(self.num_heads))
        for i in range(self.num_levels):
            if self.up_has_attn(i):
                for j in range(self.layers_per_block + 1):
                    out.append((f"up_blocks.{i}.attentions.{j}", f"up_block_{i}_attn_{j}", rev_c[i], rev_h[i]))
        return out

    def modulation_hidden_dims(self):
        """mvd_unet.py:63-80 of the reference (insertion order preserved)."""
        down = list(self.block_out_channels)
        up = list(reversed(down))
        d = {}
        for i in range(self.num_levels):
            d[f"down_{i}"] = down[i]
        for i in range(self.num_levels):
            d[f"up_{i}"] = up[i]
        d["mid"] = down[-1]
        d["output"] = 4
        return d
