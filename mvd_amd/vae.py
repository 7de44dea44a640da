"""AutoencoderKL (the SD-2.1 VAE) behind the diffusers object protocol the reference pipeline uses
(/root/reference/src/models/pipeline.py:115-116 ``vae.encode(x).latent_dist.sample() * vae.config.scaling_factor`` and
:171-176 ``vae.decode(latents / scaling_factor).sample``), on the HIP kernels of libmvd_hip.so.

``AutoencoderKLHIP`` is an ``nn.Module`` whose parameters carry diffusers' state-dict key names (``encoder.conv_in.weight``
... ``decoder.conv_out.bias``, ``quant_conv``, ``post_quant_conv``), so ``diffusion_pytorch_model.safetensors`` of a local
snapshot loads with ``load_state_dict``.  It has no torch forward: ``encode`` / ``decode`` hand device pointers to the C ABI
(``mvd_vae_encode`` / ``mvd_vae_decode``); there is no CPU fallback.

Weight slots (``pack_vae``): 3x3 convs ``[Cout][Cin/64][ky][kx][64]`` bf16 (as the UNet's); a resnet's ``conv2.w`` carries the
1x1 ``conv_shortcut`` along K (biases summed); ``conv_in.w`` ``[C][64]`` (K = 9*Cin zero padded); ``conv_out.w`` tap-major
``[Cout][ky][kx][Cin]``; attention ``q/k/v/out`` ``[C][C]`` bf16; norms, biases and the two 1x1 quant convs fp32.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .packing import _bf, _conv_w, _f32


class VAEConfig(SimpleNamespace):
    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, norm_eps=1e-6, scaling_factor=0.18215, **extra):
        super().__init__(in_channels=in_channels, latent_channels=latent_channels, block_out_channels=tuple(block_out_channels),
                         layers_per_block=layers_per_block, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                         scaling_factor=scaling_factor, **extra)


def _resnet(ci, co, groups):
    m = nn.Module()
    m.norm1, m.conv1 = nn.GroupNorm(groups, ci, eps=1e-6), nn.Conv2d(ci, co, 3, padding=1)
    m.norm2, m.conv2 = nn.GroupNorm(groups, co, eps=1e-6), nn.Conv2d(co, co, 3, padding=1)
    if ci != co:
        m.conv_shortcut = nn.Conv2d(ci, co, 1)
    return m


def _mid(c, groups):
    m = nn.Module()
    m.resnets = nn.ModuleList([_resnet(c, c, groups), _resnet(c, c, groups)])
    a = nn.Module()
    a.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
    a.to_q, a.to_k, a.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
    a.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])
    m.attentions = nn.ModuleList([a])
    return m


def _sampler(c):
    """[module with a 3x3 ``conv``]: diffusers' Downsample2D / Upsample2D hold their convolution under ``.conv``."""
    m = nn.Module()
    m.conv = nn.Conv2d(c, c, 3, padding=1)
    return nn.ModuleList([m])


class _Coder(nn.Module):
    def __init__(self, cfg: VAEConfig, decoder: bool):
        super().__init__()
        ch, g, n = list(cfg.block_out_channels), cfg.norm_num_groups, len(cfg.block_out_channels)
        if not decoder:
            self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
            self.down_blocks = nn.ModuleList()
            prev = ch[0]
            for i, c in enumerate(ch):
                b = nn.Module()
                b.resnets = nn.ModuleList([_resnet(prev if j == 0 else c, c, g) for j in range(cfg.layers_per_block)])
                if i + 1 < n:
                    b.downsamplers = _sampler(c)
                self.down_blocks.append(b)
                prev = c
            self.mid_block = _mid(ch[-1], g)
            self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
            self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)
        else:
            self.conv_in = nn.Conv2d(cfg.latent_channels, ch[-1], 3, padding=1)
            self.mid_block = _mid(ch[-1], g)
            self.up_blocks = nn.ModuleList()
            rev = list(reversed(ch))
            prev = rev[0]
            for i, c in enumerate(rev):
                b = nn.Module()
                b.resnets = nn.ModuleList([_resnet(prev if j == 0 else c, c, g) for j in range(cfg.layers_per_block + 1)])
                if i + 1 < n:
                    b.upsamplers = _sampler(c)
                self.up_blocks.append(b)
                prev = c
            self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=1e-6)
            self.conv_out = nn.Conv2d(ch[0], cfg.in_channels, 3, padding=1)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; the arithmetic runs in libmvd_hip.so")


def pack_vae(sd: Dict[str, torch.Tensor], cfg: VAEConfig, device) -> Dict[str, torch.Tensor]:
    """diffusers AutoencoderKL state dict -> the engine's weight slots (mvd_amd/csrc/vae.hip)."""
    out: Dict[str, torch.Tensor] = {}
    f = lambda k: sd[k].detach().float()   # noqa: E731

    def conv_in(slot, key):
        w = _conv_w(sd[f"{key}.weight"], tap_major=True)
        out[f"{slot}.w"] = _bf(torch.nn.functional.pad(w, (0, 64 - w.shape[1])), device)
        out[f"{slot}.b"] = _f32(sd[f"{key}.bias"], device)

    def resnet(slot, key):
        ci, co = sd[f"{key}.conv1.weight"].shape[1], sd[f"{key}.conv1.weight"].shape[0]
        for nm in ("norm1", "norm2"):
            out[f"{slot}.{nm}.g"] = _f32(sd[f"{key}.{nm}.weight"], device)
            out[f"{slot}.{nm}.b"] = _f32(sd[f"{key}.{nm}.bias"], device)
        out[f"{slot}.conv1.w"] = _bf(_conv_w(sd[f"{key}.conv1.weight"]), device)
        out[f"{slot}.conv1.b"] = _f32(sd[f"{key}.conv1.bias"], device)
        w2, b2 = _conv_w(sd[f"{key}.conv2.weight"]), f(f"{key}.conv2.bias")
        if ci != co:
            w2 = torch.cat([w2, f(f"{key}.conv_shortcut.weight").reshape(co, ci)], dim=1)
            b2 = b2 + f(f"{key}.conv_shortcut.bias")
        out[f"{slot}.conv2.w"] = _bf(w2, device)
        out[f"{slot}.conv2.b"] = _f32(b2, device)

    def mid(slot, key):
        resnet(f"{slot}.resnets.0", f"{key}.resnets.0")
        resnet(f"{slot}.resnets.1", f"{key}.resnets.1")
        a = f"{key}.attentions.0"
        out[f"{slot}.attn.norm.g"] = _f32(sd[f"{a}.group_norm.weight"], device)
        out[f"{slot}.attn.norm.b"] = _f32(sd[f"{a}.group_norm.bias"], device)
        for nm, src in (("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("out", "to_out.0")):
            out[f"{slot}.attn.{nm}.w"] = _bf(sd[f"{a}.{src}.weight"], device)
            out[f"{slot}.attn.{nm}.b"] = _f32(sd[f"{a}.{src}.bias"], device)

    def tail(slot, key):
        out[f"{slot}.norm_out.g"] = _f32(sd[f"{key}.conv_norm_out.weight"], device)
        out[f"{slot}.norm_out.b"] = _f32(sd[f"{key}.conv_norm_out.bias"], device)
        out[f"{slot}.conv_out.w"] = _bf(_conv_w(sd[f"{key}.conv_out.weight"], tap_major=True), device)
        out[f"{slot}.conv_out.b"] = _f32(sd[f"{key}.conv_out.bias"], device)

    n = len(cfg.block_out_channels)
    conv_in("encoder.conv_in", "encoder.conv_in")
    for i in range(n):
        for j in range(cfg.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", f"encoder.down_blocks.{i}.resnets.{j}")
        if i + 1 < n:
            k = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            out[f"encoder.down_blocks.{i}.down.w"] = _bf(_conv_w(sd[f"{k}.weight"]), device)
            out[f"encoder.down_blocks.{i}.down.b"] = _f32(sd[f"{k}.bias"], device)
    mid("encoder.mid_block", "encoder.mid_block")
    tail("encoder", "encoder")
    conv_in("decoder.conv_in", "decoder.conv_in")
    mid("decoder.mid_block", "decoder.mid_block")
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", f"decoder.up_blocks.{i}.resnets.{j}")
        if i + 1 < n:
            k = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            out[f"decoder.up_blocks.{i}.up.w"] = _bf(_conv_w(sd[f"{k}.weight"]), device)
            out[f"decoder.up_blocks.{i}.up.b"] = _f32(sd[f"{k}.bias"], device)
    tail("decoder", "decoder")
    for q in ("quant_conv", "post_quant_conv"):
        w = f(f"{q}.weight")
        out[f"{q}.w"] = _f32(w.reshape(w.shape[0], w.shape[1]), device)
        out[f"{q}.b"] = _f32(sd[f"{q}.bias"], device)
    return out


class DiagonalGaussianDistribution:
    """diffusers' posterior object: ``sample()`` / ``mode()`` over the (mean | logvar) moments the engine produced."""

    def __init__(self, moments: torch.Tensor):
        self.parameters = moments
        self.mean, self.logvar = moments.chunk(2, dim=1)

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, c2, H, W = self.parameters.shape
        if noise is None:
            gdev = generator.device if generator is not None else self.parameters.device
            noise = torch.randn(B, c2 // 2, H, W, generator=generator, device=gdev, dtype=torch.float32)
        noise = noise.to(self.parameters.device, torch.float32).contiguous()
        out = torch.empty_like(noise)
        L.call("mvd_op_gaussian_sample", C.c_void_p(self.parameters.data_ptr()), C.c_void_p(noise.data_ptr()), B, c2 // 2, H * W, 1.0,
               C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return out

    def mode(self) -> torch.Tensor:
        return self.mean.contiguous()


class AutoencoderKLHIP(nn.Module):
    def __init__(self, config: Optional[VAEConfig] = None):
        super().__init__()
        self.config = config or VAEConfig()
        self.encoder = _Coder(self.config, decoder=False)
        self.decoder = _Coder(self.config, decoder=True)
        lc = self.config.latent_channels
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)
        self._h = None
        self._dev = None
        self._packed: Dict[str, torch.Tensor] = {}
        self._ws = None
        self._dirty = True

    @classmethod
    def from_snapshot(cls, path: str) -> "AutoencoderKLHIP":
        """``<path>/config.json`` + ``diffusion_pytorch_model.safetensors`` of a local diffusers snapshot (nothing is fetched)."""
        raw = json.load(open(os.path.join(path, "config.json")))
        cfg = VAEConfig(**{k: raw[k] for k in ("in_channels", "latent_channels", "block_out_channels", "layers_per_block",
                                               "norm_num_groups", "scaling_factor") if k in raw})
        m = cls(cfg)
        from safetensors.torch import load_file
        m.load_state_dict(load_file(os.path.join(path, "diffusion_pytorch_model.safetensors")))
        return m

    @staticmethod
    def convert_deprecated_attention_keys(sd):
        """The published SD-2.1 VAE stores its mid-block attention under the pre-refactor names
        (``mid_block.attentions.0.{query,key,value,proj_attn}``, 1x1-conv shaped ``(C, C, 1, 1)`` in the oldest files);
        diffusers renames them on load (``_convert_deprecated_attention_blocks``).  Same mapping here."""
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        out = {}
        for k, v in sd.items():
            parts = k.split(".")
            if len(parts) >= 3 and "attentions" in parts and parts[-2] in ren:
                k = ".".join(parts[:-2] + [ren[parts[-2]], parts[-1]])
            if ".attentions." in k and k.endswith(".weight") and v.dim() == 4 and v.shape[2:] == (1, 1):
                v = v.reshape(v.shape[0], v.shape[1])
            out[k] = v
        return out

    def load_state_dict(self, sd, strict: bool = True, **kw):
        self._dirty = True
        return super().load_state_dict(self.convert_deprecated_attention_keys(sd), strict=strict, **kw)

    def to(self, *a, **k):
        self._dirty = True
        return super().to(*a, **k)

    def __del__(self):
        try:
            if self._h:
                L.lib().mvd_vae_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------ engine plumbing
    def _sync(self) -> torch.device:
        if not torch.cuda.is_available():
            raise L.MvdError("AutoencoderKLHIP needs a MI355X (there is no CPU fallback)")
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.MvdError(f"AutoencoderKLHIP is on {dev}: move it to a cuda device (there is no CPU fallback)")
        if self._h is None:
            c = L.mvd_vae_config_t()
            cfg = self.config
            c.in_channels, c.latent_channels, c.num_levels = cfg.in_channels, cfg.latent_channels, len(cfg.block_out_channels)
            for i, ch in enumerate(cfg.block_out_channels):
                c.block_out_channels[i] = ch
            c.layers_per_block, c.norm_num_groups, c.norm_eps = cfg.layers_per_block, cfg.norm_num_groups, cfg.norm_eps
            h = C.c_void_p()
            L.call("mvd_vae_create", C.byref(c), C.byref(h))
            self._h = h
        if self._dirty or self._dev != dev:
            with torch.no_grad():
                self._packed = pack_vae(self.state_dict(), self.config, dev)
            for slot, t in self._packed.items():
                dt = {torch.float32: 0, torch.bfloat16: 1}[t.dtype]
                L.call("mvd_vae_set_weight", self._h, slot.encode(), C.c_void_p(t.data_ptr()), t.numel(), dt)
            self._dirty, self._dev = False, dev
        return dev

    def _workspace(self, batch, h, w, decode):
        need = L.lib().mvd_vae_workspace_bytes(self._h, batch, h, w, int(decode))
        if need < 0:
            raise L.MvdError(f"vae workspace_bytes: {L.last_error()}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._dev)
        L.call("mvd_vae_bind_workspace", self._h, C.c_void_p(self._ws.data_ptr()), self._ws.numel())

    # ------------------------------------------------------------------ the diffusers protocol
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        dev = self._sync()
        x = x.to(dev, torch.float32).contiguous()
        B, _, H, W = x.shape
        f = 2 ** (len(self.config.block_out_channels) - 1)
        self._workspace(B, H, W, False)
        mom = torch.empty(B, 2 * self.config.latent_channels, H // f, W // f, device=dev, dtype=torch.float32)
        L.call("mvd_vae_encode", self._h, C.c_void_p(x.data_ptr()), B, H, W, C.c_void_p(mom.data_ptr()),
               C.c_void_p(torch.cuda.current_stream().cuda_stream))
        dist = DiagonalGaussianDistribution(mom)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        dev = self._sync()
        z = z.to(dev, torch.float32).contiguous()
        B, _, h, w = z.shape
        f = 2 ** (len(self.config.block_out_channels) - 1)
        self._workspace(B, h, w, True)
        img = torch.empty(B, self.config.in_channels, h * f, w * f, device=dev, dtype=torch.float32)
        L.call("mvd_vae_decode", self._h, C.c_void_p(z.data_ptr()), B, h, w, C.c_void_p(img.data_ptr()),
               C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return SimpleNamespace(sample=img) if return_dict else (img,)
