"""Oracle: the reference's own hot-path code restated functionally (CPU, fp32).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Pinned by golden vectors
captured from the reference's ``attention.py`` / ``camera_encoder.py``
(tests/golden/make_golden.py, tests/test_oracle_golden.py).

Follows (all paths relative to /root/reference):
  * ``CameraEncoder``                 src/models/camera_encoder.py:12-255
  * ``ImageCrossAttentionProcessor``  src/models/attention.py:12-265
  * ``ImageEncoder.forward``          src/models/image_encoder.py:97-112
  * ``MultiViewUNet.forward``         src/models/mvd_unet.py:179-352
State-dict key names are the reference's (last bullet).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import sd21_unet as U


# ----------------------------------------------------------------------------
# CameraEncoder (camera_encoder.py)
# ----------------------------------------------------------------------------

def modulation_hidden_dims(cfg: U.UNetConfig) -> Dict[str, int]:
    """mvd_unet.py:63-80 -- insertion order matters for parameter order only."""
    down = list(cfg.block_out_channels)
    up = list(reversed(down))
    d: Dict[str, int] = {}
    for i in range(cfg.num_levels):
        d[f"down_{i}"] = down[i]
    for i in range(cfg.num_levels):
        d[f"up_{i}"] = up[i]
    d["mid"] = down[-1]
    d["output"] = 4
    return d


def camera_param_shapes(cfg: U.UNetConfig, output_dim=1024, hidden_dim=512, simple=False):
    """camera_encoder.py:29-85 (nn.Sequential indices as state-dict keys)."""
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(k, o, i):
        s[f"{k}.weight"] = (o, i)
        s[f"{k}.bias"] = (o,)

    def ln(k, n):
        s[f"{k}.weight"] = (n,)
        s[f"{k}.bias"] = (n,)

    for enc, din in (("rotation_encoder", 9), ("translation_encoder", output_dim)):
        lin(f"{enc}.0", hidden_dim, din)
        ln(f"{enc}.1", hidden_dim)
        if simple:
            lin(f"{enc}.3", output_dim, hidden_dim)
        else:
            lin(f"{enc}.3", hidden_dim, hidden_dim)
            ln(f"{enc}.4", hidden_dim)
            lin(f"{enc}.6", output_dim, hidden_dim)
    lin("final_projection.0", output_dim, 2 * output_dim)
    ln("final_projection.1", output_dim)
    lin("final_projection.3", output_dim, output_dim)
    ln("final_projection.4", output_dim)
    ln("output_norm", output_dim)
    for name, dim in modulation_hidden_dims(cfg).items():
        lin(f"modulators.{name}.0", output_dim // 2, output_dim)
        ln(f"modulators.{name}.1", output_dim // 2)
        lin(f"modulators.{name}.3", 2 * dim, output_dim // 2)
    return s


def init_camera_params(cfg, seed=1, output_dim=1024, hidden_dim=512, simple=False):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in camera_param_shapes(cfg, output_dim, hidden_dim, simple).items():
        is_ln = len(shape) == 1 and (name.split(".")[-2] in ("1", "4") or name.startswith("output_norm"))
        if is_ln:
            t = (1.0 + 0.1 * torch.randn(shape, generator=g)) if name.endswith("weight") else 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.1 * torch.randn(shape, generator=g)
            # camera_encoder.py:93-105: modulator last layer bias = [0.5]*dim + [0]*dim; keep that
            # flavour (non-trivial scale) but perturbed so shift is exercised too.
            if name.startswith("modulators.") and name.endswith(".3.bias"):
                dim = shape[0] // 2
                t[:dim] += 0.5
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
            if name.startswith("modulators.") and name.endswith(".3.weight"):
                t = t * 0.5
        out[name] = t
    return out


def relative_transform(src: torch.Tensor, tgt: torch.Tensor):
    """camera_encoder.py:107-120.  Accepts (B,3,4) or (B,4,4) (Q8)."""
    sR, sT = src[:, :3, :3], src[:, :3, 3]
    tR, tT = tgt[:, :3, :3], tgt[:, :3, 3]
    R = torch.bmm(tR, sR.transpose(1, 2))
    T = tT - torch.bmm(R, sT.unsqueeze(2)).squeeze(2)
    return R, T


def fourier_features(T: torch.Tensor, output_dim=1024, max_freq=10) -> torch.Tensor:
    """camera_encoder.py:137-151: (B,3) -> (B, 2*3*pos_enc_dim) BEFORE the random projection."""
    n = (output_dim // 2) // 3
    freqs = torch.exp(torch.linspace(0.0, float(np.log(max_freq)), n))
    ang = T.unsqueeze(-1) * freqs[None, None, :]
    enc = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    return enc.reshape(T.shape[0], -1)


def draw_fourier_projection(output_dim=1024, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Q1: camera_encoder.py:153-155 draws ``randn(out, enc)/sqrt(enc)`` on EVERY call."""
    enc_dim = 2 * 3 * ((output_dim // 2) // 3)
    return torch.randn(output_dim, enc_dim, generator=generator) / np.sqrt(enc_dim)


def _mlp(p, prefix, x, idx_lin, idx_ln):
    """Linear/LayerNorm/SiLU stack with nn.Sequential numbering."""
    for li, ni in zip(idx_lin, idx_ln):
        x = F.linear(x, p[f"{prefix}.{li}.weight"], p[f"{prefix}.{li}.bias"])
        if ni is not None:
            n = x.shape[-1]
            x = F.layer_norm(x, (n,), p[f"{prefix}.{ni}.weight"], p[f"{prefix}.{ni}.bias"], 1e-5)
            x = F.silu(x)
    return x


def camera_embedding(p: Dict[str, torch.Tensor], src, tgt, proj: torch.Tensor, simple=False) -> torch.Tensor:
    """``encode_cameras`` + ``forward`` (camera_encoder.py:160-196).  ``proj`` = the Q1 matrix."""
    R, T = relative_transform(src.float(), tgt.float())
    out_dim = p["output_norm.weight"].shape[0]
    lin = (0, 3) if simple else (0, 3, 6)
    lns = (1, None) if simple else (1, 4, None)
    rot = _mlp(p, "rotation_encoder", R.reshape(R.shape[0], -1), lin, lns)
    enc = F.linear(fourier_features(T, out_dim), proj)
    tr = _mlp(p, "translation_encoder", enc, lin, lns)
    x = torch.cat([rot, tr], dim=-1)
    x = F.linear(x, p["final_projection.0.weight"], p["final_projection.0.bias"])
    x = F.silu(F.layer_norm(x, (out_dim,), p["final_projection.1.weight"], p["final_projection.1.bias"], 1e-5))
    x = F.linear(x, p["final_projection.3.weight"], p["final_projection.3.bias"])
    x = F.layer_norm(x, (out_dim,), p["final_projection.4.weight"], p["final_projection.4.bias"], 1e-5)
    return F.layer_norm(x, (out_dim,), p["output_norm.weight"], p["output_norm.bias"], 1e-5)


def film_scale_shift(p, name: str, emb: torch.Tensor, strength: float):
    """camera_encoder.py:215-222: (B,dim) processed scale and shift, or None if no such modulator."""
    if f"modulators.{name}.0.weight" not in p:
        return None
    x = F.linear(emb, p[f"modulators.{name}.0.weight"], p[f"modulators.{name}.0.bias"])
    n = x.shape[-1]
    x = F.silu(F.layer_norm(x, (n,), p[f"modulators.{name}.1.weight"], p[f"modulators.{name}.1.bias"], 1e-5))
    x = F.linear(x, p[f"modulators.{name}.3.weight"], p[f"modulators.{name}.3.bias"])
    scale, shift = x.chunk(2, dim=-1)
    return torch.sigmoid(scale) * 2.0 * strength, shift * strength


def apply_modulation(p, name: str, x: torch.Tensor, emb: torch.Tensor, strength: float) -> torch.Tensor:
    """camera_encoder.py:198-255 (identity for unknown modulator names, e.g. "mid_0" -- Q3)."""
    ss = film_scale_shift(p, name, emb, strength)
    if ss is None:
        return x
    scale, shift = ss
    return x * scale[:, :, None, None] + shift[:, :, None, None]


# ----------------------------------------------------------------------------
# ImageCrossAttentionProcessor (attention.py)
# ----------------------------------------------------------------------------

def normalize_reference(ref: torch.Tensor) -> torch.Tensor:
    """attention.py:95-103 (Q2): per-pixel statistics over (batch, channel) of the NCHW map."""
    r = ref - ref.mean(dim=(0, 1), keepdim=True)
    std = torch.clamp(r.std(dim=(0, 1), keepdim=True), min=1e-6)
    return r / std * 0.5


def image_cross_attention(
    w: Dict[str, torch.Tensor],
    hidden: torch.Tensor,
    ref_nchw: torch.Tensor,
    heads: int,
    dim_head: int = 64,
) -> torch.Tensor:
    """The reference branch of attention.py:83-161 BEFORE ``ref_scale``: returns (B,N,C).

    ``w`` keys: to_q_ref.weight, to_k_ref.weight, to_v_ref.weight, to_out_ref.0.{weight,bias}.
    ``ref_ln`` exists in the state dict but is never applied (attention.py:160-161).
    """
    ref = normalize_reference(ref_nchw)
    Bh = hidden.shape[0]
    Br, C, H, W = ref.shape
    ref = ref.permute(0, 2, 3, 1).reshape(Br, H * W, C)
    q = F.linear(hidden, w["to_q_ref.weight"]).view(Bh, -1, heads, dim_head).transpose(1, 2)
    # Q4: the view uses the HIDDEN batch size, silently re-chunking the reference tokens
    k = F.linear(ref, w["to_k_ref.weight"]).view(Bh, -1, heads, dim_head).transpose(1, 2)
    v = F.linear(ref, w["to_v_ref.weight"]).view(Bh, -1, heads, dim_head).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(Bh, -1, heads * dim_head)
    return F.linear(o, w["to_out_ref.0.weight"], w["to_out_ref.0.bias"])


def adapter_param_shapes(cfg: U.UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """Keys of the 32 processors relative to ``base_unet.`` (attention.py:33-43)."""
    s: Dict[str, Tuple[int, ...]] = {}

    def add(prefix, c):
        for a in ("attn1", "attn2"):
            b = f"{prefix}.transformer_blocks.0.{a}.processor"
            s[f"{b}.to_q_ref.weight"] = (c, c)
            s[f"{b}.to_k_ref.weight"] = (c, c)
            s[f"{b}.to_v_ref.weight"] = (c, c)
            s[f"{b}.ref_ln.weight"] = (c,)
            s[f"{b}.ref_ln.bias"] = (c,)
            s[f"{b}.to_out_ref.0.weight"] = (c, c)
            s[f"{b}.to_out_ref.0.bias"] = (c,)

    for i, c in enumerate(cfg.block_out_channels):
        if cfg.down_has_attn(i):
            for j in range(cfg.layers_per_block):
                add(f"down_blocks.{i}.attentions.{j}", c)
    add("mid_block.attentions.0", cfg.block_out_channels[-1])
    rev = list(reversed(cfg.block_out_channels))
    for i, c in enumerate(rev):
        if cfg.up_has_attn(i):
            for j in range(cfg.layers_per_block + 1):
                add(f"up_blocks.{i}.attentions.{j}", c)
    return s


def init_adapter_params(cfg, seed=2):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in adapter_param_shapes(cfg).items():
        if "ref_ln" in name:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif name.endswith(".bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        out[name] = t
    return out


def adapter_init_from_attention(attn_w: Dict[str, torch.Tensor], c: int) -> Dict[str, torch.Tensor]:
    """``load_original_weights`` (attention.py:199-245): initialise *_ref from the wrapped Attention.

    ``attn_w`` keys: to_q.weight, to_k.weight, to_v.weight, to_out.0.weight, to_out.0.bias.
    """
    out = {
        "to_q_ref.weight": attn_w["to_q.weight"].clone(),
        "to_out_ref.0.weight": attn_w["to_out.0.weight"].clone(),
        "to_out_ref.0.bias": attn_w["to_out.0.bias"].clone(),
    }
    for kv in ("k", "v"):
        orig = attn_w[f"to_{kv}.weight"]
        o_out, o_in = orig.shape
        if (o_out, o_in) == (c, c):
            wt = orig.clone()
        elif c >= o_in:
            wt = torch.zeros(c, c)
            wt[:, :o_in] = orig[: min(c, o_out), :]
        else:
            wt = F.linear(torch.eye(c), orig[: min(c, o_out), :c]).clone()
        out[f"to_{kv}_ref.weight"] = wt
    return out


def _processor_key(feature: str, kind: str) -> str:
    """feature name (image_encoder.py hook name) + kind -> diffusers module path of the processor."""
    parts = feature.split("_")
    if feature.startswith("mid"):
        prefix = f"mid_block.attentions.{parts[-1]}"
    else:
        prefix = f"{parts[0]}_blocks.{parts[2]}.attentions.{parts[-1]}"
    a = "attn1" if kind == "self" else "attn2"
    return f"{prefix}.transformer_blocks.0.{a}.processor"


# ----------------------------------------------------------------------------
# MultiViewUNet.forward (mvd_unet.py:179-338)
# ----------------------------------------------------------------------------

def _sub(p: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def image_encoder_forward(p_enc, cfg, latents, text) -> Dict[str, torch.Tensor]:
    """image_encoder.py:97-112: frozen UNet at t=0, plain attention, 16 captured NCHW maps."""
    cap: Dict[str, torch.Tensor] = {}
    U.unet_forward(p_enc, cfg, latents, torch.tensor([0]), text, capture=cap)
    return cap


def multiview_unet_forward(
    params: Dict[str, torch.Tensor],
    cfg: U.UNetConfig,
    sample: torch.Tensor,
    timestep,
    text: torch.Tensor,
    source_camera: Optional[torch.Tensor] = None,
    target_camera: Optional[torch.Tensor] = None,
    source_image_latents: Optional[torch.Tensor] = None,
    *,
    fourier_proj: Optional[torch.Tensor] = None,
    img_ref_scale: float = 0.3,
    cam_modulation_strength: float = 0.2,
    use_camera_conditioning: bool = True,
    use_image_conditioning: bool = True,
    simple_cam_encoder: bool = False,
    features_out: Optional[Dict[str, torch.Tensor]] = None,
) -> torch.Tensor:
    """Full reference-faithful forward.  ``params`` uses the wrapper's key prefixes
    ``base_unet.``, ``camera_encoder.``, ``image_encoder.unet.``."""
    p_base = _sub(params, "base_unet.")
    B = sample.shape[0]
    # mvd_unet.py:233-237 -- CFG: repeat text to the sample batch
    if B > text.shape[0]:
        text = text.repeat(B // text.shape[0], 1, 1)

    emb = None
    p_cam = None
    if use_camera_conditioning and target_camera is not None:
        p_cam = _sub(params, "camera_encoder.")
        if fourier_proj is None:
            fourier_proj = draw_fourier_projection(p_cam["output_norm.weight"].shape[0])
        emb = camera_embedding(p_cam, source_camera, target_camera, fourier_proj, simple_cam_encoder)
        sample = apply_modulation(p_cam, "output", sample, emb, cam_modulation_strength)  # :256-258

    ref: Optional[Dict[str, torch.Tensor]] = None
    if use_image_conditioning and source_image_latents is not None:
        bs = source_image_latents.shape[0]
        enc_text = text
        if text.shape[0] == 2 * bs:       # :280-283
            enc_text = text[bs:]
        elif text.shape[0] > bs:          # :284-285
            enc_text = text[:bs]
        ref = image_encoder_forward(_sub(params, "image_encoder.unet."), cfg, source_image_latents, enc_text)
        if features_out is not None:
            features_out.update(ref)

    heads_of = {}
    for i in range(cfg.num_levels):
        if cfg.down_has_attn(i):
            for j in range(cfg.layers_per_block):
                heads_of[f"down_block_{i}_attn_{j}"] = cfg.num_heads[i]
    heads_of["mid_block_attn_0"] = cfg.num_heads[-1]
    rev = list(reversed(cfg.num_heads))
    for i in range(cfg.num_levels):
        if cfg.up_has_attn(i):
            for j in range(cfg.layers_per_block + 1):
                heads_of[f"up_block_{i}_attn_{j}"] = rev[i]

    def attn_hook(feature, kind, hidden, attn_out):
        if ref is None or feature not in ref:
            return attn_out                                   # attention.py:72-81
        w = _sub(p_base, _processor_key(feature, kind) + ".")
        branch = image_cross_attention(w, hidden, ref[feature], heads_of[feature])
        return attn_out + img_ref_scale * branch              # attention.py:174-181

    def block_hook(name, x):
        if emb is None:
            return x
        return apply_modulation(p_cam, name, x, emb, cam_modulation_strength)  # "mid_0" -> identity (Q3)

    return U.unet_forward(p_base, cfg, sample, timestep, text, attn_hook=attn_hook, block_hook=block_hook)


def init_mvd_params(cfg: U.UNetConfig, seed=0, cam_dim=1024, cam_hidden=512, simple=False,
                    share_encoder=False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights for the whole wrapper with the reference's key names."""
    out: Dict[str, torch.Tensor] = {}
    base = U.init_params(cfg, seed)
    for k, v in base.items():
        out[f"base_unet.{k}"] = v
    for k, v in init_adapter_params(cfg, seed + 2).items():
        out[f"base_unet.{k}"] = v
    for k, v in init_camera_params(cfg, seed + 1, cam_dim, cam_hidden, simple).items():
        out[f"camera_encoder.{k}"] = v
    enc = base if share_encoder else U.init_params(cfg, seed + 3)
    for k, v in enc.items():
        out[f"image_encoder.unet.{k}"] = v
    return out
