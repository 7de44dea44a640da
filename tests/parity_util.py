"""Shared end-to-end parity harness: HIP engine (through the MultiViewUNet mirror and the C ABI)
vs the CPU oracle on identical seeded inputs and identical weights."""
from __future__ import annotations

import time

import torch

from mvd_amd.utils import look_at  # noqa: F401  (synthetic cameras: product-side helper, re-exported for the tests)
from oracle import mvd as OM
from oracle import sd21_unet as OU


def rel_l2(got: torch.Tensor, want: torch.Tensor) -> float:
    return ((got.float().cpu() - want.float()).norm() / want.float().norm().clamp_min(1e-12)).item()


def max_rel(got: torch.Tensor, want: torch.Tensor) -> float:
    return ((got.float().cpu() - want.float()).abs().max() / want.float().abs().max().clamp_min(1e-12)).item()


def make_inputs(cfg: OU.UNetConfig, batch: int, hw, text_len: int, seed: int = 0, cam_dim: int = 1024):
    """hw: latent size, an int (square) or an (H, W) pair."""
    h, w = (hw, hw) if isinstance(hw, int) else hw
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(batch, cfg.in_channels, h, w, generator=g)
    text = torch.randn(batch, text_len, cfg.cross_attention_dim, generator=g)
    lat = 0.18215 * torch.randn(batch, cfg.in_channels, h, w, generator=g)
    src = torch.stack([look_at(0.0)] * batch)
    tgt = torch.stack([look_at([45.0, 90.0, 180.0, 270.0][b % 4]) for b in range(batch)])
    proj = OM.draw_fourier_projection(cam_dim, g)
    return dict(sample=sample, text=text, lat=lat, src=src, tgt=tgt, proj=proj)


def build_pair(cfg_name: str = "tiny", seed: int = 0, cam_dim: int = 1024, cam_hidden: int = 512, img_ref_scale=0.3,
               cam_strength=0.2):
    """(oracle cfg, oracle params, MultiViewUNet mirror on cuda) sharing one set of seeded weights."""
    from mvd_amd.config import UNetConfig
    from mvd_amd.mvd_unet import MultiViewUNet
    ocfg = OU.UNetConfig.tiny() if cfg_name == "tiny" else OU.UNetConfig.sd21()
    hcfg = UNetConfig.tiny() if cfg_name == "tiny" else UNetConfig.sd21()
    params = OM.init_mvd_params(ocfg, seed, cam_dim=cam_dim, cam_hidden=cam_hidden)
    model = MultiViewUNet(None, unet_config=hcfg, init="empty", img_ref_scale=img_ref_scale,
                          cam_modulation_strength=cam_strength, cam_output_dim=cam_dim, cam_hidden_dim=cam_hidden)
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected, unexpected[:5]
    assert not missing, missing[:5]
    model = model.to("cuda")
    model.eval()
    return ocfg, params, model


_PAIRS = {}


def shared_pair(cfg_name: str):
    """One (oracle params, mirror) pair per configuration for the whole test session: building the SD-2.1 pair (1.85 G
    parameters on the host + the packed copy on the GPU) takes ~1 min."""
    if cfg_name not in _PAIRS:
        cam_dim, cam_hidden = (96, 48) if cfg_name == "tiny" else (1024, 512)
        _PAIRS[cfg_name] = build_pair(cfg_name, 0, cam_dim, cam_hidden)
    return _PAIRS[cfg_name]


def run_tiny_parity(batch: int = 2, verbose: bool = False, cfg_name: str = "tiny", hw: int = 16, text_len: int = 7,
                    timestep: int = 500, cam: bool = True, img: bool = True):
    cam_dim, cam_hidden = (96, 48) if cfg_name == "tiny" else (1024, 512)
    ocfg, params, model = shared_pair(cfg_name) if cfg_name != "tiny" else build_pair(cfg_name, 0, cam_dim, cam_hidden)
    inp = make_inputs(ocfg, batch, hw, text_len, 0, cam_dim)
    t0 = time.time()
    feats = {}
    want = OM.multiview_unet_forward(params, ocfg, inp["sample"], torch.tensor(timestep), inp["text"], inp["src"] if cam else None,
                                     inp["tgt"] if cam else None, inp["lat"] if img else None, fourier_proj=inp["proj"],
                                     img_ref_scale=0.3, cam_modulation_strength=0.2, features_out=feats)
    t_cpu = time.time() - t0
    model.fourier_projection = inp["proj"]
    with torch.no_grad():
        got = model(inp["sample"].cuda(), torch.tensor(timestep), inp["text"].cuda(),
                    source_camera=inp["src"].cuda() if cam else None, target_camera=inp["tgt"].cuda() if cam else None,
                    source_image_latents=inp["lat"].cuda() if img else None).sample
    torch.cuda.synchronize()
    stats = dict(rel_l2=rel_l2(got, want), max_rel=max_rel(got, want), cpu_seconds=t_cpu,
                 finite=bool(torch.isfinite(got).all()))
    if verbose:
        print("parity", cfg_name, f"batch={batch}", stats, flush=True)
    return stats
