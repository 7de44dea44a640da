"""Oracle: the denoising loop around the UNet (CPU, numpy/torch fp32).  TEST INFRASTRUCTURE ONLY.

* ``ddpm_step`` restates diffusers-0.32.2 ``DDPMScheduler.step`` (variance_type "fixed_small", no clipping,
  epsilon / v_prediction) -- third-party code the reference calls at /root/reference/src/models/pipeline.py:161;
  diffusers is absent, so this part is PARITY UNPINNED (the SNR-shifted betas it runs on ARE pinned: G4).
* ``denoise_loop`` restates /root/reference/src/models/pipeline.py:119-166 on top of ``oracle.mvd``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import mvd as M


def leading_timesteps(num_train: int, n: int) -> np.ndarray:
    return (np.arange(0, n) * (num_train // n)).round()[::-1].astype(np.int64)


def ddpm_step(model_out, t, sample, alphas_cumprod, num_train, n_steps, prediction_type, noise):
    prev_t = t - num_train // n_steps
    a_t = alphas_cumprod[t]
    a_prev = alphas_cumprod[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_alpha = a_t / a_prev
    cur_beta = 1 - cur_alpha
    if prediction_type == "epsilon":
        x0 = (sample - b_t ** 0.5 * model_out) / a_t ** 0.5
    elif prediction_type == "v_prediction":
        x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_out
    else:
        raise ValueError(prediction_type)
    prev = (a_prev ** 0.5 * cur_beta / b_t) * x0 + (cur_alpha ** 0.5 * b_prev / b_t) * sample
    if t > 0:
        var = torch.clamp(b_prev / b_t * cur_beta, min=1e-20)
        prev = prev + var ** 0.5 * noise
    return prev


def denoise_loop(params, cfg, betas, prompt, negative, latents, src_cam, tgt_cam, src_lat, n_steps, guidance_scale,
                 noises, fourier_projs, prediction_type="v_prediction", trace=None, **mv_kwargs):
    """``trace``: a list that receives the latents after every step (drift tests compare whole trajectories)."""
    acp = torch.cumprod(1.0 - betas, dim=0)
    T = betas.shape[0]
    use_cfg = guidance_scale > 1.0 and negative is not None
    embeds = torch.cat([negative, prompt]) if use_cfg else prompt
    for i, t in enumerate(leading_timesteps(T, n_steps).tolist()):
        x_in = torch.cat([latents] * 2) if guidance_scale > 1.0 else latents
        out = M.multiview_unet_forward(params, cfg, x_in, torch.tensor(t), embeds, src_cam, tgt_cam, src_lat,
                                       fourier_proj=None if fourier_projs is None else fourier_projs[i], **mv_kwargs)
        if guidance_scale > 1.0:
            u, c = out.chunk(2)
            out = u + guidance_scale * (c - u)
        latents = ddpm_step(out, t, latents, acp, T, n_steps, prediction_type, noises[i])
        if trace is not None:
            trace.append(latents.clone())
    return latents
