"""Host-side scheduler math for the denoising loop.

* ``compute_snr`` / ``SNR_to_betas`` / ``ShiftSNRScheduler`` mirror /root/reference/src/training/scheduler.py:16-150
  (pinned by tests/golden/g4_shift_snr.npz, captured from the reference's own functions).
* ``DDPMScheduler`` restates the part of diffusers-0.32.2 ``DDPMScheduler`` the reference uses
  (``from_config(..., trained_betas=)``, ``set_timesteps``, ``step`` with ``variance_type="fixed_small"``,
  epsilon / v_prediction, no sample clipping -- the SD-2.1 scheduler config).  diffusers is not installed here, so
  this part is unpinned (checked against a numpy restatement in oracle/scheduler.py only).

Everything here is scalar / length-1000 vector math on the host; the per-step tensor update is one fused HIP
kernel (``mvd_op_ddpm_step``) whose four coefficients are computed here, so the loop never syncs the device.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Optional

import numpy as np
import torch


def compute_snr(timesteps, noise_scheduler):
    acp = noise_scheduler.alphas_cumprod
    alpha = (acp ** 0.5)[timesteps].float()
    sigma = ((1.0 - acp) ** 0.5)[timesteps].float()
    return (alpha / sigma) ** 2


def SNR_to_betas(snr):
    alpha_t = (snr / (1 + snr)) ** 0.5
    alphas_cumprod = alpha_t ** 2
    alphas = alphas_cumprod / torch.cat([torch.ones(1, device=snr.device), alphas_cumprod[:-1]])
    return 1 - alphas


class DDPMScheduler:
    """Minimal DDPM scheduler with the diffusers attribute names the reference touches."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", trained_betas=None, prediction_type: str = "v_prediction",
                 variance_type: str = "fixed_small", clip_sample: bool = False, timestep_spacing: str = "leading",
                 steps_offset: int = 0, **_ignored):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      variance_type=variance_type, clip_sample=clip_sample,
                                      timestep_spacing=timestep_spacing, steps_offset=steps_offset)
        if trained_betas is not None:
            self.betas = torch.as_tensor(np.asarray(trained_betas), dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(f"unsupported beta_schedule {beta_schedule}")
        if variance_type != "fixed_small" or clip_sample:
            raise ValueError("only variance_type='fixed_small' without sample clipping is implemented (SD-2.1 config)")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)

    @classmethod
    def from_config(cls, config: Any, **overrides):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d.update(overrides)
        return cls(**d)

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing == "leading":
            ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        elif self.config.timestep_spacing == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        else:
            raise ValueError(f"unsupported timestep_spacing {self.config.timestep_spacing}")
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def previous_timestep(self, t: int) -> int:
        n = self.num_inference_steps or self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    def step_coefficients(self, t: int):
        """(c_x0_from_out, c_x0_from_sample, c_prev_from_x0, c_prev_from_sample, sigma) for one DDPM step:
        x0 = c0*model_out + c1*sample ; prev = c2*x0 + c3*sample + sigma*noise."""
        t = int(t)
        prev_t = self.previous_timestep(t)
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        b_t, b_prev = 1.0 - a_t, 1.0 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1.0 - cur_alpha
        if self.config.prediction_type == "epsilon":
            c0, c1 = -(b_t ** 0.5) / (a_t ** 0.5), 1.0 / (a_t ** 0.5)
        elif self.config.prediction_type == "v_prediction":
            c0, c1 = -(b_t ** 0.5), a_t ** 0.5
        else:
            raise ValueError(f"unsupported prediction_type {self.config.prediction_type}")
        c2 = (a_prev ** 0.5) * cur_beta / b_t
        c3 = (cur_alpha ** 0.5) * b_prev / b_t
        var = max(b_prev / b_t * cur_beta, 1e-20)
        sigma = var ** 0.5 if t > 0 else 0.0
        return c0, c1, c2, c3, sigma

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator: Optional[torch.Generator] = None,
             noise: Optional[torch.Tensor] = None):
        """One ancestral DDPM step on the GPU (fused HIP kernel); returns an object with ``prev_sample``."""
        from . import ops
        c0, c1, c2, c3, sigma = self.step_coefficients(int(timestep))
        if noise is None and sigma != 0.0:
            noise = torch.randn(sample.shape, generator=generator, device=sample.device, dtype=torch.float32)
        prev = ops.ddpm_step(model_output, sample, noise, c0, c1, c2, c3, sigma)
        return SimpleNamespace(prev_sample=prev)


class ShiftSNRScheduler:
    """/root/reference/src/training/scheduler.py:74-150."""

    def __init__(self, noise_scheduler, timesteps, shift_scale, scheduler_class):
        self.noise_scheduler, self.timesteps, self.shift_scale, self.scheduler_class = \
            noise_scheduler, timesteps, shift_scale, scheduler_class

    def _get_shift_scheduler(self):
        snr = compute_snr(self.timesteps, self.noise_scheduler)
        betas = SNR_to_betas(snr / self.shift_scale)
        return self.scheduler_class.from_config(self.noise_scheduler.config, trained_betas=betas.numpy())

    def _get_interpolated_shift_scheduler(self):
        snr = compute_snr(self.timesteps, self.noise_scheduler)
        shifted = snr / self.shift_scale
        w = self.timesteps.float() / (self.noise_scheduler.config.num_train_timesteps - 1)
        interp = torch.exp(torch.log(snr) * (1 - w) + torch.log(shifted) * w)
        return self.scheduler_class.from_config(self.noise_scheduler.config, trained_betas=SNR_to_betas(interp).numpy())

    @classmethod
    def from_scheduler(cls, noise_scheduler, shift_mode="default", timesteps=None, shift_scale=1.0, scheduler_class=None):
        if timesteps is None:
            timesteps = torch.arange(0, noise_scheduler.config.num_train_timesteps)
        if scheduler_class is None:
            scheduler_class = noise_scheduler.__class__
        s = cls(noise_scheduler, timesteps, shift_scale, scheduler_class)
        if shift_mode == "default":
            return s._get_shift_scheduler()
        if shift_mode == "interpolated":
            return s._get_interpolated_shift_scheduler()
        raise ValueError(f"Unknown shift_mode: {shift_mode}")
