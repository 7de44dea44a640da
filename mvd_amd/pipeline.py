"""The caller of the hot path: ``MVDPipeline.__call__`` with the signature and semantics of
/root/reference/src/models/pipeline.py:12-186 over the HIP engine, with device-resident state and no host syncs in
the loop (CFG concat -> MultiViewUNet -> CFG combine -> scheduler.step, each step a handful of kernel launches).

The reference class derives from diffusers' ``StableDiffusionPipeline`` and gets its text encoder, tokenizer and VAE
from ``from_pretrained``.  diffusers / SD-2.1 weights do not exist in this image, so this class is self-contained:

* ``unet`` (the ``MultiViewUNet`` mirror) and ``scheduler`` are always present;
* ``text_encoder`` / ``tokenizer`` / ``vae`` are optional components.  Without a text encoder the caller passes
  ``prompt_embeds`` (already a parameter of the reference signature); without a VAE the caller passes
  ``source_image_latents`` instead of ``source_images`` and asks for ``output_type="latent"``.  Asking for something a
  missing component would have to produce raises ``MvdError`` -- nothing is silently skipped.
* ``vae`` may be any object with the diffusers ``AutoencoderKL`` protocol (``encode(x).latent_dist.sample()``,
  ``decode(z).sample``, ``config.scaling_factor``); ``mvd_amd.vae.AutoencoderKLHIP`` is the engine-backed one (row N3).

Reference quirk Q7 is kept: ``ref_scale``, ``use_camera_embeddings`` and ``use_image_conditioning`` are accepted and
unused (pipeline.py:34-36, 134-135); the effective switches live on the UNet.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Optional, Union

import logging

import torch

from . import _lib as L
from . import ops


class MVDDenoiser:
    """The loop body of pipeline.py:119-166 starting from prompt embeddings / source latents."""

    def __init__(self, unet, scheduler):
        self.unet, self.scheduler = unet, scheduler

    @torch.no_grad()
    def __call__(self, prompt_embeds: torch.Tensor, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, latents: Optional[torch.Tensor] = None,
                 source_camera: Optional[torch.Tensor] = None, target_camera: Optional[torch.Tensor] = None,
                 source_image_latents: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                 height: int = 64, width: int = 64, noise_per_step=None, callback=None, callback_steps: int = 1,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None):
        dev = self.unet._exec_device()
        B = prompt_embeds.shape[0]
        cfg = guidance_scale > 1.0 and negative_prompt_embeds is not None       # pipeline.py:77-80
        embeds = torch.cat([negative_prompt_embeds.to(prompt_embeds.device), prompt_embeds]) if cfg else prompt_embeds
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
        # Q5: the reference K/V of a previous call (another object) must never be reused by this one
        if hasattr(self.unet, "reset_reference_cache"):
            self.unet.reset_reference_cache()
        ts_host = self.scheduler.timesteps.tolist()                             # host ints for the scheduler's coefficients
        ts_dev = torch.tensor(ts_host, dtype=torch.float32, device=dev)         # ONE upload: a per-step scalar upload would make
        for i, t in enumerate(ts_host):                                         # the host wait for the previous step's kernels
            x_in = torch.cat([latents] * 2) if guidance_scale > 1.0 else latents  # pipeline.py:141
            out = self.unet(sample=x_in, timestep=ts_dev[i], encoder_hidden_states=embeds,
                            cross_attention_kwargs=cross_attention_kwargs, **extra).sample
            if guidance_scale > 1.0:                                            # pipeline.py:156-158
                out = ops.cfg_combine(out.float().contiguous(), guidance_scale)
            nz = None if noise_per_step is None else noise_per_step[i]
            # pipeline.py:161 calls scheduler.step(noise_pred, t, latents) WITHOUT the generator: the ancestral noise comes from
            # torch's global RNG (of the latents' device), the caller's generator only seeds the initial latents
            latents = self.scheduler.step(out.float().contiguous(), t, latents, noise=nz).prev_sample
            if callback is not None and i % callback_steps == 0:                # pipeline.py:165-166
                callback(i, t, latents)
        return latents


logger = logging.getLogger(__name__)


class MVDPipeline:
    """Mirror of /root/reference/src/models/pipeline.py::MVDPipeline (same ``__call__`` signature and return value)."""

    def __init__(self, unet, scheduler, vae=None, text_encoder=None, tokenizer=None, vae_scale_factor: int = 8):
        self.unet, self.scheduler = unet, scheduler
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.vae_scale_factor = vae_scale_factor
        self.safety_checker = None
        self.feature_extractor = None
        # attributes create_mvd_pipeline sets on the reference pipeline (mvd_unet.py:449-451)
        self.use_camera_conditioning = getattr(unet, "use_camera_conditioning", True)
        self.use_image_conditioning = getattr(unet, "use_image_conditioning", True)
        self.img_ref_scale = getattr(unet, "img_ref_scale", 0.3)

    # ------------------------------------------------------------------ small pieces of the diffusers base class
    @property
    def device(self) -> torch.device:
        return self.unet._exec_device() if torch.cuda.is_available() else torch.device("cpu")

    def to(self, *args, **kwargs):
        self.unet.to(*args, **kwargs)
        for m in (self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(*args, **kwargs)
        return self

    def progress_bar(self, iterable):
        return iterable

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        """StableDiffusionPipeline.prepare_latents: N(0,1) of the latent shape times ``init_noise_sigma``."""
        shape = (batch_size, num_channels_latents, int(height) // self.vae_scale_factor, int(width) // self.vae_scale_factor)
        if isinstance(generator, (list, tuple)):
            # diffusers' prepare_latents / randn_tensor: one generator per latent (the signature's List[torch.Generator],
            # pipeline.py:22) -- a list of another length is an error, each row is drawn from its own generator
            if len(generator) != batch_size:
                raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                                 f"batch size of {batch_size}. Make sure the batch size matches the length of the generators.")
            if len(generator) == 1:
                generator = generator[0]
        if latents is None:
            if isinstance(generator, (list, tuple)):
                rows = [torch.randn((1,) + shape[1:], generator=g, device=g.device, dtype=torch.float32).to(device) for g in generator]
                latents = torch.cat(rows, dim=0)
            else:
                gdev = generator.device if isinstance(generator, torch.Generator) else device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(device)
        return latents.to(device) * self.scheduler.init_noise_sigma

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        return [Image.fromarray((im * 255).round().astype("uint8")) for im in images]

    def _encode_prompt(self, prompt):
        if self.text_encoder is None or self.tokenizer is None:
            raise L.MvdError("MVDPipeline has no text encoder / tokenizer (no CLIP weights in this image): "
                             "pass prompt_embeds / negative_prompt_embeds")
        ti = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                            return_tensors="pt")
        return self.text_encoder(ti.input_ids.to(self.device))[0]

    # ------------------------------------------------------------------ pipeline.py:12-186
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 latents: Optional[torch.FloatTensor] = None, prompt_embeds: Optional[torch.FloatTensor] = None,
                 negative_prompt_embeds: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
                 callback_steps: int = 1, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 source_camera: Optional[torch.Tensor] = None, target_camera: Optional[torch.Tensor] = None,
                 source_images: Optional[torch.Tensor] = None, ref_scale: float = 0.1, use_camera_embeddings: bool = True,
                 use_image_conditioning: bool = True, debug_log_file_path: Optional[str] = None, *,
                 source_image_latents: Optional[torch.Tensor] = None, noise_per_step=None):
        dev = self.device
        if prompt is not None and isinstance(prompt, str):                       # :45-50
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        elif prompt_embeds is not None:
            batch_size = prompt_embeds.shape[0]
        else:
            raise L.MvdError("MVDPipeline: neither prompt nor prompt_embeds given")
        if prompt_embeds is None:                                                # :52-62
            prompt_embeds = self._encode_prompt(prompt if prompt is not None else "")
        if negative_prompt_embeds is None and negative_prompt is not None:       # :64-75
            negative_prompt_embeds = self._encode_prompt(negative_prompt)

        height = height or self.unet.config.sample_size * self.vae_scale_factor  # :82-83
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        if latents is None:                                                      # :85-95
            latents = self.prepare_latents(batch_size * num_images_per_prompt, 4, height, width, prompt_embeds.dtype, dev,
                                           generator)

        if source_images is not None:                                            # :100-117
            if self.vae is None:
                raise L.MvdError("MVDPipeline has no VAE: pass source_image_latents (scaled by the VAE scaling factor) "
                                 "instead of source_images, or attach mvd_amd.vae.AutoencoderKLHIP")
            si = source_images.to(device=dev)
            if bool(si.min() >= 0) and bool(si.max() <= 1):                      # one host sync, outside the loop
                si = 2 * si - 1
            if si.shape[0] < batch_size:
                si = si.repeat(batch_size // si.shape[0], 1, 1, 1)
            source_image_latents = self.vae.encode(si).latent_dist.sample() * self.vae.config.scaling_factor

        den = MVDDenoiser(self.unet, self.scheduler)
        latents = den(prompt_embeds, num_inference_steps, guidance_scale, negative_prompt_embeds=negative_prompt_embeds,
                      latents=latents, source_camera=source_camera, target_camera=target_camera,
                      source_image_latents=source_image_latents, generator=generator if isinstance(generator, torch.Generator) else None,
                      noise_per_step=noise_per_step, callback=callback, callback_steps=callback_steps,
                      cross_attention_kwargs=cross_attention_kwargs or {})

        if output_type == "latent":
            image = latents
        else:                                                                    # :168-181
            if self.vae is None:
                raise L.MvdError("MVDPipeline has no VAE to decode with: use output_type='latent'")
            image = self.vae.decode(latents / self.vae.config.scaling_factor).sample
            image = (image / 2 + 0.5).clamp(0, 1)
            if output_type == "pil":
                image = self.numpy_to_pil(image.cpu().permute(0, 2, 3, 1).float().numpy())
        if not return_dict:
            return image
        return {"images": image}


def _scheduler_from_snapshot(path) -> "Any":
    """DDPM scheduler of the snapshot (``<path>/scheduler/scheduler_config.json``) or SD-2.1's published defaults."""
    from .scheduler import DDPMScheduler
    cfg = {}
    f = os.path.join(str(path), "scheduler", "scheduler_config.json") if path else ""
    if f and os.path.exists(f):
        raw = json.load(open(f))
        cfg = {k: raw[k] for k in ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type",
                                   "timestep_spacing", "steps_offset") if k in raw}
    return DDPMScheduler(**cfg)


def _optional_components(path, dtype):
    """VAE / CLIP from a local diffusers snapshot when both the libraries and the files exist; None otherwise."""
    vae = text_encoder = tokenizer = None
    if not path or not os.path.isdir(str(path)):
        return vae, text_encoder, tokenizer
    try:
        from .vae import AutoencoderKLHIP
    except ImportError:
        AutoencoderKLHIP = None
    if AutoencoderKLHIP is not None and os.path.exists(os.path.join(str(path), "vae", "diffusion_pytorch_model.safetensors")):
        try:
            vae = AutoencoderKLHIP.from_snapshot(os.path.join(str(path), "vae"))
        except Exception as e:   # a snapshot that is there but does not load is an error, not "no VAE"
            raise L.MvdError(f"VAE snapshot under {os.path.join(str(path), 'vae')} failed to load: {type(e).__name__}: {e}") from e
    te_dir, tok_dir = os.path.join(str(path), "text_encoder"), os.path.join(str(path), "tokenizer")
    if os.path.isdir(te_dir) and os.path.isdir(tok_dir):
        try:
            from transformers import CLIPTextModel, CLIPTokenizer
        except ImportError:     # no library: the caller passes prompt_embeds (the pipeline says so when asked for a prompt)
            logger.warning("snapshot %s has a text encoder but `transformers` is not importable: pass prompt_embeds", path)
            return vae, None, None
        try:
            tokenizer = CLIPTokenizer.from_pretrained(tok_dir, local_files_only=True)
            text_encoder = CLIPTextModel.from_pretrained(te_dir, local_files_only=True, torch_dtype=dtype)
        except Exception as e:   # files that are there but do not load are an error, not "no text encoder"
            raise L.MvdError(f"text encoder / tokenizer under {path} failed to load: {type(e).__name__}: {e}") from e
    return vae, text_encoder, tokenizer


def build_pipeline(pretrained_model_name_or_path, dtype, use_camera_conditioning, use_image_conditioning, img_ref_scale,
                   cam_modulation_strength, cam_output_dim, cam_hidden_dim, simple_cam_encoder, cache_dir=None,
                   unet_config=None, init: str = "default") -> MVDPipeline:
    """``create_mvd_pipeline`` (mvd_unet.py:388-453): scheduler swap to the interpolated SNR shift (scale 6, hard-coded
    there, :420-428), ``MultiViewUNet`` as ``pipeline.unet``, the three attributes of :449-451.  Nothing is fetched: the
    name resolves to local snapshot files (hub.resolve_snapshot) or the call raises."""
    from .hub import resolve_snapshot
    from .mvd_unet import MultiViewUNet
    from .scheduler import DDPMScheduler, ShiftSNRScheduler
    # MVDPipeline.from_pretrained(name, cache_dir=...) (mvd_unet.py:411-415): a directory, or a hub name whose snapshot is in a
    # local huggingface cache; a name nothing local answers to raises MvdError (never a random-initialised pipeline)
    pretrained_model_name_or_path = resolve_snapshot(pretrained_model_name_or_path, cache_dir)
    base_scheduler = _scheduler_from_snapshot(pretrained_model_name_or_path)
    scheduler = ShiftSNRScheduler.from_scheduler(noise_scheduler=base_scheduler, shift_mode="interpolated", shift_scale=6.0,
                                                 scheduler_class=DDPMScheduler)
    unet = MultiViewUNet(pretrained_model_name_or_path, dtype=dtype, img_ref_scale=img_ref_scale,
                         cam_modulation_strength=cam_modulation_strength, cam_output_dim=cam_output_dim,
                         cam_hidden_dim=cam_hidden_dim, simple_cam_encoder=simple_cam_encoder,
                         use_camera_conditioning=use_camera_conditioning, use_image_conditioning=use_image_conditioning,
                         unet_config=unet_config, init=init)
    if torch.cuda.is_available():
        unet = unet.to(device="cuda", dtype=dtype)
    vae, text_encoder, tokenizer = _optional_components(pretrained_model_name_or_path, dtype)
    # (StableDiffusionPipeline.__init__: vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1))
    vsf = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
    pipe = MVDPipeline(unet, scheduler, vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, vae_scale_factor=vsf)
    pipe.use_camera_conditioning = use_camera_conditioning
    pipe.use_image_conditioning = use_image_conditioning
    pipe.img_ref_scale = img_ref_scale
    return pipe
