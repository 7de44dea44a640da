"""Host-side mirror of the reference ``MultiViewUNet`` (/root/reference/src/models/mvd_unet.py):
same constructor kwargs, attributes, state-dict keys and ``forward`` signature, so
``pipeline.py`` / ``infer.py`` of the reference can use it unchanged.

``forward`` hands device pointers to ONE C-ABI call (``mvd_unet_forward``): camera encoder,
the frozen reference-image UNet pass, the adapter K/V projections and the main UNet all run
as hand-written gfx950 kernels.  None of the reference's ~600 per-forward host syncs
(``.item()`` / formatted tensor statistics) exist here.  There is no CPU fallback.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .attention import get_attention_processor_for_module
from .camera_encoder import CameraEncoder
from .config import UNetConfig
from .unet_params import UNet2DConditionParams

logger = logging.getLogger(__name__)


class UNetOutput(NamedTuple):
    sample: torch.FloatTensor


def _build_unet(cfg: UNetConfig, init: str) -> UNet2DConditionParams:
    if init == "empty":     # skip the (slow) default initialisation; caller loads a state dict
        with torch.device("meta"):
            m = UNet2DConditionParams(cfg)
        return m.to_empty(device="cpu")
    return UNet2DConditionParams(cfg)


def _resolve_config(path, unet_config: Optional[UNetConfig]) -> UNetConfig:
    """``path``: a snapshot DIRECTORY (already resolved by hub.resolve_snapshot) or None."""
    if unet_config is not None:
        return unet_config
    if path and os.path.isdir(str(path)):
        cfg_path = os.path.join(str(path), "unet", "config.json")
        if os.path.exists(cfg_path):
            import json
            c = json.load(open(cfg_path))
            heads = c["attention_head_dim"]
            if isinstance(heads, int):
                heads = [heads] * len(c["block_out_channels"])
            return UNetConfig(in_channels=c["in_channels"], out_channels=c["out_channels"],
                              block_out_channels=tuple(c["block_out_channels"]), layers_per_block=c["layers_per_block"],
                              num_heads=tuple(heads), cross_attention_dim=c["cross_attention_dim"],
                              norm_num_groups=c["norm_num_groups"], norm_eps=c["norm_eps"],
                              sample_size=c.get("sample_size", 96))
    return UNetConfig.sd21()


def _load_local_unet_weights(module: nn.Module, path) -> bool:
    """Load ``<path>/unet/diffusion_pytorch_model.safetensors`` when a local diffusers snapshot exists."""
    if not path or not os.path.isdir(str(path)):
        return False
    f = os.path.join(str(path), "unet", "diffusion_pytorch_model.safetensors")
    if not os.path.exists(f):
        return False
    from safetensors.torch import load_file
    missing, unexpected = module.load_state_dict(load_file(f), strict=False)
    if missing:
        logger.warning("local UNet snapshot is missing %d keys (e.g. %s)", len(missing), missing[:3])
    return True


def _load_named_unet(module: nn.Module, name, snapshot, what: str):
    """A NAMED model must bring its weights (the reference's from_pretrained would): no silent random initialisation."""
    if name is None:
        return
    if not _load_local_unet_weights(module, snapshot):
        raise L.MvdError(f"{what}: {name!r} resolved to {snapshot!r}, which has no unet/diffusion_pytorch_model.safetensors")


class ImageEncoder(nn.Module):
    """Mirror of /root/reference/src/models/image_encoder.py: frozen UNet copy whose 16
    Transformer2DModel outputs are the reference features."""

    def __init__(self, pretrained_model_name_or_path, dtype: torch.dtype = torch.float32,
                 expected_sample_size: int = None, unet_config: Optional[UNetConfig] = None, init: str = "default",
                 cache_dir=None):
        super().__init__()
        from .hub import resolve_snapshot
        snapshot = resolve_snapshot(pretrained_model_name_or_path, cache_dir)      # raises for a name nothing local answers to
        cfg = _resolve_config(snapshot, unet_config)
        self.unet = _build_unet(cfg, init)
        _load_named_unet(self.unet, pretrained_model_name_or_path, snapshot, "ImageEncoder")
        self.unet.config.sample_size = expected_sample_size
        for p in self.unet.parameters():
            p.requires_grad = False
        self.unet.eval()
        self.dtype = dtype
        self.device = "cpu"
        self.extracted_features = {}
        self._encode_fn = None   # bound MultiViewUNet._encode_reference (a plain callable, not a submodule)

    def to(self, *args, **kwargs):
        device = args[0] if args else kwargs.get("device", self.device)
        self.device = device
        if "dtype" in kwargs:
            self.dtype = kwargs["dtype"]
        return super().to(*args, **kwargs)

    def forward(self, latents, text_embeddings, timestep=None):
        """image_encoder.py:97-112: returns {hook name: NCHW feature map}.  ``timestep`` is always 0 there."""
        if self._encode_fn is None:
            raise L.MvdError("ImageEncoder is not attached to a MultiViewUNet engine; there is no CPU fallback")
        self.extracted_features = self._encode_fn(latents, text_embeddings)
        return self.extracted_features


class MultiViewUNet(nn.Module):
    def __init__(self, pretrained_model_name_or_path=None, dtype: torch.dtype = torch.float32,
                 use_memory_efficient_attention: bool = True, enable_gradient_checkpointing: bool = True,
                 img_ref_scale: float = 0.3, cam_modulation_strength: float = 0.2, cam_output_dim: int = 1024,
                 cam_hidden_dim: int = 512, use_camera_conditioning: bool = True, use_image_conditioning: bool = True,
                 simple_cam_encoder: bool = False, *, unet_config: Optional[UNetConfig] = None, init: str = "default",
                 cache_reference: bool = False, dedup_encoder_weights="auto", cache_dir=None,
                 small_batch_twins: Optional[bool] = None):
        super().__init__()
        self.use_camera_conditioning = use_camera_conditioning
        self.use_image_conditioning = use_image_conditioning
        self.cam_output_dim, self.cam_hidden_dim = cam_output_dim, cam_hidden_dim
        self.simple_cam_encoder = simple_cam_encoder
        self.cam_modulation_strength = cam_modulation_strength
        self.cache_reference = cache_reference           # Q5: reuse reference K/V when inputs are the same tensors
        # replay repeated forwards (same shapes and flags) as one hipGraphLaunch each; pays at batch 1 (hundreds of short
        # launches), not at 32 pairs (kernel time = wall time).  Not a reference attribute.
        self.use_hip_graph = False
        # N4 (training.py:60-65, train_config.yaml:43): with a frozen base UNet the image encoder holds the same weights;
        # "auto" compares the two state dicts when the engine packs them and keeps ONE packed copy if they are equal
        self.dedup_encoder_weights = dedup_encoder_weights
        self.encoder_weights_shared = False
        # the second packed copies only a batch-1 forward reads (packing.pack_unet: ~0.7 GB per weight set).  None: the engine's
        # default (on unless MVD_PACK_SMALL_BATCH_TWINS=0); False for a deployment that only runs many-image batches -- the
        # data-parallel shards of BASELINE configs[3] / [4] then pack, hold and BROADCAST 3.7 GB instead of 6.5 GB
        self.small_batch_twins = small_batch_twins

        # mvd_unet.py:46-52: UNet2DConditionModel.from_pretrained(name, subfolder="unet").  Here: a snapshot directory or a
        # hub name already in a local huggingface cache (hub.resolve_snapshot); a name nothing local answers to RAISES
        from .hub import resolve_snapshot
        snapshot = resolve_snapshot(pretrained_model_name_or_path, cache_dir)
        self.pretrained_snapshot = snapshot
        cfg = _resolve_config(snapshot, unet_config)
        self.unet_config = cfg
        self.base_unet = _build_unet(cfg, init)
        _load_named_unet(self.base_unet, pretrained_model_name_or_path, snapshot, "MultiViewUNet")
        self.config = self.base_unet.config
        self.device = torch.device("cpu")
        self.dtype = dtype
        self.img_ref_scale = img_ref_scale

        if use_camera_conditioning:
            self.camera_encoder = CameraEncoder(output_dim=cam_output_dim, hidden_dim=cam_hidden_dim,
                                                modulation_hidden_dims=cfg.modulation_hidden_dims(),
                                                modulation_strength=cam_modulation_strength,
                                                simple_encoder=simple_cam_encoder)
        else:
            self.camera_encoder = None
        if use_image_conditioning:
            self.image_encoder = ImageEncoder(snapshot, dtype=dtype,
                                              expected_sample_size=self.config.sample_size, unet_config=cfg, init=init)
            self.image_encoder._encode_fn = self._encode_reference
        else:
            self.image_encoder = None
        self.hooks = []
        self._init_image_cross_attention()
        self.to(dtype=dtype)

        self._engine = None
        self._dirty = True
        self._ref_key = None
        self._ref_hold = None              # strong references to the tensors _ref_key was computed from
        # Q2 statistics across data-parallel shards: None = replica-local (what the reference's DDP replicas
        # compute); True = the default process group, or a torch.distributed group = statistics over ALL ranks' batches
        # (one 323 KB all-gather per reference pass), which makes the sharded result equal the unsharded one
        self.reference_stats_group = None
        self.current_camera_embedding = None
        self.fourier_projection = None     # set to a (cam_dim, 6*nfreq) tensor to pin Q1's per-call random matrix

    # ------------------------------------------------------------------ mvd_unet.py:106-162
    def _init_image_cross_attention(self):
        self.attention_layer_map = {}
        self.feature_to_attention_map = {}
        with torch.no_grad():
            for key, feat, _c, _h in self.unet_config.transformers():
                tb = self.base_unet.get_submodule(key).transformer_blocks[0]
                for attn, suffix in ((tb.attn1, "_self"), (tb.attn2, "_cross")):
                    name = feat + suffix
                    proc = get_attention_processor_for_module(name, attn, img_ref_scale=self.img_ref_scale)
                    self.attention_layer_map[name] = attn
                    attn.processor = proc          # nn.Module attribute -> registered submodule ("...processor.*" keys)
                    self.feature_to_attention_map.setdefault(feat, []).append(name)

    # ------------------------------------------------------------------ nn.Module protocol
    def to(self, *args, **kwargs):
        device = args[0] if args and not isinstance(args[0], torch.dtype) else kwargs.get("device", None)
        if "dtype" in kwargs:
            self.dtype = kwargs["dtype"]
        elif args and isinstance(args[0], torch.dtype):
            self.dtype = args[0]
        if device is not None:
            self.device = torch.device(device)
        self._dirty = True
        res = super().to(*args, **kwargs)
        if self.camera_encoder is not None:      # camera path is fp32-only (Q9)
            self.camera_encoder.float()
        return res

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._dirty = True
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def mark_weights_changed(self):
        """Call after mutating parameters in place; the packed bf16 device weights are rebuilt lazily."""
        self._dirty = True

    # ------------------------------------------------------------------ engine plumbing
    def _exec_device(self) -> torch.device:
        dev = self.device if isinstance(self.device, torch.device) else torch.device(self.device)
        if dev.type != "cuda":
            raise L.MvdError(f"MultiViewUNet is on {dev}: the MVD hot path only exists as HIP kernels for MI355X; "
                             "move the module to a cuda device (there is no CPU fallback)")
        return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())

    def _sync_engine(self):
        from .engine import MVDEngine
        dev = self._exec_device()
        if self._engine is None or self._engine.device != dev:
            self._engine = MVDEngine(self.unet_config, self.cam_output_dim, self.cam_hidden_dim, self.simple_cam_encoder,
                                     self.cam_modulation_strength, device=dev, small_batch_twins=self.small_batch_twins)
            self._dirty = True
        if self._dirty:
            sd = self.base_unet.state_dict()
            self._engine.load_base(sd, adapter=True, ref_scale=self.img_ref_scale)
            if self.camera_encoder is not None:
                self._engine.load_camera(self.camera_encoder.state_dict())
                self.camera_encoder._engine = self._engine
                self.camera_encoder._sync = self._sync_engine
            self.encoder_weights_shared = False
            if self.image_encoder is not None:
                esd = self.image_encoder.unet.state_dict()
                share = self.dedup_encoder_weights
                if share == "auto":
                    share = all(k in sd and sd[k].shape == v.shape and torch.equal(sd[k], v.to(sd[k].device)) for k, v in esd.items())
                if share:
                    self._engine.share_encoder_weights(True)
                    self.encoder_weights_shared = True
                else:
                    self._engine.load_image_encoder(esd)
            self._dirty = False
            self.reset_reference_cache()
        if bool(getattr(self, "use_hip_graph", False)) != bool(getattr(self._engine, "_graph", False)):
            self._engine.set_graph(bool(getattr(self, "use_hip_graph", False)))
        return self._engine

    def reset_reference_cache(self):
        """Forget the cached reference K/V (Q5): the next forward with image conditioning re-runs the encoder pass."""
        self._ref_key = None
        self._ref_hold = None

    @staticmethod
    def _f32(t: torch.Tensor, dev) -> torch.Tensor:
        return t.to(device=dev, dtype=torch.float32).contiguous()

    def _encode_reference(self, latents, text) -> Dict[str, torch.Tensor]:
        """Stand-alone ImageEncoder.forward: run the encoder pass only and return the 16 NCHW maps."""
        eng = self._sync_engine()
        dev = eng.device
        lat, txt = self._f32(latents, dev), self._f32(text, dev)
        B = lat.shape[0]
        dummy_t = torch.zeros(B, device=dev, dtype=torch.float32)
        eng.forward(lat, dummy_t, txt, source_latents=lat, encoder_text=txt, keep_features=True)
        self.reset_reference_cache()
        return eng.features()

    # ------------------------------------------------------------------ mvd_unet.py:179-338
    def forward(self, sample: torch.FloatTensor, timestep, encoder_hidden_states: torch.FloatTensor,
                source_camera: Optional[torch.Tensor] = None, target_camera: Optional[torch.Tensor] = None,
                source_image_latents: Optional[torch.FloatTensor] = None, return_dict: bool = True,
                timestep_cond: Optional[torch.FloatTensor] = None, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None):
        if cross_attention_kwargs:
            cross_attention_kwargs.pop("debug_log_file_path", None)   # accepted, never evaluated (no host syncs)
        eng = self._sync_engine()
        dev = eng.device
        out_dtype = sample.dtype if sample.is_floating_point() else self.dtype
        x = self._f32(sample, dev)
        B = x.shape[0]
        text = self._f32(encoder_hidden_states, dev)
        if B > text.shape[0]:                                          # CFG repeat (:233-237)
            text = text.repeat(B // text.shape[0], 1, 1)
        t = torch.as_tensor(timestep, device=dev).to(torch.float32).reshape(-1)
        t = t.expand(B).contiguous() if t.numel() == 1 else t.contiguous()

        cam = {}
        self.current_camera_embedding = None
        if self.use_camera_conditioning and target_camera is not None:
            proj = self.fourier_projection
            if proj is None:
                proj = self.camera_encoder.draw_projection(dev)        # Q1: fresh matrix per call
            cam = dict(source_camera=self._f32(source_camera, dev), target_camera=self._f32(target_camera, dev),
                       fourier_proj=self._f32(proj, dev))
        img = {}
        if self.use_image_conditioning and source_image_latents is not None:
            bs = source_image_latents.shape[0]
            enc_text = text
            if text.shape[0] == 2 * bs:                                # :280-283
                enc_text = text[bs:]
            elif text.shape[0] > bs:                                   # :284-285
                enc_text = text[:bs]
            # Q5 cache key: identity + version of the two input tensors.  The tensors themselves are HELD while the key
            # is live, so the caching allocator cannot hand their addresses to different data of the same shape (a second
            # object denoised right after the first would otherwise hit the first object's K/V).
            key = (source_image_latents.data_ptr(), source_image_latents._version, tuple(source_image_latents.shape),
                   encoder_hidden_states.data_ptr(), encoder_hidden_states._version, tuple(encoder_hidden_states.shape), B)
            glob = self.reference_stats_group is not None
            hit = (self.cache_reference and self._ref_key == key
                   and eng.reference_cache_valid(B, x.shape[2], x.shape[3], text.shape[1], bs))
            if glob and self.cache_reference:
                # the recompute path below contains a collective: the decision must be the same on every rank of the group
                # (a re-bound workspace or a fresh tensor on ONE rank would otherwise leave the others out of the all-gather)
                from .distributed import any_rank
                hit = not any_rank(not hit, None if self.reference_stats_group is True else self.reference_stats_group, dev)
            if hit:
                img = dict(reuse_ref=True, keep_features=glob)
            else:
                img = dict(source_latents=self._f32(source_image_latents, dev), encoder_text=enc_text.contiguous())
                self._ref_key = key
                self._ref_hold = (source_image_latents, encoder_hidden_states) if self.cache_reference else None
                if glob:
                    # SURVEY.md 8e mode (ii): the reference pass in two halves around ONE all-gather of per-pixel statistics
                    from .distributed import merge_reference_stats
                    group = None if self.reference_stats_group is True else self.reference_stats_group
                    local = eng.reference_encode(img["source_latents"], img["encoder_text"], B)
                    eng.reference_finish(merge_reference_stats(local, group))
                    img = dict(reuse_ref=True, keep_features=True)
        out = eng.forward(x, t, text, **cam, **img)
        if cam:
            self.current_camera_embedding = eng.camera_embedding(cam["target_camera"].shape[0])
        hidden_states = out.to(out_dtype)
        if not return_dict:
            return hidden_states
        return UNetOutput(sample=hidden_states)

    # kept for API compatibility with mvd_unet.py:340-385 (the engine applies both internally)
    def _map_image_features_to_attention_layers(self, image_features):
        ref = {}
        for name, feature in image_features.items():
            for attn_name in self.feature_to_attention_map.get(name, []):
                ref[attn_name] = feature
        return ref

    def _manage_modulation_hooks(self, register: bool):
        pass


def create_mvd_pipeline(pretrained_model_name_or_path: str, dtype: torch.dtype = torch.float16,
                        use_memory_efficient_attention: bool = True, enable_gradient_checkpointing: bool = True,
                        use_camera_conditioning: bool = True, use_image_conditioning: bool = True,
                        img_ref_scale: float = 0.25, cam_modulation_strength: float = 1.0, cam_output_dim: int = 1024,
                        cam_hidden_dim: int = 512, simple_cam_encoder: bool = False, cache_dir=None,
                        scheduler_config: Optional[Dict[str, Any]] = None, *, unet_config: Optional[UNetConfig] = None,
                        init: str = "default"):
    """mvd_unet.py:388-453: returns an ``MVDPipeline`` whose ``unet`` is this ``MultiViewUNet`` and whose scheduler is the
    interpolated SNR-shifted DDPM scheduler (``scheduler_config`` is accepted and ignored like the reference, :401,
    420-421).  Text encoder / VAE are attached when a local snapshot directory provides them; nothing is downloaded.
    ``unet_config`` / ``init`` are keyword-only extensions for checkpoint-free construction (tests, synthetic weights)."""
    from .pipeline import build_pipeline
    return build_pipeline(pretrained_model_name_or_path, dtype, use_camera_conditioning, use_image_conditioning,
                          img_ref_scale, cam_modulation_strength, cam_output_dim, cam_hidden_dim, simple_cam_encoder,
                          cache_dir, unet_config=unet_config, init=init)
