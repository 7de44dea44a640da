"""Python handle on the C-ABI engine (include/mvd_hip.h).  torch is used only for device
memory, streams and the one-time weight packing; every FLOP of the forward runs in
libmvd_hip.so.  There is no CPU path: constructing an engine without a GPU raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .config import UNetConfig
from .packing import pack_camera, pack_unet


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MVDEngine:
    def __init__(self, cfg: UNetConfig, cam_output_dim: int = 1024, cam_hidden_dim: int = 512,
                 simple_cam_encoder: bool = False, cam_modulation_strength: float = 0.2, device="cuda:0",
                 small_batch_twins: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise L.MvdError("MVDEngine needs a MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        # second packed copies only a batch-1 forward reads (packing.pack_unet): on unless MVD_PACK_SMALL_BATCH_TWINS=0
        self.small_batch_twins = (os.environ.get("MVD_PACK_SMALL_BATCH_TWINS", "1") != "0") if small_batch_twins is None \
            else bool(small_batch_twins)
        c = L.mvd_config_t()
        c.in_channels, c.out_channels, c.num_levels = cfg.in_channels, cfg.out_channels, cfg.num_levels
        for i in range(cfg.num_levels):
            c.block_out_channels[i] = cfg.block_out_channels[i]
            c.num_heads[i] = cfg.num_heads[i]
        c.layers_per_block, c.cross_attention_dim = cfg.layers_per_block, cfg.cross_attention_dim
        c.norm_num_groups, c.norm_eps = cfg.norm_num_groups, cfg.norm_eps
        c.cam_output_dim, c.cam_hidden_dim = cam_output_dim, cam_hidden_dim
        c.simple_cam_encoder, c.cam_modulation_strength = int(simple_cam_encoder), cam_modulation_strength
        self.cam_output_dim = cam_output_dim
        h = C.c_void_p()
        L.call("mvd_engine_create", C.byref(c), C.byref(h))
        self._h = h
        self._weights: List[Dict[str, torch.Tensor]] = [{}, {}]   # keeps packed tensors alive
        self._arenas: Dict[torch.dtype, torch.Tensor] = {}        # one flat buffer per dtype once consolidated
        self._arenas_stale = True
        self._ws = None
        self._rc = None
        self._ws_key = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().mvd_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def _register(self, set_id: int, packed: Dict[str, torch.Tensor]):
        for slot, t in packed.items():
            assert t.is_cuda and t.is_contiguous()
            dt = {torch.float32: 0, torch.bfloat16: 1}[t.dtype]
            L.call("mvd_engine_set_weight", self._h, set_id, slot.encode(), _ptr(t), t.numel(), dt)
        self._weights[set_id].update(packed)
        self._arenas_stale = True

    def load_base(self, sd: Dict[str, torch.Tensor], adapter: bool, ref_scale: float):
        """``sd``: diffusers keys of base_unet (+ ``...processor.*`` adapter keys when ``adapter``)."""
        with torch.no_grad():
            self._register(0, pack_unet(sd, self.cfg, self.device, adapter, ref_scale, self.small_batch_twins))

    def load_camera(self, sd: Dict[str, torch.Tensor]):
        with torch.no_grad():
            self._register(0, pack_camera(sd, self.device, len(self.cfg.block_out_channels)))

    def load_image_encoder(self, sd: Dict[str, torch.Tensor]):
        with torch.no_grad():
            self._register(1, pack_unet(sd, self.cfg, self.device, False, 0.0, self.small_batch_twins))
        self.share_encoder_weights(False)

    def share_encoder_weights(self, enable: bool = True):
        """N4 (training.py:60-65): the image encoder's UNet equals the frozen base UNet -> the encoder pass reads weight
        set 0 and the second packed weight set (1.73 GB bf16) is dropped."""
        L.call("mvd_engine_share_encoder_weights", self._h, int(enable))
        if enable:
            L.call("mvd_engine_clear_weights", self._h, 1)
            self._weights[1] = {}
            self._arenas_stale = True

    def consolidate_weights(self):
        """Re-home every registered weight (both sets) in ONE flat device buffer per dtype and re-register the slots
        (same contents, 256-byte aligned views).  Makes the multi-GPU start-up broadcast a few large in-place
        transfers (mvd_amd/distributed.py) and the weights one contiguous HBM region."""
        if self._arenas and not self._arenas_stale:
            return
        from .distributed import pack_into_arenas
        flat = {f"{i}/{k}": t for i, d in enumerate(self._weights) for k, t in d.items()}
        if not flat:
            return
        self._arenas, views = pack_into_arenas(flat)
        for name, v in views.items():
            i, k = name.split("/", 1)
            self._weights[int(i)][k] = v
        for i, d in enumerate(self._weights):
            for slot, t in d.items():
                dt = {torch.float32: 0, torch.bfloat16: 1}[t.dtype]
                L.call("mvd_engine_set_weight", self._h, i, slot.encode(), _ptr(t), t.numel(), dt)
        self._arenas_stale = False

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for d in self._weights for t in d.values())

    # ------------------------------------------------------------------ workspace
    def _ensure_workspace(self, batch, h, w, text_len, ref_batch, keep_features):
        key = (batch, h, w, text_len, ref_batch, keep_features)
        if self._ws_key == key:
            return
        ws = L.lib().mvd_engine_workspace_bytes(self._h, batch, h, w, text_len, ref_batch)
        if ws < 0:
            raise L.MvdError(f"workspace_bytes: {L.last_error()}")
        rc = 0
        if ref_batch > 0:
            rc = L.lib().mvd_engine_refcache_bytes(self._h, ref_batch, h, w, int(keep_features))
            if rc < 0:
                raise L.MvdError(f"refcache_bytes: {L.last_error()}")
        if self._ws is None or self._ws.numel() < ws:
            self._ws = None
            self._ws = torch.empty(ws, dtype=torch.uint8, device=self.device)
        if rc and (self._rc is None or self._rc.numel() < rc):
            self._rc = None
            self._rc = torch.empty(rc, dtype=torch.uint8, device=self.device)
        L.call("mvd_engine_bind_workspace", self._h, _ptr(self._ws), self._ws.numel(),
               _ptr(self._rc) if rc else None, self._rc.numel() if rc else 0)
        self._ws_key = key
        self._ref_valid = None      # re-binding drops the engine's cached reference K/V

    # ------------------------------------------------------------------ forward
    def forward(self, sample: torch.Tensor, timesteps: torch.Tensor, text: torch.Tensor,
                source_camera: Optional[torch.Tensor] = None, target_camera: Optional[torch.Tensor] = None,
                fourier_proj: Optional[torch.Tensor] = None, source_latents: Optional[torch.Tensor] = None,
                encoder_text: Optional[torch.Tensor] = None, reuse_ref: bool = False,
                keep_features: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """All tensors fp32 contiguous on ``self.device``; returns fp32 NCHW."""
        B, Cin, H, W = sample.shape
        Lt = text.shape[1]
        for t in (sample, timesteps, text, source_camera, target_camera, fourier_proj, source_latents, encoder_text):
            if t is not None:
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise L.MvdError("engine.forward expects contiguous fp32 CUDA tensors")
        if text.shape[0] != B or timesteps.shape != (B,):
            raise L.MvdError(f"engine.forward: text batch {text.shape[0]} / timesteps {tuple(timesteps.shape)} vs batch {B}")
        use_cam = target_camera is not None
        use_img = source_latents is not None or reuse_ref
        ref_batch = 0
        if use_img:
            ref_batch = source_latents.shape[0] if source_latents is not None else self._last_ref_batch
            self._last_ref_batch = ref_batch
        self._ensure_workspace(B, H, W, Lt, ref_batch, keep_features)
        graph = getattr(self, "_graph", False)
        user_out = out
        if graph:
            # hipGraph replay needs the SAME device buffers every call: stage the inputs into persistent buffers (tiny
            # device-to-device copies on the caller's stream) and run on the engine's own stream (captures are illegal on the
            # legacy default stream); mvd_engine_set_graph in include/mvd_hip.h
            stage = lambda name, t: None if t is None else self._staged(name, t)   # noqa: E731
            sample, timesteps, text = stage("sample", sample), stage("timesteps", timesteps), stage("text", text)
            source_camera, target_camera = stage("src_cam", source_camera), stage("tgt_cam", target_camera)
            fourier_proj, source_latents = stage("proj", fourier_proj), stage("lat", source_latents)
            encoder_text = stage("enc_text", encoder_text)
            out = self._staged("out", torch.empty(0), shape=(B, self.cfg.out_channels, H, W))
        elif out is None:
            out = torch.empty(B, self.cfg.out_channels, H, W, dtype=torch.float32, device=self.device)
        a = L.mvd_forward_args_t()
        a.batch, a.height, a.width, a.text_len = B, H, W, Lt
        a.sample, a.timesteps, a.text = sample.data_ptr(), timesteps.data_ptr(), text.data_ptr()
        flags = 0
        if use_cam:
            if source_camera is None or fourier_proj is None:
                raise L.MvdError("camera conditioning needs source_camera, target_camera and fourier_proj")
            Bc = target_camera.shape[0]
            if source_camera.shape[0] != Bc or Bc < 1 or Bc > B or B % Bc:
                raise L.MvdError(f"camera batch {tuple(source_camera.shape)} / {tuple(target_camera.shape)} must be equal "
                                 f"and divide the sample batch {B}")
            a.source_camera, a.target_camera = source_camera.data_ptr(), target_camera.data_ptr()
            a.cam_rows = source_camera.shape[1]
            a.cam_batch = Bc        # Bc < B: the FiLM scale/shift broadcast over the sample rows (CFG, pipeline.py:141-152)
            a.fourier_proj = fourier_proj.data_ptr()
            flags |= L.MVD_USE_CAMERA
        if use_img:
            flags |= L.MVD_USE_IMAGE
            if reuse_ref:
                flags |= L.MVD_REUSE_REF
            else:
                if encoder_text is None or encoder_text.shape[0] != ref_batch:
                    raise L.MvdError("image conditioning needs encoder_text with the reference batch")
                a.source_latents, a.encoder_text = source_latents.data_ptr(), encoder_text.data_ptr()
            if keep_features:
                flags |= L.MVD_KEEP_FEATURES
        a.ref_batch, a.flags = ref_batch, flags
        a.out = out.data_ptr()
        if graph:
            cur = torch.cuda.current_stream(self.device)
            self._gstream.wait_stream(cur)
            with torch.cuda.stream(self._gstream):
                L.call("mvd_unet_forward", self._h, C.byref(a), _stream())
            cur.wait_stream(self._gstream)
            out = out.clone() if user_out is None else user_out.copy_(out)
        else:
            L.call("mvd_unet_forward", self._h, C.byref(a), _stream())
        if use_img and not reuse_ref:
            self._ref_valid = (B, H, W, Lt, ref_batch)
        return out

    # ------------------------------------------------------------------ reference pass in two halves (global Q2 statistics)
    def reference_encode(self, source_latents: torch.Tensor, encoder_text: torch.Tensor, main_batch: int) -> torch.Tensor:
        """First half (``mvd_engine_reference_encode``): the image-encoder pass on ``source_latents``; the raw features stay in
        the engine, the LOCAL per-pixel statistics come back as ``[pixels][3]`` fp32 rows (n, mean, M2) with n = ref_batch * C
        of the pixel's feature -- what ``distributed.merge_reference_stats`` exchanges.  ``main_batch`` is the batch of the
        forwards that will follow (the workspace is bound once for both)."""
        for t in (source_latents, encoder_text):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise L.MvdError("engine.reference_encode expects contiguous fp32 CUDA tensors")
        Br, _, H, W = source_latents.shape
        Lt = encoder_text.shape[1]
        if encoder_text.shape[0] != Br:
            raise L.MvdError("image conditioning needs encoder_text with the reference batch")
        self._last_ref_batch = Br
        self._ensure_workspace(main_batch, H, W, Lt, Br, True)
        npix = L.lib().mvd_engine_reference_pixels(self._h, H, W)
        stats = torch.empty(npix, 2, dtype=torch.float32, device=self.device)
        a = L.mvd_forward_args_t()
        a.batch, a.height, a.width, a.text_len = main_batch, H, W, Lt
        a.source_latents, a.encoder_text = source_latents.data_ptr(), encoder_text.data_ptr()
        a.ref_batch, a.flags = Br, L.MVD_USE_IMAGE
        L.call("mvd_engine_reference_encode", self._h, C.byref(a), _ptr(stats), _stream())
        self._ref_valid = None
        self._ref_pending = (main_batch, H, W, Lt, Br)
        n = torch.empty(npix, dtype=torch.float32, device=self.device)
        off = 0
        c_, h_, w_ = C.c_int(), C.c_int(), C.c_int()
        for i in range(L.lib().mvd_engine_num_features(self._h)):
            L.call("mvd_engine_feature_shape", self._h, i, C.byref(c_), C.byref(h_), C.byref(w_))
            n[off:off + h_.value * w_.value] = float(Br * c_.value)
            off += h_.value * w_.value
        return torch.cat([n[:, None], stats], dim=1)

    def reference_finish(self, mean_k: torch.Tensor) -> None:
        """Second half (``mvd_engine_reference_finish``): ``mean_k`` = ``[pixels][2]`` fp32 (mean, 0.5 / max(std, 1e-6)) after the
        merge.  Afterwards ``forward(..., reuse_ref=True, keep_features=True)`` runs on this reference."""
        pend = getattr(self, "_ref_pending", None)
        if pend is None:
            raise L.MvdError("reference_finish without a pending reference_encode")
        npix = L.lib().mvd_engine_reference_pixels(self._h, pend[1], pend[2])
        if not (mean_k.is_cuda and mean_k.dtype == torch.float32 and mean_k.is_contiguous() and tuple(mean_k.shape) == (npix, 2)):
            raise L.MvdError(f"reference_finish expects a contiguous fp32 CUDA tensor [{npix}, 2]")
        L.call("mvd_engine_reference_finish", self._h, _ptr(mean_k), _stream())
        self._ref_pending = None
        self._ref_valid = pend

    def set_graph(self, enable: bool) -> None:
        """Replay repeated forwards (same shapes / flags) as one hipGraphLaunch each (``mvd_engine_set_graph``)."""
        L.call("mvd_engine_set_graph", self._h, int(bool(enable)))
        self._graph = bool(enable)
        if enable and getattr(self, "_gstream", None) is None:
            self._gstream = torch.cuda.Stream(device=self.device)
            self._gbufs = {}

    def _staged(self, name: str, t: torch.Tensor, shape=None) -> torch.Tensor:
        shape = tuple(t.shape) if shape is None else tuple(shape)
        buf = self._gbufs.get((name, shape))
        if buf is None:
            buf = self._gbufs[(name, shape)] = torch.empty(shape, dtype=torch.float32, device=self.device)
        if name != "out":
            buf.copy_(t)
        return buf

    def reference_cache_valid(self, batch, h, w, text_len, ref_batch) -> bool:
        """True when the engine still holds reference K/V computed for exactly this shape (Q5 reuse is then legal)."""
        return getattr(self, "_ref_valid", None) == (batch, h, w, text_len, ref_batch)

    # ------------------------------------------------------------------ measurement
    PROFILE_CLASSES = {0: "gemm_256x160", 1: "gemm_256x128", 2: "gemm_128x160", 3: "gemm_128x128", 4: "gemm_128x64",
                       5: "gemm_64x64", 6: "gemm_pp_256x320_geglu", 7: "gemm_pp_256x320_dense", 8: "attn_1wave", 9: "attn_2wave", 10: "attn_4wave", 11: "attn_8wave",
                       12: "gemm_128x320", 13: "gemm_pp_256x320_conv3x3", 14: "gemm_pp_256x320_splitk", 15: "gemm_pp_256x320_ln_dense",
                       16: "groupnorm", 17: "layernorm",
                       # small-M kernels (gemm_sm.hip), by tile
                       20: "gemm_sm_64x64", 21: "gemm_sm_128x64", 22: "gemm_sm_64x128", 23: "gemm_sm_128x128", 24: "gemm_sm_64x160",
                       25: "gemm_sm_128x160", 26: "gemm_sm_64x320",
                       # X-stationary short-K kernels (gemm_xs.hip)
                       30: "gemm_xs_dense", 31: "gemm_xs_residual", 32: "gemm_xs_ln_dense", 33: "gemm_xs_geglu",
                       # weight-streaming convolution of one image's 8x8 / 16x16 map (conv_ws.hip)
                       34: "conv_ws"}

    def set_profiling(self, enable):
        """False / 0: off.  True / 1: per-launch HIP events on ONE stream (serial kernel times).  2: per-launch events with the
        forward's own two-stream schedule kept (durations while the two passes share the chip)."""
        L.call("mvd_engine_set_profiling", self._h, int(enable))

    def profile_shapes(self) -> str:
        """Per-shape table of the recorded launches (call before profile_summary, which resets the records)."""
        buf = C.create_string_buffer(1 << 18)
        n = L.lib().mvd_engine_profile_shapes(self._h, buf, len(buf))
        if n < 0:
            raise L.MvdError(f"profile_shapes: {L.last_error()}")
        return buf.raw[:n].decode()

    def profile_summary(self):
        """{class name: dict(launches, ms, flops, bytes)} of the launches recorded since the last call."""
        cap = 32
        cls, n_l = (C.c_int * cap)(), (C.c_int * cap)()
        ms, fl, by = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
        n = L.lib().mvd_engine_profile_summary(self._h, cap, cls, n_l, ms, fl, by)
        if n < 0:
            raise L.MvdError(f"profile_summary: {L.last_error()}")
        return {self.PROFILE_CLASSES.get(cls[i], f"class{cls[i]}"): dict(launches=n_l[i], ms=ms[i], flops=fl[i], bytes=by[i])
                for i in range(n)}

    # ------------------------------------------------------------------ introspection (parity tests)
    def features(self) -> Dict[str, torch.Tensor]:
        names = [t[1] for t in self.cfg.transformers()]
        res = {}
        for i, n in enumerate(names):
            c, h, w = C.c_int(), C.c_int(), C.c_int()
            L.call("mvd_engine_feature_shape", self._h, i, C.byref(c), C.byref(h), C.byref(w))
            t = torch.empty(self._last_ref_batch, c.value, h.value, w.value, dtype=torch.float32, device=self.device)
            L.call("mvd_engine_get_feature", self._h, i, _ptr(t), _stream())
            res[n] = t
        return res

    def camera_embedding(self, batch: int) -> torch.Tensor:
        t = torch.empty(batch, self.cam_output_dim, dtype=torch.float32, device=self.device)
        L.call("mvd_engine_get_camera_embedding", self._h, _ptr(t), _stream())
        return t
