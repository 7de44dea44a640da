"""Host-side mirror of the reference ``CameraEncoder`` (/root/reference/src/models/camera_encoder.py).

Same constructor, parameter names / shapes / initialisation and public methods
(``compute_relative_transform``, ``positional_encoding`` is folded into ``forward``,
``encode_cameras``, ``forward``, ``apply_modulation``, ``apply_modulation_to_tensor``).
The arithmetic runs in libmvd_hip.so through the engine this encoder is attached to
(``MultiViewUNet`` attaches it); the camera path stays fp32 (reference quirk Q9).

Differences from the reference, on purpose:
  * ``_current_modulation_stats`` is NOT refreshed on every call -- the reference's
    8 ``.item()`` host syncs per modulation site (camera_encoder.py:225-253) are pure
    overhead on the hot path.  Call ``collect_modulation_stats(True)`` to get them back
    for the standalone ``apply_modulation`` entry point.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


class CameraEncoder(nn.Module):
    def __init__(self, output_dim: int = 768, hidden_dim: int = 512, max_freq: int = 10,
                 modulation_hidden_dims: Dict[str, int] = None, modulation_strength: float = 1.0,
                 simple_encoder: bool = False):
        super().__init__()
        self.output_dim, self.hidden_dim, self.max_freq = output_dim, hidden_dim, max_freq
        self.simple_encoder = simple_encoder
        self.pos_enc_dim = (output_dim // 2) // 3

        def enc(din):
            if simple_encoder:
                return nn.Sequential(nn.Linear(din, hidden_dim), nn.LayerNorm(hidden_dim), nn.SiLU(),
                                     nn.Linear(hidden_dim, output_dim))
            return nn.Sequential(nn.Linear(din, hidden_dim), nn.LayerNorm(hidden_dim), nn.SiLU(),
                                 nn.Linear(hidden_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.SiLU(),
                                 nn.Linear(hidden_dim, output_dim))

        self.rotation_encoder = enc(9)
        self.translation_encoder = enc(output_dim)
        self.final_projection = nn.Sequential(nn.Linear(2 * output_dim, output_dim), nn.LayerNorm(output_dim), nn.SiLU(),
                                              nn.Linear(output_dim, output_dim), nn.LayerNorm(output_dim))
        self.output_norm = nn.LayerNorm(output_dim)
        self.modulation_hidden_dims = modulation_hidden_dims or {}
        self.modulators = nn.ModuleDict()
        for name, dim in self.modulation_hidden_dims.items():
            self.modulators[name] = nn.Sequential(nn.Linear(output_dim, output_dim // 2), nn.LayerNorm(output_dim // 2),
                                                  nn.SiLU(), nn.Linear(output_dim // 2, dim * 2))
        self.init_modulators()
        self.modulation_strength = modulation_strength
        self._current_modulation_stats = {}
        self._collect_stats = False
        self._engine = None          # set by MultiViewUNet
        self._sync = None            # callable making sure the engine holds current weights

    def init_modulators(self):       # camera_encoder.py:93-105
        for _, modulator in self.modulators.items():
            final = modulator[-1]
            nn.init.normal_(final.weight, mean=0.0, std=0.02)
            dim = final.out_features // 2
            final.bias.data[:dim].fill_(0.5)
            final.bias.data[dim:].fill_(0.0)

    def collect_modulation_stats(self, enable: bool = True):
        self._collect_stats = enable

    # ------------------------------------------------------------------ pure index math (no arithmetic kernels)
    def compute_relative_transform(self, source_camera: torch.Tensor, target_camera: torch.Tensor):
        sR, sT = source_camera[:, :3, :3], source_camera[:, :3, 3]
        tR, tT = target_camera[:, :3, :3], target_camera[:, :3, 3]
        R = torch.bmm(tR, sR.transpose(1, 2))
        T = tT - torch.bmm(R, sT.unsqueeze(2)).squeeze(2)
        return {"R": R, "T": T}

    def draw_projection(self, device) -> torch.Tensor:
        """Q1: the reference draws a fresh ``randn(out, enc)/sqrt(enc)`` on every call (camera_encoder.py:153-155)."""
        enc_dim = 6 * self.pos_enc_dim
        return torch.randn(self.output_dim, enc_dim, device=device) / np.sqrt(enc_dim)

    # ------------------------------------------------------------------ engine-backed arithmetic
    def _need_engine(self):
        if self._engine is None:
            raise L.MvdError("CameraEncoder is not attached to an MVD engine (construct it through MultiViewUNet); "
                             "there is no CPU fallback")
        if self._sync is not None:
            self._sync()
        return self._engine

    def encode_cameras(self, source_camera: torch.Tensor, target_camera: torch.Tensor,
                       fourier_proj: Optional[torch.Tensor] = None) -> torch.Tensor:
        eng = self._need_engine()
        dev = eng.device
        src = source_camera.to(device=dev, dtype=torch.float32).contiguous()
        tgt = target_camera.to(device=dev, dtype=torch.float32).contiguous()
        if fourier_proj is None:
            fourier_proj = self.draw_projection(dev)
        proj = fourier_proj.to(device=dev, dtype=torch.float32).contiguous()
        B = src.shape[0]
        eng._ensure_workspace(max(B, 1), 8, 8, 8, 0, False) if eng._ws is None else None
        out = torch.empty(B, self.output_dim, device=dev, dtype=torch.float32)
        L.call("mvd_engine_encode_cameras", eng._h, C.c_void_p(src.data_ptr()), C.c_void_p(tgt.data_ptr()),
               src.shape[1], B, C.c_void_p(proj.data_ptr()), C.c_void_p(out.data_ptr()),
               C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return out

    def forward(self, camera_data: Dict[str, torch.Tensor]) -> torch.Tensor:
        """camera_encoder.py:178-196 takes the relative transform; rebuild 3x4 cameras that produce it."""
        R, T = camera_data["R"], camera_data["T"]
        B = R.shape[0]
        src = torch.zeros(B, 3, 4, device=R.device, dtype=torch.float32)
        src[:, :, :3] = torch.eye(3, device=R.device)
        tgt = torch.cat([R.float(), T.float().unsqueeze(2)], dim=2)
        return self.encode_cameras(src, tgt)

    def apply_modulation(self, hidden_states, modulator_name: str, camera_embedding: torch.Tensor):
        if isinstance(hidden_states, tuple):   # only element 0 is modulated (camera_encoder.py:201-205)
            return (self.apply_modulation_to_tensor(hidden_states[0], modulator_name, camera_embedding),) + hidden_states[1:]
        return self.apply_modulation_to_tensor(hidden_states, modulator_name, camera_embedding)

    def apply_modulation_to_tensor(self, tensor, modulator_name, camera_embedding):
        if modulator_name not in self.modulators or camera_embedding is None:
            return tensor                      # silent identity, e.g. "mid_0" (Q3)
        eng = self._need_engine()
        dev = eng.device
        x = tensor.to(device=dev, dtype=torch.float32).contiguous()
        emb = camera_embedding.to(device=dev, dtype=torch.float32).contiguous()
        B, Cc = x.shape[0], x.shape[1]
        hw = x[0, 0].numel()
        out = torch.empty_like(x)
        rc = L.lib().mvd_engine_apply_modulation(eng._h, modulator_name.encode(), C.c_void_p(emb.data_ptr()), B,
                                                 C.c_void_p(x.data_ptr()), Cc, hw, C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc < 0:
            raise L.MvdError(f"apply_modulation: {L.last_error()}")
        if rc == 1:
            return tensor
        out = out.to(tensor.dtype)
        if self._collect_stats:
            with torch.no_grad():
                self._current_modulation_stats[modulator_name] = {
                    "before_mean": tensor.mean().item(), "before_std": tensor.std().item(),
                    "after_mean": out.mean().item(), "after_std": out.std().item(),
                }
        return out
