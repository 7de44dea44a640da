"""Host-side helpers either side of the hot path, with the names and semantics of /root/reference/src/utils.py
(``create_output_dirs`` :8-22, ``log_debug`` :25-34, ``load_image`` :36-49, ``create_camera_matrix`` :52-85) so that
``infer.py`` / ``val.py`` run against the drop-in ``src`` package (integration/src/utils.py).  Plain numpy / PIL on the
host; pinned by golden vectors G5 (cameras) and G6 (images) in tests/test_host_cpu.py.

``look_at`` is this build's own synthetic-camera generator (poses on a radius-2 sphere), used by
bench.py and the parity harness.
"""
from __future__ import annotations

import math
from datetime import datetime
from pathlib import Path

import numpy as np
import torch


def create_output_dirs(base_dir):
    run = Path(base_dir) / datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
    dirs = {name: run / name for name in ("checkpoints", "comparisons", "samples", "logs")}
    for d in dirs.values():
        d.mkdir(parents=True, exist_ok=True)
    return dirs


def log_debug(file_path, message):
    """Append ``<timestamp> - message`` to ``file_path``; never raises (a logging failure must not stop inference)."""
    if not file_path:
        return
    try:
        with open(file_path, "a") as f:
            f.write(f"{datetime.now().strftime('%Y-%m-%d %H:%M:%S.%f')} - {message}\n")
    except Exception as e:  # noqa: BLE001
        print(f"[Debug Log Error] Failed to write to {file_path}: {e}")


def load_image(image_path, target_size=(768, 768)):
    """RGB(A) file -> (1, 3, H, W) float32 in [-1, 1]: RGBA is composited on white, LANCZOS resize to ``target_size``
    (PIL's (width, height) order)."""
    from PIL import Image
    im = Image.open(image_path)
    if im.mode == "RGBA":
        white = Image.new("RGBA", im.size, (255, 255, 255, 255))
        im = Image.alpha_composite(white, im)
    im = im.convert("RGB").resize(target_size, Image.Resampling.LANCZOS)
    x = np.asarray(im, dtype=np.float32) / 127.5 - 1.0
    return torch.from_numpy(x).permute(2, 0, 1).unsqueeze(0)


def _unit(v, fallback):
    n = float(np.linalg.norm(v))
    return np.asarray(fallback, dtype=np.float64) if n < 1e-8 else v / n


def create_camera_matrix(position, target, up=None):
    """Look-at pose as a 3x4 [R | position] float tensor; columns of R = (right, up', -forward).  Degenerate inputs fall
    back like the reference: coincident position/target -> forward (0,0,-1); forward parallel to up -> right (1,0,0)."""
    position = np.asarray(position, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    up = np.asarray([0.0, 1.0, 0.0] if up is None else up, dtype=np.float64)
    fwd = _unit(target - position, (0.0, 0.0, -1.0))
    right = _unit(np.cross(fwd, up), (1.0, 0.0, 0.0))
    m = np.zeros((3, 4))
    m[:, 0], m[:, 1], m[:, 2], m[:, 3] = right, np.cross(right, fwd), -fwd, position
    return torch.from_numpy(m).float()


def look_at(azim_deg: float, elev_deg: float = 20.0, radius: float = 2.0) -> torch.Tensor:
    """4x4 camera-to-world look-at pose on a sphere around the origin (the dataset's ``matrix_world`` format, Q8)."""
    a, e = math.radians(azim_deg), math.radians(elev_deg)
    pos = torch.tensor([radius * math.cos(e) * math.sin(a), radius * math.sin(e), radius * math.cos(e) * math.cos(a)])
    fwd = -pos / pos.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]))
    right = right / right.norm()
    m = torch.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, torch.linalg.cross(right, fwd), -fwd, pos
    return m
