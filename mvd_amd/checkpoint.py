"""Checkpoint ingestion: Lightning ``.ckpt`` -> ``MultiViewUNet`` parameters with the key
rewriting of /root/reference/infer.py:46-69 (and val.py:242-268)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def remap_lightning_state_dict(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """keep ``unet.*`` keys, strip the prefix, and rewrite ``image_encoder.X`` -> ``image_encoder.unet.X``."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in state_dict.items():
        if not k.startswith("unet."):
            continue
        key = k.replace("unet.", "", 1)
        if key.startswith("image_encoder.") and not key.startswith("image_encoder.unet."):
            key = "image_encoder.unet." + key.split(".", 1)[1]
        out[key] = v
    return out


def load_lightning_checkpoint(model, path: str, map_location="cpu") -> Tuple[list, list]:
    """``model.load_state_dict(fixed, strict=False)`` exactly like infer.py; returns (missing, unexpected)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    res = model.load_state_dict(remap_lightning_state_dict(sd), strict=False)
    return list(res.missing_keys), list(res.unexpected_keys)
