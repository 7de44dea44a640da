"""Deterministic tensors shared by tests/golden/make_golden.py (which runs the
reference) and the tests (which rebuild the same inputs/weights without it).

Every tensor is a function of its *name* only, so weights never have to be
stored in the fixtures -- only the reference's outputs are.
"""
import math
import zlib

import torch


def fx(name: str, shape, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g) * scale + shift


def fx_linear(name: str, out_f: int, in_f: int) -> torch.Tensor:
    return fx(name, (out_f, in_f), 1.0 / math.sqrt(in_f))


# --- G1: ImageCrossAttentionProcessor cases ---------------------------------
# name -> (C, heads, dim_head, B_hidden, B_ref, H, W)
G1_CASES = {
    "c64_h2_d32": (64, 2, 32, 2, 2, 4, 4),
    "c128_h2_b3": (128, 2, 64, 3, 3, 4, 3),
    "c320_h5": (320, 5, 64, 2, 2, 4, 4),
    "c640_h10_b1": (640, 10, 64, 1, 1, 4, 4),
    "c1280_h20": (1280, 20, 64, 2, 2, 2, 2),
    "cfg_mismatch_q4": (320, 5, 64, 2, 1, 4, 4),
    "cfg_mismatch_q4_b2": (128, 2, 64, 4, 2, 4, 4),
}
G1_REF_SCALE = 0.3


def g1_weights(case: str, C: int):
    return {
        "to_q_ref.weight": fx_linear(f"{case}.q", C, C),
        "to_k_ref.weight": fx_linear(f"{case}.k", C, C),
        "to_v_ref.weight": fx_linear(f"{case}.v", C, C),
        "to_out_ref.0.weight": fx_linear(f"{case}.o", C, C),
        "to_out_ref.0.bias": fx(f"{case}.ob", (C,), 0.1),
    }


def g1_inputs(case: str):
    C, heads, d, Bh, Br, H, W = G1_CASES[case]
    hidden = fx(f"{case}.hidden", (Bh, H * W, C))
    ref = fx(f"{case}.ref", (Br, C, H, W), 1.7, 0.3)
    orig = fx(f"{case}.orig", (Bh, H * W, C))
    return hidden, ref, orig


# --- G2: CameraEncoder -------------------------------------------------------
G2_SEED = 1234
SD21_MOD_DIMS = {
    "down_0": 320, "down_1": 640, "down_2": 1280, "down_3": 1280,
    "up_0": 1280, "up_1": 1280, "up_2": 640, "up_3": 320, "mid": 1280, "output": 4,
}
# variant -> (output_dim, hidden_dim, simple_encoder, modulation dims, strength)
G2_VARIANTS = {
    "full": (1024, 512, False, SD21_MOD_DIMS, 0.2),
    "simple": (1024, 512, True, SD21_MOD_DIMS, 1.0),
    "small": (96, 48, False, {"down_0": 64, "up_0": 128, "mid": 128, "output": 4}, 0.5),
}


def g2_param(variant: str, key: str, shape) -> torch.Tensor:
    n = f"cam.{variant}.{key}"
    if len(shape) == 2:
        return fx_linear(n, shape[0], shape[1])
    if key.endswith("weight"):  # LayerNorm scale
        return fx(n, shape, 0.1, 1.0)
    return fx(n, shape, 0.1)


def g2_cameras(B: int = 3) -> tuple:
    """Random rigid 4x4 poses (dataset format, Q8) -- deterministic."""
    srcs, tgts = [], []
    for b in range(B):
        for lst, tag in ((srcs, "s"), (tgts, "t")):
            q, _ = torch.linalg.qr(fx(f"cam.pose.{tag}{b}", (3, 3)))
            m = torch.eye(4)
            m[:3, :3] = q
            m[:3, 3] = fx(f"cam.pos.{tag}{b}", (3,), 1.5)
            lst.append(m)
    return torch.stack(srcs), torch.stack(tgts)


def g2_mod_input(variant: str, name: str, dim: int, B: int) -> torch.Tensor:
    return fx(f"cam.{variant}.x.{name}", (B, dim, 2, 3))


# ---- G6: synthetic images for load_image (utils.py:36-49): name -> (height, width, PIL mode, target_size)
G6_CASES = {
    "rgba_64x48_to_32": (64, 48, "RGBA", (32, 32)),
    "rgb_40x40_to_64x48": (40, 40, "RGB", (64, 48)),
}


def g6_image(name: str):
    """Smooth-ish seeded uint8 image (random low-res grid upsampled + noise) so LANCZOS has structure to filter."""
    import numpy as np
    h, w, mode, _ = G6_CASES[name]
    c = len(mode)
    rng = np.random.RandomState(zlib_seed("g6." + name))
    coarse = rng.randint(0, 256, size=(h // 8 + 1, w // 8 + 1, c)).astype(np.float32)
    img = np.kron(coarse, np.ones((8, 8, 1), np.float32))[:h, :w]
    img = np.clip(img + rng.randint(-20, 21, size=img.shape), 0, 255)
    return img.astype(np.uint8)


def zlib_seed(name: str) -> int:
    import zlib
    return zlib.crc32(name.encode()) & 0x7FFFFFFF
