"""ctypes binding of libmvd_hip.so (the C ABI declared in include/mvd_hip.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MVD_HIP_LIB selects another build of the same library (A/B measurements: tools/build_variant.py)
LIB_PATH = os.environ.get("MVD_HIP_LIB") or os.path.join(HERE, "libmvd_hip.so")

MVD_MAX_LEVELS = 4
MVD_USE_CAMERA, MVD_USE_IMAGE, MVD_REUSE_REF, MVD_KEEP_FEATURES = 1, 2, 4, 8


class MvdError(RuntimeError):
    pass


class mvd_config_t(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("num_levels", C.c_int),
        ("block_out_channels", C.c_int * MVD_MAX_LEVELS), ("num_heads", C.c_int * MVD_MAX_LEVELS),
        ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int), ("norm_num_groups", C.c_int),
        ("norm_eps", C.c_float), ("cam_output_dim", C.c_int), ("cam_hidden_dim", C.c_int),
        ("simple_cam_encoder", C.c_int), ("cam_modulation_strength", C.c_float),
    ]


class mvd_vae_config_t(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("latent_channels", C.c_int), ("num_levels", C.c_int),
        ("block_out_channels", C.c_int * MVD_MAX_LEVELS), ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int),
        ("norm_eps", C.c_float),
    ]


class mvd_forward_args_t(C.Structure):
    _fields_ = [
        ("batch", C.c_int), ("height", C.c_int), ("width", C.c_int), ("text_len", C.c_int),
        ("sample", C.c_void_p), ("timesteps", C.c_void_p), ("text", C.c_void_p),
        ("source_camera", C.c_void_p), ("target_camera", C.c_void_p), ("cam_rows", C.c_int), ("cam_batch", C.c_int),
        ("fourier_proj", C.c_void_p), ("source_latents", C.c_void_p), ("encoder_text", C.c_void_p),
        ("ref_batch", C.c_int), ("flags", C.c_int), ("out", C.c_void_p),
    ]


_SIGS = {
    "mvd_last_error": (C.c_char_p, []),
    "mvd_engine_create": (C.c_int, [C.POINTER(mvd_config_t), C.POINTER(C.c_void_p)]),
    "mvd_engine_destroy": (C.c_int, [C.c_void_p]),
    "mvd_engine_set_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int64, C.c_int]),
    "mvd_engine_clear_weights": (C.c_int, [C.c_void_p, C.c_int]),
    "mvd_engine_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvd_engine_refcache_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvd_engine_bind_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "mvd_unet_forward": (C.c_int, [C.c_void_p, C.POINTER(mvd_forward_args_t), C.c_void_p]),
    "mvd_engine_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mvd_engine_profile_summary": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mvd_engine_profile_shapes": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "mvd_engine_set_graph": (C.c_int, [C.c_void_p, C.c_int]),
    "mvd_engine_reference_pixels": (C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    "mvd_engine_reference_encode": (C.c_int, [C.c_void_p, C.POINTER(mvd_forward_args_t), C.c_void_p, C.c_void_p]),
    "mvd_engine_reference_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvd_engine_share_encoder_weights": (C.c_int, [C.c_void_p, C.c_int]),
    "mvd_engine_num_features": (C.c_int, [C.c_void_p]),
    "mvd_engine_feature_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mvd_engine_get_feature": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_engine_encode_cameras": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvd_engine_apply_modulation": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_engine_get_camera_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvd_op_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_op_ln_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p,
                                   C.c_int, C.c_int, C.c_void_p]),
    "mvd_op_linear_xs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mvd_op_conv3x3_ws": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mvd_op_conv3x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_op_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "mvd_op_attention_split_ws_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvd_op_attention_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_op_groupnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvd_op_layernorm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvd_op_refnorm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_op_film": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvd_op_conv_in": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_op_conv_out": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_op_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvd_op_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_op_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mvd_op_skinny_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_void_p]),
    "mvd_gemm_num_configs": (C.c_int, []),
    "mvd_debug_last_gemm_plan": (C.c_int, [C.POINTER(C.c_int)]),
    "mvd_debug_last_gemm_nowait": (C.c_int, []),
    "mvd_debug_last_attention_plan": (C.c_int, [C.POINTER(C.c_int)]),
    "mvd_debug_pick_splitk": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvd_debug_pick_splitk_conv": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mvd_gemm_sm_num_tiles": (C.c_int, []),
    "mvd_debug_set_attention_nw": (C.c_int, [C.c_int]),
    "mvd_debug_set_flags": (C.c_int, [C.c_int]),
    "mvd_op_ddpm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "mvd_op_cfg_combine": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "mvd_vae_create": (C.c_int, [C.POINTER(mvd_vae_config_t), C.POINTER(C.c_void_p)]),
    "mvd_vae_destroy": (C.c_int, [C.c_void_p]),
    "mvd_vae_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int]),
    "mvd_vae_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvd_vae_bind_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mvd_vae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_vae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mvd_op_gaussian_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


def lib() -> C.CDLL:
    """Load libmvd_hip.so (once).  Raises if it has not been built -- no CPU fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MvdError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or python mvd_amd/_build.py)")
        # torch first: it ships its own libamdhip64 and must be the HIP runtime of the process -- if this library (linked
        # against /opt/rocm's copy) were loaded before torch, two runtimes would coexist and every launch on torch's
        # device pointers would fail ("no ROCm-capable device")
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error() -> str:
    return lib().mvd_last_error().decode()


def check(rc: int, what: str = "") -> int:
    if rc is None or rc < 0 or (rc != 0 and what.startswith("!")):
        raise MvdError(f"{what.lstrip('!')}: rc={rc}: {last_error()}")
    return rc


def call(name: str, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise MvdError(f"{name} failed (rc={rc}): {last_error()}")
    return rc
