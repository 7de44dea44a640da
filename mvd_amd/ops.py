"""Operator-level wrappers over the C ABI (the same kernels the engine schedules).

Used by the parity tests and for bring-up; tensors are bf16/fp32 CUDA tensors, token-major
(NHWC) activations.  No fallback: everything dispatches into libmvd_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "ops expect contiguous CUDA tensors"
    return C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.bfloat16, "expected bf16"


def linear(a, w, bias=None, a2=None, rowvec=None, rows_per_batch=0, res=None, alpha=1.0, geglu=False,
           out_f32=False, force_cfg=-1, splitk=1):
    """out = alpha*( [a|a2] @ w.T + bias + rowvec[row // rows_per_batch] ) + res"""
    _bf16(a, a2, w, res)
    m, k1 = a.shape
    k2 = a2.shape[1] if a2 is not None else 0
    n = w.shape[0]
    assert w.shape[1] == k1 + k2
    on = n // 2 if geglu else n
    out = torch.empty(m, on, device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    # (+ 4096 words: the tile counters of the small-M kernels' in-kernel split-K combine, MVD_OP_SPLITK_COUNTERS)
    ws = torch.empty(splitk * m * n + 4096, device=a.device, dtype=torch.float32) if splitk > 1 else None
    L.call("mvd_op_linear", _p(a), _p(a2), k1, k2, _p(w), _p(bias), _p(rowvec),
           rowvec.shape[1] if rowvec is not None else 0, rows_per_batch, _p(res), float(alpha), int(geglu),
           _p(out), int(out_f32), m, n, force_cfg, splitk, _p(ws), _s())
    return out


def ln_linear(x, w_folded, cf, eps=1e-5, geglu=False):
    """LayerNorm(x).W^T + b (optionally GEGLU) in one kernel; (w_folded, cf) from packing.fold_layernorm."""
    _bf16(x, w_folded)
    m, k = x.shape
    n = w_folded.shape[0]
    assert w_folded.shape[1] == k and cf.shape == (2, n) and cf.dtype == torch.float32 and cf.is_contiguous()
    out = torch.empty(m, n // 2 if geglu else n, device=x.device, dtype=torch.bfloat16)
    L.call("mvd_op_ln_linear", _p(x), k, _p(w_folded), _p(cf[0]), _p(cf[1]), float(eps), int(geglu), _p(out), m, n, _s())
    return out


def linear_xs(x, w_packed, geglu=False, ln=False, eps=1e-5, res=None, csplit=0, k=None):
    """The X-stationary short-K GEMM (gemm_xs.hip): out = LN?(x[:, :k]) @ W.T + b (+ res), or value * gelu(gate).
    ``w_packed`` from packing.pack_xs (bias inside; ``ln``: packed from the fold_layernorm pair)."""
    _bf16(x, w_packed, res)
    m = x.shape[0]
    units, ks1 = w_packed.shape[0], w_packed.shape[1]
    k = (ks1 - 1) * 16 if k is None else k
    n_out = units * 16 if geglu else units * 32
    out = torch.empty(m, n_out, device=x.device, dtype=torch.bfloat16)
    assert x.stride(1) == 1 and (res is None or res.stride(1) == 1)
    L.call("mvd_op_linear_xs", C.c_void_p(x.data_ptr()), x.stride(0), _p(w_packed), m, k, units, int(geglu), int(ln), float(eps),
           C.c_void_p(res.data_ptr()) if res is not None else None, res.stride(0) if res is not None else 0, _p(out), n_out,
           csplit, _s())
    return out


def conv3x3_ws(x, w_packed, bias, n, rowvec=None, res=None, shortcut=None, shortcut2=None, variant=0, upsample=False):
    """The weight-streaming 3x3 convolution of small maps (conv_ws.hip): x (B, H, W, C) bf16 with W in {8, 16, 32}, H * W % 64 == 0,
    B * H * W <= 1024, C % 128 == 0; ``variant`` 0 = the launcher's choice, 1 / 2 force 64- / 128-pixel blocks; ``upsample``: nearest 2x in front of the
    convolution (diffusers' Upsample2D; output 2H x 2W of width 16 or 32, no shortcut); ``w_packed`` from packing.pack_ws (conv weight [n][C][3][3] and, optionally, the 1x1
    shortcut weight over ``shortcut`` | ``shortcut2`` rows); stride 1, padding 1.  Returns (B, H, W, n) bf16."""
    _bf16(x, w_packed, res, shortcut, shortcut2)
    b, h, w, c = x.shape
    out = torch.empty(b, 2 * h if upsample else h, 2 * w if upsample else w, n, device=x.device, dtype=torch.bfloat16)
    assert bias.dtype == torch.float32 and bias.numel() == n and x.is_contiguous()
    assert rowvec is None or (rowvec.dtype == torch.float32 and rowvec.stride(1) == 1)
    pp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    L.call("mvd_op_conv3x3_ws", _p(x), b, h, w, c, _p(w_packed), _p(bias), pp(rowvec), rowvec.stride(0) if rowvec is not None else 0,
           pp(res), pp(shortcut), pp(shortcut2), shortcut.shape[-1] if shortcut is not None else 0,
           shortcut2.shape[-1] if shortcut2 is not None else 0, _p(out), n, int(variant) + (16 if upsample else 0), _s())
    return out


def conv3x3(x, w_packed, bias=None, stride=1, upsample=False, rowvec=None, res=None, shortcut=None,
            shortcut2=None, force_cfg=-1, splitk=1, asym_pad=False):
    """x: (B,H,W,Cin) bf16; w_packed: (Cout, 9*Cin [+ Csc]) bf16 tap-major.  asym_pad (stride 2): zero padding on the
    bottom/right edge only (the VAE's Downsample2D(padding=0))."""
    _bf16(x, w_packed, res, shortcut, shortcut2)
    B, H, W, Cin = x.shape
    cout = w_packed.shape[0]
    oh = H * 2 if upsample else (H + 1) // 2 if stride == 2 else H
    ow = W * 2 if upsample else (W + 1) // 2 if stride == 2 else W
    out = torch.empty(B, oh, ow, cout, device=x.device, dtype=torch.bfloat16)
    c1 = shortcut.shape[-1] if shortcut is not None else 0
    c2 = shortcut2.shape[-1] if shortcut2 is not None else 0
    ws = torch.empty(splitk * B * oh * ow * cout + 4096, device=x.device, dtype=torch.float32) if splitk > 1 else None
    L.call("mvd_op_conv3x3", _p(x), B, H, W, Cin, stride, int(upsample), int(asym_pad), _p(w_packed), _p(bias), _p(rowvec),
           rowvec.shape[1] if rowvec is not None else 0, _p(res), _p(shortcut), _p(shortcut2), c1, c2, _p(out), cout,
           force_cfg, splitk, _p(ws), _s())
    return out


def attention(q, k, v, heads, scale=0.125):
    """q: (B,Nq,heads*64) bf16, k/v: (B,Nk,heads*64); row strides may exceed heads*64 (views of fused buffers).
    scale=0 selects the engine's form: q already multiplied by 64^-0.5 * log2(e) (packing.QSCALE)."""
    _bf16(q, k, v)
    B, nq, _ = q.shape
    nk = k.shape[1]
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    assert q.stride(0) == nq * q.stride(1) and k.stride(0) == nk * k.stride(1) and v.stride(0) == nk * v.stride(1)
    out = torch.empty(B, nq, heads * 64, device=q.device, dtype=torch.bfloat16)
    L.call("mvd_op_attention", C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), _p(out),
           B, heads, nq, nk, q.stride(1), k.stride(1), v.stride(1), heads * 64, float(scale), _s())
    return out


def attention_split(q, k, v, heads, nsplit):
    """Split-KV attention (engine form: q carries softmax_scale * log2 e); the keys are cut into ``nsplit`` ranges."""
    _bf16(q, k, v)
    B, nq, _ = q.shape
    nk = k.shape[1]
    out = torch.empty(B, nq, heads * 64, device=q.device, dtype=torch.bfloat16)
    ws = torch.empty(int(L.lib().mvd_op_attention_split_ws_bytes(B, heads, nq, nsplit)), device=q.device, dtype=torch.uint8)
    L.call("mvd_op_attention_split", C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), _p(out),
           B, heads, nq, nk, q.stride(1), k.stride(1), v.stride(1), heads * 64, nsplit, _p(ws), _s())
    return out


def groupnorm(x, gamma, beta, groups=32, eps=1e-5, silu=False, x2=None):
    """x: (B,HW,C0) bf16 [, x2: (B,HW,C1) concatenated on channels] -> (B,HW,C0+C1)"""
    _bf16(x, x2)
    B, hw, c0 = x.shape
    c1 = x2.shape[2] if x2 is not None else 0
    y = torch.empty(B, hw, c0 + c1, device=x.device, dtype=torch.bfloat16)
    ws = torch.empty(B * 256 * groups * 2, device=x.device, dtype=torch.float32)   # MVD_GN_MAXCHUNK partial sums
    L.call("mvd_op_groupnorm", _p(x), _p(x2), c0, c1, B, hw, groups, float(eps), _p(gamma), _p(beta), int(silu), _p(y),
           _p(ws), _s())
    return y


def layernorm(x, gamma, beta, eps=1e-5):
    _bf16(x)
    rows, c = x.shape
    y = torch.empty_like(x)
    L.call("mvd_op_layernorm", _p(x), rows, c, float(eps), _p(gamma), _p(beta), _p(y), _s())
    return y


def refnorm(x):
    """(B,HW,C) bf16 -> per-pixel normalisation over (batch, channel) (attention.py:95-103 of the reference)."""
    _bf16(x)
    B, hw, c = x.shape
    y = torch.empty_like(x)
    L.call("mvd_op_refnorm", _p(x), B, hw, c, _p(y), _s())
    return y


def film(x, scale, shift):
    _bf16(x)
    B, hw, c = x.shape
    y = torch.empty_like(x)
    L.call("mvd_op_film", _p(x), B, hw, c, _p(scale), _p(shift), _p(y), _s())
    return y


def conv_in(x, w, bias):
    """x (B,H,W,Cin) bf16, w (Cout,3,3,Cin) fp32 -> (B,H,W,Cout) bf16"""
    B, H, W, cin = x.shape
    cout = w.shape[0]
    y = torch.empty(B, H, W, cout, device=x.device, dtype=torch.bfloat16)
    L.call("mvd_op_conv_in", _p(x), B, H, W, cin, _p(w), _p(bias), cout, _p(y), _s())
    return y


def conv_out(x, w, bias):
    """x (B,H,W,C) bf16, w (Cout, 9*C) bf16 -> (B,Cout,H,W) fp32"""
    B, H, W, c = x.shape
    cout = w.shape[0]
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    L.call("mvd_op_conv_out", _p(x), B, H, W, c, _p(w), _p(bias), cout, _p(y), _s())
    return y


def ddpm_step(model_out, sample, noise, c0, c1, c2, c3, sigma):
    """fp32: x0 = c0*model_out + c1*sample ; prev = c2*x0 + c3*sample + sigma*noise (noise may be None iff sigma == 0)."""
    assert model_out.dtype == torch.float32 and sample.dtype == torch.float32
    out = torch.empty_like(sample)
    L.call("mvd_op_ddpm_step", _p(model_out), _p(sample), _p(noise), float(c0), float(c1), float(c2), float(c3),
           float(sigma), _p(out), sample.numel(), _s())
    return out


def cfg_combine(uncond_cond, guidance_scale):
    """(2B, ...) fp32 [uncond | cond] -> (B, ...) uncond + g*(cond - uncond)."""
    assert uncond_cond.dtype == torch.float32 and uncond_cond.shape[0] % 2 == 0
    out = torch.empty((uncond_cond.shape[0] // 2,) + tuple(uncond_cond.shape[1:]), device=uncond_cond.device,
                      dtype=torch.float32)
    L.call("mvd_op_cfg_combine", _p(uncond_cond), float(guidance_scale), _p(out), out.numel(), _s())
    return out


def skinny_linear(x, w, bias=None, silu_in=False):
    """fp32 linear layer of the camera / time MLPs: x (B, K) fp32, w (N, K) fp32 or bf16 -> (B, N) fp32."""
    assert x.dtype == torch.float32 and x.is_cuda and x.dim() == 2 and w.dim() == 2 and w.shape[1] == x.shape[1]
    assert w.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and w.is_contiguous()
    b, k = x.shape
    n = w.shape[0]
    y = torch.empty(b, n, device=x.device, dtype=torch.float32)
    L.call("mvd_op_skinny_linear", _p(x), k, b, k, _p(w), int(w.dtype == torch.bfloat16), _p(bias), n, int(silu_in), _p(y), n, _s())
    return y


def nchw_to_nhwc(x, scale=None, shift=None):
    B, c, H, W = x.shape
    y = torch.empty(B, H, W, c, device=x.device, dtype=torch.bfloat16)
    L.call("mvd_op_nchw_to_nhwc", _p(x), B, c, H * W, _p(scale), _p(shift), _p(y), _s())
    return y


def last_gemm_plan():
    """dict(cfg, splitk, tiles, grid, per_cu) of this thread's last GEMM / conv launch."""
    out = (C.c_int * 5)()
    L.call("mvd_debug_last_gemm_plan", out)
    return dict(cfg=out[0], splitk=out[1], tiles=out[2], grid=out[3], per_cu=out[4], nowait=L.lib().mvd_debug_last_gemm_nowait())


def last_attention_plan():
    out = (C.c_int * 2)()
    L.call("mvd_debug_last_attention_plan", out)
    return dict(waves=out[0], workgroups=out[1])


def engine_splitk(m, n, k, geglu=False, conv=False):
    """The split-K factor the engine's schedule uses for this GEMM (or, ``conv=True``, 3x3 convolution) size."""
    if conv:
        return L.lib().mvd_debug_pick_splitk_conv(m, n, k)
    return L.lib().mvd_debug_pick_splitk(m, n, k, int(geglu))
