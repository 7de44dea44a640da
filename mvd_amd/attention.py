"""Host-side mirror of the reference ``ImageCrossAttentionProcessor``
(/root/reference/src/models/attention.py:12-265): same constructor, parameters
(``to_q_ref``, ``to_k_ref``, ``to_v_ref``, unused ``ref_ln``, ``to_out_ref``), the
``load_original_weights`` initialisation rules and the attention-processor call protocol.

Inside ``MultiViewUNet.forward`` the adapter branch is fused into the engine's schedule
(fused q/k/v GEMMs, one attention launch for the block's own attention + the cross-view
attention, K-concatenated out-projection).  ``__call__`` below is the stand-alone protocol
entry point: the wrapped ``original_processor`` is ``AttnProcessor2_0HIP`` (the block's own attention) and the
reference branch runs the same HIP kernels through ``mvd_amd.ops`` -- no torch arithmetic, no CPU fallback.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import ops


def _bf16c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(device="cuda", dtype=torch.bfloat16).contiguous()


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(device="cuda", dtype=torch.float32).contiguous()


class AttnProcessor2_0HIP:
    """Stand-alone equivalent of diffusers-0.32.2 ``AttnProcessor2_0`` -- the ``original_processor`` the adapter wraps
    (/root/reference/src/models/attention.py:62-70) -- on the HIP kernels: ``to_q/to_k/to_v`` GEMMs, softmax(q.k^T/8).v per
    64-wide head, ``to_out[0]`` (+bias); dropout p=0, no mask, no group norm, residual_connection False,
    rescale_output_factor 1 (the SD-2.1 attention config).  Inside ``MultiViewUNet.forward`` the same arithmetic is
    fused into the engine's schedule; this entry point serves callers that drive a processor by hand."""

    def __call__(self, attn: Any, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None, *args, **kwargs):
        if attention_mask is not None:
            raise ValueError("AttnProcessor2_0HIP: attention masks are not supported (the MVD path never passes one)")
        inner = attn.to_q.out_features
        if inner != attn.heads * 64:
            raise ValueError("the HIP attention kernel supports dim_head == 64 only")
        out_dtype, out_device = hidden_states.dtype, hidden_states.device
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, hh, ww = hidden_states.shape
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        B, N, C = hidden_states.shape
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        L = ctx.shape[1]
        h = _bf16c(hidden_states).reshape(B * N, C)
        x = _bf16c(ctx).reshape(B * L, ctx.shape[2])
        q = ops.linear(h, _bf16c(attn.to_q.weight)).view(B, N, inner)
        wkv = torch.cat([_bf16c(attn.to_k.weight), _bf16c(attn.to_v.weight)], 0).contiguous()
        kv = ops.linear(x, wkv).view(B, L, 2 * inner)
        o = ops.attention(q, kv[:, :, :inner], kv[:, :, inner:], attn.heads)
        out = ops.linear(o.reshape(B * N, inner), _bf16c(attn.to_out[0].weight), _f32c(attn.to_out[0].bias)).view(B, N, C)
        if input_ndim == 4:
            out = out.transpose(-1, -2).reshape(b, c, hh, ww)
        return out.to(device=out_device, dtype=out_dtype)


class ImageCrossAttentionProcessor(nn.Module):
    def __init__(self, name: str, query_dim: int, heads: int, dim_head: int = 64, dropout: float = 0.0,
                 img_ref_scale: float = 0.3):
        super().__init__()
        self.name, self.heads, self.dim_head = name, heads, dim_head
        self.inner_dim = heads * dim_head
        self.query_dim = query_dim
        self.original_processor = None
        self.to_q_ref = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.to_k_ref = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.to_v_ref = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.ref_ln = nn.LayerNorm(self.inner_dim)          # constructed, never applied (attention.py:160-161)
        self.feature_adapter = None
        self.to_out_ref = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=True), nn.Dropout(dropout)])
        self.ref_scale_val = img_ref_scale

    # ------------------------------------------------------------------ stand-alone processor protocol
    def reference_branch(self, hidden_states: torch.Tensor, reference_nchw: torch.Tensor) -> torch.Tensor:
        """attention.py:95-161 on the GPU: returns the un-scaled (B,N,C) branch output in bf16."""
        if self.dim_head != 64:
            raise ValueError("the HIP attention kernel supports dim_head == 64 only")
        Bh, N, Cq = hidden_states.shape
        Br, Cr, H, W = reference_nchw.shape
        ref = _bf16c(reference_nchw.permute(0, 2, 3, 1).reshape(Br, H * W, Cr))
        refn = ops.refnorm(ref)                                             # Q2 statistics over (batch, channel)
        h = _bf16c(hidden_states).reshape(Bh * N, Cq)
        q = ops.linear(h, _bf16c(self.to_q_ref.weight)).view(Bh, N, self.inner_dim)
        wkv = torch.cat([_bf16c(self.to_k_ref.weight), _bf16c(self.to_v_ref.weight)], 0).contiguous()
        kv = ops.linear(refn.reshape(Br * H * W, Cr), wkv)                  # (Br*HW, 2*inner)
        nk = Br * H * W // Bh                                               # Q4: view(batch_of_hidden, -1, heads, d)
        kv = kv.view(Bh, nk, 2 * self.inner_dim)
        o = ops.attention(q, kv[:, :, : self.inner_dim], kv[:, :, self.inner_dim:], self.heads)
        out = ops.linear(o.reshape(Bh * N, self.inner_dim), _bf16c(self.to_out_ref[0].weight), _f32c(self.to_out_ref[0].bias))
        return out.view(Bh, N, Cq)

    def __call__(self, attn: Any, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None,
                 ref_hidden_states: Optional[Dict[str, torch.Tensor]] = None, *args, **kwargs) -> torch.Tensor:
        kwargs.pop("debug_log_file_path", None)
        if self.original_processor is None:
            raise RuntimeError(f"ImageCrossAttentionProcessor '{self.name}' has no original_processor "
                               "(construct it with get_attention_processor_for_module)")
        original_output = self.original_processor(attn, hidden_states, encoder_hidden_states, attention_mask,
                                                  temb=temb, *args, **kwargs)
        if ref_hidden_states is None or self.name not in ref_hidden_states:
            return original_output                                            # attention.py:72-81
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, hh, ww = hidden_states.shape
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        branch = self.reference_branch(hidden_states, ref_hidden_states[self.name])
        if input_ndim == 4:
            branch = branch.transpose(-1, -2).reshape(b, c, hh, ww)
        return original_output + self.ref_scale_val * branch.to(device=original_output.device, dtype=original_output.dtype)

    # ------------------------------------------------------------------ attention.py:199-245
    def load_original_weights(self, attn_module):
        with torch.no_grad():
            self.to_q_ref.weight.copy_(attn_module.to_q.weight)
            self.to_out_ref[0].weight.copy_(attn_module.to_out[0].weight)
            self.to_out_ref[0].bias.copy_(attn_module.to_out[0].bias)
            for mine, orig in ((self.to_k_ref, attn_module.to_k.weight), (self.to_v_ref, attn_module.to_v.weight)):
                k_out, k_in = mine.weight.shape
                o_out, o_in = orig.shape
                if (k_out, k_in) == (o_out, o_in):
                    mine.weight.copy_(orig)
                elif k_in >= o_in:                      # wider query dim: copy, zero the tail
                    mine.weight[:, :o_in].copy_(orig[: min(k_out, o_out), :])
                    if k_in > o_in:
                        mine.weight[:, o_in:].zero_()
                else:                                   # narrower: transposed leading block
                    mine.weight.copy_(orig[: min(k_out, o_out), :k_in].t())


def get_attention_processor_for_module(name, attn_module, img_ref_scale=0.3):
    """attention.py:248-265."""
    query_dim = attn_module.to_q.in_features
    heads = attn_module.heads
    dim_head = attn_module.to_q.out_features // heads
    processor = ImageCrossAttentionProcessor(name=name, query_dim=query_dim, heads=heads, dim_head=dim_head,
                                             img_ref_scale=img_ref_scale)
    processor.original_processor = attn_module.processor
    processor.load_original_weights(attn_module)
    return processor
