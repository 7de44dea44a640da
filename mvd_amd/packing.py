"""State-dict (reference / diffusers key names) -> packed device weights for libmvd_hip.so.

One-time host work (torch is used as plumbing for the permutes / casts).  Slot layouts
(documented in DESIGN.md "Weight slots"):

  conv          [Cout][Cin/64][ky][kx][64] bf16  (K = 9*Cin; channel-slice major, taps inner)
  resnet conv2  conv2 | conv_shortcut(1x1) concatenated along K, biases summed
  attn1.qkv     [QSCALE*to_q; to_k; to_v; (QSCALE*to_q_ref)]   rows concatenated; QSCALE = 64^-0.5 * log2(e)
  attn1.out     [to_out.0 | ref_scale * to_out_ref.0]     K concatenated, bias = b + ref_scale*b_ref
  attn2.q       [QSCALE*to_q; (QSCALE*to_q_ref)]     text_kv: every attn2 site's [to_k; to_v] stacked in module order
  ref_kv        [to_k_ref(self); to_v_ref(self); to_k_ref(cross); to_v_ref(cross)]
  ff1           GEGLU rows interleaved in blocks of 16: (16 value rows, 16 gate rows)
  temb_proj     every resnet's time_emb_proj stacked in module order (one GEMM per forward)
  <slot>.wf/.cf LayerNorm-folded twins of attn1.qkv / attn2.q / ff1 (fold_layernorm): the fused LayerNorm GEMM reads the
                un-normalised rows; packed for C <= LN_FOLD_MAX_C (the levels whose GEMMs are big enough for that kernel)
"""
from __future__ import annotations

from typing import Dict


import torch

from .config import UNetConfig

# softmax scale of a 64-wide head times log2(e): folded into every query projection (see pack_unet)
QSCALE = 64 ** -0.5 * 1.4426950408889634


def _bf(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).to(torch.bfloat16).contiguous()


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _conv_w(w: torch.Tensor, tap_major: bool = False) -> torch.Tensor:
    """[Cout][Cin][3][3] -> [Cout][K].  Implicit-GEMM convs use K = [Cin/64][ky][kx][64] (channel-slice major: the
    nine taps of a 64-channel slice are consecutive K slabs -> L2-resident re-reads); conv_in / conv_out (and any
    Cin that is not a multiple of 64) use the plain tap-major [ky][kx][Cin]."""
    co, ci, kh, kw = w.shape
    w = w.detach().float()
    if tap_major or ci % 64:
        return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
    return w.reshape(co, ci // 64, 64, kh, kw).permute(0, 1, 3, 4, 2).reshape(co, kh * kw * ci)


def block_weight(w: torch.Tensor) -> torch.Tensor:
    """[N][K] (K contiguous) -> the blocked layout of the small-M kernels (gemm_sm.hip, ``MvdGemmArgs::w_blocked``):
    [N/32][K/64] blocks of 32 rows x 128 bytes stored as the LDS image itself -- the 16-byte chunk ``c`` of row ``r`` sits in
    slot ``c ^ ((r >> 1) & 7)`` -- so that one LDS-DMA instruction of a workgroup copies one contiguous 4 KB block and a work
    item's K slice of a 32-row band is one contiguous range of HBM.  Same number of elements as ``w``."""
    n, k = w.shape
    assert n % 32 == 0 and k % 64 == 0, (n, k)
    b = w.reshape(n // 32, 32, k // 64, 8, 8).permute(0, 2, 1, 3, 4)            # [nb][kb][row][chunk][8]
    r = torch.arange(32, device=w.device)
    slot = torch.arange(8, device=w.device)
    src = slot[None, :] ^ ((r[:, None] >> 1) & 7)                               # chunk held by (row, slot)
    idx = src[None, None, :, :, None].expand(n // 32, k // 64, 32, 8, 8)
    return torch.gather(b, 3, idx).contiguous().reshape(n, k)


def _geglu_rows(w: torch.Tensor) -> torch.Tensor:
    """[8C, ...] -> rows re-ordered so each block of 32 = 16 value rows then the 16 matching gate rows."""
    half = w.shape[0] // 2
    val, gate = w[:half], w[half:]
    rest = w.shape[1:]
    v = val.reshape(half // 16, 1, 16, *rest)
    g = gate.reshape(half // 16, 1, 16, *rest)
    return torch.cat([v, g], dim=1).reshape(w.shape)


XS_K = 320      # operand width of the X-stationary kernels (gemm_xs.hip): the 64x64 level of SD-2.1


def _xs_row_perm(device) -> torch.Tensor:
    """MFMA row m of a 32-row unit -> channel of the unit it must hold so that lane (token, h) ends up with the 16
    CONSECUTIVE channels 16 h .. 16 h + 15 in its 16 accumulator registers (v_mfma_f32_32x32x16_bf16: register q of half h
    is row (q & 3) + 8 (q >> 2) + 4 h)."""
    m = torch.arange(32, device=device)
    return 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3)


def pack_xs(w: torch.Tensor, bias, geglu: bool = False, device=None) -> torch.Tensor:
    """[N][K] (+ bias [N] or None) -> the weight stream of gemm_xs.hip: ``[units][K/16 + 1][64][8]`` bf16.

    A unit is a 32-row tile in the order the kernel consumes it: k-step ``s < K/16`` is one MFMA A-operand fragment (lane
    ``(r, h)`` holds ``W[row(r)][16 s + 8 h .. + 8]``, rows permuted by ``_xs_row_perm``), the last k-step carries the bias
    split into two bf16 (hi at k = 0, lo at k = 1 of half 0) against the kernel's constant 1.0 operand.  ``geglu``: ``w``
    holds the value rows then the gate rows (diffusers' GEGLU.proj); units alternate gate | value of one 32-channel tile."""
    device = device if device is not None else w.device
    w = w.detach().to(device=device, dtype=torch.float32)
    n, k = w.shape
    assert k % 16 == 0 and n % (64 if geglu else 32) == 0, (n, k)
    b = torch.zeros(n, device=device) if bias is None else bias.detach().to(device=device, dtype=torch.float32)
    if geglu:
        half = n // 2
        order = torch.stack([torch.arange(half, n, device=device).reshape(-1, 32),          # gate tile j
                             torch.arange(0, half, device=device).reshape(-1, 32)], 1).reshape(-1)   # value tile j
        w, b = w[order], b[order]
    units, ks = n // 32, k // 16
    perm = _xs_row_perm(device)
    wu = w.reshape(units, 32, k)[:, perm]                                   # [unit][m][k]: MFMA row order
    bu = b.reshape(units, 32)[:, perm]
    frag = wu.reshape(units, 32, ks, 2, 8).permute(0, 2, 3, 1, 4)           # [unit][s][h][r][8]
    out = torch.zeros(units, ks + 1, 2, 32, 8, device=device, dtype=torch.bfloat16)
    out[:, :ks] = frag.to(torch.bfloat16)
    hi = bu.to(torch.bfloat16)
    lo = (bu - hi.float()).to(torch.bfloat16)
    out[:, ks, 0, :, 0] = hi
    out[:, ks, 0, :, 1] = lo
    return out.reshape(units, ks + 1, 64, 8).contiguous()


def pack_ws(w4: torch.Tensor, wsc=None, device=None) -> torch.Tensor:
    """conv weight ``[N][C][3][3]`` (+ 1x1 shortcut weight ``[N][Csc]`` or None) -> the weight stream of conv_ws.hip:
    per 16-channel column tile, the convolution's rounds ``[C/128][wave 4][tap 9][lane 64][8]`` then the shortcut's rounds
    ``[Csc/128][wave 4][lane 64][8]`` (bf16).  Lane ``(i, h) = (lane & 15, lane >> 4)`` of a 1 KB block holds the MFMA A-operand
    fragment ``W[16 ct + i][128 rd + 32 wave + 8 h .. + 8]`` of tap ``3 ky + kx`` (input pixel (y + ky - 1, x + kx - 1))."""
    device = device if device is not None else w4.device
    w4 = w4.detach().to(device=device, dtype=torch.float32)
    n, c = w4.shape[0], w4.shape[1]
    assert n % 16 == 0 and c % 128 == 0 and tuple(w4.shape[2:]) == (3, 3), tuple(w4.shape)
    t = w4.permute(0, 2, 3, 1).reshape(n // 16, 16, 9, c // 128, 4, 4, 8)        # [ct][i][tap][rd][wave][h][8]
    conv = t.permute(0, 3, 4, 2, 5, 1, 6).reshape(n // 16, -1)                   # [ct][rd][wave][tap][h][i][8]
    parts = [conv]
    if wsc is not None:
        wsc = wsc.detach().to(device=device, dtype=torch.float32).reshape(n, -1)
        sc = wsc.shape[1]
        assert sc % 128 == 0, sc
        u = wsc.reshape(n // 16, 16, sc // 128, 4, 4, 8)                         # [ct][i][rd][wave][h][8]
        parts.append(u.permute(0, 2, 3, 4, 1, 5).reshape(n // 16, -1))           # [ct][rd][wave][h][i][8]
    return torch.cat(parts, 1).to(torch.bfloat16).contiguous().reshape(-1)


# LayerNorm fold: only the 64x64 / 32x32 levels (C = 320 / 640 in SD-2.1) ever reach the fused kernel (it needs >= 200
# tiles of 256x320, i.e. many rows); deeper levels keep ln_kernel + the plain GEMM and get no folded twin
LN_FOLD_MAX_C = 1 << 30     # every level: the small-M kernels (batch 1) fold at C = 1280 too (the M = 32-images kernels stop at 640)
LN_FOLD_LARGE_BATCH_MAX_C = 640   # what the many-images kernels (gemm_pp / gemm_xs: mvd_gemm_ln_fold_ok) ever read


def fold_layernorm(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias, device):
    """LayerNorm(x).W^T + b  ==  rstd*(x.Wf^T - mean*c1) + c2  with  Wf = W.diag(gamma) (bf16), c1 = Wf.1 (summed over the
    bf16 values, so that the mean term cancels exactly what the GEMM accumulated), c2 = W.beta + b.  Returns (wf, cf[2][N])."""
    w, gamma, beta = w.detach().float(), gamma.detach().float(), beta.detach().float()
    wf = (w * gamma[None, :]).to(torch.bfloat16)
    c1 = wf.float().sum(1)
    c2 = w @ beta
    if bias is not None:
        c2 = c2 + bias.detach().float()
    return wf.to(device).contiguous(), torch.stack([c1, c2], 0).to(device=device, dtype=torch.float32).contiguous()


def ws_twin_shapes_ok(c: int, n: int, scc0: int, scc1: int) -> bool:
    """The channel rules of conv_ws.hip ``mvd_conv_ws_applicable``: input width a multiple of 128 (a round), output width a multiple
    of 16 (a column tile), and EACH source of a fused 1x1 shortcut a multiple of 128 on its own (the halves of a skip
    concatenation are separate tensors)."""
    return c > 0 and c % 128 == 0 and n > 0 and n % 16 == 0 and scc0 % 128 == 0 and scc1 % 128 == 0 and not (scc1 and not scc0)


def pack_unet(sd: Dict[str, torch.Tensor], cfg: UNetConfig, device, adapter: bool, ref_scale: float = 0.0,
              small_batch_twins: bool = True) -> Dict[str, torch.Tensor]:
    """``sd`` holds diffusers keys (no wrapper prefix) and, when ``adapter``, the ``...processor.*`` keys.

    ``small_batch_twins=False`` leaves out the second copies only a batch-1 forward reads -- the weight-streaming ``.ws``
    convolution twins (conv_ws.hip) and the LayerNorm-folded ``.wf/.cf`` twins above C = 640 (gemm_sm.hip): ~1.4 GB per SD-2.1
    weight set.  The engine looks every twin up by name and takes the tiled convolution / ln_kernel + plain GEMM route when it is
    absent (engine.hip ``try_ws`` / ``ln_gemm``), so a deployment that only ever runs many-image batches loses nothing."""
    fold_max_c = LN_FOLD_MAX_C if small_batch_twins else LN_FOLD_LARGE_BATCH_MAX_C
    out: Dict[str, torch.Tensor] = {}
    w_in = _conv_w(sd["conv_in.weight"], tap_major=True)                       # [C0][9*Cin] -> zero padded to K = 64 (one MFMA slab)
    out["conv_in.w"] = _bf(torch.nn.functional.pad(w_in, (0, 64 - w_in.shape[1])), device)
    out["conv_in.b"] = _f32(sd["conv_in.bias"], device)
    out["time.l1.w"] = _bf(sd["time_embedding.linear_1.weight"], device)
    out["time.l1.b"] = _f32(sd["time_embedding.linear_1.bias"], device)
    out["time.l2.w"] = _bf(sd["time_embedding.linear_2.weight"], device)
    out["time.l2.b"] = _f32(sd["time_embedding.linear_2.bias"], device)
    tw, tb = [], []
    split = cfg.resnet_input_split()
    for key, cin, cout in cfg.resnets():
        out[f"{key}.norm1.g"] = _f32(sd[f"{key}.norm1.weight"], device)
        out[f"{key}.norm1.b"] = _f32(sd[f"{key}.norm1.bias"], device)
        out[f"{key}.conv1.w"] = _bf(_conv_w(sd[f"{key}.conv1.weight"]), device)
        out[f"{key}.conv1.b"] = _f32(sd[f"{key}.conv1.bias"], device)
        out[f"{key}.norm2.g"] = _f32(sd[f"{key}.norm2.weight"], device)
        out[f"{key}.norm2.b"] = _f32(sd[f"{key}.norm2.bias"], device)
        w2 = _conv_w(sd[f"{key}.conv2.weight"])
        b2 = sd[f"{key}.conv2.bias"].detach().float()
        if cin != cout:
            w2 = torch.cat([w2, sd[f"{key}.conv_shortcut.weight"].detach().float().reshape(cout, cin)], dim=1)
            b2 = b2 + sd[f"{key}.conv_shortcut.bias"].detach().float()
        out[f"{key}.conv2.w"] = _bf(w2, device)
        out[f"{key}.conv2.b"] = _f32(b2, device)
        # weight-streaming twins (conv_ws.hip) for every level but the first -- the 32x32, 16x16 and 8x8 maps of a 64x64 latent,
        # where a batch-1 launch is a weight stream -- and only where the kernel's shape predicate can ever say yes
        # (ws_twin_shapes_ok = mvd_conv_ws_applicable's channel rules: a twin nothing can read is not packed, nor broadcast)
        if small_batch_twins and cout > cfg.block_out_channels[0]:
            if ws_twin_shapes_ok(cin, cout, 0, 0):
                out[f"{key}.conv1.ws"] = pack_ws(sd[f"{key}.conv1.weight"], None, device)
            if cin == cout:
                if ws_twin_shapes_ok(cout, cout, 0, 0):
                    out[f"{key}.conv2.ws"] = pack_ws(sd[f"{key}.conv2.weight"], None, device)
            elif ws_twin_shapes_ok(cout, cout, *split[key]):
                out[f"{key}.conv2.ws"] = pack_ws(sd[f"{key}.conv2.weight"], sd[f"{key}.conv_shortcut.weight"].reshape(cout, cin), device)
        tw.append(sd[f"{key}.time_emb_proj.weight"].detach().float())
        tb.append(sd[f"{key}.time_emb_proj.bias"].detach().float())
    out["temb_proj.w"] = _bf(torch.cat(tw, 0), device)
    out["temb_proj.b"] = _f32(torch.cat(tb, 0), device)

    tkv = []   # text K/V projections of every attn2 site, stacked in module order (one GEMM per pass)
    for key, _feat, C, _heads in cfg.transformers():
        b = f"{key}.transformer_blocks.0"
        out[f"{key}.norm.g"] = _f32(sd[f"{key}.norm.weight"], device)
        out[f"{key}.norm.b"] = _f32(sd[f"{key}.norm.bias"], device)
        out[f"{key}.proj_in.w"] = _bf(sd[f"{key}.proj_in.weight"], device)
        out[f"{key}.proj_in.b"] = _f32(sd[f"{key}.proj_in.bias"], device)
        out[f"{key}.proj_out.w"] = _bf(sd[f"{key}.proj_out.weight"], device)
        out[f"{key}.proj_out.b"] = _f32(sd[f"{key}.proj_out.bias"], device)
        for i in (1, 2, 3):
            out[f"{key}.ln{i}.g"] = _f32(sd[f"{b}.norm{i}.weight"], device)
            out[f"{key}.ln{i}.b"] = _f32(sd[f"{b}.norm{i}.bias"], device)
        f = lambda k: sd[k].detach().float()  # noqa: E731
        # every query projection carries the softmax scale and log2(e): the attention kernel then works in the exp2
        # domain without a per-score multiply (attn_kernel<.., PRE>); the factor is applied in fp32 before the bf16
        # rounding of the weights
        qkv = [QSCALE * f(f"{b}.attn1.to_q.weight"), f(f"{b}.attn1.to_k.weight"), f(f"{b}.attn1.to_v.weight")]
        q2 = [QSCALE * f(f"{b}.attn2.to_q.weight")]
        for a in ("attn1", "attn2"):
            wo, bo = f(f"{b}.{a}.to_out.0.weight"), f(f"{b}.{a}.to_out.0.bias")
            if adapter:
                pr = f"{b}.{a}.processor"
                out[f"{key}.{a}.out.b0"] = _f32(bo, device)      # plain bias for passes without the adapter
                wo = torch.cat([wo, ref_scale * f(f"{pr}.to_out_ref.0.weight")], dim=1)
                bo = bo + ref_scale * f(f"{pr}.to_out_ref.0.bias")
            out[f"{key}.{a}.out.w"] = _bf(wo, device)
            out[f"{key}.{a}.out.b"] = _f32(bo, device)
        if adapter:
            p1, p2 = f"{b}.attn1.processor", f"{b}.attn2.processor"
            qkv.append(QSCALE * f(f"{p1}.to_q_ref.weight"))
            q2.append(QSCALE * f(f"{p2}.to_q_ref.weight"))
            out[f"{key}.ref_kv.w"] = _bf(torch.cat([f(f"{p1}.to_k_ref.weight"), f(f"{p1}.to_v_ref.weight"),
                                                    f(f"{p2}.to_k_ref.weight"), f(f"{p2}.to_v_ref.weight")], 0), device)
        out[f"{key}.attn1.qkv.w"] = _bf(torch.cat(qkv, 0), device)
        out[f"{key}.attn2.q.w"] = _bf(torch.cat(q2, 0), device)
        tkv.append(torch.cat([f(f"{b}.attn2.to_k.weight"), f(f"{b}.attn2.to_v.weight")], 0))
        ff1_w, ff1_b = _geglu_rows(f(f"{b}.ff.net.0.proj.weight")), _geglu_rows(f(f"{b}.ff.net.0.proj.bias"))
        out[f"{key}.ff1.w"] = _bf(ff1_w, device)
        out[f"{key}.ff1.b"] = _f32(ff1_b, device)
        if C <= fold_max_c:
            for slot, w, bias, i in ((f"{key}.attn1.qkv", torch.cat(qkv, 0), None, 1), (f"{key}.attn2.q", torch.cat(q2, 0), None, 2),
                                     (f"{key}.ff1", ff1_w, ff1_b, 3)):
                out[f"{slot}.wf"], out[f"{slot}.cf"] = fold_layernorm(w, f(f"{b}.norm{i}.weight"), f(f"{b}.norm{i}.bias"), bias, device)
        if C == XS_K:
            # twins for the X-stationary kernels (gemm_xs.hip): the K = 320 projections of the 64x64 level
            out[f"{key}.proj_in.wx"] = pack_xs(f(f"{key}.proj_in.weight"), f(f"{key}.proj_in.bias"), device=device)
            out[f"{key}.proj_out.wx"] = pack_xs(f(f"{key}.proj_out.weight"), f(f"{key}.proj_out.bias"), device=device)
            for a in ("attn1", "attn2"):   # the out-projection of a pass WITHOUT the adapter branch (K = C)
                out[f"{key}.{a}.out.wx"] = pack_xs(f(f"{b}.{a}.to_out.0.weight"), f(f"{b}.{a}.to_out.0.bias"), device=device)
            if adapter:
                out[f"{key}.ref_kv.wx"] = pack_xs(out[f"{key}.ref_kv.w"].float(), None, device=device)
            for slot, geglu in ((f"{key}.attn1.qkv", False), (f"{key}.attn2.q", False), (f"{key}.ff1", True)):
                wf, cf = out[f"{slot}.wf"].float(), out[f"{slot}.cf"]
                if geglu:    # .wf / .cf are in the 16 | 16 interleaved row order of the ping-pong kernel: undo it
                    inv = torch.argsort(_geglu_rows(torch.arange(wf.shape[0], device=wf.device)))
                    wf, cf = wf[inv], cf[:, inv]
                out[f"{slot}.wx"] = pack_xs(wf, cf[1], geglu=geglu, device=device)
        out[f"{key}.ff2.w"] = _bf(sd[f"{b}.ff.net.2.weight"], device)
        out[f"{key}.ff2.b"] = _f32(sd[f"{b}.ff.net.2.bias"], device)

    out["text_kv.w"] = _bf(torch.cat(tkv, 0), device)
    n = cfg.num_levels
    for i in range(n - 1):
        out[f"down_blocks.{i}.down.w"] = _bf(_conv_w(sd[f"down_blocks.{i}.downsamplers.0.conv.weight"]), device)
        out[f"down_blocks.{i}.down.b"] = _f32(sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], device)
        out[f"up_blocks.{i}.up.w"] = _bf(_conv_w(sd[f"up_blocks.{i}.upsamplers.0.conv.weight"]), device)
        wu = sd[f"up_blocks.{i}.upsamplers.0.conv.weight"]
        if small_batch_twins and wu.shape[0] > cfg.block_out_channels[0] and wu.shape[0] % 128 == 0 and wu.shape[1] % 128 == 0:
            out[f"up_blocks.{i}.up.ws"] = pack_ws(wu, None, device)            # conv_ws.hip with the 2x upsampling in front
        out[f"up_blocks.{i}.up.b"] = _f32(sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], device)
    out["conv_norm_out.g"] = _f32(sd["conv_norm_out.weight"], device)
    out["conv_norm_out.b"] = _f32(sd["conv_norm_out.bias"], device)
    out["conv_out.w"] = _bf(_conv_w(sd["conv_out.weight"], tap_major=True), device)
    out["conv_out.b"] = _f32(sd["conv_out.bias"], device)
    return out


def pack_camera(sd: Dict[str, torch.Tensor], device, num_levels: int = 4) -> Dict[str, torch.Tensor]:
    """Camera encoder stays fp32 (Q9): slots are the reference keys prefixed with ``cam.``.  In addition the modulator MLPs the
    hooks address (down_i, up_i, output -- the engine's order; ``mid`` is never addressed, Q3) are stored CONCATENATED
    (``cam.modcat.*``), so that all of them run as four launches instead of four each (the camera path is ~50 tiny launches
    in front of a batch-1 forward)."""
    out = {f"cam.{k}": _f32(v, device) for k, v in sd.items()}
    names = [f"down_{i}" for i in range(num_levels)] + [f"up_{i}" for i in range(num_levels)] + ["output"]
    keys = [f"modulators.{n}.{j}.{t}" for n in names for j in (0, 1, 3) for t in ("weight", "bias")]
    if all(k in sd for k in keys):
        cat = lambda j, t: torch.cat([sd[f"modulators.{n}.{j}.{t}"].detach().float() for n in names], 0)   # noqa: E731
        out["cam.modcat.w0"], out["cam.modcat.b0"] = _f32(cat(0, "weight"), device), _f32(cat(0, "bias"), device)
        out["cam.modcat.g1"], out["cam.modcat.be1"] = _f32(cat(1, "weight"), device), _f32(cat(1, "bias"), device)
        out["cam.modcat.w3"], out["cam.modcat.b3"] = _f32(cat(3, "weight"), device), _f32(cat(3, "bias"), device)
    return out
