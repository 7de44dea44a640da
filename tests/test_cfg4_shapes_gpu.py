"""Parity AT THE BENCHMARKED SHAPES (BASELINE.json configs[3]: 32 pairs, 64x64 latent, 77 text tokens).

Every kernel class of the cfg4 bench line is run here through the HEURISTIC path (force_cfg = -1, the engine's own
split-K choice) at the exact per-launch shapes of a B=32 forward and compared with an independent fp32 reference
computed by torch on the GPU (matmul / conv2d / SDPA / group_norm / layer_norm in fp32 on the same bf16-rounded
operands).  Each GEMM test ASSERTS, from the launch plan the library records, that there were more work items than
workgroups -- i.e. that the persistent cross-tile pipeline of the 256x320 kernel (next tile's first slab in flight under
the current tile's last MFMAs and epilogue, XCD-chunked tile walk) is the code that ran under the checker.

Tolerance: |err| <= 2^-7 * max|ref| (bf16 output, fp32 accumulate), GEGLU 2^-6 (product of two rounded factors).
The end-to-end test compares the whole forward at B=32 (camera + image conditioning on, cold) with the CPU oracle:
rel-L2 <= 2e-2, max-abs <= 5e-2 * max|ref| (the tolerances of the B=1 full-size test).
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

B32 = 32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd import ops as O
    return O


def grnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cuda").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(dtype)


def close(got, want, tol=2 ** -7, what=""):
    got, want = got.float(), want.float()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    rel_l2 = ((got - want).norm() / want.norm().clamp_min(1e-12)).item()
    assert err <= tol * ref + 1e-6, f"{what}: max-abs {err:.4g} vs ref max {ref:.4g} (rel-L2 {rel_l2:.3g})"
    assert rel_l2 <= 6e-3, f"{what}: rel-L2 {rel_l2:.3g}"      # bf16 rounding alone is ~2.3e-3
    return rel_l2


def _pack(w):
    from mvd_amd.packing import _conv_w
    return _conv_w(w, False).to(torch.bfloat16)


def _assert_persistent(ops, cfg_expected, what, n=0, m=0):
    """The 256x320 kernel with more tiles than workgroups (the persistent cross-tile pipeline).  The N = 640 shapes of the
    32x32 level have exactly 256 tiles = one per CU (that IS what the bench runs); every other shape must be multi-tile."""
    plan = ops.last_gemm_plan()
    assert plan["cfg"] == cfg_expected, (what, plan)
    if m == 32768 and n == 640:
        assert plan["tiles"] == plan["grid"] == 256, (what, plan)
    else:
        assert plan["tiles"] > plan["grid"], f"{what}: {plan} -- every workgroup owned one tile, the cross-tile pipeline did not run"
    return plan


# ------------------------------------------------------------------------------- dense GEMMs at M = 32*4096, 32*1024
@pytest.mark.parametrize("m,n,k,res", [
    (131072, 320, 320, False),     # proj_in / to_q(text) at L0: 105 launches per step
    (131072, 320, 320, True),      # proj_out (+ residual)
    (131072, 1280, 320, False),    # fused q|k|v|q_ref
    (131072, 320, 1280, True),     # ff2 (+ residual)
    (131072, 320, 640, True),      # to_out || ref_scale*to_out_ref (K-concatenated, two A sources) + residual
    (32768, 640, 640, False),      # L1 proj_in
    (32768, 2560, 640, False),     # L1 fused q|k|v|q_ref
    (32768, 640, 2560, True),      # L1 ff2
])
def test_linear_cfg4_shapes(ops, m, n, k, res):
    a, w = grnd(m, k, seed=1), grnd(n, k, scale=1 / math.sqrt(k), seed=2)
    bias = grnd(n, seed=3, dtype=torch.float32)
    r = grnd(m, n, seed=4) if res else None
    want = a.float() @ w.float().T + bias
    if res:
        want += r.float()
    if k == 640 and n == 320:     # the engine's out-projection form: A = [o_self | o_ref]
        got = ops.linear(a[:, :320].contiguous(), w, bias, a2=a[:, 320:].contiguous(), res=r)
    else:
        got = ops.linear(a, w, bias, res=r)
    _assert_persistent(ops, 7, f"linear {m}x{n}x{k}", n, m)
    close(got, want, what=f"linear {m}x{n}x{k} res={res}")


@pytest.mark.parametrize("m,c", [(131072, 320), (32768, 640), (8192, 1280)])
def test_geglu_cfg4_shapes(ops, m, c):
    """ff1 + GEGLU: N = 8C packed rows (16 value | 16 gate), tile config 6 (256x320, wave tile 64x160)."""
    from mvd_amd.packing import _geglu_rows
    k, n = c, 8 * c
    a, w = grnd(m, k, seed=5), grnd(n, k, scale=1 / math.sqrt(k), seed=6)
    bias = grnd(n, seed=7, dtype=torch.float32)
    h = a.float() @ w.float().T + bias
    want = h[:, : n // 2] * F.gelu(h[:, n // 2:])
    del h
    got = ops.linear(a, _geglu_rows(w).contiguous(), _geglu_rows(bias).contiguous(), geglu=True)
    _assert_persistent(ops, 6, f"geglu {m}x{n}x{k}")
    close(got, want, tol=2 ** -6, what=f"geglu M={m} C={c}")


@pytest.mark.parametrize("n,geglu", [(5120, True), (2560, False)])
def test_column_group_tile_walk_is_bit_identical(ops, n, geglu):
    """Round 5 (gemm_pp.hip, MvdGemmArgs::walk_cg): the 32x32-level launches whose weights exceed an XCD's L2 (GEGLU N 5120, the fused
    q|k|v|q_ref N 2560; K = 640, M = 32768) walk an XCD's row blocks in groups of 4 column tiles instead of 16 / 8 side by side.
    Same tiles, same arithmetic per tile: the output must equal the row-major walk's (debug flag 131072) bit for bit, and the fp32
    reference within tolerance."""
    from mvd_amd import _lib as L
    from mvd_amd.packing import _geglu_rows
    m, k = 32768, 640
    a, w = grnd(m, k, seed=11), grnd(n, k, scale=1 / math.sqrt(k), seed=12)
    bias = grnd(n, seed=13, dtype=torch.float32)
    if geglu:
        w, bias = _geglu_rows(w).contiguous(), _geglu_rows(bias).contiguous()
    got = ops.linear(a, w, bias, geglu=geglu)
    L.lib().mvd_debug_set_flags(131072)
    try:
        plain = ops.linear(a, w, bias, geglu=geglu)
    finally:
        L.lib().mvd_debug_set_flags(0)
    assert torch.equal(got, plain), "the column-group walk changed the result"
    if not geglu:
        close(got, a.float() @ w.float().T + bias, what=f"walk linear {m}x{n}x{k}")


# ------------------------------------------------------------------------------- implicit-GEMM convolutions at B = 32
@pytest.mark.parametrize("hw,cin,cout,extra", [
    (64, 320, 320, "rowvec"),          # L0 resnet conv1 (+ time-embedding row vector): M 131072, K 2880
    (64, 320, 320, "res"),             # L0 resnet conv2 (+ identity residual)
    (64, 960, 320, "rowvec"),          # up_blocks.3 conv1 on cat(hidden 640, skip 320): K 8640
    (32, 640, 640, "res"),             # L1: M 32768, K 5760
    (32, 1920, 640, "rowvec"),         # up_blocks.2 conv1: K 17280
])
def test_conv_cfg4_shapes(ops, hw, cin, cout, extra):
    x = grnd(B32, cin, hw, hw, seed=11)
    w = grnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin), seed=12)
    bias = grnd(cout, seed=13, dtype=torch.float32)
    want = F.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    kw = {}
    if extra == "rowvec":
        rv = grnd(B32, cout, seed=14, dtype=torch.float32)
        kw["rowvec"] = rv
        want = want + rv[:, None, None, :]
    else:
        r = grnd(B32, hw, hw, cout, seed=15)
        kw["res"] = r
        want = want + r.float()
    got = ops.conv3x3(xn, _pack(w), bias, **kw)
    _assert_persistent(ops, 7, f"conv {hw}x{hw} {cin}->{cout}", cout, B32 * hw * hw)
    close(got, want, what=f"conv {hw}^2 {cin}->{cout} {extra}")


def test_conv_plus_shortcut_cfg4_shape(ops):
    """up_blocks.3.resnets.0 conv2 || 1x1 conv_shortcut on cat(hidden 640, skip 320): one GEMM, K = 9*320 + 960."""
    cout, c1, c2, hw = 320, 640, 320, 64
    t2 = grnd(B32, cout, hw, hw, seed=21)
    s1, s2 = grnd(B32, hw, hw, c1, seed=22), grnd(B32, hw, hw, c2, seed=23)
    w = grnd(cout, cout, 3, 3, scale=1 / math.sqrt(9 * cout), seed=24)
    wsc = grnd(cout, c1 + c2, scale=1 / math.sqrt(c1 + c2), seed=25)
    bias = grnd(cout, seed=26, dtype=torch.float32)
    want = F.conv2d(t2.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1) + torch.cat([s1, s2], -1).float() @ wsc.float().T
    wp = torch.cat([_pack(w), wsc], dim=1).contiguous()
    got = ops.conv3x3(t2.permute(0, 2, 3, 1).contiguous(), wp, bias, shortcut=s1, shortcut2=s2)
    _assert_persistent(ops, 7, "conv+shortcut")
    close(got, want, what="conv2 || shortcut (640|320 -> 320)")


def test_conv_stride2_and_upsample_cfg4_shapes(ops):
    """down_blocks.0 downsampler (64^2 -> 32^2, stride 2) and up_blocks.2 upsampler (nearest 2x fused, 32^2 -> 64^2)."""
    x = grnd(B32, 320, 64, 64, seed=31)
    w = grnd(320, 320, 3, 3, scale=1 / math.sqrt(9 * 320), seed=32)
    bias = grnd(320, seed=33, dtype=torch.float32)
    want = F.conv2d(x.float(), w.float(), bias, padding=1, stride=2).permute(0, 2, 3, 1)
    got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous(), _pack(w), bias, stride=2)
    close(got, want, what="stride-2 conv 320 @64^2")
    x = grnd(B32, 640, 32, 32, seed=34)
    w = grnd(640, 640, 3, 3, scale=1 / math.sqrt(9 * 640), seed=35)
    bias = grnd(640, seed=36, dtype=torch.float32)
    want = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous(), _pack(w), bias, upsample=True)
    _assert_persistent(ops, 7, "upsample conv")
    close(got, want, what="upsample conv 640 @32^2->64^2")


@pytest.mark.parametrize("cin,sc", [(1280, 0), (2560, 0), (1280, 2560)])
def test_deep_level_conv_cfg4_shapes(ops, cin, sc):
    """The 8x8 level at 32 images (M = 2048, K = 11520 ... 23040, mid block and up_blocks.0): round 5 runs these through the 256x320
    tile cut EIGHT ways along K (256 workgroups) instead of 512 work items of the 128x160 tile at split 4 -- the split the
    engine's heuristic picks, the launch plan it leads to, and the result against fp32 conv2d (+ the fused 1x1 shortcut)."""
    hw, cout = 8, 1280
    k = 9 * cin + sc
    sk = ops.engine_splitk(B32 * hw * hw, cout, k, conv=True)
    assert sk == 8, sk
    x = grnd(B32, cin, hw, hw, seed=51)
    w = grnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin), seed=52)
    bias = grnd(cout, seed=53, dtype=torch.float32)
    want = F.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    wp, kw = _pack(w), {}
    if sc:
        s1, s2 = grnd(B32, hw, hw, sc // 2, seed=54), grnd(B32, hw, hw, sc // 2, seed=55)
        wsc = grnd(cout, sc, scale=1 / math.sqrt(sc), seed=56)
        want = want + torch.cat([s1, s2], -1).float() @ wsc.float().T
        wp, kw = torch.cat([wp, wsc], dim=1).contiguous(), dict(shortcut=s1, shortcut2=s2)
    got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous(), wp, bias, splitk=sk, **kw)
    plan = ops.last_gemm_plan()
    assert plan["cfg"] == 7 and plan["splitk"] == 8 and plan["tiles"] == 8 * 4 * 8, plan
    close(got, want, what=f"8x8-level conv {cin}->{cout} +sc{sc} split {sk}")
    # the operator entry point's own default (no split asked for) keeps the 128x160 tile: 32 unsplit big tiles would idle 224 CUs
    ops.conv3x3(x.permute(0, 2, 3, 1).contiguous(), wp, bias, **kw)
    assert ops.last_gemm_plan()["cfg"] == 2, ops.last_gemm_plan()


@pytest.mark.parametrize("kind", ["conv", "dense"])
def test_splitk_cfg4_shapes(ops, kind):
    """M = 8192 (16x16 level at B=32): the 256x320 grid is 128 tiles, so the engine cuts K in two (fp32 partials +
    reduce/epilogue pass).  conv 1280->1280 (K 11520) and ff2 (K 5120), with the split the ENGINE's heuristic picks."""
    if kind == "conv":
        x = grnd(B32, 1280, 16, 16, seed=41)
        w = grnd(1280, 1280, 3, 3, scale=1 / math.sqrt(9 * 1280), seed=42)
        bias = grnd(1280, seed=43, dtype=torch.float32)
        r = grnd(B32, 16, 16, 1280, seed=44)
        sk = ops.engine_splitk(8192, 1280, 11520)
        assert sk == 2, sk
        want = F.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1) + r.float()
        got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous(), _pack(w), bias, res=r, splitk=sk)
    else:
        a, w = grnd(8192, 5120, seed=45), grnd(1280, 5120, scale=1 / math.sqrt(5120), seed=46)
        bias, r = grnd(1280, seed=47, dtype=torch.float32), grnd(8192, 1280, seed=48)
        sk = ops.engine_splitk(8192, 1280, 5120)
        assert sk == 2, sk
        want = a.float() @ w.float().T + bias + r.float()
        got = ops.linear(a, w, bias, res=r, splitk=sk)
    plan = ops.last_gemm_plan()
    assert plan["cfg"] == 7 and plan["splitk"] == 2 and plan["tiles"] == 256, plan
    close(got, want, what=f"split-K {kind}")


def test_short_k_m8192_uses_128x160_without_split(ops):
    """K < 4096 at M = 8192: the heuristic takes 128x160 tiles (two workgroups per CU) and no split."""
    a, w = grnd(8192, 1280, seed=51), grnd(1280, 1280, scale=1 / math.sqrt(1280), seed=52)
    bias, r = grnd(1280, seed=53, dtype=torch.float32), grnd(8192, 1280, seed=54)
    assert ops.engine_splitk(8192, 1280, 1280) == 1
    got = ops.linear(a, w, bias, res=r)
    plan = ops.last_gemm_plan()
    assert plan["cfg"] == 2 and plan["splitk"] == 1, plan
    close(got, a.float() @ w.float().T + bias + r.float(), what="M 8192 N 1280 K 1280")


# ------------------------------------------------------------------------------- attention at B = 32
def _sdpa_ref(q, k, v, heads, chunk=8):
    B, nq, _ = q.shape
    out = torch.empty(B, nq, heads * 64, device=q.device, dtype=torch.float32)
    for b0 in range(0, B, chunk):
        sl = slice(b0, b0 + chunk)
        qq = q[sl].float().view(-1, nq, heads, 64).transpose(1, 2)
        kk = k[sl].float().reshape(qq.shape[0], -1, heads, 64).transpose(1, 2)
        vv = v[sl].float().reshape(qq.shape[0], -1, heads, 64).transpose(1, 2)
        s = torch.softmax(qq @ kk.transpose(-1, -2) * 0.125, dim=-1)
        out[sl] = (s @ vv).transpose(1, 2).reshape(-1, nq, heads * 64)
    return out


@pytest.mark.parametrize("heads,nq,nk", [(5, 4096, 4096), (5, 4096, 77), (10, 1024, 1024), (10, 1024, 77), (20, 256, 256)])
def test_attention_cfg4_shapes(ops, heads, nq, nk):
    """Self / adapter (nk = nq) and text (nk = 77, ragged second key tile) attention of a B=32 forward, consumed in
    place from a fused q|k|v buffer (row stride 3C) like the engine does; 4-wave kernel, grid >> 256 CUs."""
    C = heads * 64
    if nk == nq:
        qkv = grnd(B32, nq, 3 * C, seed=61)
        q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    else:
        q = grnd(B32, nq, C, seed=62)
        kv = grnd(B32, nk, 2 * C, seed=63)
        k, v = kv[:, :, :C], kv[:, :, C:]
    got = ops.attention(q, k, v, heads)
    plan = ops.last_attention_plan()
    assert plan["waves"] == 4 and plan["workgroups"] == B32 * heads * ((nq + 127) // 128), plan
    close(got, _sdpa_ref(q, k, v, heads), what=f"attention {heads}x{nq}x{nk}")


def test_attention_prescaled_cfg4_shape(ops):
    """The engine's form at the dominant site: q pre-multiplied by 64^-0.5 * log2(e) (packing.QSCALE), exp2-domain scores."""
    from mvd_amd.packing import QSCALE
    heads, n = 5, 4096
    C = heads * 64
    q, k, v = grnd(B32, n, C, seed=71), grnd(B32, n, C, seed=72), grnd(B32, n, C, seed=73)
    qs = (q.float() * QSCALE).to(torch.bfloat16)
    got = ops.attention(qs, k, v, heads, scale=0.0)
    want = _sdpa_ref((qs.float() / QSCALE), k, v, heads)
    close(got, want, what="prescaled attention 5x4096x4096")


@pytest.mark.parametrize("heads,nq,nk,strided", [(5, 4096, 77, False), (5, 1000, 333, False), (10, 1024, 1024, True), (5, 2050, 65, True)])
def test_attention_engine_form_ragged_and_strided(ops, heads, nq, nk, strided):
    """The engine's attention kernel (prescaled q, LDS-DMA K/V staging, dot2c denominators, four waves per SIMD, XCD-aware block
    order) on ragged key / query counts -- keys >= nk reach the kernel as buffer-range zeros and must be masked -- and on K/V
    that are column slices of a wider fused buffer (row stride 4C, as the q|k|v|q_ref GEMM output is consumed in place)."""
    from mvd_amd.packing import QSCALE
    C = heads * 64
    B = 32 if nq * nk <= 1024 * 1024 else 16
    q = grnd(B, nq, C, seed=81)
    if strided:
        fused = grnd(B, nk, 4 * C, seed=82)
        k, v = fused[:, :, C:2 * C], fused[:, :, 2 * C:3 * C]
    else:
        k, v = grnd(B, nk, C, seed=82), grnd(B, nk, C, seed=83)
    qs = (q.float() * QSCALE).to(torch.bfloat16)
    got = ops.attention(qs, k, v, heads, scale=0.0)
    assert ops.last_attention_plan()["waves"] == 4, ops.last_attention_plan()
    want = _sdpa_ref(qs.float() / QSCALE, k.contiguous(), v.contiguous(), heads)
    close(got, want, what=f"engine-form attention {heads}x{nq}x{nk} strided={strided}")


# ------------------------------------------------------------------------------- norms at B = 32
@pytest.mark.parametrize("hw,c0,c1,silu", [(4096, 320, 0, True), (4096, 640, 320, True), (4096, 320, 320, True), (1024, 640, 0, False),
                                           (1024, 640, 320, True), (1024, 1280, 640, True), (256, 1280, 1280, True), (64, 1280, 0, True)])
def test_groupnorm_cfg4_shapes(ops, hw, c0, c1, silu):
    x0 = grnd(B32, hw, c0, scale=2.0, seed=81) + 0.5
    x1 = grnd(B32, hw, c1, scale=1.5, seed=82) - 0.25 if c1 else None
    g, b = grnd(c0 + c1, seed=83, dtype=torch.float32), grnd(c0 + c1, seed=84, dtype=torch.float32)
    x = torch.cat([x0, x1], -1) if c1 else x0
    want = F.group_norm(x.float().transpose(1, 2), 32, g, b, 1e-5).transpose(1, 2)
    if silu:
        want = F.silu(want)
    got = ops.groupnorm(x0.contiguous(), g, b, 32, 1e-5, silu, x2=x1.contiguous() if c1 else None)
    close(got, want, tol=2 ** -6, what=f"groupnorm hw={hw} c={c0}+{c1}")


@pytest.mark.parametrize("batch", [4, 32])      # 4: the two-kernel form (shifted sums, Chan merge); 32: the one-pass slice kernel (two-pass in registers)
def test_groupnorm_large_offset(ops, batch):
    """Trained SD weights give channels with |mean| >> std.  mean 50 / std 1: a one-pass E[x^2] - mean^2 variance loses
    ~3.4 decimal digits to cancellation in fp32; the kernel must stay within bf16 rounding of the fp64 reference."""
    hw, c = 4096, 320
    x = (grnd(batch, hw, c, seed=91, dtype=torch.float32) + 50.0).to(torch.bfloat16)
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    want = F.group_norm(x.double().transpose(1, 2), 32, g.double(), b.double(), 1e-5).transpose(1, 2)
    got = ops.groupnorm(x.contiguous(), g, b, 32, 1e-5, False)
    err = (got.double() - want).abs().max().item()
    assert err <= 2 ** -7 * want.abs().max().item() + 1e-3, err


@pytest.mark.parametrize("rows,c", [(131072, 320), (32768, 640), (8192, 1280)])
def test_layernorm_cfg4_shapes(ops, rows, c):
    x = grnd(rows, c, scale=1.7, seed=95) + 0.3
    g, b = grnd(c, seed=96, dtype=torch.float32), grnd(c, seed=97, dtype=torch.float32)
    close(ops.layernorm(x, g, b), F.layer_norm(x.float(), (c,), g, b, 1e-5), tol=2 ** -6, what=f"layernorm {rows}x{c}")


# ------------------------------------------------------------------------------- LayerNorm folded into the GEMM (64^2 / 32^2 levels)
@pytest.mark.parametrize("m,c,nmul,geglu,offset", [
    (131072, 320, 4, False, 0.3),     # ln1 -> [q; k; v; q_ref]
    (131072, 320, 1, False, 0.3),     # ln2 -> q (no adapter)
    (32768, 640, 4, False, 0.3),
    (32768, 640, 2, False, 8.0),      # rows with |mean| = 8 std: the E[x^2] - mean^2 form must hold up
    (32768, 640, 2, False, 100.0),    # |mean| ~ 77 std (massive-activation rows): fp32 sums of bf16 squares keep ~11 bits of the
    (131072, 320, 1, False, -60.0),   #   variance there, and acc - mean * c1 cancels exactly (c1 from the same bf16 weights)
    (131072, 320, 8, True, -2.0),     # ln3 -> ff1 + GEGLU (the C = 640 GEGLU keeps the separate LayerNorm: measured slower fused)
])
def test_layernorm_fold_cfg4_shapes(ops, m, c, nmul, geglu, offset):
    """LayerNorm(x).W^T + b as ONE kernel (gemm_pp_kernel<..., LNF>: statistics from the MFMA fragments, gamma folded into W,
    epilogue rstd*(acc - mean*c1) + c2) against fp32 layer_norm + matmul on the same bf16 x and the UNfolded fp32 weights."""
    from mvd_amd.packing import _geglu_rows, fold_layernorm
    n = nmul * c
    x = grnd(m, c, scale=1.3, seed=101) + offset
    x[:7] *= 0.01                                      # a few low-variance rows (rstd ~ 1e2)
    w = grnd(n, c, scale=1 / math.sqrt(c), seed=102).float()
    gamma = 1.0 + 0.3 * grnd(c, seed=103, dtype=torch.float32)
    beta = 0.2 * grnd(c, seed=104, dtype=torch.float32)
    bias = grnd(n, seed=105, dtype=torch.float32) if geglu else None
    h = F.layer_norm(x.float(), (c,), gamma, beta, 1e-5) @ w.T
    if geglu:
        h = h + bias
        want = h[:, : n // 2] * F.gelu(h[:, n // 2:])
        wf, cf = fold_layernorm(_geglu_rows(w), gamma, beta, _geglu_rows(bias), "cuda")
    else:
        want = h
        wf, cf = fold_layernorm(w, gamma, beta, None, "cuda")
    del h
    got = ops.ln_linear(x, wf, cf, geglu=geglu)
    _assert_persistent(ops, 6 if geglu else 7, f"ln-fold {m}x{n}x{c}", n=n, m=m)
    close(got, want, tol=2 ** -6, what=f"ln-fold M={m} C={c} N={n} geglu={geglu}")


def test_layernorm_fold_rejects_shapes_no_kernel_takes(ops):
    """Shapes neither fused kernel takes (256x320 form: N not a multiple of 320; small-M form: M > 4608; C = 1280 at a size only
    the 256x320 form could run) are refused, not mis-run."""
    from mvd_amd._lib import MvdError
    from mvd_amd.packing import fold_layernorm
    for m, c, n in [(131072, 320, 384), (8192, 1280, 3840)]:
        wf, cf = fold_layernorm(grnd(n, c).float(), torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), None, "cuda")
        with pytest.raises(MvdError):
            ops.ln_linear(grnd(m, c), wf, cf)


# ------------------------------------------------------------------------------- the whole forward at B = 32
def test_sd21_full_size_parity_b32():
    """configs[3] end to end: 32 pairs, full SD-2.1 shapes, camera FiLM + cross-view adapter, cold forward, vs the CPU
    oracle on identical weights and inputs (MVD_E2E_BATCH overrides the batch; the oracle needs a few minutes)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.parity_util import run_tiny_parity
    batch = int(os.environ.get("MVD_E2E_BATCH", "32"))
    stats = run_tiny_parity(batch=batch, verbose=True, cfg_name="sd21", hw=64, text_len=77)
    assert stats["finite"]
    assert stats["rel_l2"] <= 2e-2, stats
    assert stats["max_rel"] <= 5e-2, stats
