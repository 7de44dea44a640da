"""Operator-level parity: every HIP kernel vs a plain PyTorch fp32 reference of the same op,
on bf16-rounded inputs (so the only differences are accumulation order and the final bf16
rounding).  Tolerance: |err| <= 2^-7 * max|ref| (bf16 has 8 significant bits) unless stated."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from mvd_amd import ops as O
    return O


def rnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def close(got, want, tol=2 ** -7, what=""):
    got = got.float().cpu()
    want = want.float()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    rel_l2 = ((got - want).norm() / want.norm().clamp_min(1e-12)).item()
    assert err <= tol * ref + 1e-6, f"{what}: max-abs {err:.4g} vs ref max {ref:.4g} (rel-L2 {rel_l2:.3g})"
    return rel_l2


def _pack(w, tap_major=False):
    from mvd_amd.packing import _conv_w
    return _conv_w(w, tap_major).to(torch.bfloat16)


# ------------------------------------------------------------------------------- GEMM
# tile configs 2..5 (128x160, 128x128, 128x64, 64x64) use register staging, 10..13 the same tiles with LDS-DMA (global_load_lds)
# staging; 6 / 7 = the 256x320 tile, ping-pong kernels (gemm_pp.hip; 6 = the GEGLU wave grid).
# (Configs the heuristic never picks -- 256x160, 256x128, 128x320, the forced lock-step forms 16 / 17 of the 256x320 tile -- and
#  the ring-pipelined experiment gemm_ring.hip are in probe builds only, not part of the product library.)
@pytest.mark.parametrize("cfg", [-1, 2, 3, 4, 5, 7, 10, 11, 12, 13])
@pytest.mark.parametrize("m,n,k", [(300, 640, 320), (1024, 1280, 192), (77, 640, 1024), (5, 1920, 64), (2100, 320, 128)])
def test_linear_configs(ops, cfg, m, n, k):
    tiles = {2: 160, 3: 128, 4: 64, 5: 64, 7: 320, 6: 320}
    if cfg >= 0 and n % tiles[cfg % 8]:
        pytest.skip("N not divisible by this tile")
    a, w = rnd(m, k, seed=1), rnd(n, k, scale=1 / math.sqrt(k), seed=2)
    bias = rnd(n, seed=3, dtype=torch.float32)
    want = a.float() @ w.float().T + bias
    got = ops.linear(a.cuda(), w.cuda(), bias.cuda(), force_cfg=cfg)
    close(got, want, what=f"linear cfg{cfg}")


@pytest.mark.parametrize("splitk", [2, 3, 5])
@pytest.mark.parametrize("cfg", [-1, 7, 10, 13])
def test_linear_and_conv_splitk(ops, splitk, cfg):
    """Split-K: K slabs divided over several work items per tile, fp32 partials, fused reduce + epilogue."""
    m, n, k, rpb = 100, 320, 640, 50
    a, w = rnd(m, k, seed=1), rnd(n, k, scale=1 / math.sqrt(k), seed=2)
    bias, rowvec, res = rnd(n, seed=3, dtype=torch.float32), rnd(m // rpb, n, seed=4, dtype=torch.float32), rnd(m, n, seed=5)
    want = (a.float() @ w.float().T + bias + rowvec.repeat_interleave(rpb, 0)) + res.float()
    got = ops.linear(a.cuda(), w.cuda(), bias.cuda(), rowvec=rowvec.cuda(), rows_per_batch=rpb, res=res.cuda(),
                     force_cfg=cfg, splitk=splitk)
    close(got, want, what=f"linear splitk={splitk}")
    B, H, W, cin, cout = 2, 6, 6, 128, 320
    x = rnd(B, cin, H, W, seed=6)
    wc = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin), seed=7)
    want = F.conv2d(x.float(), wc.float(), bias, padding=1).permute(0, 2, 3, 1)
    got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), _pack(wc).cuda(), bias.cuda(), force_cfg=cfg, splitk=splitk)
    close(got, want, what=f"conv splitk={splitk}")


@pytest.mark.parametrize("cfg,m", [(-1, 512), (6, 8192)])
def test_gelu_of_the_geglu_epilogue(ops, cfg, m):
    """The GEGLU epilogue's erf GELU (common.h: Abramowitz-Stegun 7.1.26, |error| 1.5e-7) swept through a GEMM whose value half
    is the constant 1 (bias) and whose gate half passes one input column through -- out = gelu(x) -- against fp64
    0.5 x (1 + erf(x / sqrt 2)) over |x| <= 12, tails and the region around zero included: within half a bf16 ulp of the output
    (+ 1e-6 absolute)."""
    from mvd_amd.packing import _geglu_rows
    k, n_out = 64, 160
    x = torch.linspace(-12.0, 12.0, m * n_out).view(m, n_out)
    x = x.to(torch.bfloat16)
    a = torch.zeros(m, k, dtype=torch.bfloat16)
    out_all = torch.empty(m, n_out)
    w = torch.zeros(2 * n_out, k)
    w[n_out:, 0] = 1.0                                    # gate_j = a[:, 0] for every output column j
    bias = torch.cat([torch.ones(n_out), torch.zeros(n_out)])
    wp, bp = _geglu_rows(w).to(torch.bfloat16), _geglu_rows(bias)
    for j in range(0, n_out, 40):                         # each pass sweeps another slice of the range through column 0
        a[:, 0] = x[:, j]
        got = ops.linear(a.cuda(), wp.cuda(), bp.cuda(), geglu=True, force_cfg=cfg).float().cpu()
        xj = x[:, j].double()
        want = 0.5 * xj * (1.0 + torch.erf(xj / math.sqrt(2.0)))
        err = (got[:, 0].double() - want).abs()
        bound = want.abs() * 2.0 ** -8 + 1e-6
        assert bool((err <= bound).all()), f"gelu: worst excess {(err - bound).max().item():.3g} at x = {xj[(err - bound).argmax()].item():.4g}"
        assert torch.equal(got, got[:, :1].expand_as(got)), "all output columns see the same gate"


def test_probe_only_configs_are_rejected_and_lockstep_fallback_works(ops):
    """256x160 / 256x128 / 128x320 tiles and the forced lock-step 256x320 forms are not in the product library; the lock-step
    256x320 kernel itself remains as the fallback of the ping-pong kernels (here: an fp32 output, which those do not write)."""
    from mvd_amd import _lib as L
    a, w = rnd(300, 320, seed=1), rnd(640, 320, scale=1 / math.sqrt(320), seed=2)
    for cfg in (0, 1, 8, 9, 14, 16, 17):
        with pytest.raises(L.MvdError, match="probe builds only"):
            ops.linear(a.cuda(), w.cuda(), force_cfg=cfg)
    got = ops.linear(a.cuda(), w.cuda(), out_f32=True, force_cfg=7)
    close(got, a.float() @ w.float().T, tol=1e-4, what="lock-step 256x320 fallback, fp32 out")


def test_linear_epilogues(ops):
    m, n, k1, k2, rpb = 384, 320, 128, 192, 96
    a, a2 = rnd(m, k1, seed=1), rnd(m, k2, seed=2)
    w = rnd(n, k1 + k2, scale=1 / math.sqrt(k1 + k2), seed=3)
    bias = rnd(n, seed=4, dtype=torch.float32)
    rowvec = rnd(m // rpb, n, seed=5, dtype=torch.float32)
    res = rnd(m, n, seed=6)
    want = 0.3 * (torch.cat([a, a2], 1).float() @ w.float().T + bias + rowvec.repeat_interleave(rpb, 0)) + res.float()
    got = ops.linear(a.cuda(), w.cuda(), bias.cuda(), a2=a2.cuda(), rowvec=rowvec.cuda(), rows_per_batch=rpb,
                     res=res.cuda(), alpha=0.3)
    close(got, want, what="linear dual-source+rowvec+res+alpha")
    got32 = ops.linear(a.cuda(), w[:, :k1].contiguous().cuda(), bias.cuda(), out_f32=True)
    close(got32, a.float() @ w[:, :k1].float().T + bias, tol=1e-4, what="linear fp32 out")


@pytest.mark.parametrize("cfg", [-1, 3, 4, 5, 6, 11, 12, 13])
def test_linear_geglu(ops, cfg):
    from mvd_amd.packing import _geglu_rows
    m, c = 300, 320
    a = rnd(m, c, seed=1)
    w = rnd(8 * c, c, scale=1 / math.sqrt(c), seed=2)
    bias = rnd(8 * c, seed=3, dtype=torch.float32)
    f = a.float() @ w.float().T + bias
    val, gate = f.chunk(2, -1)
    want = val * F.gelu(gate)
    got = ops.linear(a.cuda(), _geglu_rows(w).contiguous().cuda(), _geglu_rows(bias).contiguous().cuda(), geglu=True,
                     force_cfg=cfg)
    close(got, want, what=f"geglu cfg{cfg}")


# ------------------------------------------------------------------------------- conv

@pytest.mark.parametrize("cfg", [-1, 5, 7, 13])
@pytest.mark.parametrize("stride,ups", [(1, False), (2, False), (1, True)])
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 16, 64, 128), (1, 8, 12, 192, 64), (3, 6, 6, 128, 320)])
def test_conv3x3(ops, stride, ups, B, H, W, cin, cout, cfg):
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin), seed=2)
    bias = rnd(cout, seed=3, dtype=torch.float32)
    xin = x.float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    want = F.conv2d(xin, w.float(), bias, stride=stride, padding=1).permute(0, 2, 3, 1)
    if cfg == 7 and cout % 320:
        pytest.skip("N not divisible by this tile")
    got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), _pack(w).cuda(), bias.cuda(), stride=stride, upsample=ups,
                      force_cfg=cfg)
    close(got, want, what=f"conv3x3 s{stride} ups{ups}")


@pytest.mark.parametrize("cfg", [-1, 2, 4, 5, 7, 10, 12, 13])
def test_conv3x3_resnet_fusions(ops, cfg):
    """conv1 (+time-embedding row vector) and conv2 (+1x1 shortcut over a 2-source concat / + residual)."""
    B, H, W, c0, c1, cout = 2, 8, 8, 128, 64, 320
    x0, x1 = rnd(B, c0, H, W, seed=1), rnd(B, c1, H, W, seed=2)
    h = rnd(B, cout, H, W, seed=3)
    w2 = rnd(cout, cout, 3, 3, scale=1 / math.sqrt(9 * cout), seed=4)
    wsc = rnd(cout, c0 + c1, 1, 1, scale=1 / math.sqrt(c0 + c1), seed=5)
    bias = rnd(cout, seed=6, dtype=torch.float32)
    temb = rnd(B, cout, seed=7, dtype=torch.float32)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()  # noqa: E731
    want = F.conv2d(h.float(), w2.float(), bias, padding=1) + F.conv2d(torch.cat([x0, x1], 1).float(), wsc.float())
    wp = torch.cat([_pack(w2), wsc.reshape(cout, c0 + c1)], 1).contiguous()
    got = ops.conv3x3(nhwc(h), wp.cuda(), bias.cuda(), shortcut=nhwc(x0), shortcut2=nhwc(x1), force_cfg=cfg)
    close(got, want.permute(0, 2, 3, 1), what="conv2+shortcut")
    want = F.conv2d(h.float(), w2.float(), bias, padding=1) + temb[:, :, None, None] + h.float()
    got = ops.conv3x3(nhwc(h), _pack(w2).cuda(), bias.cuda(), rowvec=temb.cuda(), res=nhwc(h), force_cfg=cfg)
    close(got, want.permute(0, 2, 3, 1), what="conv+temb+residual")


def test_conv_in_out(ops):
    B, H, W, c = 2, 16, 12, 64
    x = rnd(B, 4, H, W, seed=1)
    w = rnd(c, 4, 3, 3, scale=1 / 6, seed=2, dtype=torch.float32)
    b = rnd(c, seed=3, dtype=torch.float32)
    want = F.conv2d(x.float(), w, b, padding=1).permute(0, 2, 3, 1)
    got = ops.conv_in(x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), b.cuda())
    close(got, want, what="conv_in")
    y = rnd(B, c, H, W, seed=4)
    w2 = rnd(4, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=5)
    b2 = rnd(4, seed=6, dtype=torch.float32)
    want = F.conv2d(y.float(), w2.float(), b2, padding=1)
    got = ops.conv_out(y.permute(0, 2, 3, 1).contiguous().cuda(), _pack(w2, tap_major=True).cuda(), b2.cuda())
    close(got, want, tol=1e-4, what="conv_out")
    # a width that is not a multiple of the four pixels a wave takes, full channel count of the UNet's last level
    y = rnd(1, 320, 9, 10, seed=7)
    w3 = rnd(4, 320, 3, 3, scale=1 / math.sqrt(9 * 320), seed=8)
    want = F.conv2d(y.float(), w3.float(), b2, padding=1)
    got = ops.conv_out(y.permute(0, 2, 3, 1).contiguous().cuda(), _pack(w3, tap_major=True).cuda(), b2.cuda())
    close(got, want, tol=1e-4, what="conv_out ragged width")


# ------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,heads,nq,nk", [
    (2, 5, 256, 256),      # self-attention, multiple of every tile
    (1, 2, 64, 77),        # text cross-attention: ragged key tile
    (3, 1, 16, 16),        # fewer queries than one wave tile
    (2, 2, 144, 144),      # 768^2-style ragged query count
    (1, 20, 64, 32),       # Q4: reference tokens re-chunked (nk = N/2)
    (1, 5, 1024, 1024),    # multi-tile, 8-wave path
    (1, 1, 4, 4),
])
def test_attention(ops, B, heads, nq, nk):
    C = heads * 64
    q, k, v = rnd(B, nq, C, seed=1), rnd(B, nk, C, seed=2), rnd(B, nk, C, seed=3)
    qh = q.float().view(B, nq, heads, 64).transpose(1, 2)
    kh = k.float().view(B, nk, heads, 64).transpose(1, 2)
    vh = v.float().view(B, nk, heads, 64).transpose(1, 2)
    want = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, nq, C)
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads)
    close(got, want, tol=2 ** -6, what=f"attention {B}x{heads}x{nq}x{nk}")


@pytest.mark.parametrize("B,heads,nq,nk,shift", [
    (2, 5, 256, 256, 0.0), (1, 2, 64, 77, 0.0), (3, 1, 16, 16, 0.0), (2, 2, 144, 144, 0.0), (1, 20, 64, 32, 0.0),
    (1, 5, 1024, 1024, 0.0), (1, 1, 4, 4, 0.0),
    (1, 2, 160, 200, -40.0),   # every score far below zero: the first tile must still set the running max
    (1, 2, 160, 200, 25.0),    # large positive scores
])
def test_attention_prescaled(ops, B, heads, nq, nk, shift):
    """Engine form: q carries softmax_scale*log2(e) (folded into to_q by the packing); kernel works in the exp2 domain."""
    from mvd_amd.packing import QSCALE
    C = heads * 64
    q, k, v = rnd(B, nq, C, seed=1), rnd(B, nk, C, seed=2), rnd(B, nk, C, seed=3)
    if shift:   # add a constant direction so that q.k is shifted by about `shift` (pre-softmax, natural-log units)
        k = (k.float() + 1.0).to(torch.bfloat16)
        q = (q.float() + shift * 8.0 / 64.0).to(torch.bfloat16)
    qs = (q.float() * QSCALE).to(torch.bfloat16)          # what the scaled to_q GEMM would emit
    sp = lambda t, n: t.float().view(B, n, heads, 64).transpose(1, 2)  # noqa: E731
    # reference on the SAME rounded operand: softmax over ln2 * (qs . k)
    want = F.scaled_dot_product_attention(sp(qs, nq) * 0.6931471805599453, sp(k, nk), sp(v, nk), scale=1.0)
    got = ops.attention(qs.cuda(), k.cuda(), v.cuda(), heads, scale=0.0)
    close(got, want.transpose(1, 2).reshape(B, nq, C), tol=2 ** -6, what=f"prescaled attention {B}x{heads}x{nq}x{nk} shift {shift}")


def test_attention_strided_views_and_spike(ops):
    """Fused-QKV strides, plus a forced online-softmax rescale (one key spikes late in the sequence)."""
    B, heads, n = 2, 2, 320
    C = heads * 64
    qkv = rnd(B, n, 3 * C, seed=5)
    qkv[:, 300, C:2 * C] *= 6.0          # late key with a huge score -> running max jumps at the last tile
    g = qkv.cuda()
    q, k, v = g[:, :, :C], g[:, :, C:2 * C], g[:, :, 2 * C:]
    got = ops.attention(q, k, v, heads)
    f = qkv.float()
    sp = lambda t: t.reshape(B, n, heads, 64).transpose(1, 2)  # noqa: E731
    want = F.scaled_dot_product_attention(sp(f[:, :, :C]), sp(f[:, :, C:2 * C]), sp(f[:, :, 2 * C:]))
    close(got, want.transpose(1, 2).reshape(B, n, C), tol=2 ** -6, what="attention strided+spike")


# ------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,hw,c0,c1", [(2, 256, 320, 0), (1, 64, 1280, 640), (3, 16, 64, 64), (2, 4096, 64, 0), (1, 4, 1280, 1280),
                                        # one-pass slice kernel: XCD-swizzled block order (batch % 8 == 0), a slice straddling the
                                        # two sources, 16 / 21 vectors per thread, a ragged pixel count
                                        (8, 1024, 640, 320), (16, 64, 1280, 0), (8, 4096, 320, 0), (3, 1000, 320, 320)])
@pytest.mark.parametrize("silu", [False, True])
def test_groupnorm(ops, B, hw, c0, c1, silu):
    x = rnd(B, hw, c0, seed=1, scale=2.0) + 0.5
    x = x.to(torch.bfloat16)
    x2 = rnd(B, hw, c1, seed=2) if c1 else None
    C = c0 + c1
    g = 1 + 0.1 * rnd(C, seed=3, dtype=torch.float32)
    b = 0.1 * rnd(C, seed=4, dtype=torch.float32)
    full = torch.cat([x, x2], 2) if c1 else x
    want = F.group_norm(full.float().permute(0, 2, 1), 32, g, b, 1e-5).permute(0, 2, 1)
    if silu:
        want = F.silu(want)
    got = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), silu=silu, x2=x2.cuda() if c1 else None)
    close(got, want, what="groupnorm")


@pytest.mark.parametrize("B,hw,C", [(1, 1296, 640), (1, 1296, 1920), (1, 784, 1920), (2, 676, 2560), (1, 5184, 320), (1, 9216, 960)])
def test_groupnorm_two_kernel_ragged_chunks_large_mean(ops, B, hw, C):
    """Two-kernel form (gn_stats / gn_apply) at pixel counts the chunk size does not divide (72x72, 56x56, 52x52 latents at batch
    1-2), with |mean| >> std per group: the per-chunk (mean, M2) merge must not see chunks of negative size (round-2 advisor)."""
    x = (rnd(B, hw, C, seed=11, dtype=torch.float32) + 60.0).to(torch.bfloat16)
    g = 1 + 0.1 * rnd(C, seed=3, dtype=torch.float32)
    b = 0.1 * rnd(C, seed=4, dtype=torch.float32)
    want = F.group_norm(x.double().permute(0, 2, 1), 32, g.double(), b.double(), 1e-5).permute(0, 2, 1)
    got = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), silu=False)
    close(got, want.float(), what=f"groupnorm {B}x{hw}x{C} mean 60")


@pytest.mark.parametrize("rows,c", [(100, 320), (7, 640), (33, 1280), (5, 64)])
def test_layernorm(ops, rows, c):
    x = rnd(rows, c, seed=1, scale=3.0)
    g = 1 + 0.1 * rnd(c, seed=2, dtype=torch.float32)
    b = 0.1 * rnd(c, seed=3, dtype=torch.float32)
    want = F.layer_norm(x.float(), (c,), g, b, 1e-5)
    close(ops.layernorm(x.cuda(), g.cuda(), b.cuda()), want, what="layernorm")


@pytest.mark.parametrize("B,hw,c", [(1, 64, 320), (4, 16, 128), (32, 4, 64)])
def test_refnorm(ops, B, hw, c):
    """Q2 of SURVEY.md: per-pixel statistics over (batch, channel), unbiased std, clamp, x0.5."""
    x = (rnd(B, hw, c, seed=1, scale=1.7) + 0.3).to(torch.bfloat16)
    nchw = x.float().permute(0, 2, 1).reshape(B, c, hw, 1)
    r = nchw - nchw.mean(dim=(0, 1), keepdim=True)
    want = (r / torch.clamp(r.std(dim=(0, 1), keepdim=True), min=1e-6) * 0.5).reshape(B, c, hw).permute(0, 2, 1)
    close(ops.refnorm(x.cuda()), want, what="refnorm")


def test_film_and_layout(ops):
    B, hw, c = 3, 32, 128
    x = rnd(B, hw, c, seed=1)
    s, t = rnd(B, c, seed=2, dtype=torch.float32), rnd(B, c, seed=3, dtype=torch.float32)
    close(ops.film(x.cuda(), s.cuda(), t.cuda()), x.float() * s[:, None] + t[:, None], what="film")
    xi = rnd(2, 4, 8, 8, seed=4, dtype=torch.float32)
    s4, t4 = rnd(2, 4, seed=5, dtype=torch.float32), rnd(2, 4, seed=6, dtype=torch.float32)
    want = (xi * s4[:, :, None, None] + t4[:, :, None, None]).permute(0, 2, 3, 1)
    close(ops.nchw_to_nhwc(xi.cuda(), s4.cuda(), t4.cuda()), want, what="nchw_to_nhwc+film")


def test_bad_shapes_are_rejected(ops):
    """The host must refuse shapes the kernels do not support (never launch a faulting kernel)."""
    from mvd_amd._lib import MvdError
    a, w = rnd(8, 96).cuda(), rnd(64, 96).cuda()         # K not a multiple of 64
    with pytest.raises(MvdError):
        ops.linear(a, w)
    a, w = rnd(8, 64).cuda(), rnd(96, 64).cuda()         # N not a multiple of 64
    with pytest.raises(MvdError):
        ops.linear(a, w)


# ------------------------------------------------------------------------------- small-M kernels (gemm_sm.hip, the batch-1 path)
# force_cfg = 100 + 10 * tile + ring depth; tiles: 0 64x64, 1 128x64, 2 64x128, 3 128x128, 4 64x160, 5 128x160, 6 64x320
SM_BN = {0: 64, 1: 64, 2: 128, 3: 128, 4: 160, 5: 160, 6: 320}


@pytest.mark.parametrize("nstage", [2, 3, 4, 8])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("m,n,k", [(300, 640, 320), (1024, 1280, 192), (77, 640, 1024), (5, 1920, 64), (2100, 320, 128)])
def test_sm_linear(ops, tile, nstage, m, n, k):
    if n % SM_BN[tile]:
        pytest.skip("N not divisible by this tile")
    a, w = rnd(m, k, seed=1), rnd(n, k, scale=1 / math.sqrt(k), seed=2)
    bias = rnd(n, seed=3, dtype=torch.float32)
    want = a.float() @ w.float().T + bias
    got = ops.linear(a.cuda(), w.cuda(), bias.cuda(), force_cfg=100 + 10 * tile + nstage)
    close(got, want, what=f"sm linear tile{tile} ns{nstage}")


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6])
def test_sm_linear_epilogues(ops, tile):
    """two-source A, bias + per-batch row vector (tile inside one batch element and straddling two) + alpha + residual; fp32 out"""
    for m, rpb in ((384, 128), (384, 96)):
        n, k1, k2 = 640, 128, 192
        a, a2 = rnd(m, k1, seed=1), rnd(m, k2, seed=2)
        w = rnd(n, k1 + k2, scale=1 / math.sqrt(k1 + k2), seed=3)
        bias = rnd(n, seed=4, dtype=torch.float32)
        rowvec = rnd(m // rpb, n, seed=5, dtype=torch.float32)
        res = rnd(m, n, seed=6)
        want = 0.3 * (torch.cat([a, a2], 1).float() @ w.float().T + bias + rowvec.repeat_interleave(rpb, 0)) + res.float()
        got = ops.linear(a.cuda(), w.cuda(), bias.cuda(), a2=a2.cuda(), rowvec=rowvec.cuda(), rows_per_batch=rpb,
                         res=res.cuda(), alpha=0.3, force_cfg=100 + 10 * tile + 3)
        close(got, want, what=f"sm tile{tile} dual-source+rowvec(rpb {rpb})+res+alpha")
    got32 = ops.linear(a.cuda(), w[:, :k1].contiguous().cuda(), bias.cuda(), out_f32=True, force_cfg=100 + 10 * tile + 4)
    close(got32, a.float() @ w[:, :k1].float().T + bias, tol=1e-4, what="sm linear fp32 out")


def test_sm_splitk_grid_beyond_the_chip_takes_the_nowait_combine(ops):
    """a forced split whose grid cannot be resident at once (here 16 x 20 tiles x 12 slices = 3840 workgroups of 64 KB
    LDS) must not let its resident slices poll ~1 ms each for slices that cannot start: the launcher's occupancy hint switches
    such a launch to the no-wait combine.  Result correct and bit-stable; a grid that fits keeps the rendezvous."""
    m, n, k = 1024, 1280, 1536
    a, w = rnd(m, k, seed=1), rnd(n, k, scale=1 / math.sqrt(k), seed=2)
    bias = rnd(n, seed=3, dtype=torch.float32)
    want = a.float() @ w.float().T + bias
    args = (a.cuda(), w.cuda(), bias.cuda())
    got = ops.linear(*args, force_cfg=100 + 10 * 0 + 4, splitk=12)
    plan = ops.last_gemm_plan()
    assert plan["grid"] == 16 * 20 * 12 and plan["nowait"] == 1, plan
    close(got, want, what="sm linear split 12 beyond the chip")
    assert torch.equal(got, ops.linear(*args, force_cfg=100 + 10 * 0 + 4, splitk=12))
    ops.linear(a[:128].cuda(), w[:256].contiguous().cuda(), bias[:256].contiguous().cuda(), force_cfg=100 + 10 * 0 + 4, splitk=4)
    plan = ops.last_gemm_plan()
    assert plan["grid"] == 2 * 4 * 4 and plan["nowait"] == 0, plan


@pytest.mark.parametrize("splitk", [2, 3, 5, 10])
@pytest.mark.parametrize("tile", [0, 1, 3, 5, 6])
def test_sm_splitk_in_kernel_combine(ops, tile, splitk):
    """Split-K combined inside the kernel (last-arriving slice sums the write-through partials in slice order): same epilogue
    as the unsplit form, and two launches give the same bits (the combine order does not depend on who arrives last)."""
    m, n, k, rpb = 200, 640, 640, 100
    a, w = rnd(m, k, seed=1), rnd(n, k, scale=1 / math.sqrt(k), seed=2)
    bias, rowvec, res = rnd(n, seed=3, dtype=torch.float32), rnd(m // rpb, n, seed=4, dtype=torch.float32), rnd(m, n, seed=5)
    want = (a.float() @ w.float().T + bias + rowvec.repeat_interleave(rpb, 0)) + res.float()
    args = (a.cuda(), w.cuda(), bias.cuda())
    kw = dict(rowvec=rowvec.cuda(), rows_per_batch=rpb, res=res.cuda(), force_cfg=100 + 10 * tile + 3, splitk=splitk)
    got = ops.linear(*args, **kw)
    close(got, want, what=f"sm linear tile{tile} splitk={splitk}")
    for _ in range(3):
        assert torch.equal(got, ops.linear(*args, **kw)), "split-K combine is not bit-deterministic"
    B, H, W, cin, cout = 2, 6, 6, 128, 640
    x = rnd(B, cin, H, W, seed=6)
    wc = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin), seed=7)
    want = F.conv2d(x.float(), wc.float(), bias[:cout], padding=1).permute(0, 2, 3, 1)
    got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), _pack(wc).cuda(), bias[:cout].contiguous().cuda(),
                      force_cfg=100 + 10 * tile + 4, splitk=splitk)
    close(got, want, what=f"sm conv tile{tile} splitk={splitk}")


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 6])
def test_sm_geglu(ops, tile):
    from mvd_amd.packing import _geglu_rows
    m, c = 300, 320
    a = rnd(m, c, seed=1)
    w = rnd(8 * c, c, scale=1 / math.sqrt(c), seed=2)
    bias = rnd(8 * c, seed=3, dtype=torch.float32)
    val, gate = (a.float() @ w.float().T + bias).chunk(2, -1)
    got = ops.linear(a.cuda(), _geglu_rows(w).contiguous().cuda(), _geglu_rows(bias).contiguous().cuda(), geglu=True,
                     force_cfg=100 + 10 * tile + 3)
    close(got, val * F.gelu(gate), what=f"sm geglu tile{tile}")


@pytest.mark.parametrize("tile", [0, 1, 3, 4, 5, 6])
@pytest.mark.parametrize("stride,ups,asym", [(1, False, False), (2, False, False), (1, True, False), (2, False, True)])
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 16, 64, 128), (1, 8, 12, 192, 64), (3, 6, 6, 128, 320)])
def test_sm_conv3x3(ops, stride, ups, asym, B, H, W, cin, cout, tile):
    if cout % SM_BN[tile]:
        pytest.skip("N not divisible by this tile")
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin), seed=2)
    bias = rnd(cout, seed=3, dtype=torch.float32)
    xin = x.float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if asym:
        want = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w.float(), bias, stride=2, padding=0).permute(0, 2, 3, 1)
    else:
        want = F.conv2d(xin, w.float(), bias, stride=stride, padding=1).permute(0, 2, 3, 1)
    got = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), _pack(w).cuda(), bias.cuda(), stride=stride, upsample=ups,
                      asym_pad=asym, force_cfg=100 + 10 * tile + 3)
    close(got, want, what=f"sm conv3x3 s{stride} ups{ups} asym{asym} tile{tile}")


@pytest.mark.parametrize("tile", [0, 1, 3, 4, 5, 6])
def test_sm_conv3x3_resnet_fusions(ops, tile):
    """conv1 (+time-embedding row vector) and conv2 (+1x1 shortcut over a 2-source concat / + residual), also split over K."""
    B, H, W, c0, c1, cout = 2, 8, 8, 128, 64, 320
    if cout % SM_BN[tile]:
        pytest.skip("N not divisible by this tile")
    x0, x1 = rnd(B, c0, H, W, seed=1), rnd(B, c1, H, W, seed=2)
    h = rnd(B, cout, H, W, seed=3)
    w2 = rnd(cout, cout, 3, 3, scale=1 / math.sqrt(9 * cout), seed=4)
    wsc = rnd(cout, c0 + c1, 1, 1, scale=1 / math.sqrt(c0 + c1), seed=5)
    bias = rnd(cout, seed=6, dtype=torch.float32)
    temb = rnd(B, cout, seed=7, dtype=torch.float32)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()  # noqa: E731
    wp = torch.cat([_pack(w2), wsc.reshape(cout, c0 + c1)], 1).contiguous()
    for sk in (1, 4):
        want = F.conv2d(h.float(), w2.float(), bias, padding=1) + F.conv2d(torch.cat([x0, x1], 1).float(), wsc.float())
        got = ops.conv3x3(nhwc(h), wp.cuda(), bias.cuda(), shortcut=nhwc(x0), shortcut2=nhwc(x1), force_cfg=100 + 10 * tile + 4, splitk=sk)
        close(got, want.permute(0, 2, 3, 1), what=f"sm conv2+shortcut splitk {sk}")
        want = F.conv2d(h.float(), w2.float(), bias, padding=1) + temb[:, :, None, None] + h.float()
        got = ops.conv3x3(nhwc(h), _pack(w2).cuda(), bias.cuda(), rowvec=temb.cuda(), res=nhwc(h), force_cfg=100 + 10 * tile + 4, splitk=sk)
        close(got, want.permute(0, 2, 3, 1), what=f"sm conv+temb+residual splitk {sk}")


def test_rows_beyond_m_are_never_written(ops):
    """Every GEMM form on a ragged M: the rows behind the output (a guard region of the same allocation) stay untouched."""
    m, n, k = 300, 640, 320
    a, w = rnd(m, k, seed=1).cuda(), rnd(n, k, scale=1 / math.sqrt(k), seed=2).cuda()
    import ctypes as C
    from mvd_amd import _lib as L
    for cfg in (-1, 7, 10, 13, 103, 133, 153, 163):
        buf = torch.full((m + 600, n), 7.0, device="cuda", dtype=torch.bfloat16)
        L.call("mvd_op_linear", C.c_void_p(a.data_ptr()), None, k, 0, C.c_void_p(w.data_ptr()), None, None, 0, 0, None, 1.0, 0,
               C.c_void_p(buf.data_ptr()), 0, m, n, cfg, 1, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert bool((buf[m:] == 7.0).all()), f"cfg {cfg} wrote rows >= M"
        close(buf[:m], a.float().cpu() @ w.float().cpu().T, what=f"cfg {cfg}")


@pytest.mark.parametrize("m,c,nmul,geglu,offset", [(300, 320, 3, False, 0.3), (4096, 320, 1, False, 0.3), (1024, 640, 3, False, 4.0),
                                                   (256, 1280, 3, False, 0.3), (64, 1280, 1, False, -2.0), (4096, 320, 8, True, 0.3),
                                                   (1024, 640, 8, True, 0.3), (256, 1280, 8, True, 1.0), (64, 1280, 8, True, 0.3),
                                                   (1024, 640, 3, False, 100.0), (256, 1280, 3, False, -130.0),   # |mean| ~ 60-75 std
                                                   (4096, 320, 8, True, 90.0)])
def test_sm_ln_linear(ops, m, c, nmul, geglu, offset):
    """LayerNorm folded into the small-M GEMM (batch-1 shapes of every level): LayerNorm(x).W^T + b from the un-normalised rows."""
    from mvd_amd.packing import fold_layernorm, _geglu_rows
    n = nmul * c
    x = (rnd(m, c, seed=1, scale=1.7, dtype=torch.float32) + offset).to(torch.bfloat16)
    w = rnd(n, c, scale=1 / math.sqrt(c), seed=2, dtype=torch.float32)
    gamma = 1 + 0.1 * rnd(c, seed=3, dtype=torch.float32)
    beta = 0.1 * rnd(c, seed=4, dtype=torch.float32)
    bias = rnd(n, seed=5, dtype=torch.float32)
    ref = F.layer_norm(x.float(), (c,), gamma, beta, 1e-5) @ w.to(torch.bfloat16).float().T + bias
    if geglu:
        val, gate = ref.chunk(2, -1)
        ref = val * F.gelu(gate)
        w, bias = _geglu_rows(w), _geglu_rows(bias)
    wf, cf = fold_layernorm(w, gamma, beta, bias, "cuda")
    xc = x.cuda()
    got = ops.ln_linear(xc, wf, cf, geglu=geglu)
    close(got, ref, tol=2 ** -6, what=f"sm ln_linear {m}x{c}->{n} geglu={geglu}")
    for _ in range(5):
        assert torch.equal(got, ops.ln_linear(xc, wf, cf, geglu=geglu)), "fused LayerNorm GEMM is not bit-deterministic"


@pytest.mark.parametrize("nq,nk,jump_at,nsplit", [(256, 640, 64, 1), (200, 1111, 512, 1), (256, 1280, 700, 2), (4096, 4096, 3000, 1)])
def test_attention_running_max_fallback(ops, nq, nk, jump_at, nsplit):
    """The engine form takes its running max from the first key tile and checks nothing afterwards; scores that later rise by more
    than 2^128 over it make the denominators non-finite, which the workgroup detects behind its loop and answers by re-running
    its tiles with the checked (deferred-rescale) loop.  Here the scores of keys >= jump_at lie ~190 (exp2 domain) above the
    earlier ones for the first half of the queries (the other half never overflows: both paths in one launch)."""
    B, heads = 2, 2
    C = heads * 64
    g = torch.Generator().manual_seed(5)
    q = 0.05 * torch.randn(B, nq, C, generator=g)
    k = 0.05 * torch.randn(B, nk, C, generator=g)
    v = torch.randn(B, nk, C, generator=g)
    q[:, : nq // 2] += 1.0                    # q.k = 64 * 1.0 * (+-1.5) = +-96 for those queries
    k[:, :jump_at] -= 1.5
    k[:, jump_at:] += 1.5
    qs, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    sp = lambda t, n: t.double().view(B, n, heads, 64).transpose(1, 2)  # noqa: E731
    want = F.scaled_dot_product_attention(sp(qs, nq) * 0.6931471805599453, sp(k, nk), sp(v, nk), scale=1.0)
    want = want.transpose(1, 2).reshape(B, nq, C).float()
    if nsplit > 1:
        got = ops.attention_split(qs.cuda(), k.cuda(), v.cuda(), heads, nsplit)
    else:
        got = ops.attention(qs.cuda(), k.cuda(), v.cuda(), heads, scale=0.0)
    close(got, want, tol=2 ** -6, what=f"attention, scores jumping at key {jump_at} of {nk}")


def test_attention_running_max_fallback_when_only_o_would_overflow(ops):
    """the overflow can sit in the UNCHECKED accumulators, not in the denominators.  Scores of the later keys lie ~2^90
    (exp2 domain) above the first tile's maximum -- the denominators reach 2^90, finite and below the former acceptance bound of
    1e30 -- and those keys' values are ~3e9: l * |v| = 4e36 * nk overflows fp32.  The bound of 2^64 on l sends such a workgroup
    to the checked loop; the result is the plain softmax average (finite, ~3e9)."""
    B, heads, nq, nk, jump_at = 1, 2, 128, 512, 64
    C = heads * 64
    g = torch.Generator().manual_seed(9)
    q = 0.02 * torch.randn(B, nq, C, generator=g) + 1.0
    k = 0.02 * torch.randn(B, nk, C, generator=g)
    k[:, :jump_at] -= 0.703125                # q.k = 64 * (-0.703) = -45  |  +45 behind the jump: 90 apart
    k[:, jump_at:] += 0.703125
    v = torch.randn(B, nk, C, generator=g)
    v[:, jump_at:] *= 3.0e9
    qs, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    sp = lambda t, n: t.double().view(B, n, heads, 64).transpose(1, 2)  # noqa: E731
    want = F.scaled_dot_product_attention(sp(qs, nq) * 0.6931471805599453, sp(k, nk), sp(v, nk), scale=1.0)
    want = want.transpose(1, 2).reshape(B, nq, C).float()
    got = ops.attention(qs.cuda(), k.cuda(), v.cuda(), heads, scale=0.0)
    close(got, want, tol=2 ** -6, what="attention, overflow in O only")


@pytest.mark.parametrize("B,heads,nq,nk,nsplit,shift", [
    (1, 5, 4096, 4096, 4, 0.0),        # the 64x64-level self-attention of a batch-1 forward
    (1, 10, 1024, 1024, 4, 0.0), (1, 10, 1024, 1024, 2, 0.0),
    (1, 2, 200, 1000, 3, 0.0),         # ragged query block and a ragged last key tile, 16 tiles over 3 ranges
    (2, 3, 384, 2048, 8, 0.0),
    (1, 2, 256, 640, 2, -40.0),        # every score far below zero: each range's first tile sets its running max
    (1, 2, 256, 640, 2, 25.0),
])
def test_attention_split_kv(ops, B, heads, nq, nk, nsplit, shift):
    """Split-KV (batch 1): per-range partials merged by the last workgroup; same tolerance as the unsplit engine form, and the
    merge is bit-deterministic (fixed range order) whoever arrives last."""
    from mvd_amd.packing import QSCALE
    C = heads * 64
    q, k, v = rnd(B, nq, C, seed=1), rnd(B, nk, C, seed=2), rnd(B, nk, C, seed=3)
    if shift:
        k = (k.float() + 1.0).to(torch.bfloat16)
        q = (q.float() + shift * 8.0 / 64.0).to(torch.bfloat16)
    qs = (q.float() * QSCALE).to(torch.bfloat16)
    sp = lambda t, n: t.float().view(B, n, heads, 64).transpose(1, 2)  # noqa: E731
    want = F.scaled_dot_product_attention(sp(qs, nq) * 0.6931471805599453, sp(k, nk), sp(v, nk), scale=1.0)
    qc, kc, vc = qs.cuda(), k.cuda(), v.cuda()
    got = ops.attention_split(qc, kc, vc, heads, nsplit)
    close(got, want.transpose(1, 2).reshape(B, nq, C), tol=2 ** -6, what=f"split-KV attention {B}x{heads}x{nq}x{nk} / {nsplit}")
    for _ in range(4):
        assert torch.equal(got, ops.attention_split(qc, kc, vc, heads, nsplit)), "split-KV merge is not bit-deterministic"


@pytest.mark.parametrize("batch", [1, 8, 11, 32, 40])
@pytest.mark.parametrize("k,n,wbf16,silu", [(1020, 1024, False, False), (512, 1000, False, True), (2048, 1024, False, False),
                                            (320, 1280, True, False), (1280, 1280, True, True), (512, 14088, False, False)])
def test_skinny_linear_matrix_pipe_form(ops, batch, k, n, wbf16, silu):
    """Round 5: the camera / time MLP layers on the fp32 matrix pipe (misc.hip skinny_mfma_kernel): every layer shape of the front
    matter (Fourier projection K = 1020, the 2048-wide concat layer, the time MLPs with bf16 weights, the 14088-wide modulator
    layer), batches of 1 (the vector form: the matrix-pipe form starts at 8 rows) / 8 / 11 / 32 / 40 rows (ragged tiles, two tiles), N not a multiple of the 32-feature tile, SiLU on the
    inputs -- against torch fp64, and bit-identical run to run; the vector form (debug flag 8388608) agrees within fp32 rounding."""
    from mvd_amd import _lib as L
    g = torch.Generator().manual_seed(batch * 7 + k + n)
    x = torch.randn(batch, k, generator=g)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k))
    w = w.to(torch.bfloat16) if wbf16 else w
    bias = torch.randn(n, generator=g)
    xin = F.silu(x.double()) if silu else x.double()
    want = (xin @ w.double().T + bias.double()).float()
    got = ops.skinny_linear(x.cuda(), w.cuda(), bias.cuda(), silu_in=silu)
    err = (got.cpu() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err
    assert torch.equal(got, ops.skinny_linear(x.cuda(), w.cuda(), bias.cuda(), silu_in=silu))
    L.lib().mvd_debug_set_flags(8388608)
    try:
        old = ops.skinny_linear(x.cuda(), w.cuda(), bias.cuda(), silu_in=silu)
    finally:
        L.lib().mvd_debug_set_flags(0)
    assert (old - got).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
