"""Weight-streaming 3x3 convolution of small maps (mvd_amd/csrc/conv_ws.hip) vs a PyTorch fp32 conv2d of the same op on
bf16-rounded inputs.  Tolerance |err| <= 2^-7 * max|ref| as for the other GEMM / conv kernels (tests/test_ops_gpu.py): the
output is bf16, accumulation fp32 in a different order than the reference's.  Covers the map widths 8, 16, 32 (and 12, 24: the maps of a 96 x 96 latent) and the block heights of each, several images
per launch, the fused dense shortcut with one and two sources, the time-embedding row vector, the residual, the map borders
(a one-hot input makes every tap land on a known pixel) and agreement with the implicit-GEMM kernel the engine used before."""
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
    got, want = got.float().cpu(), want.float()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err, ref = (got - want).abs().max().item(), want.abs().max().item()
    assert err <= tol * ref + 1e-6, f"{what}: max-abs {err:.4g} vs ref max {ref:.4g}"


def reference(x, w4, bias, wsc=None, sc=None, rowvec=None, res=None):
    """x (B,H,W,C) bf16, w4 (N,C,3,3) bf16, wsc (N,Csc) bf16 over the rows sc (B,H,W,Csc)"""
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w4.float(), padding=1).permute(0, 2, 3, 1) + bias
    if wsc is not None:
        y = y + sc.float() @ wsc.float().T
    if rowvec is not None:
        y = y + rowvec[:, None, None, :]
    if res is not None:
        y = y + res.float()
    return y


def variants(h, w):
    """kernel forms that take an h x w map: 0 = the launcher's choice, 1 = 64-pixel blocks, 2 = 128-pixel blocks (16-wide maps),
    3 = 48- / 96-pixel blocks of the 12- / 24-wide maps of a 96 x 96 latent"""
    if w in (12, 24):
        return [0, 3]
    return [0, 1] + ([2] if w == 16 and (h * w) % 128 == 0 else [])


@pytest.mark.parametrize("b,h,w,c,n", [(1, 8, 8, 1280, 1280), (1, 16, 16, 640, 1280), (2, 8, 8, 256, 64), (1, 16, 16, 128, 48),
                                       (3, 8, 8, 128, 16), (1, 8, 16, 256, 32), (2, 16, 16, 128, 160), (1, 32, 32, 640, 640),
                                       (1, 4, 32, 128, 16), (1, 24, 16, 128, 32),
                                       # round 5: the 12 x 12 and 24 x 24 maps of the reference's 768 x 768 default (96 x 96 latents)
                                       (1, 12, 12, 1280, 1280), (1, 24, 24, 1280, 1280), (2, 12, 12, 256, 48), (1, 8, 24, 128, 32),
                                       (3, 4, 12, 128, 16), (1, 24, 24, 128, 64), (2, 16, 24, 128, 64)])
def test_conv_ws_plain_rowvec_residual(ops, b, h, w, c, n):
    from mvd_amd.packing import pack_ws
    x, w4 = rnd(b, h, w, c, seed=1), rnd(n, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=2)
    bias, rowvec, res = rnd(n, seed=3, dtype=torch.float32), rnd(b, n, seed=4, dtype=torch.float32), rnd(b, h, w, n, seed=5)
    wp = pack_ws(w4).cuda()
    assert wp.numel() == n * 9 * c
    for v in variants(h, w):
        close(ops.conv3x3_ws(x.cuda(), wp, bias.cuda(), n, variant=v), reference(x, w4, bias), what=f"ws plain, variant {v}")
        close(ops.conv3x3_ws(x.cuda(), wp, bias.cuda(), n, rowvec=rowvec.cuda(), res=res.cuda(), variant=v),
              reference(x, w4, bias, rowvec=rowvec, res=res), what=f"ws rowvec + residual, variant {v}")


@pytest.mark.parametrize("b,h,w,c,n,s0,s1", [(1, 8, 8, 1280, 1280, 1280, 1280), (1, 16, 16, 1280, 1280, 640, 0), (2, 8, 8, 128, 64, 128, 256),
                                             (1, 16, 16, 256, 32, 384, 0), (1, 32, 32, 128, 32, 256, 128),
                                             (1, 12, 12, 1280, 1280, 1280, 1280), (1, 24, 24, 256, 64, 384, 128), (2, 12, 12, 128, 32, 256, 0)])
def test_conv_ws_fused_shortcut(ops, b, h, w, c, n, s0, s1):
    """conv2 | conv_shortcut of a channel-changing resnet: the 1x1 shortcut over the block input (one tensor, or the two halves
    of a skip concatenation) is a second K segment of the same launch."""
    from mvd_amd.packing import pack_ws
    x, w4 = rnd(b, h, w, c, seed=1), rnd(n, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=2)
    sc0 = rnd(b, h, w, s0, seed=6)
    sc1 = rnd(b, h, w, s1, seed=7) if s1 else None
    wsc = rnd(n, s0 + s1, scale=1 / math.sqrt(s0 + s1), seed=8)
    bias = rnd(n, seed=3, dtype=torch.float32)
    wp = pack_ws(w4, wsc).cuda()
    sc = torch.cat([sc0, sc1], -1) if s1 else sc0
    for v in variants(h, w):
        got = ops.conv3x3_ws(x.cuda(), wp, bias.cuda(), n, shortcut=sc0.cuda(), shortcut2=sc1.cuda() if s1 else None, variant=v)
        close(got, reference(x, w4, bias, wsc=wsc, sc=sc), what=f"ws + shortcut, variant {v}")


@pytest.mark.parametrize("b,h,w,c,n", [(1, 8, 8, 1280, 1280), (1, 16, 16, 256, 64), (2, 4, 8, 128, 48), (1, 2, 16, 128, 16), (3, 8, 8, 128, 32),
                                       (1, 12, 12, 1280, 1280), (2, 4, 12, 128, 32)])
def test_conv_ws_upsample_in_front(ops, b, h, w, c, n):
    """diffusers' Upsample2D: nearest-neighbour 2x, then the 3x3 convolution -- the slab holds the INPUT rows, a tap reads
    pixel (uy >> 1, ux >> 1).  Output maps 16 and 32 wide, both block heights where they apply."""
    from mvd_amd.packing import pack_ws
    x, w4 = rnd(b, h, w, c, seed=31), rnd(n, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=32)
    bias = rnd(n, seed=33, dtype=torch.float32)
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    want = F.conv2d(up, w4.float(), padding=1).permute(0, 2, 3, 1) + bias
    wp = pack_ws(w4).cuda()
    for v in variants(2 * h, 2 * w):
        close(ops.conv3x3_ws(x.cuda(), wp, bias.cuda(), n, variant=v, upsample=True), want, what=f"ws upsample, variant {v}")


def test_conv_ws_upsample_taps_exact(ops):
    """one-hot input pixel and one-hot weights under the upsampling: exact 0 / 1 pattern at every border"""
    from mvd_amd.packing import pack_ws
    h = w = 8
    c, n = 128, 16
    for (py, px) in [(0, 0), (h - 1, w - 1), (3, 0), (0, w - 1), (4, 5)]:
        x = torch.zeros(1, h, w, c)
        x[0, py, px, 70] = 1.0
        w4 = torch.zeros(n, c, 3, 3)
        for t in range(9):
            w4[t, 70, t // 3, t % 3] = 1.0
        up = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
        want = F.conv2d(up, w4, padding=1).permute(0, 2, 3, 1)
        for v in (1, 2):
            got = ops.conv3x3_ws(x.to(torch.bfloat16).cuda(), pack_ws(w4).cuda(), torch.zeros(n).cuda(), n, variant=v, upsample=True).float().cpu()
            assert torch.equal(got, want), (py, px, v)


@pytest.mark.parametrize("h,w", [(8, 8), (16, 16), (8, 32), (12, 12), (8, 24)])
def test_conv_ws_taps_and_borders_exact(ops, h, w):
    """A one-hot pixel through one-hot weights: out[y][x][n] = 1 exactly where (y, x) = pixel - tap offset lies in the map --
    every tap, both borders, every 16-pixel block, each wave's channel quarter (exact in bf16: single products of 1)."""
    from mvd_amd.packing import pack_ws
    c, n = 128, 16
    for (py, px) in [(0, 0), (h - 1, w - 1), (3, 0), (0, w - 1), (h // 2, w // 2)]:
        for ch in (5, 37, 70, 127):                           # one channel in each wave's quarter
            x = torch.zeros(1, h, w, c)
            x[0, py, px, ch] = 1.0
            w4 = torch.zeros(n, c, 3, 3)
            for t in range(9):
                w4[t, ch, t // 3, t % 3] = 1.0                # output channel t picks tap t
            want = torch.zeros(1, h, w, n)
            for t in range(9):
                y, xx = py - (t // 3 - 1), px - (t % 3 - 1)
                if 0 <= y < h and 0 <= xx < w:
                    want[0, y, xx, t] = 1.0
            for v in variants(h, w):
                got = ops.conv3x3_ws(x.to(torch.bfloat16).cuda(), pack_ws(w4).cuda(), torch.zeros(n).cuda(), n, variant=v).float().cpu()
                assert torch.equal(got, want), (py, px, ch, v)


def test_conv_ws_matches_the_implicit_gemm_kernel(ops):
    """Same problem through the kernel the engine used for it before (gemm_sm / gemm split-K via ops.conv3x3): both within
    tolerance of the reference and of each other."""
    from mvd_amd.packing import pack_ws, _conv_w
    b, h, w, c, n = 1, 8, 8, 1280, 1280
    x, w4 = rnd(b, h, w, c, seed=11), rnd(n, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=12)
    bias = rnd(n, seed=13, dtype=torch.float32)
    got = ops.conv3x3_ws(x.cuda(), pack_ws(w4).cuda(), bias.cuda(), n)
    old = ops.conv3x3(x.cuda(), _conv_w(w4).to(torch.bfloat16).cuda(), bias.cuda())
    close(got, old.float().cpu(), tol=2 ** -6, what="ws vs implicit GEMM")


def test_conv_ws_is_bit_deterministic(ops):
    from mvd_amd.packing import pack_ws
    b, h, w, c, n = 1, 16, 16, 640, 1280
    x, w4 = rnd(b, h, w, c, seed=21).cuda(), rnd(n, c, 3, 3, scale=0.02, seed=22)
    wp, bias = pack_ws(w4).cuda(), rnd(n, seed=23, dtype=torch.float32).cuda()
    for v in (1, 2):
        first = ops.conv3x3_ws(x, wp, bias, n, variant=v).clone()
        for _ in range(20):
            assert torch.equal(ops.conv3x3_ws(x, wp, bias, n, variant=v), first)


def test_conv_ws_rejects_shapes_it_does_not_take(ops):
    from mvd_amd import _lib as L
    from mvd_amd.packing import pack_ws
    w4 = rnd(16, 128, 3, 3)
    wp, bias = pack_ws(w4).cuda(), torch.zeros(16).cuda()
    for shape, v in [((1, 20, 20, 128), 0), ((1, 64, 64, 128), 0), ((32, 8, 8, 128), 0),      # 20-wide map, 64-wide map, more than 1024 rows
                     ((1, 6, 12, 128), 0),                                                     # 72 pixels: not whole 48-pixel blocks
                     ((1, 32, 32, 128), 2), ((1, 8, 8, 128), 2), ((1, 8, 8, 128), 3), ((1, 12, 12, 128), 1), ((1, 8, 8, 128), 4)]:        # forms that do not take the width / do not exist
        with pytest.raises(L.MvdError, match="conv_ws"):
            ops.conv3x3_ws(rnd(*shape).cuda(), wp, bias, 16, variant=v)
