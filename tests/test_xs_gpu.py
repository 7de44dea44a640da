"""X-stationary short-K GEMM (mvd_amd/csrc/gemm_xs.hip) vs a PyTorch fp32 reference of the same op on bf16-rounded inputs.
Tolerance |err| <= 2^-7 * max|ref| as for the other GEMMs (tests/test_ops_gpu.py).  Covers every instantiation (plain,
residual, LayerNorm-in-registers, GEGLU with / without LayerNorm), ragged row counts (rows beyond M are never written),
column splits, strided operands and the cfg4 shapes of the 64x64 level."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

K = 320


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


@pytest.mark.parametrize("csplit", [0, 1, 2])
@pytest.mark.parametrize("m,n", [(256, 320), (1000, 640), (77, 320), (4096 + 33, 1280)])
def test_xs_plain_and_residual(ops, m, n, csplit):
    from mvd_amd.packing import pack_xs
    if csplit and (n // 64) % csplit:
        pytest.skip("column parts are whole store groups (unit pairs)")
    x, w = rnd(m, K, seed=1), rnd(n, K, scale=1 / math.sqrt(K), seed=2)
    bias, res = rnd(n, seed=3, dtype=torch.float32), rnd(m, n, seed=4)
    wp = pack_xs(w.float(), bias).cuda()
    want = x.float() @ w.float().T + bias
    close(ops.linear_xs(x.cuda(), wp, csplit=csplit), want, what="xs plain")
    close(ops.linear_xs(x.cuda(), wp, res=res.cuda(), csplit=csplit), want + res.float(), what="xs residual")
    # no bias: an all-zero bias k-step
    close(ops.linear_xs(x.cuda(), pack_xs(w.float(), None).cuda(), csplit=csplit), x.float() @ w.float().T, what="xs no bias")


def test_xs_bias_keeps_16_bits(ops):
    """The bias rides the matrix pipe as hi + lo bf16: with zero weights the output is bf16(bias) exactly."""
    from mvd_amd.packing import pack_xs
    m, n = 300, 320
    bias = rnd(n, seed=9, dtype=torch.float32) * 3
    got = ops.linear_xs(rnd(m, K, seed=1).cuda(), pack_xs(torch.zeros(n, K), bias).cuda())
    assert torch.equal(got.cpu(), bias.to(torch.bfloat16)[None].expand(m, n))


def test_xs_rows_beyond_m_untouched_and_strided_operands(ops):
    from mvd_amd.packing import pack_xs
    from mvd_amd import _lib as L
    import ctypes as C
    m, n = 700, 320
    big = rnd(m, 3 * K, seed=5).cuda()                       # operand = a column slice of a wider buffer (ldx = 960)
    x = big[:, K:2 * K]
    w = rnd(n, K, scale=1 / math.sqrt(K), seed=2)
    out = torch.full((m + 300, 2 * n), 7.0, device="cuda", dtype=torch.bfloat16)   # ldo = 640, guard rows behind M
    wp = pack_xs(w.float(), None).cuda()
    L.call("mvd_op_linear_xs", C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(wp.data_ptr()), m, K, n // 32, 0, 0, 1e-5,
           None, 0, C.c_void_p(out.data_ptr()), out.stride(0), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    close(out[:m, :n], x.float().cpu() @ w.float().T, what="xs strided")
    assert (out[m:] == 7.0).all() and (out[:, n:] == 7.0).all(), "wrote outside [M][N]"


@pytest.mark.parametrize("offset", [0.0, 60.0])
@pytest.mark.parametrize("m,n", [(512, 960), (1000, 320)])
def test_xs_layernorm(ops, m, n, offset):
    """LayerNorm in registers; rows with |mean| = 60 sigma exercise the E[x^2] - mean^2 cancellation."""
    from mvd_amd.packing import fold_layernorm, pack_xs
    x = (rnd(m, K, seed=1).float() + offset).to(torch.bfloat16)
    w = rnd(n, K, scale=1 / math.sqrt(K), seed=2).float()
    gamma, beta = 1 + 0.1 * rnd(K, seed=3, dtype=torch.float32), 0.1 * rnd(K, seed=4, dtype=torch.float32)
    bias = rnd(n, seed=5, dtype=torch.float32)
    wf, cf = fold_layernorm(w, gamma, beta, bias, "cpu")
    wp = pack_xs(wf.float(), cf[1]).cuda()
    want = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.T + bias
    tol = 2 ** -7 if offset == 0 else 2 ** -5      # (a 60-sigma offset leaves bf16 inputs ~2 bits of the deviation)
    close(ops.linear_xs(x.cuda(), wp, ln=True), want, tol=tol, what="xs layernorm")


@pytest.mark.parametrize("ln", [False, True])
@pytest.mark.parametrize("m,c", [(512, 320), (900, 160)])
def test_xs_geglu(ops, m, c, ln):
    """ff.net.0 of BasicTransformerBlock: proj to 8C', chunk (value, gate), value * gelu_erf(gate)."""
    from mvd_amd.packing import fold_layernorm, pack_xs
    n = 8 * c
    x = rnd(m, K, seed=1)
    w = rnd(n, K, scale=1 / math.sqrt(K), seed=2).float()
    bias = rnd(n, seed=3, dtype=torch.float32)
    if ln:
        gamma, beta = 1 + 0.1 * rnd(K, seed=4, dtype=torch.float32), 0.1 * rnd(K, seed=5, dtype=torch.float32)
        wf, cf = fold_layernorm(w, gamma, beta, bias, "cpu")
        wp = pack_xs(wf.float(), cf[1], geglu=True).cuda()
        h = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.T + bias
    else:
        wp = pack_xs(w, bias, geglu=True).cuda()
        h = x.float() @ w.T + bias
    val, gate = h.chunk(2, dim=-1)
    close(ops.linear_xs(x.cuda(), wp, geglu=True, ln=ln), val * F.gelu(gate), what="xs geglu")


@pytest.mark.parametrize("n,geglu,ln,res", [(320, False, False, True), (960, False, True, False), (1280, False, True, False),
                                            (1280, False, False, False), (2560, True, True, False)])
def test_xs_cfg4_shapes(ops, n, geglu, ln, res):
    """The launches of a 32-pair forward at the 64x64 level (M = 131072), through the heuristic column split; checked on a
    row sample against fp32 torch and for run-to-run bit stability."""
    from mvd_amd.packing import fold_layernorm, pack_xs
    m = 131072
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(m, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16).float()
    bias = torch.randn(n, device="cuda", generator=g)
    r = torch.randn(m, n, device="cuda", generator=g).to(torch.bfloat16) if res else None
    gamma, beta = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
    if ln:
        wf, cf = fold_layernorm(w, gamma, beta, bias, "cuda")
        wp = pack_xs(wf.float(), cf[1], geglu=geglu)
    else:
        wp = pack_xs(w, bias, geglu=geglu)
    got = ops.linear_xs(x, wp, geglu=geglu, ln=ln, res=r)
    again = ops.linear_xs(x, wp, geglu=geglu, ln=ln, res=r)
    assert torch.equal(got, again), "not bit-stable"
    plan = ops.last_gemm_plan()
    assert plan["cfg"] == 9 and plan["tiles"] >= 512, plan
    rows = torch.cat([torch.arange(0, 300), torch.arange(65536 - 40, 65536 + 40), torch.arange(m - 300, m)]).cuda()
    xs = x[rows].float()
    h = (F.layer_norm(xs, (K,), gamma, beta, 1e-5) if ln else xs) @ w.T + bias
    if geglu:
        v, gt = h.chunk(2, dim=-1)
        h = v * F.gelu(gt)
    if res:
        h = h + r[rows].float()
    close(got[rows], h.cpu(), what=f"xs cfg4 n={n}")
