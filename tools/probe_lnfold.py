#!/usr/bin/env python3
"""LayerNorm kernel + GEMM vs the fused LayerNorm GEMM (gemm_pp_kernel<..., LNF>) at the cfg4 shapes of the 64^2 / 32^2 levels."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
from mvd_amd.packing import fold_layernorm, _geglu_rows

def rnd(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (m, c, nmul, geglu) in [(131072, 320, 4, False), (131072, 320, 2, False), (131072, 320, 8, True), (32768, 640, 4, False), (32768, 640, 2, False), (32768, 640, 8, True)]:
    n = nmul * c
    xs = [rnd(m, c) for _ in range(4)]                      # rotate operands: no L2 / MALL-resident inputs
    w = rnd(n, c, scale=1 / math.sqrt(c)).float()
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    bias = torch.randn(n, device="cuda") if geglu else None
    if geglu:
        wp, bp = _geglu_rows(w).to(torch.bfloat16).contiguous(), _geglu_rows(bias).contiguous()
        wf, cf = fold_layernorm(_geglu_rows(w), g, b, _geglu_rows(bias), "cuda")
    else:
        wp, bp = w.to(torch.bfloat16), None
        wf, cf = fold_layernorm(w, g, b, None, "cuda")
    it = [0]
    def two():
        it[0] += 1
        ops.linear(ops.layernorm(xs[it[0] % 4], g, b), wp, bp, geglu=geglu)
    def lin_only():
        it[0] += 1
        ops.linear(xs[it[0] % 4], wp, bp, geglu=geglu)
    def fused():
        it[0] += 1
        ops.ln_linear(xs[it[0] % 4], wf, cf, geglu=geglu)
    t2, tl, tf = timeit(two), timeit(lin_only), timeit(fused)
    print(f"M={m} C={c} N={n} geglu={int(geglu)}: layernorm+gemm {t2:7.1f} us (gemm alone {tl:7.1f})  fused {tf:7.1f} us  ({2e-6 * m * n * c / tf:.0f} TFLOP/s)", flush=True)
