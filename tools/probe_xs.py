#!/usr/bin/env python3
"""X-stationary kernels (gemm_xs.hip) vs the ping-pong kernels at the 64x64-level shapes of a 32-pair forward (M = 131072, K = 320)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
from mvd_amd.packing import fold_layernorm, _geglu_rows, pack_xs

M, K = 131072, 320
def rnd(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

xs = [rnd(M, K) for _ in range(4)]
g, b = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
csplits = [int(c) for c in os.environ.get("XS_CSPLIT", "0").split(",")]
SHAPES = [(320, False, False, False), (320, False, False, True), (320, False, True, False), (640, False, True, False),
                            (960, False, True, False), (1280, False, True, False), (1280, False, False, False), (2560, True, True, False)]
sel = [int(i) for i in os.environ['XS_SHAPES'].split(',')] if os.environ.get('XS_SHAPES') else range(len(SHAPES))
for (n, geglu, ln, res) in [SHAPES[i] for i in sel]:
    w = rnd(n, K, scale=1 / math.sqrt(K)).float()
    bias = torch.randn(n, device="cuda")
    r = rnd(M, n) if res else None
    it = [0]
    if ln:
        if geglu:
            wf, cf = fold_layernorm(_geglu_rows(w), g, b, _geglu_rows(bias), "cuda")
        else:
            wf, cf = fold_layernorm(w, g, b, bias, "cuda")
        def old():
            it[0] += 1; ops.ln_linear(xs[it[0] % 4], wf, cf, geglu=geglu)
        wf2, cf2 = fold_layernorm(w, g, b, bias, "cuda")
        wp = pack_xs(wf2.float(), cf2[1], geglu=geglu)
    else:
        wb = w.to(torch.bfloat16)
        def old():
            it[0] += 1; ops.linear(xs[it[0] % 4], wb, bias, res=r)
        wp = pack_xs(w, bias, geglu=geglu)
    t_old = timeit(old) if not os.environ.get('XS_ONLY') else 0.0
    line = f"N={n:5d} geglu={int(geglu)} ln={int(ln)} res={int(res)}: ping-pong {t_old:7.1f} us"
    fl = 2e-6 * M * n * K
    by = (M * K * 2 + M * (n // 2 if geglu else n) * 2 * (2 if res else 1)) / 1e6
    for cs in csplits:
        groups = n // 64
        if cs > 0 and groups % cs: continue
        def new():
            it[0] += 1; ops.linear_xs(xs[it[0] % 4], wp, geglu=geglu, ln=ln, res=r, csplit=cs)
        t = timeit(new)
        line += f" | xs cs={cs}: {t:7.1f} us ({fl / t:.0f} TF, {by / t:.2f} TB/s)"
    print(line, flush=True)
