#!/usr/bin/env python3
"""The X-stationary K = 320 kernels (gemm_xs.hip) on ONE image's 64x64 map (M = 4096 rows: 16 row blocks) against the small-M
kernel the engine uses there (gemm_sm.hip, through mvd_op_linear / mvd_op_ln_linear's heuristic).  Cold weights, warm rows, HIP
events around the launch, empty bracket subtracted (as tools/tune_sm.py)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
from mvd_amd.packing import pack_xs, fold_layernorm

dev = "cuda"
flush = torch.empty(150 * 1024 * 1024, device=dev, dtype=torch.float32).normal_()
ITERS = 9


def bracket(fn, warm):
    ts = []
    for _ in range(ITERS + 1):
        flush.sum()
        for t in warm:
            t.view(torch.int16).max()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[1:])
    return v[len(v) // 2]


EMPTY = bracket(lambda: None, [])
print(f"# empty bracket {EMPTY:.2f} us (subtracted); median of {ITERS}")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)   # noqa: E731
for M in (4096, 8192, 16384):
    x = rnd(M, 320)
    for name, n, geglu, ln, res in (("proj / out  N=320 + residual", 320, False, False, True), ("q2  LN N=320", 320, False, True, False),
                                    ("qkv LN N=960", 960, False, True, False), ("qkv+ref LN N=1280", 1280, False, True, False), ("ff1 GEGLU LN N=2560", 2560, True, True, False)):
        w = torch.randn(n, 320, generator=g) / math.sqrt(320)
        b = torch.randn(n, generator=g)
        r = rnd(M, n) if res else None
        if ln:
            gam, bet = 1 + 0.1 * torch.randn(320, generator=g), 0.1 * torch.randn(320, generator=g)
            wf, cf = fold_layernorm(w, gam, bet, b, dev)
            wp = pack_xs(wf.float().cpu(), cf[1].cpu(), geglu=geglu).to(dev)
            old = None        # (the engine's LayerNorm-folded small-M launch has no op-level entry: compare with tools/tune_sm.py's table)
        else:
            wp = pack_xs(w, b).to(dev)
            wb, bb = w.to(torch.bfloat16).to(dev), b.to(dev)
            old = lambda: ops.linear(x, wb, bb, res=r, force_cfg=103)   # noqa: E731  (gemm_sm 64x64 tiles, ring depth 3: the engine's plan)
        out = []
        for cs in (0, 1, 5 if n == 320 else 2):
            try:
                new = lambda: ops.linear_xs(x, wp, geglu=geglu, ln=ln, res=r, csplit=cs)   # noqa: E731
                new()
                out.append(f"xs csplit {cs}: {bracket(new, [x] + ([r] if res else [])) - EMPTY:5.1f} us")
            except Exception as e:      # noqa: BLE001
                out.append(f"xs csplit {cs}: n/a")
        so = "(see tune_sm)"
        if old is not None:
            try:
                so = f"{bracket(old, [x] + ([r] if res else [])) - EMPTY:5.1f} us"
            except Exception as e:      # noqa: BLE001
                so = f"n/a ({str(e)[:40]})"
        print(f"M={M:6d} {name:28s} | engine's small-M / tiled kernel {so} | " + "  ".join(out), flush=True)
