#!/usr/bin/env python3
"""(tile config, split-K) sweep on the GEMM / conv shapes of a SMALL-batch forward (TUNE_B pairs, default 1)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops

B = int(os.environ.get("TUNE_B", "1"))
TILES = {7: (256, 320), 8: (256, 160), 10: (128, 160), 11: (128, 128), 12: (128, 64), 13: (64, 64)}
SPLITS = [1, 2, 4, 8]

def time_fn(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)

shapes = []
for lvl, (hw, c) in enumerate([(64, 320), (32, 640), (16, 1280), (8, 1280)]):
    M = B * hw * hw
    shapes.append(("conv", f"L{lvl} conv {c}->{c} M={M}", dict(hw=hw, cin=c, cout=c)))
    if lvl < 3:
        shapes.append(("lin", f"L{lvl} sq   M={M} K={c} N={c}", dict(M=M, K=c, N=c)))
        shapes.append(("lin", f"L{lvl} qkvq M={M} K={c} N={4*c}", dict(M=M, K=c, N=4 * c)))
        shapes.append(("lin", f"L{lvl} out  M={M} K={2*c} N={c}", dict(M=M, K=2 * c, N=c)))
        shapes.append(("lin", f"L{lvl} ff2  M={M} K={4*c} N={c}", dict(M=M, K=4 * c, N=c)))
shapes.append(("conv", "L1 up conv 1920->640", dict(hw=32, cin=1920, cout=640)))
shapes.append(("conv", "L2 up conv 2560->1280", dict(hw=16, cin=2560, cout=1280)))

for kind, name, p in shapes:
    res = []
    for cfg, (bm, bn) in TILES.items():
        for sk in SPLITS:
            try:
                if kind == "lin":
                    if p["N"] % bn or sk > p["K"] // 64: continue
                    a, w = rnd(p["M"], p["K"]), rnd(p["N"], p["K"])
                    fn = lambda: ops.linear(a, w, force_cfg=cfg, splitk=sk)
                    fl = 2.0 * p["M"] * p["N"] * p["K"]
                else:
                    if p["cout"] % bn: continue
                    x = rnd(B, p["hw"], p["hw"], p["cin"])
                    k = 9 * p["cin"]
                    w = rnd(p["cout"], k)
                    fn = lambda: ops.conv3x3(x, w, force_cfg=cfg, splitk=sk)
                    fl = 2.0 * B * p["hw"] ** 2 * p["cout"] * k
                us = time_fn(fn) * 1e3
                res.append((us, cfg, sk))
            except Exception:
                pass
    res.sort()
    auto = time_fn((lambda: ops.linear(a, w)) if kind == "lin" else (lambda: ops.conv3x3(x, w))) * 1e3
    print(f"{name:34s} auto(no split) {auto:6.1f}us | best " + " ".join(f"c{c}/s{s}:{u:.1f}" for u, c, s in res[:5]), flush=True)
