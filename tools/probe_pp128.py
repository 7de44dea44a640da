#!/usr/bin/env python3
"""Small-M shapes (16x16 and 8x8 levels at B=32): the 128x320 ping-pong tile (force_cfg 18) against the round-1 choices
(128x160 lock-step = force_cfg 10, 256x320 + split-K 2 = force_cfg 7)."""
import math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
from mvd_amd.packing import _conv_w

def rnd(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def med(fn): return statistics.median(timeit(fn) for _ in range(5))

for (m, n, k, res) in [(8192, 1280, 1280, 1), (8192, 1280, 2560, 1), (8192, 3840, 1280, 0), (8192, 1280, 5120, 1), (2048, 1280, 1280, 1), (2048, 5120, 1280, 0), (2048, 1280, 5120, 1)]:
    a, w, b = rnd(m, k), rnd(n, k, scale=1 / math.sqrt(k)), torch.randn(n, device="cuda")
    r = rnd(m, n) if res else None
    out = []
    for cfg, sk in [(18, 1), (18, 2), (18, 4), (10, 1), (10, 2), (10, 4), (7, 2), (-1, ops.engine_splitk(m, n, k))]:
        if sk > k // 64: continue
        t = med(lambda: ops.linear(a, w, b, res=r, force_cfg=cfg, splitk=sk))
        out.append(f"c{cfg}/s{sk}:{t:6.1f}")
    print(f"dense M={m} N={n} K={k} res={res} | " + " ".join(out), flush=True)
for (hw, cin, cout) in [(16, 1280, 1280), (16, 2560, 1280), (8, 1280, 1280), (8, 2560, 1280)]:
    x = rnd(32, hw, hw, cin)
    w = _conv_w(torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin)).to(torch.bfloat16).cuda()
    b = torch.randn(cout, device="cuda")
    m = 32 * hw * hw
    out = []
    for cfg, sk in [(18, 1), (18, 2), (18, 4), (10, 2), (10, 4), (7, 2), (7, 4), (-1, ops.engine_splitk(m, cout, 9 * cin))]:
        t = med(lambda: ops.conv3x3(x, w, b, force_cfg=cfg, splitk=sk))
        out.append(f"c{cfg}/s{sk}:{t:6.1f}")
    print(f"conv {hw}^2 {cin}->{cout} (M={m} K={9*cin}) | " + " ".join(out), flush=True)
