#!/usr/bin/env python3
"""Tile-config sweep for the short-K linear GEMMs of the transformer blocks (q/k/v/out projections, proj_in/out,
ff2) with rotating operands so that A never sits in the 256 MB Infinity Cache (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops

CFGS = [int(c) for c in os.environ.get("PROBE_CFGS", "7,8,10,11,12").split(",")]
TILES = {15: (256, 320), 7: (256, 320), 8: (256, 160), 9: (256, 128), 10: (128, 160), 11: (128, 128), 12: (128, 64), 13: (64, 64)}
ROT = 6

def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)

shapes = []
for M, c in [(131072, 320), (32768, 640), (8192, 1280)]:
    shapes += [(M, c, c, False), (M, c, c, True), (M, 3 * c, c, False), (M, 4 * c, c, False), (M, 2 * c, c, False),
               (M, c, 2 * c, True), (M, c, 4 * c, True)]

for M, N, K, res in shapes:
    n_rot = max(2, min(ROT, int(600e6 // (M * K * 2)) + 1))
    As = [rnd(M, K) for _ in range(n_rot)]
    R = rnd(M, N) if res else None
    w = rnd(N, K)
    row = []
    for cfg in CFGS:
        bm, bn = TILES[cfg]
        if N % bn: continue
        i = [0]
        def fn():
            i[0] += 1
            return ops.linear(As[i[0] % n_rot], w, res=R, force_cfg=cfg)
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 12
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        gb = (M * K + M * N * (2 if res else 1) + N * K) * 2 / 1e9
        row.append(f"c{cfg}:{us:6.1f}us {2.0*M*N*K/us/1e6:5.0f}TF {gb/us*1e3:5.2f}TB/s")
    print(f"M={M:6d} N={N:5d} K={K:5d} res={int(res)} | " + " | ".join(row), flush=True)
    del As, R
