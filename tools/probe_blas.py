#!/usr/bin/env python3
"""How fast does the vendor BLAS behind torch (hipBLASLt / rocBLAS) run the short-K projection shapes?  (reference point
for the hand-written GEMM; rotating A operands as in tools/probe_shortk.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mvd_amd import ops

def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)

def bench(fn, iters=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

shapes = [(131072, 320, 320), (131072, 1280, 320), (131072, 320, 1280), (32768, 640, 640), (32768, 2560, 640),
          (32768, 640, 2560), (8192, 1280, 1280), (8192, 5120, 1280), (8192, 1280, 5120), (131072, 320, 2880), (32768, 640, 5760)]
for M, N, K in shapes:
    n_rot = max(2, min(6, int(600e6 // (M * K * 2)) + 1))
    As = [rnd(M, K) for _ in range(n_rot)]
    w = rnd(N, K)
    i = [0]
    def mine():
        i[0] += 1
        return ops.linear(As[i[0] % n_rot], w)
    def blas():
        i[0] += 1
        return F.linear(As[i[0] % n_rot], w)
    t_m, t_b = bench(mine), bench(blas)
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d} | mine {t_m:7.1f}us {fl/t_m/1e6:5.0f}TF | torch/BLAS {t_b:7.1f}us {fl/t_b/1e6:5.0f}TF", flush=True)
    del As
