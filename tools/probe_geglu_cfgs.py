#!/usr/bin/env python3
"""GEGLU ff1 at the 64x64 level (M = 131072, N = 2560, K = 320) through every tile configuration that supports it:
ping-pong 256x320 (cfg 6, one 8-wave workgroup per CU, common epilogue phase), its LayerNorm-folded form (the engine's), and
the lock-step 256x128 / 128x128 kernels (cfg 1 / 3: independent workgroups per CU, one's epilogue under the other's MFMAs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvd_amd import ops, packing

def t_us(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (M, C) in ((131072, 320), (32768, 640), (8192, 1280)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, C, device="cuda", generator=g).bfloat16()
    w = (torch.randn(8 * C, C, device="cuda", generator=g) / C ** 0.5).bfloat16()
    b = torch.randn(8 * C, device="cuda", generator=g)
    fl = 2.0 * M * 8 * C * C
    for cfg in (6, 1, 3):
        us = t_us(lambda: ops.linear(x, w, b, geglu=True, force_cfg=cfg))
        print(f"M={M} C={C} cfg={cfg}: {us:8.1f} us  {fl / us / 1e6:7.0f} TF", flush=True)
    us = t_us(lambda: ops.layernorm(x, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")))
    print(f"M={M} C={C} layernorm alone: {us:8.1f} us", flush=True)
    if C <= packing.LN_FOLD_MAX_C:
        wf, cf = packing.fold_layernorm(w, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), b, "cuda")
        try:
            us = t_us(lambda: ops.ln_linear(x, wf, cf, geglu=True))
            print(f"M={M} C={C} ln-fold cfg 6: {us:8.1f} us  {fl / us / 1e6:7.0f} TF", flush=True)
        except Exception as e:
            print("ln-fold:", str(e)[:100])
