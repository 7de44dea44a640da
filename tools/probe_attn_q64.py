#!/usr/bin/env python3
"""The engine's attention form at 64 queries per wave (attn_q64_kernel: two 32-query sub-tiles share every K / V^T fragment,
Q fragments re-read from LDS, two waves per SIMD) against the product kernel (32 queries per wave, four waves per SIMD), on the
32-pair forward's attention shapes.  mvd_debug_set_attention_nw(34 / 33) routes launches to the 4- / 2-wave form of it (probe builds only:
   python tools/build_variant.py probe -DMVD_PROBE; MVD_HIP_LIB=mvd_amd/libmvd_hip_probe.so python tools/probe_attn_q64.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvd_amd import ops, _lib as L

def time_fn(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
for name, B, h, n, nk in [("L0 self 4096x4096", 32, 5, 4096, 4096), ("L1 self 1024x1024", 32, 10, 1024, 1024), ("L2 self 256x256", 32, 20, 256, 256),
                          ("L0 text 4096x77", 32, 5, 4096, 77), ("ragged 1000x1111", 8, 5, 1000, 1111)]:
    q, k, v = rnd(B, n, h * 64) * 0.4, rnd(B, nk, h * 64), rnd(B, nk, h * 64)
    ref = None
    for tag, nw in (("product 32 q/wave", -1), ("64 q/wave, 4 waves", 34), ("64 q/wave, 2 waves", 33)):
        L.lib().mvd_debug_set_attention_nw(nw)
        out = ops.attention(q, k, v, h, scale=0.0)
        if ref is None: ref = out.float()
        err = (out.float() - ref).abs().max().item()
        ms = time_fn(lambda: ops.attention(q, k, v, h, scale=0.0))
        print(f"{name:20s} {tag:20s}: {ms*1e3:8.1f} us {4.0*B*h*n*nk*64/ms/1e9:7.0f} TF/s   max |diff| vs product {err:.3g}", flush=True)
    L.lib().mvd_debug_set_attention_nw(-1)
