#!/usr/bin/env python3
"""A few launches of the hot kernels on UNet shapes, for `rocprofv3 --pmc` passes (see tools/pmc_ops_summary.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
rnd = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
B = 32
which = os.environ.get("PMC_OPS", "attn,conv,lin,geglu").split(",")
if "attn" in which:
    q, k, v = rnd(B, 4096, 320), rnd(B, 4096, 320), rnd(B, 4096, 320)
    for _ in range(3): ops.attention(q, k, v, 5, scale=0.0)      # the engine's form: prescaled q, LDS-DMA staging, dot2c denominators
if "conv" in which:
    x, w = rnd(B, 32, 32, 640), rnd(640, 9 * 640)
    for _ in range(3): ops.conv3x3(x, w, force_cfg=7)
if "lin" in which:
    a, w = rnd(131072, 320), rnd(320, 320)
    for _ in range(3): ops.linear(a, w, force_cfg=7)
    a2, w2 = rnd(131072, 1280), rnd(320, 1280)
    for _ in range(3): ops.linear(a2, w2, force_cfg=7)
    w3 = rnd(1280, 320)
    for _ in range(3): ops.linear(a, w3, force_cfg=7)
if "geglu" in which:
    a, w = rnd(131072, 320), rnd(2560, 320)
    for _ in range(3): ops.linear(a, w, geglu=True, force_cfg=6)
torch.cuda.synchronize()
