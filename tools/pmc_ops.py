#!/usr/bin/env python3
"""A few launches of the hot kernels on UNet shapes, for `rocprofv3 --pmc` passes (see tools/pmc_ops_summary.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
rnd = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
B = 32
which = os.environ.get("PMC_OPS", "attn,conv,lin,geglu,xs,ws").split(",")
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
if "xs" in which:      # the X-stationary kernels (gemm_xs.hip) at the 64x64-level shapes: N = 320 + residual, LayerNorm N = 1280, GEGLU
    import math
    from mvd_amd.packing import pack_xs
    a = rnd(131072, 320)
    for n, geglu, ln, res in ((320, False, False, True), (1280, False, True, False), (2560, True, True, False)):
        wp = pack_xs(torch.randn(n, 320, device="cuda") / math.sqrt(320), torch.randn(n, device="cuda"), geglu=geglu)
        r = rnd(131072, n) if res else None
        for _ in range(3): ops.linear_xs(a, wp, geglu=geglu, ln=ln, res=r)
if "ws" in which:      # the weight-streaming convolution (conv_ws.hip) at batch 1: 8x8 / 16x16 (C = N = 1280) and 32x32 (C = N = 640),
    import math        # a fresh copy of the weights per launch (a forward streams every weight once: cold)
    from mvd_amd.packing import pack_ws
    for hw, c in ((8, 1280), (16, 1280), (32, 640)):
        x = rnd(1, hw, hw, c)
        wp = pack_ws((torch.randn(c, c, 3, 3) / math.sqrt(9 * c)).to(torch.bfloat16)).cuda()
        wps = [wp.clone() for _ in range(3)]
        bias = torch.randn(c, device="cuda")
        for i in range(3): ops.conv3x3_ws(x, wps[i], bias, c)
torch.cuda.synchronize()
