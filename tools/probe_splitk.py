#!/usr/bin/env python3
"""Time the split-K launches of the 32-pair forward in isolation (convolutions of the 16x16 and 8x8 levels at the schedule's
split factor, and the same launches unsplit): MVD_HIP_LIB=<lib> python tools/probe_splitk.py   -- for same-box A/B of two builds."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvd_amd import ops

def bench(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g = torch.Generator().manual_seed(0)
for (B, H, cin, cout) in ((32, 16, 640, 1280), (32, 16, 1280, 1280), (32, 16, 2560, 1280), (32, 8, 1280, 1280), (32, 8, 2560, 1280)):
    x = (torch.randn(B, H, H, cin, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn(cout, 9 * cin, generator=g) / math.sqrt(9 * cin)).to(torch.bfloat16).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    temb = torch.randn(B, cout, generator=g).cuda()
    res = torch.randn(B, H, H, cout, generator=g).to(torch.bfloat16).cuda()
    sk = ops.engine_splitk(B * H * H, cout, 9 * cin)
    fl = 2.0 * B * H * H * cout * 9 * cin
    t_s = bench(lambda: ops.conv3x3(x, w, bias, rowvec=temb, res=res, splitk=sk))
    plan = ops.last_gemm_plan()
    t_1 = bench(lambda: ops.conv3x3(x, w, bias, rowvec=temb, res=res, splitk=1))
    print(f"conv M={B*H*H} N={cout} K={9*cin} cfg={plan['cfg']} split={sk}: {t_s:7.1f} us ({fl/t_s/1e6:6.0f} TF/s)   unsplit {t_1:7.1f} us ({fl/t_1/1e6:6.0f} TF/s)", flush=True)
