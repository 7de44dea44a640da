#!/usr/bin/env python3
"""Round 5, timing only: the 3x3 convolutions of a cfg4 forward (32 images) through the engine's launch choice.  Run once with the
product library and once with a gemm_pp variant built with -DPP_ABLATE_A_TAPS (A operand fetched for one tap in nine: an upper
bound of what an LDS-resident activation patch could save; results of that build are WRONG by construction):
    tools/build_pp_variant.sh ablate_a -DPP_ABLATE_A_TAPS;  tools/ab_probe.sh tools/probe_r05_conv_ablation.py mvd_amd/libmvd_hip.so mvd_amd/libmvd_hip_ablate_a.so"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mvd_amd import ops
from mvd_amd.packing import _conv_w

rnd = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)


def time_fn(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B = 32
for hw, cin, cout in [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 640, 640), (32, 1280, 640), (32, 1920, 640), (16, 1280, 1280), (16, 2560, 1280)]:
    x = rnd(B, hw, hw, cin)
    w = _conv_w((torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin))).to(torch.bfloat16).cuda()
    bias = torch.randn(cout, device="cuda")
    sk = ops.engine_splitk(B * hw * hw, cout, 9 * cin, conv=True)
    best = min(time_fn(lambda: ops.conv3x3(x, w, bias, splitk=sk)) for _ in range(3))
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    print(f"conv {hw}x{hw} x{B} {cin:4d} -> {cout:4d} (split {sk}): {best:7.1f} us  {fl / best * 1e-6:6.0f} TF   plan {ops.last_gemm_plan()}", flush=True)
