#!/usr/bin/env python3
"""Diagnosis of the reverted four-pixel conv_out: run it on a FIXED input many times -- alone or with another
process on the GPU -- and report (a) launches whose output differs from the first, (b) where (x mod 4 of the differing
pixels), (c) its distance from the product (one pixel per wave) kernel's output on the same input.
    MVD_HIP_LIB=mvd_amd/libmvd_hip_co4.so MVD_CONV_OUT4=1 python tools/probe_conv_out4.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
REPS = int(os.environ.get("REPS", "300"))
torch.manual_seed(0)
B = 32
x = (torch.randn(B, 64, 64, 320, device="cuda") * 0.5).to(torch.bfloat16)
w4 = (torch.randn(4, 9 * 320, device="cuda") * 0.5).to(torch.bfloat16)
b4 = torch.randn(4, device="cuda")
xh = int(x.view(torch.int16).long().sum())
first = ops.conv_out(x, w4, b4).clone()
bad, pos = 0, [0, 0, 0, 0]
for i in range(REPS):
    out = ops.conv_out(x, w4, b4)
    d = out != first
    if d.any():
        bad += 1
        for xx in d.nonzero()[:, 3].tolist():
            pos[xx % 4] += 1
assert int(x.view(torch.int16).long().sum()) == xh, "the INPUT changed"
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w4.float().reshape(4, 3, 3, 320).permute(0, 3, 1, 2), b4, padding=1)
err = (first - ref).abs()
print(f"conv_out variant MVD_CONV_OUT4={os.environ.get('MVD_CONV_OUT4', '0')} lib={os.path.basename(os.environ.get('MVD_HIP_LIB', 'libmvd_hip.so'))}: "
      f"{bad}/{REPS} launches differ from the first; differing elements by (x mod 4): {pos}; vs fp32 conv2d: max |err| {float(err.max()):.3e}, "
      f"elements with |err| > 1e-3: {int((err > 1e-3).sum())} by (x mod 4): {[int((err[..., k::4] > 1e-3).sum()) for k in range(4)]}", flush=True)
