#!/usr/bin/env python3
"""Round 5: same-process A/B of (1) the column-group tile walk of the ping-pong GEMM (gemm_pp.hip walk_cg; off = debug flag
131072) at the 32x32-level shapes whose weights exceed an XCD's L2, and (2) the 8x8-level convolution rule (256x320 tile at
split 8; off = MVD_GEMM_DEEP_CONV_SPLIT=0 needs a second process, so the old choice is forced through force_cfg / splitk).
PMC_MODE=1: three launches per variant and nothing else (for `rocprofv3 --pmc FETCH_SIZE` / WRITE_SIZE passes)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mvd_amd import _lib as L
from mvd_amd import ops
from mvd_amd.packing import _geglu_rows

PMC = os.environ.get("PMC_MODE") == "1"
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


def ab(name, fn, flops, rounds=3):
    if PMC:
        for flag in (131072, 0):
            L.lib().mvd_debug_set_flags(flag)
            for _ in range(3):
                fn()
        L.lib().mvd_debug_set_flags(0)
        torch.cuda.synchronize()
        return
    res = {0: [], 131072: []}
    for _ in range(rounds):
        for flag in (131072, 0):
            L.lib().mvd_debug_set_flags(flag)
            res[flag].append(time_fn(fn))
    L.lib().mvd_debug_set_flags(0)
    old, new = min(res[131072]), min(res[0])
    print(f"{name:46s} row-major walk {old:7.1f} us ({flops / old * 1e-6:5.0f} TF)   column groups {new:7.1f} us ({flops / new * 1e-6:5.0f} TF)   "
          f"{(old / new - 1) * 100:+5.1f} %   plan {ops.last_gemm_plan()}", flush=True)


M = 32768
a = rnd(M, 640)
for n, geglu, tag in ((5120, True, "GEGLU ff1 N 5120 K 640"), (2560, False, "q|k|v|q_ref N 2560 K 640"), (1920, False, "q|k|v N 1920 K 640 (not taken: 6 tiles)")):
    w = rnd(n, 640) * (1 / math.sqrt(640) / 0.5)
    bias = torch.randn(n, device="cuda")
    if geglu:
        w, bias = _geglu_rows(w).contiguous(), _geglu_rows(bias).contiguous()
    ab(f"M {M} {tag}", lambda: ops.linear(a, w, bias, geglu=geglu), 2.0 * M * n * 640)
# LayerNorm-folded forms of the same launches (what the engine runs at the 32x32 level)
from mvd_amd.packing import fold_layernorm
for n, tag in ((2560, "LN + q|k|v|q_ref N 2560"), (1920, "LN + q|k|v N 1920")):
    wf, cf = fold_layernorm(torch.randn(n, 640, device="cuda") / math.sqrt(640), torch.ones(640, device="cuda"), torch.zeros(640, device="cuda"), None, "cuda")
    ab(f"M {M} {tag}", lambda: ops.ln_linear(a, wf, cf), 2.0 * M * n * 640)

if not PMC:
    # (2) the 8x8-level convolutions at 32 images: heuristic (now 256x320 tile, split 8) against the old choice (128x160, split 4)
    from mvd_amd.packing import _conv_w
    pack = lambda w: _conv_w(w.float().cpu()).to(torch.bfloat16).cuda()       # noqa: E731
    for ci in (1280, 2560):
        x = rnd(32, 8, 8, ci)
        w = pack((torch.randn(1280, ci, 3, 3, device="cuda") / math.sqrt(9 * ci)).to(torch.bfloat16))
        bias = torch.randn(1280, device="cuda")
        fl = 2.0 * 2048 * 1280 * 9 * ci
        for r in range(2):
            t_new = time_fn(lambda: ops.conv3x3(x, w, bias))
            plan = ops.last_gemm_plan()
            t_old = time_fn(lambda: ops.conv3x3(x, w, bias, force_cfg=10, splitk=4))
            print(f"conv 8x8 x32 {ci} -> 1280: heuristic {t_new:7.1f} us ({fl / t_new * 1e-6:5.0f} TF) plan {plan}   old 128x160 split 4 {t_old:7.1f} us "
                  f"({fl / t_old * 1e-6:5.0f} TF)   {(t_old / t_new - 1) * 100:+5.1f} %", flush=True)
