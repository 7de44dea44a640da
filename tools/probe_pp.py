#!/usr/bin/env python3
"""A/B of the ping-pong 256x320 kernels (gemm_pp.hip: force_cfg 7 / 6) against the lock-step round-1 kernels
(force_cfg 17 / 16) on the cfg4 (B=32) shapes; interleaved rounds in one process, median of the per-round times."""
import math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
from mvd_amd.packing import _conv_w, _geglu_rows

def rnd(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)
def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

cases = []
for (m, n, k, res) in [(131072, 320, 320, 0), (131072, 320, 320, 1), (131072, 1280, 320, 0), (131072, 320, 1280, 1), (131072, 320, 640, 1),
                       (32768, 640, 640, 0), (32768, 2560, 640, 0), (32768, 640, 2560, 1), (8192, 5120, 1280, 0)]:
    a, w, b = rnd(m, k), rnd(n, k, scale=1 / math.sqrt(k)), torch.randn(n, device="cuda")
    r = rnd(m, n) if res else None
    cases.append((f"dense M={m} N={n} K={k} res={res}", 2.0 * m * n * k, lambda c, a=a, w=w, b=b, r=r: ops.linear(a, w, b, res=r, force_cfg=c), (7, 17)))
for (m, c) in [(131072, 320), (32768, 640), (8192, 1280)]:
    a, w, b = rnd(m, c), _geglu_rows(rnd(8 * c, c, scale=1 / math.sqrt(c))).contiguous(), torch.randn(8 * c, device="cuda")
    cases.append((f"geglu M={m} C={c}", 2.0 * m * 8 * c * c, lambda cf, a=a, w=w, b=b: ops.linear(a, w, b, geglu=True, force_cfg=cf), (6, 16)))
for (hw, cin, cout, ups) in [(64, 320, 320, 0), (64, 960, 320, 0), (32, 640, 640, 0), (32, 1920, 640, 0), (32, 640, 640, 1), (16, 1280, 1280, 0)]:
    x = rnd(32, hw, hw, cin)
    w = _conv_w(torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin)).to(torch.bfloat16).cuda()
    b = torch.randn(cout, device="cuda")
    oh = hw * 2 if ups else hw
    sk = ops.engine_splitk(32 * oh * oh, cout, 9 * cin)
    cases.append((f"conv {hw}^2 {cin}->{cout} ups={ups} splitk={sk}", 2.0 * 32 * oh * oh * cout * 9 * cin,
                  lambda c, x=x, w=w, b=b, ups=ups, sk=sk: ops.conv3x3(x, w, b, upsample=bool(ups), force_cfg=c, splitk=sk), (7, 17)))
for name, fl, fn, (new, old) in cases:
    tn, to = [], []
    for _ in range(5):
        tn.append(timeit(lambda: fn(new))); to.append(timeit(lambda: fn(old)))
    a, b = statistics.median(tn), statistics.median(to)
    same = torch.equal(fn(new), fn(old))
    print(f"{name:44s} pp {a:8.1f} us {fl / a / 1e6:6.0f} TF | lock-step {b:8.1f} us {fl / b / 1e6:6.0f} TF | x{b / a:.2f} bit-equal={same}", flush=True)
