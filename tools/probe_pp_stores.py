#!/usr/bin/env python3
"""Ping-pong kernels at the short-K shapes that stay on them at 32 pairs (timing probe for epilogue store experiments)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
def rnd(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (m, n, k, res) in [(131072, 320, 640, True), (131072, 320, 1280, True), (32768, 640, 640, True), (32768, 640, 1280, True), (32768, 640, 2560, True), (32768, 2560, 640, False)]:
    xs = [rnd(m, k) for _ in range(3)]
    w = rnd(n, k, scale=1 / math.sqrt(k)); b = torch.randn(n, device="cuda"); r = rnd(m, n) if res else None
    it = [0]
    def f():
        it[0] += 1; ops.linear(xs[it[0] % 3], w, b, res=r)
    t = timeit(f)
    print(f"linear M={m} N={n} K={k} res={int(res)}: {t:7.1f} us ({2e-6 * m * n * k / t:.0f} TF)", flush=True)
x = rnd(32, 64, 64, 320); wc = rnd(320, 9 * 320, scale=0.02); bc = torch.randn(320, device="cuda")
t = timeit(lambda: ops.conv3x3(x, wc, bc, res=x))
print(f"conv3x3 32x64x64 320->320 + res: {t:7.1f} us ({2e-6 * 131072 * 320 * 2880 / t:.0f} TF)")
x2 = rnd(32, 32, 32, 640); wc2 = rnd(640, 9 * 640, scale=0.02); bc2 = torch.randn(640, device="cuda")
t = timeit(lambda: ops.conv3x3(x2, wc2, bc2, res=x2))
print(f"conv3x3 32x32x32 640->640 + res: {t:7.1f} us ({2e-6 * 32768 * 640 * 5760 / t:.0f} TF)")
