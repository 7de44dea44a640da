#!/usr/bin/env python3
"""Practical HBM ceilings on the box: device copy, read-only reduction, write-only fill (torch kernels)."""
import torch
def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for mb in (84, 168, 512, 2048):
    n = mb * 1000 * 1000 // 2
    xs = [torch.randn(n, device="cuda").to(torch.bfloat16) for _ in range(3)]
    y = torch.empty_like(xs[0])
    i = [0]
    def cp(): i[0] += 1; y.copy_(xs[i[0] % 3])
    def rd(): i[0] += 1; return xs[i[0] % 3].sum()
    def wr(): y.fill_(1.0)
    def ad(): i[0] += 1; torch.add(xs[i[0] % 3], xs[(i[0] + 1) % 3], out=y)
    print(f"{mb:5d} MB: copy {2*n*2/t(cp)/1e12:5.2f} TB/s | read(sum) {n*2/t(rd)/1e12:5.2f} | fill {n*2/t(wr)/1e12:5.2f} | add(2r1w) {3*n*2/t(ad)/1e12:5.2f}", flush=True)
