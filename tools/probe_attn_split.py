#!/usr/bin/env python3
"""Batch-1 self-attention sites: unsplit forms (1/2/4 waves per workgroup) vs split-KV over 2..8 key ranges (warm inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops, _lib as L
dev = "cuda"
flush = torch.empty(64 * 1024 * 1024, device=dev)


def bracket(fn, warm, iters=9):
    ts = []
    for _ in range(iters + 1):
        flush.sum()
        for t in warm:
            t.view(torch.int16).max()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[1:])
    return v[len(v) // 2]


EMPTY = bracket(lambda: None, [])
print(f"empty bracket {EMPTY:.2f} us")
for (B, heads, n) in [(1, 5, 4096), (1, 10, 1024), (1, 20, 256), (2, 5, 4096), (1, 5, 9216), (2, 5, 9216), (1, 10, 2304), (2, 10, 2304)]:
    C = heads * 64
    qkv = (torch.randn(B, n, 3 * C, device=dev) * 0.5).to(torch.bfloat16)
    q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    res = []
    for nw in (0, 1, 2):
        L.lib().mvd_debug_set_attention_nw(nw)
        res.append(f"{1 << nw}w {bracket(lambda: ops.attention(q, k, v, heads, scale=0.0), [qkv]) - EMPTY:6.1f}")
    L.lib().mvd_debug_set_attention_nw(18)
    res.append(f"4w-pipe {bracket(lambda: ops.attention(q, k, v, heads, scale=0.0), [qkv]) - EMPTY:6.1f}")
    L.lib().mvd_debug_set_attention_nw(-1)
    for ns in (2, 3, 4, 6, 8):
        if n // ns < 256:
            continue
        res.append(f"split{ns} {bracket(lambda: ops.attention_split(q, k, v, heads, ns), [qkv]) - EMPTY:6.1f}")
    fl = 4.0 * B * heads * n * n * 64
    print(f"B={B} heads={heads} n={n} ({fl / 1e9:.1f} GFLOP): " + " | ".join(res) + " us", flush=True)
