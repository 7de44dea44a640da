#!/usr/bin/env python3
"""Why do the deep-level split-K convolutions stream their weights at 1.3 TB/s?  Same weight bytes (29.5 MB), different
access patterns, cold weights / warm activations (tune_sm.py's bracket)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from mvd_amd import _lib as L
from mvd_amd.packing import block_weight
dev = "cuda"
flush = torch.empty(150 * 1024 * 1024, device=dev, dtype=torch.float32).normal_()
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)   # noqa: E731
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)         # noqa: E731


def bracket(fn, warm, iters=7, cold=True):
    ts = []
    for _ in range(iters + 1):
        if cold:
            flush.sum()
        for t in warm:
            t.view(torch.int16).max()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[1:])
    return v[len(v) // 2]


EMPTY = bracket(lambda: None, [], 20)
print(f"empty bracket {EMPTY:.2f} us")


def lin(M, N, K, cfg, sk, blocked=False, cold=True):
    a, w = rnd(M, K), rnd(N, K)
    wb = block_weight(w) if blocked else w
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(max(sk, 1) * M * N + 4096, device=dev, dtype=torch.float32)
    fn = lambda: L.call("mvd_op_linear", ptr(a), None, K, 0, ptr(wb), None, None, 0, 0, None, 1.0, 0, ptr(out), 0, M, N,
                        cfg + (1000 if blocked else 0), sk, ptr(ws), st())   # noqa: E731
    fn(); torch.cuda.synchronize()
    ref = a.float() @ w.float().T
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    t = bracket(fn, [a], cold=cold) - EMPTY
    return t, err


def conv(hw, cin, cout, cfg, sk, blocked=False, cold=True):
    x, w = rnd(1, hw, hw, cin), rnd(cout, 9 * cin)
    wb = block_weight(w) if blocked else w
    out = torch.empty(hw * hw, cout, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(max(sk, 1) * hw * hw * cout + 4096, device=dev, dtype=torch.float32)
    fn = lambda: L.call("mvd_op_conv3x3", ptr(x), 1, hw, hw, cin, 1, 0, 0, ptr(wb), None, None, 0, None, None, None, 0, 0, ptr(out), cout,
                        cfg + (1000 if blocked else 0), sk, ptr(ws), st())   # noqa: E731
    return bracket(fn, [x], cold=cold) - EMPTY


MB = 1280 * 11520 * 2 / 1e6
for sk in (4, 8, 12, 16):
    for ns in (3, 6):
        t1 = conv(8, 1280, 1280, 100 + ns, sk)
        t2, e2 = lin(64, 1280, 11520, 100 + ns, sk)
        t3, e3 = lin(64, 1280, 11520, 100 + ns, sk, blocked=True)
        t4 = conv(8, 1280, 1280, 100 + ns, sk, blocked=True)
        t5, _ = lin(64, 1280, 11520, 100 + ns, sk, blocked=True, cold=False)
        print(f"M=64 N=1280 K=11520 S={sk:2d} ns={ns}: conv {t1:5.1f} us ({MB / t1:.2f} TB/s) | dense {t2:5.1f} ({MB / t2:.2f}) err {e2:.1e} | dense blockedW {t3:5.1f} ({MB / t3:.2f}) err {e3:.1e}"
              f" | conv blockedW {t4:5.1f} ({MB / t4:.2f}) | dense blockedW WARM {t5:5.1f}", flush=True)
for ns in (3, 6):
    t, e = lin(64, 11520, 1280, 100 + ns, 1)
    tb, eb = lin(64, 11520, 1280, 100 + ns, 1, blocked=True)
    print(f"M=64 N=11520 K=1280 S=1 ns={ns} (same bytes, whole rows per work item): {t:5.1f} us ({MB / t:.2f} TB/s) err {e:.1e} | blockedW {tb:5.1f} ({MB / tb:.2f}) err {eb:.1e}", flush=True)
# M = 256 levels
MB2 = 1280 * 23040 * 2 / 1e6
for tile in (0, 2):
    for sk in (3, 6, 12):
        t1 = conv(16, 2560, 1280, 100 + 10 * tile + 4, sk)
        t4 = conv(16, 2560, 1280, 100 + 10 * tile + 4, sk, blocked=True)
        print(f"M=256 N=1280 K=23040 tile{tile} S={sk:2d}: conv {t1:5.1f} us ({MB2 / t1:.2f} TB/s) | blockedW {t4:5.1f} ({MB2 / t4:.2f})", flush=True)
