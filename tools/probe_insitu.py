#!/usr/bin/env python3
"""Why does a small GEMM take ~2x longer inside a forward than in an isolated bracket?  Same launch, different states in front
of it: A warm-read (tune_sm.py's state) / A just WRITTEN by another kernel (dirty lines in other XCDs' L2s) / a train of other
kernels in front (instruction caches cold) / both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from mvd_amd import ops, _lib as L
dev = "cuda"
flush = torch.empty(150 * 1024 * 1024, device=dev)
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)   # noqa: E731
small = [torch.randn(4096, device=dev) for _ in range(4)]


def other_kernels():
    # a dozen different small kernels (different code objects): evicts the instruction caches, costs ~100 us of GPU time
    x = small[0]
    x = torch.sin(x); x = torch.exp(-x.abs()); x = torch.tanh(x); x = torch.sigmoid(x); x = torch.sqrt(x.abs() + 1)
    x = torch.cumsum(x, 0); x = torch.sort(x)[0]; x = torch.flip(x, [0]); x = torch.softmax(x, 0); x = torch.log1p(x)
    x = torch.erf(x); x = torch.floor(x * 7)
    return x


def bracket(fn, prep, iters=11):
    ts = []
    for _ in range(iters + 1):
        flush.sum()
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[1:])
    return v[len(v) // 2]


EMPTY = bracket(lambda: None, lambda: None)
print(f"empty bracket {EMPTY:.2f} us")
for (M, N, K, cfg) in [(4096, 320, 320, 103), (1024, 640, 640, 104), (256, 1280, 1280, 104), (64, 1280, 1280, 104), (4096, 320, 1280, 103)]:
    a, a_src, w = rnd(M, K), rnd(M, K), rnd(N, K)
    bias = torch.randn(N, device=dev)
    res = rnd(M, N)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: L.call("mvd_op_linear", C.c_void_p(a.data_ptr()), None, K, 0, C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, 0, 0,
                        C.c_void_p(res.data_ptr()), 1.0, 0, C.c_void_p(out.data_ptr()), 0, M, N, cfg, 1, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))   # noqa: E731
    t_warm = bracket(fn, lambda: (a.view(torch.int16).max(), res.view(torch.int16).max())) - EMPTY
    t_dirty = bracket(fn, lambda: (a.copy_(a_src), res.view(torch.int16).max())) - EMPTY
    t_ic = bracket(fn, lambda: (a.view(torch.int16).max(), res.view(torch.int16).max(), other_kernels())) - EMPTY
    t_both = bracket(fn, lambda: (other_kernels(), res.view(torch.int16).max(), a.copy_(a_src))) - EMPTY
    t_back = bracket(lambda: (fn(), fn(), fn(), fn()), lambda: (a.view(torch.int16).max(), res.view(torch.int16).max())) - EMPTY
    print(f"M={M} N={N} K={K}: A warm-read {t_warm:5.1f} | A just written {t_dirty:5.1f} | other kernels in front {t_ic:5.1f} | both {t_both:5.1f} | 4 back to back {t_back / 4:5.1f} each  us", flush=True)
