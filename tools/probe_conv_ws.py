#!/usr/bin/env python3
"""Weight-streaming convolution (conv_ws.hip) against the implicit-GEMM launch the engine made for the same problem, on the
batch-1 shapes of the 8x8 / 16x16 levels, with the operand state of a real forward: weights COLD (a 600 MB read in front),
activations warm.  HIP events around the launch only, the empty bracket subtracted (as tools/tune_sm.py).
    python tools/probe_conv_ws.py > gpurun_out/probe_conv_ws.log"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from mvd_amd import ops, _lib as L
from mvd_amd.packing import pack_ws, _conv_w

dev = "cuda"
flush = torch.empty(150 * 1024 * 1024, device=dev, dtype=torch.float32).normal_()
ITERS = int(os.environ.get("ITERS", "7"))


def bracket(fn, warm):
    ts = []
    for _ in range(ITERS + 1):
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
print(f"# empty bracket {EMPTY:.2f} us (subtracted); median of {ITERS}; cold weights, warm activations")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)   # noqa: E731
tot_old = tot_new = 0.0
for name, count, hw, cin, sc, n in (("L3 conv 1280->1280", 9, 8, 1280, 0, 1280), ("L3 conv1 2560->1280", 3, 8, 2560, 0, 1280), ("L3 conv2+sc 1280 (+2560)", 3, 8, 1280, 2560, 1280),
                                    ("L2 conv1 640->1280", 1, 16, 640, 0, 1280), ("L2 conv 1280->1280", 3, 16, 1280, 0, 1280), ("L2 conv2+sc 1280 (+640)", 1, 16, 1280, 640, 1280),
                                    ("L2 conv1 2560->1280", 2, 16, 2560, 0, 1280), ("L2 conv2+sc 1280 (+2560)", 2, 16, 1280, 2560, 1280),
                                    ("L2 conv1 1920->1280", 1, 16, 1920, 0, 1280), ("L2 conv2+sc 1280 (+1920)", 1, 16, 1280, 1920, 1280),
                                    ("L1 conv 640->640", 2, 32, 640, 0, 640), ("L1 conv1 1280->640", 1, 32, 1280, 0, 640), ("L1 conv1 1920->640", 1, 32, 1920, 0, 640),
                                    ("L3 up 8->16 1280", 1, -8, 1280, 0, 1280), ("L2 up 16->32 1280", 1, -16, 1280, 0, 1280)):
    ups = hw < 0          # (negative size: the INPUT map of an upsampling convolution)
    hw = abs(hw)
    x = rnd(1, hw, hw, cin)
    w4 = (torch.randn(n, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(torch.bfloat16)
    wsc = (torch.randn(n, sc, generator=g) / math.sqrt(sc)).to(torch.bfloat16) if sc else None
    bias = torch.randn(n, generator=g).to(dev)
    rowvec = torch.randn(1, n, generator=g).to(dev)
    s0 = rnd(1, hw, hw, sc - 1280 if sc > 1280 else sc) if sc else None
    s1 = rnd(1, hw, hw, 1280) if sc > 1280 else None
    wp = pack_ws(w4, wsc).to(dev)
    wold = _conv_w(w4)
    if sc:
        wold = torch.cat([wold, wsc.float()], 1)
    wold = wold.to(torch.bfloat16).to(dev)
    M, K = hw * hw * (4 if ups else 1), 9 * cin + sc
    sk = ops.engine_splitk(M, n, K)
    warm = [x] + [t for t in (s0, s1) if t is not None]
    if ups:
        rowvec = None
    old = lambda: ops.conv3x3(x, wold, bias, rowvec=rowvec, shortcut=s0, shortcut2=s1, splitk=sk, upsample=ups)   # noqa: E731
    b = old().float()
    t_old = bracket(old, warm) - EMPTY
    mb = wp.numel() * 2 / 1e6
    res = {}
    for v in (0, 1, 2):
        new = lambda: ops.conv3x3_ws(x, wp, bias, n, rowvec=rowvec, shortcut=s0, shortcut2=s1, variant=v, upsample=ups)   # noqa: E731
        try:
            a = new().float()
        except L.MvdError:
            continue
        err = (a - b).abs().max().item() / b.abs().max().item()
        res[v] = (bracket(new, warm) - EMPTY, err)
    t_new = res[0][0]
    tot_old += count * t_old; tot_new += count * t_new
    print(f"{name:28s} x{count} M={M:4d} K={K:6d} weights {mb:5.1f} MB | implicit GEMM (split {sk:2d}) {t_old:6.1f} us {mb / t_old:5.2f} TB/s | "
          f"conv_ws {t_new:6.1f} us {mb / t_new:5.2f} TB/s | " + "  ".join(f"v{v}: {t:5.1f} us (rel diff {e:.0e})" for v, (t, e) in res.items() if v), flush=True)
print(f"# per batch-1 forward: implicit GEMM {tot_old / 1e3:.3f} ms, conv_ws {tot_new / 1e3:.3f} ms   (variants: 1 = 64-pixel blocks, 2 = 128-pixel blocks; conv_ws column = the launcher's choice)")
