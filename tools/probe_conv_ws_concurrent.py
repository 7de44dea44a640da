#!/usr/bin/env python3
"""conv_ws launches on one stream while another stream keeps the chip busy: is every launch bit-identical to the first?
(The two-stream forward -- encoder pass beside the main pass -- is where a timing-dependent wait shows.)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
from mvd_amd.packing import pack_ws

dev = "cuda"
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)   # noqa: E731
REPS = int(os.environ.get("REPS", "200"))
side = torch.cuda.Stream()
big_a, big_b = rnd(8192, 8192), rnd(8192, 8192)
flush = torch.empty(64 * 1024 * 1024, device=dev, dtype=torch.float32).normal_()
for name, hw, cin, n, s0, s1 in (("32x32 640->640", 32, 640, 640, 0, 0), ("32x32 640->640 + sc 640", 32, 640, 640, 640, 0),
                                 ("32x32 640->640 + sc 1280|640", 32, 640, 640, 1280, 640), ("32x32 640->640 + sc 640|640", 32, 640, 640, 640, 640),
                                 ("16x16 1280->1280 + sc 1280|1280", 16, 1280, 1280, 1280, 1280), ("8x8 1280->1280 + sc 1280|1280", 8, 1280, 1280, 1280, 1280),
                                 ("16x16 1280->1280", 16, 1280, 1280, 0, 0),
                                 # round 5: the 12- / 24-wide forms (96 x 96 latents)
                                 ("24x24 1280->1280", 24, 1280, 1280, 0, 0), ("24x24 1280->1280 + sc 1280|1280", 24, 1280, 1280, 1280, 1280),
                                 ("12x12 1280->1280", 12, 1280, 1280, 0, 0), ("12x12 1280->1280 + sc 1280|1280", 12, 1280, 1280, 1280, 1280))[int(os.environ.get("FROM", "0")):]:
    x = rnd(1, hw, hw, cin)
    w4 = (torch.randn(n, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).to(torch.bfloat16)
    sc = s0 + s1
    wsc = (torch.randn(n, sc, generator=g) / math.sqrt(sc)).to(torch.bfloat16) if sc else None
    bias = torch.randn(n, generator=g).to(dev)
    a0 = rnd(1, hw, hw, s0) if s0 else None
    a1 = rnd(1, hw, hw, s1) if s1 else None
    wp = pack_ws(w4, wsc).to(dev)
    run = lambda: ops.conv3x3_ws(x, wp, bias, n, shortcut=a0, shortcut2=a1)   # noqa: E731
    first = run().clone()
    torch.cuda.synchronize()
    # a second conv_ws problem of its own for the other stream
    x2 = rnd(1, hw, hw, cin); wp2 = wp.clone(); a02 = rnd(1, hw, hw, s0) if s0 else None; a12 = rnd(1, hw, hw, s1) if s1 else None
    run2 = lambda: ops.conv3x3_ws(x2, wp2, bias, n, shortcut=a02, shortcut2=a12)   # noqa: E731
    for mode in ("alone", "beside a GEMM stream", "beside a GEMM stream, cold caches", "beside conv_ws launches of another stream",
                 "alternating with a second problem (stale LDS differs), beside a GEMM stream, cold caches"):
        bad = worst = 0
        for i in range(REPS):
            if mode.startswith("beside conv_ws"):
                with torch.cuda.stream(side):
                    run2(); run2(); run2()
            elif mode != "alone":
                with torch.cuda.stream(side):
                    big_a @ big_b
            if mode.endswith("cold caches"):
                flush.sum()
            if mode.startswith("alternating"):
                run2()
            out = run()
            if not torch.equal(out, first):
                bad += 1
                worst = max(worst, int((out != first).sum()))
        torch.cuda.synchronize()
        print(f"{'FAIL' if bad else 'ok  '} {name:34s} {mode:36s}: {bad}/{REPS} launches differ (most differing elements: {worst})", flush=True)
