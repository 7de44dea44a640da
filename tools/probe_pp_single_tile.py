#!/usr/bin/env python3
"""Timeline of the ping-pong GEMM on shapes with ONE tile per workgroup (the 32x32 level's N = 640 projections): kernel time by
events, first-to-last stamp and the work time of every phase (probe build, MVD_GEMM_DEBUG=32; see tools/probe_pp_stamps.py).
Finding (round 2): the read phases of these HBM / MALL-fed dense shapes take 1000-1700 cycles against 640-1012 on the
L2-fed 3x3 convolutions of the same size -- 5.8 k cycles per slab instead of 3.8 k."""
import ctypes as C, math, os, statistics, sys
sys.path.insert(0, os.getcwd())
import torch
from mvd_amd import _lib as L
def rnd(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)
def p(t): return C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (m, n, k, res) in [(32768, 640, 640, 0), (32768, 640, 1280, 1)]:
    a, w, b = rnd(m, k), rnd(n, k, scale=1 / math.sqrt(k)), torch.randn(n, device="cuda")
    r = rnd(m, n) if res else None
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(64 * 2 * 512, device="cuda", dtype=torch.int64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(4):
        if it == 3: e0.record()
        L.call("mvd_op_linear", p(a), None, k, 0, p(w), p(b), None, 0, 0, p(r), 1.0, 0, p(out), 0, m, n, 7, 1, C.c_void_p(ws.data_ptr()), st)
        if it == 3: e1.record()
    torch.cuda.synchronize()
    s = ws.view(64, 2, 512).cpu()
    nslab = k // 64
    print(f"== M={m} N={n} K={k} res={res}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {nslab} slabs")
    for g in (0, 1):
        tot, first, lastph = [], [], []
        for wg in range(64):
            cnt = int(s[wg, g, 0]); v = s[wg, g, 1:1 + cnt].tolist()
            tot.append(v[-1] - v[0]); first.append(v[1] - v[0])
            pairs = [(v[i], v[i + 1]) for i in range(0, cnt - 1, 2)]
            lastph.append([pairs[i][0] - pairs[i - 1][1] for i in range(1, len(pairs))])
        med = [statistics.median(x[i] for x in lastph if len(x) > i) for i in range(min(len(x) for x in lastph))]
        print(f"  group {g}: first stamp -> last stamp {statistics.median(tot):.0f} cycles; phase work:", [int(x) for x in med])
