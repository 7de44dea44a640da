#!/usr/bin/env python3
"""Phase stamps of the ping-pong kernel on 3x3 convolutions (probe build, MVD_GEMM_DEBUG=32): median work / wait cycles per
phase over the steady-state slabs of a tile (see tools/probe_pp_stamps.py)."""
import ctypes as C, math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import _lib as L
from mvd_amd.packing import _conv_w

def rnd(*s, scale=1.0): return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)
def p(t): return C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

for (hw, cin, cout) in [(64, 320, 320), (32, 640, 640)]:
    x = rnd(32, hw, hw, cin)
    w = _conv_w(torch.randn(cout, cin, 3, 3) / math.sqrt(9 * cin)).to(torch.bfloat16).cuda()
    b = torch.randn(cout, device="cuda")
    out = torch.empty(32, hw, hw, cout, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(64 * 2 * 512, device="cuda", dtype=torch.int64)
    for _ in range(3):
        L.call("mvd_op_conv3x3", p(x), 32, hw, hw, cin, 1, 0, 0, p(w), p(b), None, 0, None, None, None, 0, 0, p(out), cout, 7, 1,
               C.c_void_p(ws.data_ptr()), st)
    torch.cuda.synchronize()
    s = ws.view(64, 2, 512).cpu()
    nslab = 9 * cin // 64
    print(f"== conv {hw}^2 {cin}->{cout}: {nslab} slabs per tile")
    for g in (0, 1):
        ph = {0: [], 1: [], 2: [], 3: []}; wt = {0: [], 1: [], 2: [], 3: []}
        for wg in range(64):
            cnt = int(s[wg, g, 0])
            v = s[wg, g, 1:1 + cnt].tolist()
            pairs = [(v[i], v[i + 1]) for i in range(0, cnt - 1, 2)][(1 if g == 0 else 2):]
            # first tile only has no E phase: use phases 8 .. 4*nslab-8 of it (steady state)
            for i in range(8, min(len(pairs), 4 * nslab - 8)):
                ph[i % 4].append(pairs[i][0] - pairs[i - 1][1]); wt[i % 4].append(pairs[i][1] - pairs[i][0])
        names = ["R0", "M0", "R1", "M1"]
        print(f"  group {g}: " + "  ".join(f"{names[k]} work {statistics.median(ph[k]):.0f} + wait {statistics.median(wt[k]):.0f}" for k in range(4))
              + f"   slab total {sum(statistics.median(ph[k]) + statistics.median(wt[k]) for k in range(4)):.0f} cycles")
