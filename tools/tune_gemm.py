#!/usr/bin/env python3
"""GEMM / implicit-conv tile-config sweep on the shapes of the SD2.1+MVD forward (run on the GPU box)."""
import sys, os, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops

B = int(os.environ.get("TUNE_B", "32"))
TILES = {0: (256, 160), 1: (256, 128), 2: (128, 160), 3: (128, 128), 4: (128, 64), 5: (64, 64)}
TILES.update({k + 8: v for k, v in list(TILES.items())})
TILES[7] = (256, 320)
TILES[6] = (256, 320)
TILES[14] = (128, 320)
TILES[15] = (256, 320)   # ring-pipelined kernel
if os.environ.get('TUNE_CFGS'):
    TILES = {int(c): TILES[int(c)] for c in os.environ['TUNE_CFGS'].split(',')}

def time_fn(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def rnd(*s): return (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)

shapes = []
for lvl, (hw, c) in enumerate([(64, 320), (32, 640), (16, 1280), (8, 1280)]):
    M = B * hw * hw
    shapes.append(("conv", f"L{lvl} conv {c}->{c}", dict(hw=hw, cin=c, cout=c)))
    if lvl < 3:
        shapes.append(("lin", f"L{lvl} qkvq M={M} K={c} N={4*c}", dict(M=M, K=c, N=4 * c)))
        shapes.append(("lin", f"L{lvl} out  M={M} K={2*c} N={c}", dict(M=M, K=2 * c, N=c)))
        shapes.append(("lin", f"L{lvl} ff1g M={M} K={c} N={8*c}", dict(M=M, K=c, N=8 * c, geglu=True)))
        shapes.append(("lin", f"L{lvl} ff2  M={M} K={4*c} N={c}", dict(M=M, K=4 * c, N=c)))
        shapes.append(("lin", f"L{lvl} textkv M={B*77} K=1024 N={2*c}", dict(M=B * 77, K=1024, N=2 * c)))
shapes.append(("conv", "L0 up conv 960->320 (+shortcut)", dict(hw=64, cin=320, cout=320, sc=960)))
shapes.append(("conv", "L1 up conv 1920->640 conv1", dict(hw=32, cin=1920, cout=640)))
shapes.append(("conv", "L2 up conv 2560->1280 conv1", dict(hw=16, cin=2560, cout=1280)))
shapes.append(("lin", f"temb_proj M={B} K=1280 N=20160", dict(M=B, K=1280, N=20160)))

res = {}
for kind, name, p in shapes:
    row = {}
    for cfg, (bm, bn) in TILES.items():
        try:
            if kind == "lin":
                if p["N"] % bn or (p.get("geglu") and cfg in (0, 2, 7, 8, 10, 14, 15)): continue
                a, w = rnd(p["M"], p["K"]), rnd(p["N"], p["K"])
                fn = lambda: ops.linear(a, w, geglu=p.get("geglu", False), force_cfg=cfg)
                fl = 2.0 * p["M"] * p["N"] * p["K"]
            else:
                if p["cout"] % bn: continue
                x = rnd(B, p["hw"], p["hw"], p["cin"])
                k = 9 * p["cin"] + p.get("sc", 0)
                w = rnd(p["cout"], k)
                sc = rnd(B, p["hw"], p["hw"], p["sc"]) if "sc" in p else None
                fn = lambda: ops.conv3x3(x, w, shortcut=sc, force_cfg=cfg)
                fl = 2.0 * B * p["hw"] ** 2 * p["cout"] * k
            ms = time_fn(fn)
            row[cfg] = (round(ms * 1e3, 1), round(fl / ms / 1e9, 0))
        except Exception as e:
            row[cfg] = ("err", str(e)[:40])
    res[name] = row
    best = max((v[1], k) for k, v in row.items() if v[0] != "err")
    print(f"{name:45s} best cfg{best[1]} {best[0]:6.0f} TF | " + " ".join(f"c{k}:{v[1]}" for k, v in row.items()), flush=True)
json.dump(res, open(os.path.join("gpurun_out", f"tune_B{B}.json"), "w"))
