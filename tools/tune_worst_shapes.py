#!/usr/bin/env python3
"""Sweep (tile config x split-K) through the operator entry points for the GEMM / convolution shapes of a cfg4 forward that sit
furthest below their bounds (profiles/r04_shapes_cfg4.txt: the 8x8-level convolutions at 890-1000 TFLOP/s, the N = K = 1280
projections of the 16x16 level at 705, N = K = 640 of the 32x32 level at 650), each beside what the engine's heuristic picks.
Measurement only: prints a table, changes nothing.  Run on the GPU box: python tools/tune_worst_shapes.py > gpurun_out/...log"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mvd_amd import _lib as L  # noqa: E402
from mvd_amd import ops  # noqa: E402

B = 32
NCFG = L.lib().mvd_gemm_num_configs()
SPLITS = (1, 2, 3, 4, 6, 8)
T0 = time.time()
BUDGET_S = float(os.environ.get("TUNE_BUDGET_S", "240"))


def bench(fn, n=12):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


def rnd(*s, scale=0.5):
    return (torch.randn(*s, device="cuda") * scale).to(torch.bfloat16)


def sweep(name, flops, make_fn, m, n, k):
    sk = ops.engine_splitk(m, n, k)
    t_h = bench(make_fn(-1, sk))
    plan = ops.last_gemm_plan()
    rows = []
    for cfg in range(NCFG):
        for s in SPLITS:
            if time.time() - T0 > BUDGET_S:
                break
            try:
                t = bench(make_fn(cfg, s))
                p = ops.last_gemm_plan()
                rows.append((t, cfg, s, p["tiles"], p["grid"]))
            except L.MvdError:
                continue
    rows.sort()
    best = " | ".join(f"cfg{c} split{s}: {t:6.1f} us ({flops / t / 1e6:5.0f} TF, {tl} tiles / {g} wg)" for t, c, s, tl, g in rows[:4])
    gain = (t_h / rows[0][0] - 1.0) * 100 if rows else 0.0
    print(f"{name:38s} heuristic cfg{plan['cfg']} split{plan['splitk']}: {t_h:6.1f} us ({flops / t_h / 1e6:5.0f} TF)  best +{gain:4.1f} %  || {best}", flush=True)


def conv_case(hw, cin, cout, sc=0):
    x = rnd(B, hw, hw, cin)
    k = 9 * cin + sc
    w = rnd(cout, k, scale=1.0 / math.sqrt(k))
    bias = torch.randn(cout, device="cuda")
    temb = torch.randn(B, cout, device="cuda")
    res = rnd(B, hw, hw, cout) if not sc else None
    scx = rnd(B, hw, hw, sc) if sc else None
    m = B * hw * hw
    mk = lambda cfg, s: (lambda: ops.conv3x3(x, w, bias, rowvec=temb, res=res, shortcut=scx, force_cfg=cfg, splitk=s))   # noqa: E731
    sweep(f"conv {hw}x{hw} {cin}->{cout}" + (f" +sc{sc}" if sc else ""), 2.0 * m * cout * k, mk, m, cout, k)


def lin_case(m, n, k, res=True):
    a = rnd(m, k)
    w = rnd(n, k, scale=1.0 / math.sqrt(k))
    bias = torch.randn(n, device="cuda")
    r = rnd(m, n) if res else None
    mk = lambda cfg, s: (lambda: ops.linear(a, w, bias, res=r, force_cfg=cfg, splitk=s))   # noqa: E731
    sweep(f"linear M={m} N={n} K={k}" + (" +res" if res else ""), 2.0 * m * n * k, mk, m, n, k)


if __name__ == "__main__":
    print(f"# {NCFG} tile configs x split-K {SPLITS}, B = {B}; time of the whole operator call (split-K: incl. its reduce pass)", flush=True)
    conv_case(8, 1280, 1280)                 # K = 11520: 18 launches / step
    conv_case(8, 2560, 1280)                 # K = 23040: 6 / step
    conv_case(8, 1280, 1280, sc=2560)        # K = 14080: 6 / step
    lin_case(B * 256, 1280, 1280)            # 16x16 level out-projections: 35 / step at 705 TF
    lin_case(B * 256, 1280, 2560)            # adapter out-projection (K = 2 C): 10 / step
    lin_case(B * 1024, 640, 640)             # 32x32 level: 30 / step at 648 TF
    lin_case(B * 1024, 640, 1280)            # 10 / step
    lin_case(B * 64, 1280, 1280)             # 8x8 level (small-M kernel): 7 / step at 262 TF
    print(f"# done in {time.time() - T0:.0f} s", flush=True)
