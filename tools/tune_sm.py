#!/usr/bin/env python3
"""Sweep of the small-M kernels (gemm_sm.hip: tile x ring depth x split-K) over every GEMM / conv shape of a batch-B forward
(TUNE_B, default 1), against the launch the engine's current heuristic makes.  Each timed launch runs with the operand state
of a real forward: weights COLD (a 600 MB read in front evicts the Infinity Cache: a forward streams 1.7 GB of weights), the
activation operand warm (re-read just before).  HIP events around the launch(es) only; the empty bracket is subtracted.

    python tools/tune_sm.py [--quick] [--filter L2] > gpurun_out/tune_sm.log
"""
import argparse, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from mvd_amd import ops, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--filter", default="")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--adapter", action="store_true", help="cfg3 shapes (adapter rows / columns) instead of the base UNet's")
ap.add_argument("--warm-weights", action="store_true", help="re-read the WEIGHT operand too just before the launch (what a prefetcher running ahead of the launch chain would give)")
ap.add_argument("--auto-only", action="store_true", help="time the heuristic's launch only, no sweep")
args = ap.parse_args()
B = int(os.environ.get("TUNE_B", "1"))
SM_TILES = {0: (64, 64), 1: (128, 64), 2: (64, 128), 3: (128, 128), 4: (64, 160), 5: (128, 160), 6: (64, 320)}
dev = "cuda"
flush = torch.empty(150 * 1024 * 1024, device=dev, dtype=torch.float32).normal_()


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def bracket(fn, warm, iters):
    """GPU time of fn() alone: [flush][warm operands][e0] fn [e1]"""
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


EMPTY = bracket(lambda: None, [], 20)


def unet_shapes(adapter):
    """(kind, name, params, count per forward) of a batch-B forward at a 64x64 latent"""
    ch, out = [320, 640, 1280, 1280], {}

    def add(kind, name, **p):
        key = (kind, tuple(sorted(p.items())))
        if key in out:
            out[key][3] += 1
        else:
            out[key] = [kind, name, p, 1]

    def resnet(hw, cin, cout, lvl):
        add("conv", f"L{lvl} conv1 {cin}->{cout}", hw=hw, cin=cin, cout=cout, sc=0, stride=1, ups=0, rowvec=1, res=0)
        if cin != cout:
            add("conv", f"L{lvl} conv2+sc {cout}+{cin}->{cout}", hw=hw, cin=cout, cout=cout, sc=cin, stride=1, ups=0, rowvec=0, res=0)
        else:
            add("conv", f"L{lvl} conv2 {cout}->{cout}", hw=hw, cin=cout, cout=cout, sc=0, stride=1, ups=0, rowvec=0, res=1)

    def transformer(hw, c, lvl):
        M = B * hw * hw
        nq = 4 if adapter else 3
        add("lin", f"L{lvl} proj/q/out C->C", M=M, K=c, N=c, res=1, geglu=0)
        add("lin", f"L{lvl} proj/q/out C->C", M=M, K=c, N=c, res=1, geglu=0)      # proj_in, proj_out (res), attn2.q: counted below
        add("lin", f"L{lvl} qkv", M=M, K=c, N=nq * c, res=0, geglu=0)
        add("lin", f"L{lvl} attn out", M=M, K=(2 if adapter else 1) * c, N=c, res=1, geglu=0)
        add("lin", f"L{lvl} attn out", M=M, K=(2 if adapter else 1) * c, N=c, res=1, geglu=0)
        add("lin", f"L{lvl} q2", M=M, K=c, N=(2 if adapter else 1) * c, res=0, geglu=0)
        add("lin", f"L{lvl} ff1 geglu", M=M, K=c, N=8 * c, res=0, geglu=1)
        add("lin", f"L{lvl} ff2", M=M, K=4 * c, N=c, res=1, geglu=0)

    hw, prev = 64, 320
    skips = [320]
    for i in range(4):
        co = ch[i]
        for j in range(2):
            resnet(hw, prev if j == 0 else co, co, i)
            if i < 3:
                transformer(hw, co, i)
            skips.append(co)
        prev = co
        if i < 3:
            add("conv", f"L{i} down {co}", hw=hw, cin=co, cout=co, sc=0, stride=2, ups=0, rowvec=0, res=0)
            hw //= 2
            skips.append(co)
    resnet(hw, 1280, 1280, 3); transformer(hw, 1280, 3); resnet(hw, 1280, 1280, 3)
    prev_out = 1280
    for i in range(4):
        co = ch[3 - i]
        for j in range(3):
            sk = skips.pop()
            resnet(hw, (prev_out if j == 0 else co) + sk, co, 3 - i)
            if i > 0:
                transformer(hw, co, 3 - i)
        prev_out = co
        if i < 3:
            add("conv", f"L{3 - i} up {co}", hw=hw, cin=co, cout=co, sc=0, stride=1, ups=1, rowvec=0, res=0)
            hw *= 2
    add("lin", "temb_proj f32", M=B, K=1280, N=20160, res=0, geglu=0, f32=1)
    add("lin", "text_kv", M=B * 77, K=1024, N=24960, res=0, geglu=0)
    add("lin", "conv_in K64", M=B * 4096, K=64, N=320, res=0, geglu=0)
    return list(out.values())


def make(kind, p):
    """returns (run(force_cfg, splitk) -> None, warm tensors, flops, N, K)"""
    if kind == "lin":
        M, K, N = p["M"], p["K"], p["N"]
        a, w = rnd(M, K), rnd(N, K)
        bias = torch.randn(N, device=dev)
        res = rnd(M, N // 2 if p["geglu"] else N) if p.get("res") else None
        f32 = bool(p.get("f32"))
        out = torch.empty(M, N // 2 if p["geglu"] else N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        ws = torch.empty(32 * M * N + 4096, device=dev, dtype=torch.float32)

        def run(cfg, sk):
            L.call("mvd_op_linear", C.c_void_p(a.data_ptr()), None, K, 0, C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), None, 0, 0,
                   C.c_void_p(res.data_ptr()) if res is not None else None, 1.0, int(p["geglu"]), C.c_void_p(out.data_ptr()), int(f32), M, N,
                   cfg, sk, C.c_void_p(ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return run, [a] + ([res] if res is not None else []) + ([w] if args.warm_weights else []), 2.0 * M * N * K, N, K, M
    hw, cin, cout, sc = p["hw"], p["cin"], p["cout"], p["sc"]
    oh = hw * 2 if p["ups"] else (hw // 2 if p["stride"] == 2 else hw)
    M, K = B * oh * oh, 9 * cin + sc
    x = rnd(B, hw, hw, cin)
    w = rnd(cout, K)
    bias = torch.randn(cout, device=dev)
    rowvec = torch.randn(B, cout, device=dev) if p["rowvec"] else None
    res = rnd(M, cout) if p["res"] else None
    scx = rnd(M, sc) if sc else None
    out = torch.empty(M, cout, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(32 * M * cout + 4096, device=dev, dtype=torch.float32)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731

    def run(cfg, sk):
        L.call("mvd_op_conv3x3", ptr(x), B, hw, hw, cin, p["stride"], p["ups"], 0, ptr(w), ptr(bias), ptr(rowvec), cout if rowvec is not None else 0,
               ptr(res), ptr(scx), None, sc, 0, ptr(out), cout, cfg, sk, ptr(ws), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    return run, [x] + [t for t in (res, scx) if t is not None] + ([w] if args.warm_weights else []), 2.0 * M * cout * K, cout, K, M


print(f"# B={B} adapter={args.adapter} empty bracket {EMPTY:.2f} us (subtracted); times in us, median of {args.iters}", flush=True)
total_auto = total_best = 0.0
for kind, name, p, count in unet_shapes(args.adapter):
    if args.filter and args.filter not in name:
        continue
    run, warm, fl, N, K, M = make(kind, p)
    geglu = int(p.get("geglu", 0))
    S_auto = L.lib().mvd_debug_pick_splitk(M, N, K, geglu)
    auto = bracket(lambda: run(-1, S_auto), warm, args.iters) - EMPTY
    res = []
    nkt = K // 64
    for tile, (bm, bn) in SM_TILES.items():
        if args.auto_only:
            break
        if N % bn or (geglu and (bn // 32) % 2):
            continue
        tiles = ((M + bm - 1) // bm) * (N // bn)
        for sk in ([1] if geglu else [1, 2, 3, 4, 6, 8, 12, 16, 24, 32]):
            if sk > 1 and (sk > nkt // 2 or tiles * sk > 520 or tiles >= 256):
                continue
            if tiles * sk > 4096 or (sk > 1 and tiles > 4096):
                continue
            for ns in ([3, 6] if args.quick else [2, 3, 4, 6]):
                if ns > 2 and ns - 1 > math.ceil(nkt / sk) + 1:
                    continue
                if ns * (bm + bn) * 128 > 160 * 1024:
                    continue
                try:
                    us = bracket(lambda: run(100 + 10 * tile + ns, sk), warm, args.iters) - EMPTY
                    res.append((us, tile, ns, sk))
                except L.MvdError:
                    pass
    res.sort()
    best = res[0][0] if res else float("nan")
    total_auto += count * auto
    total_best += count * min(best, auto)
    print(f"{name:30s} x{count:2d} M={M:5d} N={N:5d} K={K:5d} | auto(S={S_auto}) {auto:6.1f} ({fl / auto / 1e6:5.0f} TF) | best "
          + "  ".join(f"t{t}/n{n}/s{s}:{u:.1f}" for u, t, n, s in res[:6]), flush=True)
print(f"# sum over a forward: auto {total_auto / 1e3:.3f} ms, best-of {total_best / 1e3:.3f} ms")
