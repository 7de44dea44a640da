#!/usr/bin/env python3
"""Run every batch-1 operator form at its real shape many times with other work interleaved; report any launch whose bits
differ from the first (races show up here long before they show up in a parity tolerance)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from mvd_amd import ops, _lib as L
from mvd_amd.packing import fold_layernorm
dev = "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)   # noqa: E731
junk = torch.empty(64 * 1024 * 1024, device=dev)
REPS = int(os.environ.get("REPS", "30"))


def check(name, fn):
    try:
        first = fn().clone()
    except L.MvdError as e:
        print(f"skip {name}: {str(e)[-90:]}")
        return
    bad = 0
    for i in range(REPS):
        if i % 3 == 0:
            junk.normal_()                       # other traffic / evictions in between
        out = fn()
        if not torch.equal(out, first):
            bad += 1
    nan = int(torch.isnan(first.float()).sum())
    print(f"{'FAIL' if bad or nan else 'ok  '} {name}: {bad}/{REPS} launches differ, {nan} NaNs", flush=True)


g, b = torch.ones(2560, device=dev), torch.zeros(2560, device=dev)
for (B, hw, c) in [(1, 4096, 320), (1, 4096, 960), (1, 1024, 640), (1, 1024, 1920), (1, 256, 1280), (1, 256, 2560), (2, 4096, 320), (1, 9216, 320)]:
    x = rnd(B, hw, c)
    check(f"groupnorm {B}x{hw}x{c}", lambda: ops.groupnorm(x, g[:c].contiguous(), b[:c].contiguous(), silu=True))

lib = L.lib()
for (M, N, K, conv) in [(1024, 640, 5760, 32), (256, 1280, 11520, 16), (64, 1280, 11520, 8), (64, 1280, 23040, 8), (256, 1280, 5120, 0),
                        (1024, 320, 2880, 64), (4096, 320, 1280, 0), (256, 1280, 1280, 0), (1024, 640, 640, 0), (4096, 960, 320, 0)]:
    # what the engine's planner would launch
    import ctypes
    if conv:
        cin = K // 9
        x, w = rnd(1, conv, conv, cin), rnd(N, K)
        bias = torch.randn(N, device=dev)
        res = rnd(1, conv, conv, N) if conv != 64 else None
        for cfg, sk in ((100 + 0 * 10 + 4, 12), (100 + 2 * 10 + 4, 6), (100 + 4 * 10 + 4, 4), (100 + 4, 3)):
            if N % (64, 64, 128, 128, 160, 160, 320)[(cfg - 100) // 10]:
                continue
            if ((M + 63) // 64) * (N // 64) * sk > 256:
                continue
            check(f"conv M={M} N={N} K={K} cfg={cfg} S={sk}", lambda: ops.conv3x3(x, w, bias, res=res, force_cfg=cfg, splitk=sk))
    else:
        a, w = rnd(M, K), rnd(N, K)
        bias = torch.randn(N, device=dev)
        res = rnd(M, N)
        for cfg, sk in ((104, 1), (103, 1), (104, 3), (114, 1), (124, 1)):
            if sk > 1 and ((M + 63) // 64) * (N // 64) * sk > 256:
                continue
            if N % (64, 64, 128)[(cfg - 100) // 10]:
                continue
            check(f"linear M={M} N={N} K={K} cfg={cfg} S={sk}", lambda: ops.linear(a, w, bias, res=res, force_cfg=cfg, splitk=sk))

for (m, c, nmul, geglu) in [(4096, 320, 3, False), (4096, 320, 8, True), (1024, 640, 3, False), (1024, 640, 8, True), (256, 1280, 3, False),
                            (256, 1280, 8, True), (64, 1280, 8, True), (64, 1280, 1, False)]:
    n = nmul * c
    x = rnd(m, c)
    w = torch.randn(n, c) / math.sqrt(c)
    wf, cf = fold_layernorm(w, torch.ones(c), torch.zeros(c), torch.zeros(n), dev)
    check(f"ln_linear {m}x{c}->{n} geglu={geglu}", lambda: ops.ln_linear(x, wf, cf, geglu=geglu))
