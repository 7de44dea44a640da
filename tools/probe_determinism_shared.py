#!/usr/bin/env python3
"""Bit-determinism of the B = 32 operator forms while ANOTHER PROCESS keeps the same GPU busy (two ranks on one GPU: the
bench rehearsal showed 2-5 differing output elements in some runs).  Start a load in the background first, e.g.
    python bench.py --steps 300 --no-check --no-cpu-baseline --no-profile > /dev/null &
"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mvd_amd import ops
dev = "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)   # noqa: E731
REPS = int(os.environ.get("REPS", "60"))


def check(name, fn):
    first = fn().clone()
    bad = 0
    worst = 0
    for i in range(REPS):
        out = fn()
        if not torch.equal(out, first):
            bad += 1
            worst = max(worst, int((out != first).sum()))
    print(f"{'FAIL' if bad else 'ok  '} {name}: {bad}/{REPS} launches differ (most differing elements in one launch: {worst})", flush=True)


B = 32
x = rnd(B, 64, 64, 320)
w4 = rnd(4, 9 * 320); b4 = torch.randn(4, device=dev)
check("conv_out 32x64x64x320", lambda: ops.conv_out(x, w4, b4))
g, b = torch.ones(320, device=dev), torch.zeros(320, device=dev)
xf = x.reshape(B, 4096, 320)
check("groupnorm+silu 32x4096x320 (slice kernel)", lambda: ops.groupnorm(xf, g, b, silu=True))
x2 = rnd(B, 4096, 640)
g2, b2 = torch.ones(960, device=dev), torch.zeros(960, device=dev)
check("groupnorm 32x4096x(320+640) two sources", lambda: ops.groupnorm(xf, g2, b2, silu=True, x2=x2))
wc = rnd(320, 9 * 320); bc = torch.randn(320, device=dev)
check("conv3x3 32x64x64 320->320 (ping-pong)", lambda: ops.conv3x3(x, wc, bc, res=x))
wsc = rnd(320, 9 * 320 + 640)
xs = rnd(B, 64, 64, 640)
check("conv3x3 + 1x1 shortcut (ping-pong, 2 segments)", lambda: ops.conv3x3(x, wsc, bc, shortcut=xs))
a = rnd(B * 4096, 320); wl = rnd(320, 320)
check("linear 131072x320x320 + residual (ping-pong dense)", lambda: ops.linear(a, wl, bc, res=a))
wq = rnd(960, 320)
check("linear 131072x960x320", lambda: ops.linear(a, wq))
qkv = rnd(B, 4096, 960)
check("attention 32x5x4096x4096 (engine form)", lambda: ops.attention(qkv[:, :, :320], qkv[:, :, 320:640], qkv[:, :, 640:], 5, scale=0.0))
sc, sh = torch.randn(B, 320, device=dev), torch.randn(B, 320, device=dev)
check("film 32x4096x320", lambda: ops.film(xf, sc, sh) if hasattr(ops, "film") else xf)

# ---- round 4: the in-kernel rendezvous forms with a second tenant on the GPU -- the small-M kernels' split-K
# combine (bounded wait + claims, gemm_sm.hip), the split-KV attention merge (ticket, last arriver), and the X-stationary
# kernels.  A batch-1 forward's shapes: the deep-level convolutions / projections split 4-16 ways.
from mvd_amd.packing import pack_xs
x8 = rnd(1, 8, 8, 1280)
w8 = rnd(1280, 9 * 1280); b8 = torch.randn(1280, device=dev)
for S in (4, 12, 16):
    check(f"gemm_sm conv3x3 1x8x8 1280->1280 split-K {S} (in-kernel rendezvous)", lambda S=S: ops.conv3x3(x8, w8, b8, force_cfg=100, splitk=S))
a1 = rnd(256, 5120); w1 = rnd(1280, 5120)
check("gemm_sm linear 256x1280x5120 split-K 8 + residual", lambda: ops.linear(a1, w1, b8, res=rnd(256, 1280) * 0 + 1, force_cfg=100, splitk=8))
q1 = rnd(1, 4096, 960)
for ns in (2, 3):
    check(f"attention 1x5x4096x4096 split-KV {ns} (in-kernel merge)", lambda ns=ns: ops.attention_split(q1[:, :, :320], q1[:, :, 320:640], q1[:, :, 640:], 5, ns))
wx = pack_xs(torch.randn(1280, 320, device=dev) / math.sqrt(320), torch.randn(1280, device=dev))
check("gemm_xs 131072x1280x320 (X-stationary, LayerNorm)", lambda: ops.linear_xs(a, wx, ln=True))
wg = pack_xs(torch.randn(2560, 320, device=dev) / math.sqrt(320), torch.randn(2560, device=dev), geglu=True)
check("gemm_xs GEGLU 131072x2560x320", lambda: ops.linear_xs(a, wg, geglu=True, ln=True))
