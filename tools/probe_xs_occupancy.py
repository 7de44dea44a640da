import math, os, sys
sys.path.insert(0, os.getcwd())
import torch
from mvd_amd import ops
from mvd_amd.packing import pack_xs
K=320
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n in (320, 1280):
    w = torch.randn(n, K, device="cuda") / math.sqrt(K)
    wp = pack_xs(w, torch.randn(n, device="cuda"))
    for M in (16384, 32768, 65536, 98304, 131072, 196608, 262144):
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        t = timeit(lambda: ops.linear_xs(x, wp, csplit=1))
        print(f"N={n} M={M} ({M//256} workgroups): {t:.1f} us", flush=True)
