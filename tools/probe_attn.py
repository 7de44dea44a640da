import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
def time_fn(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
rnd = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
for name, B, h, n, nk in [("L0 self", 32, 5, 4096, 4096), ("L1 self", 32, 10, 1024, 1024), ("L2 self", 32, 20, 256, 256), ("L0 text", 32, 5, 4096, 77)]:
    q, k, v = rnd(B, n, h * 64), rnd(B, nk, h * 64), rnd(B, nk, h * 64)
    ms = time_fn(lambda: ops.attention(q, k, v, h, scale=float(os.environ.get('PROBE_SCALE', '0'))))
    print(f"pipe={os.environ.get('MVD_ATTN_PIPE','1')} NW={os.environ.get('MVD_ATTN_NW','auto')} {name}: {ms*1e3:8.1f} us {4.0*B*h*n*nk*64/ms/1e9:7.0f} TF", flush=True)
