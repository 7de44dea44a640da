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
rnd = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
for name, M, K, N in [("L0 qkvq", 131072, 320, 1280), ("L0 out", 131072, 640, 320), ("L0 ff2", 131072, 1280, 320), ("L1 qkvq", 32768, 640, 2560)]:
    a, w = rnd(M, K), rnd(N, K)
    for cfg in (10, 8):
        ms = time_fn(lambda: ops.linear(a, w, force_cfg=cfg))
        print(f"dbg={os.environ.get('MVD_GEMM_DEBUG','0')} {name} cfg{cfg}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.0f} TF  {(M*K+M*N)*2/ms/1e6:7.0f} GB/s", flush=True)
