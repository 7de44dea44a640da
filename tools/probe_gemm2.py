import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops
def time_fn(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
rnd = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
for name, M, K, N in [("L0 sq", 131072, 320, 320), ("L1 sq", 32768, 640, 640), ("L2 sq", 8192, 1280, 1280)]:
    a, w, r = rnd(M, K), rnd(N, K), rnd(M, N)
    b = torch.randn(N, device="cuda")
    for cfg in (7, 10, 12, 13):
        for res in (None, r):
            try:
                ms = time_fn(lambda: ops.linear(a, w, b, res=res, force_cfg=cfg))
            except Exception as e:
                continue
            byt = (M * K + M * N * (2 if res is not None else 1)) * 2
            print(f"dbg={os.environ.get('MVD_GEMM_DEBUG','0')} {name} cfg{cfg} res={res is not None}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF {byt/ms/1e6:6.0f} GB/s", flush=True)
