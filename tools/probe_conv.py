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
zer = lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16)
B = 32
for name, hw, c in [("L1 conv 640", 32, 640), ("L0 conv 320", 64, 320)]:
    for data in ("rand",):
        mk = rnd if data == "rand" else zer
        x, w = mk(B, hw, hw, c), mk(c, 9 * c)
        for cfg in (10, 6):
            ms = time_fn(lambda: ops.conv3x3(x, w, force_cfg=cfg))
            fl = 2.0 * B * hw * hw * c * 9 * c
            print(f"dbg={os.environ.get('MVD_GEMM_DEBUG','0')} {data} {name} cfg{cfg}: {ms*1e3:8.1f} us {fl/ms/1e9:7.0f} TF", flush=True)
