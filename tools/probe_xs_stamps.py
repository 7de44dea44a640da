#!/usr/bin/env python3
"""Phase stamps of the X-stationary kernels (build: tools/build_xs_variant.sh stamps -DXS_STAMPS; run with MVD_HIP_LIB=...):
cycles wave 0 of a workgroup spends waiting for its DMAs (counted vmcnt), at the barrier, issuing DMAs, multiplying, in the epilogue."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvd_amd import ops, _lib as L
import ctypes as C
from mvd_amd.packing import fold_layernorm, pack_xs
M, K = int(os.environ.get('XS_M', 131072)), 320
g, b = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
x = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
for (n, geglu, ln, cs) in [(320, False, False, 1), (1280, False, False, 1), (1280, False, False, 2), (1280, False, True, 1), (2560, True, True, 1), (2560, True, True, 2)]:
    w = (torch.randn(n, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(n, device="cuda")
    if ln:
        wf, cf = fold_layernorm(w, g, b, bias, "cuda"); wp = pack_xs(wf.float(), cf[1], geglu=geglu)
    else:
        wp = pack_xs(w, bias, geglu=geglu)
    no = n // 2 if geglu else n
    nwg = (M // 256) * cs
    out = torch.zeros(M + (nwg * 64 + no * 2 - 1) // (no * 2) + 1, no, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        L.call("mvd_op_linear_xs", C.c_void_p(x.data_ptr()), K, C.c_void_p(wp.data_ptr()), M, K, n // 32, int(geglu), int(ln), 1e-5,
               None, 0, C.c_void_p(out.data_ptr()), no, cs, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    st = out[M:].reshape(-1).view(torch.int64)[: nwg * 8].reshape(nwg, 8).double().cpu()
    nit = (n // 32) // cs
    beg = (st[:, 7] - st[:, 7].min()) / 100.0      # us (s_memrealtime: 100 MHz, chip-wide)
    hist = torch.histc(beg.float(), bins=8, min=0, max=float(beg.max()) + 1e-3).int().tolist()
    print(f"   workgroup start times: last starts {beg.max().item():.1f} us after the first; histogram over that span {hist}")
    m = st.mean(0)
    print(f"   shader clock while the workgroups ran: {m[6] / m[5] * 0.1:.2f} GHz (s_memtime / s_memrealtime); workgroup duration {m[5] / 100:.1f} us")
    print(f"N={n} geglu={int(geglu)} ln={int(ln)} cs={cs}: units/WG {nit:.0f}; total {m[6]:.0f} cyc; per unit: "
          f"dma-wait {m[0]/nit:.0f}  barrier {m[1]/nit:.0f}  dma-issue {m[2]/nit:.0f}  multiply {m[3]/nit:.0f}  epilogue {m[4]/nit:.0f}", flush=True)
