#!/usr/bin/env python3
"""Round-5 pricing of Winograd F(2x2, 3x3) for the stride-1 resnet convolutions, part 1 (CPU, no GPU needed): what the bf16
rounding of the TRANSFORMED operands costs against the direct bf16 convolution the engine runs today.

Both forms: inputs and weights as the engine holds them (bf16 activations = SiLU(GroupNorm(x)), weights from fp32 masters), fp32
accumulation, bf16 output.  Winograd: V = B^T d B computed in fp32 from the bf16 activations and rounded to bf16 (the MFMA operand),
U = G g G^T computed from the fp32 MASTER weights and rounded to bf16 (pack time), M = sum_c U.V in fp32, Y = A^T M A in fp32.
A second variant keeps the transformed operands in fp16 (same MFMA rate on gfx950, 11 significand bits instead of 8).
Reference: fp32 convolution of the same bf16 activations with the fp32 master weights' bf16 rounding (what the GPU tests compare
against), and with the fp32 masters themselves.

    python tools/probe_winograd_numerics.py > profiles/r05_probe_winograd_numerics.log
"""
import math

import torch
import torch.nn.functional as F

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd(x, w, dt):
    """x [B,C,H,W] fp32 values (already bf16-representable), w [N,C,3,3] fp32 masters; transformed operands rounded to `dt`."""
    B, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                  # [B,C,H/2,W/2,4,4]
    V = torch.einsum("ij,bchwjk,lk->bchwil", Bt, t, Bt).to(dt).float()
    U = torch.einsum("ij,ncjk,lk->ncil", G, w, G).to(dt).float()
    M = torch.einsum("bchwil,ncil->bnhwil", V, U)
    Y = torch.einsum("ij,bnhwjk,lk->bnhwil", At, M, At)                     # [B,N,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def main():
    torch.manual_seed(0)
    print("# shape, then rel-L2 of the bf16-rounded output against the fp32 convolution with bf16-rounded weights")
    for (C, N, HW) in [(320, 320, 32), (640, 640, 16), (1280, 1280, 8), (960, 320, 16)]:
        raw = torch.randn(2, C, HW, HW) * 1.3 + 0.2
        x = F.silu(F.group_norm(raw, 32)).to(torch.bfloat16).float()
        w = torch.randn(N, C, 3, 3) / math.sqrt(9 * C)
        wb = w.to(torch.bfloat16).float()
        ref = F.conv2d(x, wb, padding=1)
        ref_master = F.conv2d(x, w, padding=1)
        direct = ref.to(torch.bfloat16).float()                             # the engine's kernel: exact products, fp32 sum, one rounding
        wino_bf = winograd(x, w, torch.bfloat16).to(torch.bfloat16).float()
        wino_h = winograd(x, w, torch.float16).to(torch.bfloat16).float()
        wino_exact = winograd(x, w, torch.float32)
        print(f"C {C:4d} -> N {N:4d}, {HW}x{HW}:  direct bf16 {rel(direct, ref):.2e} (vs fp32 masters {rel(direct, ref_master):.2e})   "
              f"winograd bf16 operands {rel(wino_bf, ref):.2e} (vs masters {rel(wino_bf, ref_master):.2e})   "
              f"winograd fp16 operands {rel(wino_h, ref):.2e} (vs masters {rel(wino_h, ref_master):.2e})   "
              f"[algebra check, fp32 operands vs masters: {rel(wino_exact, ref_master):.1e}]")
    print("# reading: the direct kernel's error against the fp32-master convolution is the weights' bf16 rounding (~2.3e-3); Winograd with")
    print("# bf16 transformed operands rounds V (sums of 4 activations) and U once more -> the per-convolution error roughly doubles;")
    print("# fp16 transformed operands (same MFMA rate) would be MORE accurate than today's direct bf16 kernel.")


if __name__ == "__main__":
    main()
