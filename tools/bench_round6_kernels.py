#!/usr/bin/env python3
"""a few launches of the two kernel forms added at the end of round 6, at the timed step's size (for PMC passes, tools/pmc_kernel.sh):
hipie_gemm_ln (gemm_kernel<256, true, 6>: fp32 A rows, 8 x 21760 tokens) and the two launches of ops.bi_i2t_folded
(gemm_kernel<256, true, 8>: logits + softmax over 8 heads x 194 text tokens; gemm_kernel<256, true, 0> batched with K = 8 x 224)."""
import sys
import torch
sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402

torch.set_grad_enabled(False)


class Owner:
    pass


B, Nv, C, H, hd, L = 8, 21760, 256, 8, 256, 194
E = H * hd
x = torch.randn(B, Nv, C, device="cuda")
w = torch.randn(256, C, device="cuda") * C ** -0.5
b = torch.randn(256, device="cuda")
g, be = torch.randn(256, device="cuda"), torch.randn(256, device="cuda")
own = Owner()
for _ in range(4):
    ops.split_linear_ln(x.view(-1, C), own, "w", w, b, x.view(-1, C), g, be, 1e-5)
vh = ops.to_hl8(x)
k = torch.randn(B, L, E, device="cuda") * 0.5
vl = torch.randn(B, L, E, device="cuda")
wq = torch.randn(E, C, device="cuda") * (0.5 / C ** 0.5)
bq = torch.randn(E, device="cuda") * 0.1
wo = torch.randn(C, E, device="cuda") * E ** -0.5
bo = torch.randn(C, device="cuda")
kh = k.view(B, L, H, hd).permute(0, 2, 1, 3)
M = torch.matmul(kh, wq.view(1, H, hd, C)).contiguous()
cb = (kh * bq.view(1, H, 1, hd)).sum(-1).contiguous()
mask = torch.ones(B, L, dtype=torch.bool, device="cuda")
for _ in range(4):
    ops.bi_i2t_folded(vh, M, cb, vl, mask, H, wo, bo, resid=x)
torch.cuda.synchronize()
