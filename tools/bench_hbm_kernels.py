#!/usr/bin/env python3
"""The three HBM- / gather-bound head kernels of the split3 policy at the bench geometry (B = 8, 1024^2), a few launches each -- the command
tools/pmc_kernel.sh profiles (one rocprofv3 pass per counter set): the mask contraction (hipie_mask_einsum_ws, post-ReLU-like features),
the fused deformable-attention sampling on fp32 values (encoder call) and the split-operand dynamic mask head."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hipie_amd import ops  # noqa: E402
from hipie_amd.modeling.transformer import encoder_reference_points  # noqa: E402


def main():
    dev, n = "cuda", int(os.environ.get("LAUNCHES", "5"))
    g = torch.Generator().manual_seed(0)
    B, Q, C, H = 8, 300, 256, 256
    emb = torch.randn(B, Q, C, generator=g).to(dev)
    feat = torch.randn(B, C, H, H, generator=g).clamp_(min=0).to(dev)          # GroupNorm + ReLU output: half zeros
    rb = torch.randn(B, Q, generator=g).to(dev)
    for _ in range(n):
        ops.mask_einsum(emb, feat, precision=1, row_bias=rb)
    M, D, L, P = 8, 32, 4, 4
    shapes = [(128, 128), (64, 64), (32, 32), (16, 16)]
    S = sum(h * w for h, w in shapes)
    value = torch.randn(B, S, M, D, generator=g).to(dev)
    proj = torch.randn(B, S, M * L * P * 3, generator=g)
    proj[..., :M * L * P * 2] *= 1.5
    proj = proj.to(dev)
    off = proj[..., :M * L * P * 2].unflatten(-1, (M, L, P, 2))
    lg = proj[..., M * L * P * 2:].unflatten(-1, (M, L * P))
    ref = encoder_reference_points(shapes, torch.ones(B, L, 2), "cpu").to(dev)
    ss = torch.tensor(shapes, device=dev)
    ls = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    for _ in range(n):
        ops.msda_fused(value, ss, ls, ref, off, lg)
    nq = 910
    feats = torch.randn(B, 8, 128, 128, generator=g).to(dev)
    refs = (torch.rand(B * nq, 2, generator=g) * 1024).to(dev)
    params = torch.randn(B * nq, 169, generator=g).to(dev)
    for _ in range(n):
        ops.dynamic_mask(feats, refs, params, nq, stride=8, up=2, out_dtype=torch.float32, mlp_dtype="split")
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
