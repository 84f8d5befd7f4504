#!/usr/bin/env python3
"""CPU: how accurate would the split GEMM be with its two CROSS terms on FP8 operands?  (docs/next_round.md: the cross terms W_lo.A_hi and
W_hi.A_lo are 2^-11 of the main term, so ~5 operand bits keep the product at the 2^-15 class, and gfx950's FP8 MFMA runs at twice the fp16 rate.)
Products of random operands at the ViT-H shapes' K, against fp64:  max|a - b| / max|b|  (the tests' metric) and the RMS relative error.
  three      W_hi.A_hi + W_lo.A_hi + W_hi.A_lo on fp16 operands            (what hipie_gemm computes today: 3 fp16 MFMAs)
  fp8 cross  W_hi.A_hi on fp16 + q8(W_lo).q8(A_hi) + q8(W_hi).q8(A_lo)     (1 fp16 + 2 fp8 MFMAs = 2 fp16-equivalents), q8 = e4m3 with a
             power-of-two scale per 32-element block along K (the block scales of v_mfma_scale_f32_*_f8f6f4)
  one        W_hi.A_hi alone                                              (the 'fast' policy's single product)
python tools/fp8_cross_terms.py"""
import torch

torch.manual_seed(0)


def split(x):
    hi = x.half().float()
    lo = (x - hi).half().float()
    return hi, lo


def q8(x, block=32):
    """e4m3 rounding with a power-of-two scale per `block` consecutive elements of the last dimension"""
    shp = x.shape
    xb = x.reshape(-1, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(448.0 / amax)))
    return ((xb * scale).to(torch.float8_e4m3fn).float() / scale).reshape(shp)


def err(a, ref):
    d = (a.double() - ref)
    return float(d.abs().max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())


for M, N, K in ((512, 1280, 1280), (512, 1280, 5120), (512, 256, 256)):
    A = torch.randn(M, K) * 1.5
    W = torch.randn(N, K) * K ** -0.5
    ref = A.double() @ W.double().t()
    ah, al = split(A)
    wh, wl = split(W)
    main = ah.double() @ wh.double().t()
    three = main + ah.double() @ wl.double().t() + al.double() @ wh.double().t()
    f8 = main + q8(ah).double() @ q8(wl).double().t() + q8(al).double() @ q8(wh).double().t()
    print("M=%d N=%d K=%d:  three %.1e (rms %.1e)   fp8 cross %.1e (rms %.1e)   one %.1e (rms %.1e)" % (
        (M, N, K) + err(three, ref) + err(f8, ref) + err(main, ref)))
