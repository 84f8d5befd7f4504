#!/usr/bin/env python3
"""How exact are the library's fp32 convolutions at the head's full-size shapes?  F.conv2d / conv_transpose2d on the GPU (MIOpen, the
mode the product uses: HIPIE_MIOPEN_FIND=0 = immediate mode) vs the same convolution in double on the CPU."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import hipie_amd  # noqa: F401,E402  (sets the MIOpen mode the product uses)
from hipie_amd import hipie_img  # noqa: F401,E402

torch.manual_seed(0)
CASES = [("lay1 3x3 256->64 @128^2", 256, 64, 3, 1, 1, 128), ("lay2 3x3 64->8 @128^2", 64, 8, 3, 1, 1, 128),
         ("dcn 3x3 256->256 @128^2", 256, 256, 3, 1, 1, 128), ("lay4 3x3 256->256 @64^2", 256, 256, 3, 1, 1, 64),
         ("lay3 3x3 256->256 @32^2", 256, 256, 3, 1, 1, 32), ("input_proj 3x3 s2 1280->256 @32^2", 1280, 256, 3, 2, 1, 32),
         ("1x1 640->256 @128^2", 640, 256, 1, 1, 0, 128), ("layer_1 3x3 256->256 @128^2 (NormConv)", 256, 256, 3, 1, 1, 128)]
for name, cin, cout, k, stride, pad, hw in CASES:
    for fmt in ("nchw", "nhwc", "nchw find", "nhwc find"):
        torch.backends.cudnn.benchmark = fmt.endswith("find")
        x = torch.randn(1, cin, hw, hw)
        w = torch.randn(cout, cin, k, k) * (cin * k * k) ** -0.5
        b = torch.randn(cout)
        want = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
        xg = x.cuda()
        if fmt.startswith("nhwc"):
            xg = xg.contiguous(memory_format=torch.channels_last)
        got = F.conv2d(xg, w.cuda(), b.cuda(), stride=stride, padding=pad).cpu().double()
        err = float((got - want).abs().max() / want.abs().max())
        print("%-44s %-9s  max|err|/max|ref| = %.2e" % (name, fmt, err), flush=True)
x = torch.randn(1, 256, 128, 128)
w = torch.randn(256, 256, 2, 2) * 256 ** -0.5
want = F.conv_transpose2d(x.double(), w.double(), None, stride=2)
got = F.conv_transpose2d(x.cuda(), w.cuda(), None, stride=2).cpu().double()
print("%-44s nchw  max|err|/max|ref| = %.2e" % ("mask_features convT 2x2 s2 256->256 @128^2", float((got - want).abs().max() / want.abs().max())))
print("MIOPEN env:", {k: v for k, v in os.environ.items() if "MIOPEN" in k})
