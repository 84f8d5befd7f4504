#!/usr/bin/env python3
"""hipie_selftest probe 4: is the v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16 form of the fp16-pair split bit-identical to the C++ form?"""
import os
import struct
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipie_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
n = 4096 * 256
for name, x in (("N(0,1)", torch.randn(2 * n, generator=g)), ("probabilities [0, 64]", torch.rand(2 * n, generator=g) * 64),
                ("log-uniform 1e-6 .. 1e4", torch.exp(torch.rand(2 * n, generator=g) * 23 - 13.8) * torch.sign(torch.randn(2 * n, generator=g))),
                ("tiny 1e-9 .. 1e-4", torch.exp(torch.rand(2 * n, generator=g) * 11.5 - 20.7))):
    x = x.to(dev).contiguous()
    out = torch.zeros(256, device=dev)
    assert lib.hipie_selftest(4, x.data_ptr(), None, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    o = out.cpu()
    msg = ""
    if o[0] > 0:
        w = [struct.unpack("I", struct.pack("f", float(v)))[0] for v in o[4:8]]
        msg = "  first: a=%r b=%r  hi asm %08x c++ %08x | lo asm %08x c++ %08x" % (float(o[2]), float(o[3]), w[0], w[1], w[2], w[3])
    print("%-26s %d of %d pairs differ%s" % (name, int(o[0]), n, msg))
