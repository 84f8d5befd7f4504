#!/usr/bin/env python3
"""Does a kernel write into LDS it does not own?  A canary kernel (hipie_selftest probe 2: every workgroup fills 16.5 KB of LDS with a
pattern, sleeps, re-checks) runs on a side stream while the kernel under suspicion runs on the main stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipie_amd import _lib, ops  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def rn(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dev)


M = 174080
bx = ops.to_hl8(rn(M, 256))
bxf = rn(M, 256)
bw = ops.hl8_pack(rn(256, 256, scale=0.06)).to(dev)
bw384 = ops.hl8_pack(rn(384, 256, scale=0.06)).to(dev)
vx = ops.to_hl8(rn(32768, 1280))
vw = ops.hl8_pack(rn(1280, 1280, scale=0.03)).to(dev)
t32 = rn(8, 300, 256)
lin1 = torch.nn.Linear(256, 2048).to(dev)
lin2 = torch.nn.Linear(2048, 256).to(dev)
BG = {
    "none": lambda: None,
    "gemm<256> HL8 rows (K 256, N 256)": lambda: ops.gemm(bx, bw, None, split=True, out_fmt=ops.F32),
    "gemm<256> fp32 rows (K 256, N 256)": lambda: ops.gemm(bxf, bw, None, split=True, out_fmt=ops.F32),
    "gemm k256 thin kernel (N 384)": lambda: ops.gemm(bx, bw384, None, split=True, out_fmt=ops.F32),
    "gemm<320> ViT": lambda: ops.gemm(vx, vw, None, split=True, out_fmt=ops.F32),
    "gemm_small": lambda: ops.gemm(t32, bw, None, split=True, out_fmt=ops.F32),
    "ffn_fused": lambda: ops.ffn_fused(bx.view(8, 21760, 512), lin1, lin2),
}
lib = _lib.load()
idx = torch.arange(1 << 20, dtype=torch.int64)
pat = ((idx * 2654435761) & 0xFFFFFFFF) ^ 0xA5A5A5A5
pat = pat.to(torch.int64).to(dev).to(torch.int32) if False else torch.tensor((pat.numpy().astype("uint32")).view("int32"), device=dev)
cfg = torch.tensor([4224, 40], dtype=torch.int16, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
for name, bg in BG.items():
    out = torch.zeros(256, device=dev)
    for it in range(6):
        side.wait_stream(main)
        for _ in range(4):
            bg()
        with torch.cuda.stream(side):
            rc = lib.hipie_selftest(2, cfg.data_ptr(), None, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        main.wait_stream(side)
        torch.cuda.synchronize()
    print("canary beside %-38s: %d corrupted words in %d workgroups" % (name, int(out[0]), int(out[1])), flush=True)
    out = torch.zeros(256, device=dev)
    for it in range(6):
        side.wait_stream(main)
        for _ in range(4):
            bg()
        with torch.cuda.stream(side):
            rc = lib.hipie_selftest(3, pat.data_ptr(), None, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        main.wait_stream(side)
        torch.cuda.synchronize()
    print("   class canaries: L1 dword loads %d | LDS b128 in-wave exchange %d | ds_bpermute %d | 128-B gathers %d   (%d workgroups)" % (
        int(out[4]), int(out[5]), int(out[6]), int(out[7]), int(out[1])), flush=True)
