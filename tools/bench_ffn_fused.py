#!/usr/bin/env python3
"""encoder FFN at bs 8 (174080 tokens): the two split GEMMs (linear1 -> ReLU -> HL8, linear2) against hipie_ffn_fused.
    python tools/bench_ffn_fused.py [two|fused]    (one side only: for PMC passes)"""
import torch, types, sys
sys.path.insert(0, ".")
from hipie_amd import ops
M=174080
x=ops.to_hl8(torch.randn(M,256,device="cuda"))
l1=types.SimpleNamespace(weight=torch.randn(2048,256,device="cuda")/16, bias=torch.randn(2048,device="cuda"))
l2=types.SimpleNamespace(weight=torch.randn(256,2048,device="cuda")/45, bias=torch.randn(256,device="cuda"))
w1=ops.hl8_pack(l1.weight); w2=ops.hl8_pack(l2.weight)
def two():
    h=ops.gemm(x,w1,l1.bias,out_fmt=ops.HL8,act=ops.ACT_RELU,split=True)
    return ops.gemm(h,w2,l2.bias,out_fmt=ops.F32,split=True)
def one(): return ops.ffn_fused(x,l1,l2)
only = sys.argv[1] if len(sys.argv) > 1 else None
for f,n in ((two,"two GEMMs"),(one,"fused")):
    if only and not n.startswith(only):
        continue
    for _ in range(3): f()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    print(n, "%.3f ms" % (a.elapsed_time(b)/10))
