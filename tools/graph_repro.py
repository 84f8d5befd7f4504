"""capture + replay a hipGraph of a small slice of the model (encoder layer with MSDeformAttn, a ViT block) to check that
every custom op is capture-safe."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipie_amd.config import HipieConfig, Precision
from hipie_amd.modeling.transformer import DeformableTransformerEncoderLayer, cast_head, level_tensors, encoder_reference_points
from hipie_amd.modeling.vit import Block

torch.set_grad_enabled(False)
which = sys.argv[1] if len(sys.argv) > 1 else "enc"
prec = Precision.fast()
dev = "cuda"
if which == "enc":
    layer = DeformableTransformerEncoderLayer(256, 2048, 4, 8, 4, prec.value).to(dev)
    cast_head(layer, prec.head, prec.act)
    shapes = [(128, 128), (64, 64), (32, 32), (16, 16)]
    S = sum(h * w for h, w in shapes)
    B = 4
    src = torch.randn(B, S, 256, device=dev).bfloat16()
    pos = torch.randn(B, S, 256, device=dev).bfloat16()
    ss, ls = level_tensors(shapes, torch.device(dev))
    refs = encoder_reference_points(shapes, torch.ones(B, 4, 2, device=dev), torch.device(dev))
    f = lambda: layer(src, pos, refs, ss, ls, None)
else:
    blk = Block(1280, 16, 4.0, 14 if which == "win" else 0, (64, 64), prec).to(dev)
    for m in blk.modules():
        if isinstance(m, torch.nn.Linear):
            m.to(prec.gemm)
    x = torch.randn(4, 64, 64, 1280, device=dev).bfloat16()
    f = lambda: blk(x, None)
for _ in range(2):
    out = f()
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    f()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gout = f()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
ref = f()
a = gout[0] if isinstance(gout, tuple) else gout
b = ref[0] if isinstance(ref, tuple) else ref
print(which, "graph replay ok; max diff", float((a.float() - b.float()).abs().max()))
