import sys, torch
sys.path.insert(0, ".")
from hipie_amd import ops
torch.manual_seed(0)
for shape in ((2, 256, 32, 40), (1, 256, 64, 64), (3, 64, 8, 8)):
    C = shape[1]
    x = (torch.randn(shape) * 2 + 0.5).cuda().contiguous(memory_format=torch.channels_last)
    w, b = torch.randn(C).cuda(), torch.randn(C).cuda()
    want = ops.group_norm(x, C // 8, w, b, 1e-5, relu=True).contiguous()
    got = ops.group_norm(x, C // 8, w, b, 1e-5, relu=True, out_nchw=True)
    d = (got - want).abs()
    print(shape, got.shape, got.is_contiguous(), "max diff", d.max().item(), "bad frac", (d > 0).float().mean().item())
    bad = (d > 0).nonzero()
    print(bad[:5].tolist(), bad[-3:].tolist())
    print("per-channel bad", (d > 0).float().mean((0, 2, 3))[:16].tolist())
    print("per-pixel bad", (d > 0).float().mean((0, 1)).flatten()[:70].tolist())
