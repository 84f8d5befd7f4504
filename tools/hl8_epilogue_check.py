import sys, torch
sys.path.insert(0, ".")
from hipie_amd import ops
torch.manual_seed(0)
x = torch.randn(4096, 256, device="cuda") * torch.logspace(-4, 1, 256, device="cuda")[None]
eye = ops.hl8_pack(torch.eye(256)).cuda()
got = ops.gemm(ops.to_hl8(x), eye, None, split=True, out_fmt=ops.HL8)
want = ops.to_hl8(x)
same = torch.equal(got, want)
d = (ops.hl8_unpack(got) - ops.hl8_unpack(want)).abs()
print("GEMM HL8 epilogue == to_hl8: %s; max |diff| %.3e; lo halves equal: %s" % (same, float(d.max()), torch.equal(got.view(-1, 16, 2, 8)[:, :, 1], want.view(-1, 16, 2, 8)[:, :, 1])))
lo = want.view(-1, 16, 2, 8)[:, :, 1].float().abs()
sub = (lo > 0) & (lo < 6.1e-5)
gl = got.view(-1, 16, 2, 8)[:, :, 1].float().abs()
print("subnormal lo halves in the reference: %d; of those flushed to zero by the epilogue: %d" % (int(sub.sum()), int(((gl == 0) & sub).sum())))
gh, gl_ = got.view(-1, 16, 2, 8)[:, :, 0].float(), got.view(-1, 16, 2, 8)[:, :, 1].float()
wh, wl = want.view(-1, 16, 2, 8)[:, :, 0].float(), want.view(-1, 16, 2, 8)[:, :, 1].float()
xs = x.view(-1, 16, 8)
print("hi halves equal: %s (%d differ); lo halves differ in %d of %d" % (torch.equal(gh, wh), int((gh != wh).sum()), int((gl_ != wl).sum()), gl_.numel()))
idx = torch.nonzero(gl_ != wl)[:8]
for i in idx.tolist():
    a, b, c = i
    print("x %.9e | epilogue hi %.9e lo %.9e (sum err %.2e) | to_hl8 hi %.9e lo %.9e (sum err %.2e)" % (
        float(xs[a, b, c]), float(gh[a, b, c]), float(gl_[a, b, c]), float(xs[a, b, c].double() - gh[a, b, c].double() - gl_[a, b, c].double()),
        float(wh[a, b, c]), float(wl[a, b, c]), float(xs[a, b, c].double() - wh[a, b, c].double() - wl[a, b, c].double())))
e_got = (xs.double() - gh.double() - gl_.double()).abs() / xs.double().abs().clamp_min(1e-30)
e_want = (xs.double() - wh.double() - wl.double()).abs() / xs.double().abs().clamp_min(1e-30)
print("relative split error |x - hi - lo| / |x|: epilogue max %.3e mean %.3e | to_hl8 max %.3e mean %.3e" % (float(e_got.max()), float(e_got.mean()), float(e_want.max()), float(e_want.mean())))
