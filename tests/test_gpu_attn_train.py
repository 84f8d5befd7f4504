"""GPU: the fused training attention (hipie_attn_train_forward / _backward, csrc/attn_train.hip) against the materialised formulation in double.
The operands are what hipie_amd/training/net.vit_attention builds: q' = [scale q, rel_h, rel_w], k' = [k, one-hot key row, one-hot key column]
(Attention.forward + add_decomposed_rel_pos, hipie/backbone/vit.py:69-80, utils.py:96-125)."""
import pytest
import torch

from util import rel_err

gpu = pytest.mark.gpu


def _operands(BH, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    N, hd = H * W, 80
    q = torch.randn(BH, N, hd, generator=g, dtype=torch.float64)
    k = torch.randn(BH, N, hd, generator=g, dtype=torch.float64)
    v = torch.randn(BH, N, hd, generator=g, dtype=torch.float64)
    rel_h = torch.randn(BH, N, H, generator=g, dtype=torch.float64) * 0.7
    rel_w = torch.randn(BH, N, W, generator=g, dtype=torch.float64) * 0.7
    n = torch.arange(N)
    ind = torch.zeros(N, H + W, dtype=torch.float64)
    ind[n, n // W] = 1
    ind[n, H + n % W] = 1
    qa = torch.cat((q * hd ** -0.5, rel_h, rel_w), -1)
    ka = torch.cat((k, ind.expand(BH, -1, -1)), -1)
    return qa, ka, v


@gpu
@pytest.mark.parametrize("BH,H,W", [(3, 16, 16), (2, 32, 32), (1, 8, 16), (5, 64, 64)])
def test_attn_train_forward_vs_materialised(BH, H, W):
    from hipie_amd import ops
    qa, ka, v = _operands(BH, H, W, BH * 100 + H)
    s = qa @ ka.transpose(1, 2)
    want = torch.softmax(s, -1) @ v
    want_lse = torch.logsumexp(s, -1)
    f = lambda t: t.float().cuda()
    out, lse = ops.attn_train_forward(ops.f16_pair(f(qa), 224), ops.f16_pair(f(ka), 224), ops.f16_pair(f(v)))
    assert rel_err(out.cpu(), want) < 3e-6, rel_err(out.cpu(), want)
    assert float((lse.cpu().double() - want_lse).abs().max()) < 2e-5


@gpu
@pytest.mark.parametrize("BH,H,W,gscale", [(3, 16, 16, 1.0), (2, 32, 32, 1e-4), (1, 8, 16, 30.0), (2, 64, 64, 1e-2)])
def test_attn_train_function_gradients_vs_autograd_in_double(BH, H, W, gscale):
    """FusedAttentionFunction (forward + both backward kernels) against torch.autograd of the materialised formulation in double: d q'
    (whose columns 80.. are d rel_h | d rel_w), d k (the first 80 columns of k'; the indicator columns are constants), d v; upstream
    gradients of very different magnitudes (the function scales dO into fp16's range itself)."""
    from hipie_amd.training.functions import FusedAttentionFunction, fused_attention_ok
    qa, ka, v = _operands(BH, H, W, BH * 7 + W)
    g = torch.Generator().manual_seed(5)
    go = torch.randn(v.shape, generator=g, dtype=torch.float64) * gscale
    with torch.enable_grad():
        ql, kl, vl = (t.clone().requires_grad_(True) for t in (qa, ka, v))
        want_o = torch.softmax(ql @ kl.transpose(1, 2), -1) @ vl
        want = torch.autograd.grad(want_o, (ql, kl, vl), go)
        dq, dk, dv = (t.float().cuda().requires_grad_(True) for t in (qa, ka, v))
        assert fused_attention_ok(dq, dk, dv)
        out = FusedAttentionFunction.apply(dq, dk, dv)
        got = torch.autograd.grad(out, (dq, dk, dv), go.float().cuda())
    assert rel_err(out.detach().cpu(), want_o.detach()) < 3e-6
    assert rel_err(got[0].cpu(), want[0]) < 1e-5, ("dq'", rel_err(got[0].cpu(), want[0]))
    assert rel_err(got[1][..., :80].cpu(), want[1][..., :80]) < 1e-5, ("dk", rel_err(got[1][..., :80].cpu(), want[1][..., :80]))
    assert float(got[1][..., 80:].abs().max()) == 0.0
    assert rel_err(got[2].cpu(), want[2]) < 1e-5, ("dv", rel_err(got[2].cpu(), want[2]))


@gpu
def test_f16_pair_kernel_matches_the_torch_formulation():
    """hipie_to_f16_pair (pad + optional device scale + hi / lo split in one pass) == the same arithmetic in torch on the host, bit for bit; values
    beyond fp16's range saturate; a row-strided input view"""
    from hipie_amd import ops

    def host_pair(x, cols, scale=None):
        x = torch.nn.functional.pad(x, (0, cols - x.shape[-1])).float() * (1.0 if scale is None else scale)
        x = x.clamp(-65504.0, 65504.0)
        hi = x.half()
        return hi, (x - hi.float()).half()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 50, 208, generator=g) * torch.logspace(-6, 3, 208)
    x[0, 0, 0], x[0, 0, 1] = 1e6, -1e6
    for cols, scale in ((224, None), (208, None), (224, torch.tensor([2.0 ** -7]))):
        want = host_pair(x, cols, scale)
        got = ops.f16_pair(x.cuda(), cols, None if scale is None else scale.cuda())
        assert got[0].shape[-1] == cols and torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1])
    with pytest.raises(RuntimeError):
        ops.f16_pair(x, 224)                                   # no host path
    wide = torch.randn(40, 96, generator=g)
    want = host_pair(wide[:, :80], 96)
    got = ops.f16_pair(wide.cuda()[:, :80], 96)
    assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1])
    assert float(got[0][:, 80:].abs().max()) == 0.0


@gpu
@pytest.mark.parametrize("BH,H,W", [(2, 50, 76), (3, 33, 35)])
def test_fused_attention_pads_token_counts_that_are_not_a_multiple_of_128(BH, H, W):
    """functions.fused_attention on token grids that are not a multiple of 128 (50 x 76: an 800 x 1216 image; 33 x 35): padding keys masked
    through one more operand column, padding query rows dropped -- output and gradients against the materialised formulation in double on
    the UNPADDED operands.  The 196-token windows are left to the materialised formulation (None)."""
    from hipie_amd.training.functions import fused_attention
    w = _operands(4, 14, 14, 1)
    assert fused_attention(*(t.float().cuda() for t in w)) is None
    qa, ka, v = _operands(BH, H, W, BH + H)
    g = torch.Generator().manual_seed(9)
    go = torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1
    with torch.enable_grad():
        ql, kl, vl = (t.clone().requires_grad_(True) for t in (qa, ka, v))
        want_o = torch.softmax(ql @ kl.transpose(1, 2), -1) @ vl
        want = torch.autograd.grad(want_o, (ql, kl, vl), go)
        dq, dk, dv = (t.float().cuda().requires_grad_(True) for t in (qa, ka, v))
        out = fused_attention(dq, dk, dv)
        assert out is not None and out.shape == (BH, H * W, 80)
        got = torch.autograd.grad(out, (dq, dk, dv), go.float().cuda())
    assert rel_err(out.detach().cpu(), want_o.detach()) < 3e-6
    assert rel_err(got[0].cpu(), want[0]) < 1e-5 and rel_err(got[1][..., :80].cpu(), want[1][..., :80]) < 1e-5 and rel_err(got[2].cpu(), want[2]) < 1e-5
    assert float(got[1][..., 80:].abs().max()) == 0.0
