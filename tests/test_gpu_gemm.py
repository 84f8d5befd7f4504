"""GPU: hipie_gemm (plain fp16 and split-fp16 "HL8" operands) against fp64 torch.matmul.

Tolerances (max|a-b| / max|b|):
  split (HL8 x HL8, three MFMA products, fp32 accumulation)  3e-6 against the fp64 product of the ORIGINAL fp32 operands --
        the reference runs these linears in fp32 (hipie/backbone/vit.py:67-83, deformable_transformer_dino.py:378-394);
  plain fp16                                                  3e-6 against the fp64 product of the fp16-ROUNDED operands (the only
        error left is the fp32 accumulation order), 1e-3 against the product of the unrounded operands (the operand rounding).
"""
import pytest
import torch

from util import rel_err

torch.set_grad_enabled(False)


def test_hl8_pack_roundtrip_cpu():
    from hipie_amd import ops
    x = torch.randn(7, 64) * torch.logspace(-6, 3, 64)
    p = ops.hl8_pack(x)
    assert p.shape == (7, 128) and p.dtype == torch.float16
    back = ops.hl8_unpack(p)
    big = x.abs() > 0.25                 # lo = fp16(x - hi) is a NORMAL fp16 there: the pair carries 22 bits
    assert ((back - x).abs()[big] <= x.abs()[big] * 2.0 ** -21).all()
    assert ((back - x).abs() <= 2.0 ** -25 + x.abs() * 2.0 ** -21).all()      # below: lo is subnormal, absolute error 2^-25
    # layout: group g of 8 values -> 8 hi then 8 lo
    assert torch.equal(p[:, 0:8], x[:, 0:8].half())
    assert torch.equal(p[:, 16:24], x[:, 8:16].half())


gpu = pytest.mark.gpu


def _ref(a, w, bias, resid, act, alpha, oscale):
    y = alpha * (a.double() @ w.double().t())
    if bias is not None:
        y = y + bias.double()
    if act == 1:
        y = torch.nn.functional.gelu(y)
    elif act == 2:
        y = torch.relu(y)
    elif act == 3:                                   # QuickGELU (open_clip, the OpenAI CLIP weights)
        y = y * torch.sigmoid(1.702 * y)
    if resid is not None:
        y = y + resid.double()
    return y * oscale


@gpu
def test_to_hl8_kernel_matches_torch_pack():
    from hipie_amd import ops
    x = (torch.randn(301, 1280, device="cuda") * 3).contiguous()
    assert torch.equal(ops.to_hl8(x), ops.hl8_pack(x))
    assert torch.equal(ops.to_hl8(x, 16.0), ops.hl8_pack(x, 16.0))
    xs = torch.randn(50, 512, device="cuda")[:, :256]          # row-strided view
    assert torch.equal(ops.to_hl8(xs), ops.hl8_pack(xs))
    xh = torch.randn(33, 64, device="cuda").half()
    assert torch.equal(ops.to_hl8(xh), ops.hl8_pack(xh))


@gpu
@pytest.mark.parametrize("M,C,stride", [(8192, 1280, None), (333, 64, None), (1000, 320, 512), (7, 5, None), (130, 70, 72)])
def test_to_hl8_t_is_the_pack_of_the_transpose(M, C, stride):
    """hipie_to_hl8_t (round 6: both operands of the weight gradients) == hl8_pack(x^T) with the rows padded to 32, bit for bit: full tiles,
    ragged rows and columns, a row-strided input, columns that are not a multiple of 4 (the unaligned load path)"""
    from hipie_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + C)
    base = torch.randn(M, stride or C, device="cuda", generator=g) * 3
    x = base[:, :C]
    Mp = -(-M // 32) * 32
    want = torch.zeros(C, Mp, device="cuda")
    want[:, :M] = x.t()
    got = ops.to_hl8_t(x, 32)
    assert got.shape == (C, 2 * Mp) and torch.equal(got, ops.hl8_pack(want))
    assert torch.equal(ops.to_hl8_t(x, 32, 0.25), ops.hl8_pack(want, 0.25))


SHAPES = [  # M, N, K
    (300, 256, 256),          # M tail, one N tile of 256
    (512, 1280, 1280),        # the 320-wide tile (ViT-H proj), two full M tiles
    (700, 384, 768),          # N tail inside a 256 tile
    (257, 640, 2048),         # 320 tile, M tail of one row
    (64, 2048, 256),          # small M
    (1000, 3840, 1280),       # ViT-H qkv
]


@gpu
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_split_fp32_class(M, N, K):
    from hipie_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g) * 2.0
    w = torch.randn(N, K, device="cuda", generator=g) * (K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    out = ops.gemm(ops.hl8_pack(a), ops.hl8_pack(w), bias, split=True)
    err = rel_err(out.cpu(), _ref(a, w, bias, None, 0, 1.0, 1.0).cpu())
    print("split M=%d N=%d K=%d err %.2e" % (M, N, K, err))
    assert err < 3e-6


@gpu
def test_gemm_split_small_magnitudes():
    """operands whose lo halves are fp16 SUBNORMALS (|x| ~ 1e-2): the matrix pipe must not flush them (error would be ~1e-4)."""
    from hipie_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(512, 512, device="cuda", generator=g) * 1e-2
    w = torch.randn(256, 512, device="cuda", generator=g) * 1e-2
    out = ops.gemm(ops.hl8_pack(a), ops.hl8_pack(w), None, split=True)
    err = rel_err(out.cpu(), _ref(a, w, None, None, 0, 1.0, 1.0).cpu())
    print("split, subnormal lo parts: err %.2e" % err)
    assert err < 2e-5


@gpu
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_plain_fp16(M, N, K):
    from hipie_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M * 3 + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * (K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    a16, w16 = a.half(), w.half()
    out = ops.gemm(a16, w16, bias, split=False)
    e16 = rel_err(out.cpu(), _ref(a16, w16, bias, None, 0, 1.0, 1.0).cpu())
    e32 = rel_err(out.cpu(), _ref(a, w, bias, None, 0, 1.0, 1.0).cpu())
    print("plain M=%d N=%d K=%d err vs rounded operands %.2e, vs fp32 operands %.2e" % (M, N, K, e16, e32))
    assert e16 < 3e-6 and e32 < 1e-3


@gpu
@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("out_fmt", ["f32", "f16", "hl8"])
@pytest.mark.parametrize("act,with_res", [(0, False), (1, False), (2, True), (0, True), (3, False), (3, True)])
def test_gemm_epilogues(split, out_fmt, act, with_res):
    from hipie_amd import ops
    M, N, K = 333, 640, 512
    g = torch.Generator(device="cuda").manual_seed(17)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * (K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g) if with_res else None
    alpha, osc = 0.5, (4.0 if out_fmt != "f32" else 1.0)
    if split:
        A, W = ops.hl8_pack(a), ops.hl8_pack(w)
        ar, wr = a, w
    else:
        A, W = a.half(), w.half()
        ar, wr = A, W
    fmt = {"f32": ops.F32, "f16": ops.F16, "hl8": ops.HL8}[out_fmt]
    out = ops.gemm(A, W, bias, resid, out_fmt=fmt, act=act, alpha=alpha, oscale=osc, split=split)
    ref = _ref(ar, wr, bias, resid, act, alpha, osc).cpu()
    if out_fmt == "hl8":
        assert out.shape == (M, 2 * N) and out.dtype == torch.float16
        got = ops.hl8_unpack(out).cpu()
        tol = 3e-6
        pl = out.cpu().reshape(M, N // 8, 2, 8).float()
        hi_plane, lo_plane = pl[:, :, 0, :].reshape(M, N), pl[:, :, 1, :].reshape(M, N)
        # hi is the fp16 rounding of the value and lo the remainder: |lo| <= half an ulp of hi (2^-11 relative), never a second "hi"
        assert (lo_plane.abs() <= hi_plane.abs() * 2.0 ** -11 + 2.0 ** -24).all()
    elif out_fmt == "f16":
        got, tol = out.float().cpu(), 6e-4          # one fp16 rounding of the result
    else:
        got, tol = out.cpu(), 3e-6
    err = rel_err(got, ref)
    print("epilogue split=%s out=%s act=%d res=%s err %.2e" % (split, out_fmt, act, with_res, err))
    assert err < tol


@gpu
def test_gemm_strided_rows_and_chaining():
    """A as a column block of a wider tensor (row stride > K) and an HL8 output fed straight into the next GEMM (fc1 -> GELU -> fc2)."""
    from hipie_amd import ops
    g = torch.Generator(device="cuda").manual_seed(23)
    M, K, Hd = 520, 256, 1024
    x = torch.randn(M, K, device="cuda", generator=g)
    w1 = torch.randn(Hd, K, device="cuda", generator=g) * (K ** -0.5)
    w2 = torch.randn(K, Hd, device="cuda", generator=g) * (Hd ** -0.5)
    b1 = torch.randn(Hd, device="cuda", generator=g)
    b2 = torch.randn(K, device="cuda", generator=g)
    wide = torch.zeros(M, 3 * 2 * K, dtype=torch.float16, device="cuda")
    wide[:, 2 * K:4 * K] = ops.hl8_pack(x)
    h = ops.gemm(wide[:, 2 * K:4 * K], ops.hl8_pack(w1), b1, out_fmt=ops.HL8, act=ops.ACT_GELU, split=True)
    y = ops.gemm(h, ops.hl8_pack(w2), b2, resid=x, split=True)
    ref = torch.nn.functional.gelu(x.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double() + x.double()
    err = rel_err(y.cpu(), ref.cpu())
    print("chained split MLP err %.2e" % err)
    assert err < 3e-6


@gpu
def test_gemm_rejects_bad_arguments():
    from hipie_amd import ops
    a = torch.zeros(8, 96, dtype=torch.float16, device="cuda")
    w = torch.zeros(16, 96, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm(a, w, split=False)             # K = 96 is not a multiple of 64
    with pytest.raises(RuntimeError):
        ops.gemm(a.float(), w, split=False)
    w2 = torch.zeros(12, 128, dtype=torch.float16, device="cuda")
    a2 = torch.zeros(8, 128, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm(a2, w2, split=False)           # N = 12 is not a multiple of 8


@gpu
def test_gemm_row_map_and_in_place_residual():
    """out_row: product row m lands in (and takes its residual from) row out_row[m], negative entries are dropped; out aliases the
    residual (the ViT stream is updated in place; window_unpartition is the store index)."""
    from hipie_amd import ops
    from hipie_amd.modeling.vit import window_row_maps
    g = torch.Generator(device="cuda").manual_seed(31)
    B, H, W, C, ws = 2, 9, 9, 256, 7
    out_src, delta_row, nwin = window_row_maps(B, H, W, ws, torch.device("cuda"))
    M = nwin * ws * ws
    a = torch.randn(M, C, device="cuda", generator=g)
    w = torch.randn(C, C, device="cuda", generator=g) * C ** -0.5
    bias = torch.randn(C, device="cuda", generator=g)
    stream = torch.randn(B * H * W, C, device="cuda", generator=g)
    want = stream.double().clone()
    full = a.double() @ w.double().t() + bias.double()
    valid = out_src >= 0
    want[out_src[valid].long()] += full[valid]
    got = stream.clone()
    ops.gemm(ops.hl8_pack(a), ops.hl8_pack(w), bias, resid=got, out=got, out_row=out_src, split=True)
    assert rel_err(got.cpu(), want.float().cpu()) < 3e-6


@gpu
@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (1000, 1280, 1280), (64, 2048, 256), (7200, 264, 256)])
def test_gemm_fp32_rows_equal_the_converted_form_bit_for_bit(M, N, K):
    """in_fmt HIPIE_F32: the A rows are split inside the kernel -- identical to hipie_to_hl8 + the HL8 form, including a strided view,
    values beyond the fp16 range (saturated) and fp16-subnormal remainders."""
    from hipie_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + 2 * N + K)
    wide = torch.randn(M, K + 64, device="cuda", generator=g) * 2.0
    a = wide[:, 32:32 + K]                                   # row stride K + 64, 128-byte offset
    a[0, :4] = torch.tensor([7e4, -1e5, 1e-3, 65504.0], device="cuda")
    w = ops.hl8_pack(torch.randn(N, K, device="cuda", generator=g) * (K ** -0.5))
    bias = torch.randn(N, device="cuda", generator=g)
    want = ops.gemm(ops.to_hl8(a), w, bias, split=True, act=ops.ACT_RELU)
    got = ops.gemm(a, w, bias, split=True, act=ops.ACT_RELU)
    assert torch.equal(got, want)
    want_h = ops.gemm(ops.to_hl8(a), w, bias, split=True, out_fmt=ops.HL8)
    got_h = ops.gemm(a, w, bias, split=True, out_fmt=ops.HL8)
    assert torch.equal(got_h, want_h)


@gpu
@pytest.mark.parametrize("cin,cout,hw", [(640, 256, (128, 128)), (1280, 256, (64, 64)), (256, 256, (20, 36))])
def test_pointwise_conv_on_the_split_gemm(cin, cout, hw):
    """PConv2d with `split`: a 1x1 convolution of a channels-last map runs as hipie_gemm on the pixel rows (input_proj of both heads,
    maskdino_encoder.py:213-236 / deformable_detr.py:139-160) -- vs F.conv2d in double; an NCHW-contiguous input keeps the library path."""
    from hipie_amd.modeling.transformer import PConv2d
    g = torch.Generator(device="cuda").manual_seed(cin + cout)
    conv = PConv2d(cin, cout, kernel_size=1).cuda()
    conv.split = True
    x = torch.randn(2, hw[0], hw[1], cin, device="cuda", generator=g).permute(0, 3, 1, 2)        # logical NCHW, channels-last memory
    want = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double())
    from hipie_amd import ops
    ops.PROFILE.enable("gemm")
    y = conv(x)
    n_gemm = len(ops.PROFILE.events.get("gemm", []))
    ops.PROFILE.disable()
    assert n_gemm == 1 and y.shape == want.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert rel_err(y.cpu(), want.float().cpu()) < 3e-6
    ops.PROFILE.enable("gemm")
    y2 = conv(x.contiguous())                      # NCHW memory: not a row-major pixel matrix -> MIOpen
    assert len(ops.PROFILE.events.get("gemm", [])) == 0
    ops.PROFILE.disable()
    assert rel_err(y2.cpu(), want.float().cpu()) < 1e-4


@gpu
@pytest.mark.parametrize("M", [4096, 21760 + 37, 128 * 3 + 5])
def test_ffn_fused_matches_two_split_gemms(M):
    """hipie_ffn_fused (linear1 -> ReLU -> linear2 of the deformable encoder layers in one launch, hidden activations in registers) against
    the fp64 evaluation and against the two hipie_gemm launches it replaces (same operand splits: identical up to the fp32 summation order
    inside a 16-wide MFMA step); ragged token count, biases, negative pre-activations."""
    import types
    from hipie_amd import ops
    g = torch.Generator().manual_seed(M)
    D, F_ = 256, 2048
    x = torch.randn(M, D, generator=g)
    lin1 = types.SimpleNamespace(weight=(torch.randn(F_, D, generator=g) * D ** -0.5).cuda(), bias=(torch.randn(F_, generator=g) * 0.3).cuda())
    lin2 = types.SimpleNamespace(weight=(torch.randn(D, F_, generator=g) * F_ ** -0.5).cuda(), bias=(torch.randn(D, generator=g) * 0.3).cuda())
    xs = ops.to_hl8(x.cuda())
    if M >= 4096:
        assert ops.ffn_fused_ok(xs, lin1, lin2)
    got = ops.ffn_fused(xs, lin1, lin2).cpu()
    want = (torch.relu(x.double() @ lin1.weight.cpu().double().t() + lin1.bias.cpu().double()) @ lin2.weight.cpu().double().t()
            + lin2.bias.cpu().double())
    e64 = float((got.double() - want).abs().max() / want.abs().max())
    h = ops.gemm(xs, ops.hl8_pack(lin1.weight), lin1.bias, out_fmt=ops.HL8, act=ops.ACT_RELU, split=True)
    two = ops.gemm(h, ops.hl8_pack(lin2.weight), lin2.bias, out_fmt=ops.F32, split=True).cpu()
    e2 = float((got - two).abs().max() / two.abs().max())
    print("ffn_fused M=%d: vs fp64 %.2e, vs the two split GEMMs %.2e" % (M, e64, e2))
    assert e64 < 5e-6 and e2 < 5e-6


@gpu
@pytest.mark.parametrize("B,C,N,H,W,bias", [(2, 256, 256, 128, 128, True), (1, 64, 256, 33, 47, False), (3, 32, 512, 64, 20, True)])
def test_conv3x3_split_matches_fp64(B, C, N, H, W, bias):
    """hipie_conv3x3_split (3 x 3 / stride 1 / padding 1 convolution as an implicit GEMM on a zero-padded pixel grid, three-product split
    arithmetic) against F.conv2d in double: ragged maps, borders, bias, NCHW and channels-last inputs."""
    import torch.nn.functional as F
    from hipie_amd import ops
    g = torch.Generator().manual_seed(B * 100 + C + H)
    conv = torch.nn.Conv2d(C, N, 3, padding=1, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5)
        if bias:
            conv.bias.copy_(torch.randn(N, generator=g) * 0.3)
    conv = conv.cuda()
    x = torch.randn(B, C, H, W, generator=g)
    want = F.conv2d(x.double(), conv.weight.detach().cpu().double(), None if not bias else conv.bias.detach().cpu().double(), padding=1)
    for xin in (x.cuda(), x.cuda().contiguous(memory_format=torch.channels_last)):
        got = ops.conv3x3_split(xin, conv)
        assert got.shape == (B, N, H, W)
        e = float((got.double().cpu() - want).abs().max() / want.abs().max())
        print("conv3x3_split %dx%dx%dx%d -> %d: %.2e" % (B, C, H, W, N, e))
        assert e < 3e-6


@gpu
def test_gemm_gather_and_scatter_rows():
    """hipie_gemm_gather: product row m reads operand row a_row[m] and (with out_row) lands in output row out_row[m] -- the windowed ViT
    blocks run their linears over the real tokens of the zero-padded window layout this way; rows that are not written keep their content."""
    from hipie_amd import ops
    g = torch.Generator().manual_seed(11)
    rows, M, K, N = 1200, 700, 320, 640
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    a_row = torch.randperm(rows, generator=g)[:M].to(torch.int32)
    out_row = torch.randperm(rows, generator=g)[:M].to(torch.int32)
    xs, ws = ops.to_hl8(x.cuda()), ops.hl8_pack(w.cuda())
    want = x[a_row.long()].double() @ w.double().t() + b.double()
    got = ops.gemm(xs, ws, b.cuda(), split=True, a_row=a_row.cuda()).cpu()
    assert got.shape == (M, N) and rel_err(got, want.float()) < 3e-6
    out = torch.full((rows, N), 7.0, device="cuda")
    ops.gemm(xs, ws, b.cuda(), split=True, a_row=a_row.cuda(), out_row=out_row.cuda(), out=out)
    ref = torch.full((rows, N), 7.0, dtype=torch.float64)
    ref[out_row.long()] = want
    assert rel_err(out.cpu(), ref.float()) < 3e-6
    xf = x.cuda()                                   # fp32 operand rows split in the kernel
    got32 = ops.gemm(xf, ws, b.cuda(), split=True, a_row=a_row.cuda()).cpu()
    assert rel_err(got32, want.float()) < 3e-6


@gpu
@pytest.mark.parametrize("M,N,f32", [(8300, 384, False), (20000, 2304, True), (8192, 1024, True)])
def test_gemm_thin_k256_kernel(M, N, f32):
    """gemm_k256.hip (K = 256, fp32 out, many rows: X rows in registers, the weight streamed through LDS in 32-feature chunks; what
    hipie_gemm runs for N >= 384) against fp64 and against the tile kernel on the first 256 output features (N = 256 takes the tile kernel);
    a row-strided fp32 operand, a missing bias, an M tail."""
    from hipie_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    K = 256
    wide = torch.randn(M, K + 64, device="cuda", generator=g) * 2.0
    x = wide[:, 32:32 + K]
    w = torch.randn(N, K, device="cuda", generator=g) * (K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    a = x if f32 else ops.to_hl8(x)
    tile = ops.gemm(a, ops.hl8_pack(w[:256]), bias[:256].contiguous(), split=True)
    thin = ops.gemm(a, ops.hl8_pack(w), bias, split=True)
    nob = ops.gemm(a, ops.hl8_pack(w), None, split=True)
    want = _ref(x, w, bias, None, 0, 1.0, 1.0)
    assert rel_err(thin.cpu(), want.cpu()) < 3e-6
    assert rel_err(thin[:, :256].cpu(), tile.cpu()) < 1e-6
    assert rel_err((nob + bias).cpu(), want.cpu()) < 3e-6
    pad = torch.full((M + 3, N + 8), 7.0, device="cuda")                     # strided output rows, nothing written outside them
    ops.gemm(a, ops.hl8_pack(w), bias, split=True, out=pad[1:M + 1, :N])
    assert torch.equal(pad[1:M + 1, :N], thin) and bool((pad[0] == 7).all()) and bool((pad[M + 1:] == 7).all()) and bool((pad[:, N:] == 7).all())


@gpu
@pytest.mark.parametrize("M,N,K,bias", [(1000, 320, 256, True), (4096, 1280, 1280, True), (333, 64, 2048, False), (8192, 3840, 1280, True), (2080, 512, 5120, False)])
def test_split_linear_function_forward_and_backward(M, N, K, bias):
    """row f-4 (3a): hipie_gemm as an autograd Function -- y = x W^T + b, dx = dy W, dW = dy^T x, db = sum dy, all three products on the
    split-fp16 GEMM (training/functions.SplitLinearFunction) -- against torch.autograd of F.linear in double.  dW cuts the token rows into
    up to 16 chunks that run as one hipie_gemm_batched launch (16 / 16 / 1 / 4 / 1 chunks in these cases; 2080 = 65 x 32 rows do not halve)."""
    from hipie_amd.training.functions import SplitLinearFunction
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g, dtype=torch.float64)
    w = torch.randn(N, K, generator=g, dtype=torch.float64) * K ** -0.5
    b = torch.randn(N, generator=g, dtype=torch.float64) if bias else None
    go = torch.randn(M, N, generator=g, dtype=torch.float64)
    with torch.enable_grad():
        leaves = [t.clone().requires_grad_(True) for t in (x, w) + ((b,) if bias else ())]
        want_y = torch.nn.functional.linear(leaves[0], leaves[1], leaves[2] if bias else None)
        want = torch.autograd.grad(want_y, leaves, go)
        dl = [t.float().cuda().requires_grad_(True) for t in (x, w) + ((b,) if bias else ())]
        owner = type("_W", (), {})()
        y = SplitLinearFunction.apply(dl[0], dl[1], dl[2] if bias else None, owner, "w")
        got = torch.autograd.grad(y, dl, go.float().cuda())
    assert rel_err(y.detach().cpu(), want_y.detach()) < 3e-6
    for a, c, name in zip(got, want, ("dx", "dW", "db")):
        assert a.shape == c.shape and rel_err(a.cpu(), c) < 5e-6, (name, rel_err(a.cpu(), c))


class _Owner:
    pass


@gpu
@pytest.mark.parametrize("M,K,hl8_in", [(1, 256, False), (255, 256, True), (1000, 256, False), (21760, 256, False), (4100, 2048, True)])
def test_gemm_ln_matches_projection_then_layernorm(M, K, hl8_in):
    """hipie_gemm_ln (output_proj + residual + norm1 of the deformable encoder layer in ONE launch: deformable_transformer_dino.py:387-389)
    against the two launches it replaces (hipie_gemm, then hipie_add_layernorm_dec) and against fp64; the HL8 output is the split of the fp32
    output bit for bit; in place on the residual stream."""
    from hipie_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x = torch.randn(M, K, device="cuda", generator=g) * 2.0
    w = torch.randn(256, K, device="cuda", generator=g) * (K ** -0.5)
    b = torch.randn(256, device="cuda", generator=g)
    resid = torch.randn(M, 256, device="cuda", generator=g) * 3.0 + 0.7
    gamma = torch.randn(256, device="cuda", generator=g)
    beta = torch.randn(256, device="cuda", generator=g)
    eps = 1e-5
    a = ops.to_hl8(x) if hl8_in else x
    own = _Owner()
    assert ops.split_linear_ln_ok(x, w, gamma)
    n32, n16 = ops.split_linear_ln(a, own, "w", w, b, resid, gamma, beta, eps, x_hl8=hl8_in)
    proj = ops.gemm(a, ops.hl8_pack(w), b, split=True)
    want32, want16, _ = ops.add_layernorm_dec(resid, proj, gamma, beta, eps, "hl8", want16=True)
    assert rel_err(n32.cpu(), want32.cpu()) < 2e-6
    assert torch.equal(n16, ops.to_hl8(n32))
    ref = torch.nn.functional.layer_norm(resid.double() + x.double() @ w.double().t() + b.double(), (256,), gamma.double(), beta.double(), eps)
    assert rel_err(n32.cpu(), ref.float().cpu()) < 5e-6
    only32, none16 = ops.split_linear_ln(a, own, "w", w, b, resid, gamma, beta, eps, x_hl8=hl8_in, want_hl8=False)
    assert none16 is None and torch.equal(only32, n32)


@gpu
def test_decoder_values_batched_in_the_split_policy():
    """split policy: the value projections of all decoder layers as ONE thin-K split GEMM over the shared memory (transformer.
    batched_decoder_values) equal the per-layer projections (ms_deform_attn.py:95-99 once per layer), and the decoder output does not move."""
    import hipie_amd.modeling.transformer as T
    torch.manual_seed(3)
    dec = T.DeformableTransformerDecoder(256, T.DeformableTransformerDecoderLayer(256, 512, 4, 8, 4, torch.float32), 3).cuda()
    for p in dec.parameters():
        torch.nn.init.normal_(p, std=0.05)
    dec.bbox_embed = torch.nn.ModuleList([T.MLP(256, 256, 4, 3).cuda() for _ in range(3)])
    T.set_split(dec)
    shapes = [(32, 40), (16, 20), (8, 10), (4, 5)]
    S = sum(h * w for h, w in shapes)
    ss = torch.tensor(shapes, device="cuda")
    ls = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    B, Q = 6, 50                                        # 6 x 1700 rows: the thin-K kernel (M >= 8192)
    src = torch.randn(B, S, 256, device="cuda")
    mask = torch.zeros(B, S, dtype=torch.bool, device="cuda")
    mask[1, -37:] = True
    assert T.decoder_split_values(dec, src)
    vals = T.batched_decoder_values(dec, dec.layers, src, mask)
    for lid, layer in enumerate(dec.layers):
        want = layer.cross_attn.project_value(src, mask)
        assert vals[lid].shape == want.shape and rel_err(vals[lid].cpu(), want.cpu()) < 1e-6
    tgt = torch.randn(B, Q, 256, device="cuda")
    ref = torch.rand(B, Q, 4, device="cuda") * 0.5 + 0.25
    vr = torch.ones(B, 4, 2, device="cuda")
    got, got_refs = dec(tgt, ref, src, ss, ls, vr, mask)
    keep = T.decoder_split_values
    T.decoder_split_values = lambda *a: False
    try:
        want, want_refs = dec(tgt, ref, src, ss, ls, vr, mask)
    finally:
        T.decoder_split_values = keep
    assert rel_err(got.cpu(), want.cpu()) < 2e-6 and rel_err(got_refs.cpu(), want_refs.cpu()) < 2e-6
