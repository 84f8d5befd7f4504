"""Operator layer: torch.Tensor in, torch.Tensor out, raw device pointers + current HIP stream across the C ABI.

Mirrors the reference's operator boundary: ``ms_deform_attn_forward`` has the signature and error
behaviour of the pybind module ``MultiScaleDeformableAttention`` (contiguity / device checks raise RuntimeError, CPU
tensors raise "Not implemented on the CPU", ops/src/ms_deform_attn.h:28-39).  All functions are inference-only.
"""
import torch

from . import _lib

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
F32, F16, BF16, HL8 = 0, 1, 2, 4          # include/hipie_mi355.h: HIPIE_F32 / F16 / BF16 / HL8
OUT_F32 = 0x100                           # HIPIE_OUT_F32
K_HL8_HI = 0x200       # hipie_flash_attn: keys are the hi halves of an HL8 buffer (HIPIE_K_HL8_HI)


class _Profile(object):
    """Optional live timing of the hand-written kernels with HIP events recorded on the launch stream (torch's current
    stream, the one every entry point is launched on).  Used by bench.py for the `roofline` object."""

    def __init__(self):
        self.tags, self.events = set(), {}
        self.shapes = False            # tools/stage_times.py: tag the generic GEMM launches by (M, N, K, formats)

    def enable(self, *tags):
        self.tags, self.events = set(tags), {}

    def disable(self):
        self.tags = set()

    def begin(self, tag):
        if tag not in self.tags and "all" not in self.tags:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        self.events.setdefault(tag, []).append((a, b))
        return b

    def mean_ms(self, tag):
        ev = self.events.get(tag, [])
        if not ev:
            return None, 0
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in ev]
        return sum(ts) / len(ts), len(ts)

    def summary(self):
        return {t: self.mean_ms(t) + (sum(a.elapsed_time(b) for a, b in self.events[t]),) for t in self.events}


PROFILE = _Profile()


import os as _os
import sys as _sys

_DEBUG_SYNC = _os.environ.get("HIPIE_DEBUG_SYNC") == "1"      # debugging aid: synchronise + log around every custom op


def _timed(tag):
    def deco(fn):
        def wrapper(*a, **k):
            if _DEBUG_SYNC:
                torch.cuda.synchronize()
                print("[hipie] > %s %s" % (fn.__name__, [tuple(t.shape) for t in a if torch.is_tensor(t)]), file=_sys.stderr, flush=True)
                out = fn(*a, **k)
                torch.cuda.synchronize()
                print("[hipie] < %s" % fn.__name__, file=_sys.stderr, flush=True)
                return out
            end = PROFILE.begin(tag(*a, **k) if callable(tag) else tag) if PROFILE.tags else None
            out = fn(*a, **k)
            if end is not None:
                end.record()
            return out
        wrapper.__doc__ = fn.__doc__
        wrapper.__name__ = fn.__name__
        return wrapper
    return deco


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _code(dtype):
    """C-ABI element code of an output type: a torch dtype, or the string "hl8" (split fp16, HIPIE_HL8)."""
    return HL8 if dtype == "hl8" else _DT[dtype]


def _alloc(shape, dtype, device):
    """output buffer of logical ``shape`` in ``dtype``; "hl8": fp16 with the last dimension doubled."""
    if dtype == "hl8":
        return torch.empty(*shape[:-1], 2 * shape[-1], dtype=torch.float16, device=device)
    return torch.empty(shape, dtype=dtype, device=device)


def _chk(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError("Not implemented on the CPU (%s must be a CUDA/HIP tensor)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s tensor has to be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t.data_ptr()


def _check_im2col_step(B, im2col_step):
    """the reference op processes the batch in chunks of min(B, im2col_step) images and asserts that the chunk divides the batch
    (ms_deform_attn_cuda.cu:50-52, :112-114); this kernel has no chunking (one launch over B), but a call the reference would refuse
    is refused here too, with its message."""
    step = min(int(B), int(im2col_step))
    if step <= 0 or B % step != 0:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (B, step))


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """MSDA.ms_deform_attn_forward (ops/src/vision.cpp:13-16): value (B,S,M,D), shapes (L,2) i64, level_start (L,) i64,
    sampling_loc (B,Lq,M,L,P,2), attn_weight (B,Lq,M,L,P) -> (B,Lq,M*D).  value may be f32/f16/bf16 with f32 loc/attn, or all f64."""
    lib = _lib.load()
    B, S, M, D = value.shape
    _check_im2col_step(B, im2col_step)
    _, Lq, _, L, P, _ = sampling_loc.shape
    f64 = value.dtype == torch.float64              # the reference op's double instantiation (ops/test.py checks it)
    if value.dtype not in _DT and not f64:
        raise RuntimeError("ms_deform_attn_forward: unsupported dtype %s" % value.dtype)
    aux = torch.float64 if f64 else torch.float32
    out = torch.empty(B, Lq, M * D, dtype=value.dtype, device=value.device)
    rc = lib.hipie_msda_forward(_chk(value, "value"), _chk(spatial_shapes, "spatial_shapes", torch.int64),
                                _chk(level_start_index, "level_start_index", torch.int64),
                                _chk(sampling_loc, "sampling_loc", aux),
                                _chk(attn_weight, "attn_weight", aux), out.data_ptr(),
                                B, S, M, D, L, Lq, P, 3 if f64 else _DT[value.dtype], _stream())
    _lib.check(rc, "hipie_msda_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step=64):
    """MSDA.ms_deform_attn_backward (ops/src/vision.cpp:13-16; ms_deform_attn_cuda.cu:83-153): the gradients of the op above with
    respect to value, sampling_loc and attn_weight, all f32 or all f64 -> (grad_value, grad_sampling_loc, grad_attn_weight).
    Runs hipie_msda_backward_ws: the gather form for fp32 and D = 32, the atomic kernel of hipie_msda_backward otherwise."""
    lib = _lib.load()
    B, S, M, D = value.shape
    _check_im2col_step(B, im2col_step)
    _, Lq, _, L, P, _ = sampling_loc.shape
    dt = value.dtype
    if dt not in (torch.float32, torch.float64):
        raise RuntimeError("ms_deform_attn_backward: float32 / float64 only (got %s)" % dt)
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU (ms_deform_attn_backward)")
    gv = torch.empty_like(value, memory_format=torch.contiguous_format)
    gl = torch.empty(sampling_loc.shape, dtype=dt, device=value.device)
    ga = torch.empty(attn_weight.shape, dtype=dt, device=value.device)
    need = int(lib.hipie_msda_backward_workspace(B, S, M, L, Lq, P)) if dt == torch.float32 and D == 32 and B * Lq > 0 else 0
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=value.device)      # the gather form's bins and corner records (fp32, D = 32)
    rc = lib.hipie_msda_backward_ws(_chk(value, "value", dt), _chk(spatial_shapes, "spatial_shapes", torch.int64),
                                    _chk(level_start_index, "level_start_index", torch.int64), _chk(sampling_loc, "sampling_loc", dt),
                                    _chk(attn_weight, "attn_weight", dt), _chk(grad_output.contiguous(), "grad_output", dt),
                                    gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, S, M, D, L, Lq, P, 3 if dt == torch.float64 else F32,
                                    ws.data_ptr(), need, _stream())
    _lib.check(rc, "hipie_msda_backward")
    return gv, gl, ga


@_timed("msda")
def msda_fused(value, spatial_shapes, level_start_index, ref, offsets, logits):
    """value (B,S,M,D); ref (B,Lq,L,2|4) f32; offsets (B,Lq,M,L,P,2); logits (B,Lq,M,L*P) (f32/f16/bf16, same dtype; they
    may be column slices of ONE projection output: only the last dim block must be contiguous) -> (B,Lq,M*D)."""
    lib = _lib.load()
    B, S, M, D = value.shape
    vrow = M * D
    if not value.is_contiguous():             # a column block of a wider (B, S, n * M * D) projection output: sampled in place
        vrow = value.stride(1)
        if value.stride(3) != 1 or value.stride(2) != D or value.stride(0) != S * vrow or vrow % 8 or vrow < M * D:
            raise RuntimeError("msda_fused: value must be dense or a column block of a dense (B, S, k*M*D) tensor")
    _, Lq, _, L, P, _ = offsets.shape
    if offsets.dtype != logits.dtype or offsets.dtype not in _DT:
        raise RuntimeError("msda_fused: offsets/logits must share a dtype in f32/f16/bf16")
    if B == 0 or Lq == 0:
        return torch.empty(B, Lq, M * D, dtype=value.dtype, device=value.device)
    off_stride, lg_stride = offsets.stride(1), logits.stride(1)
    if offsets.stride(0) != Lq * off_stride or logits.stride(0) != Lq * lg_stride or \
            offsets[0, 0].stride() != (L * P * 2, P * 2, 2, 1) or logits[0, 0].stride() != (L * P, 1):
        raise RuntimeError("msda_fused: offsets/logits rows must be dense (M,L,P,2)/(M,L*P) blocks")
    for t, n in ((offsets, "offsets"), (logits, "logits")):
        if not t.is_cuda:
            raise RuntimeError("Not implemented on the CPU (%s)" % n)
    out = torch.empty(B, Lq, M * D, dtype=value.dtype, device=value.device)
    if not value.is_cuda or value.dtype not in _DT:
        raise RuntimeError("Not implemented on the CPU (value)" if not value.is_cuda else "msda_fused: bad value dtype")
    # dense values go in with row stride 0 (= M * D inside the library): the % 8 requirement applies to strided column blocks only
    rc = lib.hipie_msda_fused_forward_strided(value.data_ptr(), 0 if value.is_contiguous() else vrow,
                                              _chk(spatial_shapes, "spatial_shapes", torch.int64),
                                              _chk(level_start_index, "level_start_index", torch.int64),
                                              _chk(ref, "ref", torch.float32), offsets.data_ptr(), logits.data_ptr(), out.data_ptr(),
                                              B, S, M, D, L, Lq, P, ref.shape[-1], _DT[value.dtype], _DT[offsets.dtype],
                                              off_stride, lg_stride, _stream())
    _lib.check(rc, "hipie_msda_fused_forward")
    return out


@_timed("flash_attn")
def flash_attn(q, k, v, scale, bias_h=None, bias_w=None, key_mask=None, clamp=0.0, out_f32=False, k_hl8=False):
    """q (B,Nq,H,hd), k,v (B,Nk,H,hd) 16-bit (may be strided views with hd contiguous) -> (B,Nq,H*hd) in the operand dtype, or
    fp32 with out_f32.  bias_h (B*H,kh,Nq) / bias_w (B*H,Nq,kw) f32 decomposed rel-pos bias; key_mask (B,Nk) uint8/bool.
    k_hl8: k is a contiguous (B, Nk, 2*H*hd) fp16 HL8 buffer whose hi halves are the keys (read in place, HIPIE_K_HL8_HI)."""
    lib = _lib.load()
    B, Nq, H, hd = q.shape
    Nk = k.shape[1]
    if k_hl8:
        if k.dtype != torch.float16 or q.dtype != torch.float16 or not k.is_contiguous() or tuple(k.shape) != (B, Nk, 2 * H * hd):
            raise RuntimeError("flash_attn: k_hl8 expects a contiguous (B, Nk, 2*H*hd) fp16 HL8 buffer and fp16 q / v")
        ks = (Nk * 2 * H * hd, 2 * H * hd, 2 * hd)
    for t, n in ((q, "q"), (v, "v")) + (() if k_hl8 else ((k, "k"),)):
        if not t.is_cuda:
            raise RuntimeError("Not implemented on the CPU (%s)" % n)
        if t.stride(-1) != 1 or t.dtype != q.dtype or t.dtype not in (torch.float16, torch.bfloat16):
            raise RuntimeError("flash_attn: %s must be fp16/bf16 with contiguous head_dim" % n)
    out = torch.empty(B, Nq, H * hd, dtype=torch.float32 if out_f32 else q.dtype, device=q.device)
    kh = kw = 0
    bhp = bwp = mp = None
    if bias_h is not None:
        kh, kw = bias_h.shape[-2], bias_w.shape[-1]
        bhp, bwp = _chk(bias_h, "bias_h", torch.float32), _chk(bias_w, "bias_w", torch.float32)
    if key_mask is not None:
        key_mask = key_mask.to(torch.uint8).contiguous()
        mp = _chk(key_mask, "key_mask")
    rc = lib.hipie_flash_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Nq, Nk, hd,
                              q.stride(0), q.stride(1), q.stride(2), *(ks if k_hl8 else (k.stride(0), k.stride(1), k.stride(2))),
                              v.stride(0), v.stride(1), v.stride(2), Nq * H * hd, H * hd, hd,
                              bhp, bwp, kh, kw, mp, float(scale), float(clamp),
                              _DT[q.dtype] | (OUT_F32 if out_f32 else 0) | (K_HL8_HI if k_hl8 else 0), _stream())
    _lib.check(rc, "hipie_flash_attn")
    return out


@_timed("bi_i2t_split")
def bi_i2t_split(q_hl8, k, vl, text_mask, heads, clamp=50000.0, n_keys=None):
    """image -> text direction of the vision-language fusion at fp32-class accuracy (fuse_helper.py:77-121):
        out_v[b, i, h] = softmax_j( clamp(q[b,i,h] . k[b,j,h]) over the text tokens kept by text_mask ) . vl[b, j, h]
    q_hl8 (B, Nv, 2*E) HL8 (E = heads * hd; the scaled v_proj output straight from the GEMM epilogue), k, vl (B, L, E) fp32,
    text_mask (B, L) -> (B, Nv, E) fp32.  Texts of up to 256 tokens: TWO launches -- the batched logits GEMM with the masked row softmax in
    its epilogue (P as HL8; S never exists) and P_h.V_h as the second batched GEMM; longer texts: S = Q_h.K_h^T, hipie_softmax_hl8, P_h.V_h; the text-side operands (L x E) are padded to a
    multiple of 32 tokens and split on the way.  n_keys (host int, optional): every attended token sits in the first n_keys columns --
    the masked keys behind them have probability exactly 0 (-9e15 before the softmax) and are left out, so a PAD_MAX prompt (4096
    columns, 194 of them attended) costs what its real tokens cost and nothing of size Nv x L is allocated."""
    lib = _lib.load()
    B, Nv, E2 = q_hl8.shape
    E = E2 // 2
    hd = E // heads
    if n_keys is not None and 0 < n_keys < k.shape[1]:
        k, vl, text_mask = k[:, :n_keys], vl[:, :n_keys], text_mask[:, :n_keys]
    L = k.shape[1]
    Lp = (L + 31) // 32 * 32
    if q_hl8.dtype != torch.float16 or not q_hl8.is_contiguous() or k.dtype != torch.float32 or vl.dtype != torch.float32 or hd % 32:
        raise RuntimeError("bi_i2t_split: q HL8 (fp16) contiguous, k / vl fp32, head_dim a multiple of 32")
    dev = q_hl8.device
    kp = torch.zeros(B, Lp, E, dtype=torch.float32, device=dev)
    kp[:, :L] = k
    k_hl8 = to_hl8(kp)                                                        # (B, Lp, 2E): head h = columns [2 hd h, 2 hd (h + 1))
    vt = torch.zeros(B, heads, hd, Lp, dtype=torch.float32, device=dev)
    vt[..., :L] = vl.reshape(B, L, heads, hd).permute(0, 2, 3, 1)
    vt_hl8 = to_hl8(vt)                                                       # (B, heads, hd, 2 Lp): V_h^T, K = tokens
    P = torch.empty(B, heads, Nv, 2 * Lp, dtype=torch.float16, device=dev)
    mk = text_mask.to(torch.uint8).contiguous()
    if Lp <= 256:
        # a text of up to 256 tokens is ONE column tile: the masked softmax runs in the logits GEMM's epilogue, S never exists
        rc = lib.hipie_gemm_batched_softmax(q_hl8.data_ptr(), 2 * E, Nv * 2 * E, 2 * hd, k_hl8.data_ptr(), 2 * E, Lp * 2 * E, 2 * hd,
                                            P.data_ptr(), 2 * Lp, heads * Nv * 2 * Lp, Nv * 2 * Lp, B, heads, Nv, Lp, hd, mk.data_ptr(), L,
                                            float(clamp), 1.0, _stream())
        _lib.check(rc, "hipie_gemm_batched_softmax")
    else:
        S = torch.empty(B, heads, Nv, Lp, dtype=torch.float32, device=dev)
        rc = lib.hipie_gemm_batched(q_hl8.data_ptr(), 2 * E, Nv * 2 * E, 2 * hd, k_hl8.data_ptr(), 2 * E, Lp * 2 * E, 2 * hd,
                                    S.data_ptr(), Lp, heads * Nv * Lp, Nv * Lp, B, heads, Nv, Lp, hd, F32, 1.0, _stream())
        _lib.check(rc, "hipie_gemm_batched")
        rc = lib.hipie_softmax_hl8(S.data_ptr(), Lp, P.data_ptr(), 2 * Lp, B * heads * Nv, L, Lp, mk.data_ptr(), heads * Nv, float(clamp), _stream())
        _lib.check(rc, "hipie_softmax_hl8")
    out = torch.empty(B, Nv, E, dtype=torch.float32, device=dev)
    rc = lib.hipie_gemm_batched(P.data_ptr(), 2 * Lp, heads * Nv * 2 * Lp, Nv * 2 * Lp, vt_hl8.data_ptr(), 2 * Lp, heads * hd * 2 * Lp, hd * 2 * Lp,
                                out.data_ptr(), E, Nv * E, hd, B, heads, Nv, hd, Lp, F32, 1.0, _stream())
    _lib.check(rc, "hipie_gemm_batched")
    return out


def bi_i2t_folded_ok(v, L_text, n_keys=None):
    """the folded vision-language attention applies: at most 256 attended text columns (one column tile of the softmax epilogue), a visual
    width the split GEMM takes (and the fp16 attention kernel as a head dim: 256), inference"""
    L = L_text if not n_keys else min(int(n_keys), L_text)
    return v.is_cuda and v.dtype == torch.float32 and v.shape[-1] == 256 and 0 < L <= 256 and not torch.is_grad_enabled()


@_timed("bi_i2t_folded")
def bi_i2t_folded(v_hl8, M, cb, vl, text_mask, heads, wo, bo, resid=None, clamp=50000.0):
    """image -> text direction of the vision-language fusion (fuse_helper.py:62-139) at fp32-class accuracy WITHOUT the two visual-side
    projections of width E = heads * hd:
        logits[b,h][i,j] = (wq_h x_i + bq_h) . k[b,j,h] = x_i . M[b,h,j] + cb[b,h,j]       M = k_h wq_h (L x C), cb = k_h . bq_h
        out[b,i]         = wo concat_h(P_h[i] vl_h) + bo (+ resid) = sum_h P_h[i] . U[b,h] + bo      U = vl_h wo_h^T (L x C_out)
    v_hl8 (B, Nv, 2C) HL8 of the (layer-normed) visual stream x; M (B, H, L, C), cb (B, H, L) fp32 (the caller's small text-side products,
    already cut to the attended columns); vl (B, L, E) fp32 text values; wo (C_out, E), bo (C_out) the output projection; resid (B, Nv, C_out)
    fp32 or None -> (B, Nv, C_out) fp32.  TWO launches over the visual tokens (hipie_gemm_batched_softmax_bias: P as HL8, the logits never
    exist; hipie_gemm_batched_resid: K = heads * Lp).  L <= 256.  The reassociation moves the rounding of the logits at the 1e-7 level."""
    lib = _lib.load()
    B, Nv, C2 = v_hl8.shape
    C = C2 // 2
    L = M.shape[2]
    E = vl.shape[-1]
    hd = E // heads
    Lp = (L + 31) // 32 * 32
    Co = wo.shape[0]
    if v_hl8.dtype != torch.float16 or not v_hl8.is_contiguous() or M.dtype != torch.float32 or vl.dtype != torch.float32 or Lp > 256 or Co % 8 \
            or tuple(M.shape) != (B, heads, L, C) or tuple(cb.shape) != (B, heads, L) or vl.shape[1] != L:
        raise RuntimeError("bi_i2t_folded: x HL8 (fp16) contiguous, M (B, H, L, C) / cb (B, H, L) / vl (B, L, E) fp32, at most 256 text columns")
    dev = v_hl8.device
    Mp = torch.zeros(B, heads, Lp, C, dtype=torch.float32, device=dev)
    Mp[:, :, :L] = M
    cbp = torch.zeros(B, heads, Lp, dtype=torch.float32, device=dev)
    cbp[:, :, :L] = cb
    m_hl8 = to_hl8(Mp)                                                                 # (B, H, Lp, 2C)
    U = torch.zeros(B, Co, heads, Lp, dtype=torch.float32, device=dev)
    U[..., :L] = torch.einsum("nhd,blhd->bnhl", wo.float().view(Co, heads, hd), vl.reshape(B, L, heads, hd))
    u_hl8 = to_hl8(U.view(B, Co, heads * Lp))                                          # (B, C_out, 2 H Lp): K = (head, text token)
    P = torch.empty(B, Nv, heads * 2 * Lp, dtype=torch.float16, device=dev)
    mk = text_mask.to(torch.uint8).contiguous()
    rc = lib.hipie_gemm_batched_softmax_bias(v_hl8.data_ptr(), 2 * C, Nv * 2 * C, 0, m_hl8.data_ptr(), 2 * C, heads * Lp * 2 * C, Lp * 2 * C,
                                             P.data_ptr(), heads * 2 * Lp, Nv * heads * 2 * Lp, 2 * Lp, B, heads, Nv, Lp, C, mk.data_ptr(), L,
                                             cbp.data_ptr(), float(clamp), 1.0, _stream())
    _lib.check(rc, "hipie_gemm_batched_softmax_bias")
    out = torch.empty(B, Nv, Co, dtype=torch.float32, device=dev)
    r = None
    if resid is not None:
        r = resid.reshape(B, Nv, Co)
        if r.dtype != torch.float32 or not r.is_contiguous():
            raise RuntimeError("bi_i2t_folded: resid must be contiguous fp32 (B, Nv, C_out)")
    bo32 = None if bo is None else bo.float().contiguous()
    KK = heads * Lp
    rc = lib.hipie_gemm_batched_resid(P.data_ptr(), 2 * KK, Nv * 2 * KK, 0, u_hl8.data_ptr(), 2 * KK, Co * 2 * KK, 0,
                                      None if bo32 is None else bo32.data_ptr(), None if r is None else r.data_ptr(),
                                      Co, Nv * Co, 0, out.data_ptr(), Co, Nv * Co, 0, B, 1, Nv, Co, KK, 1.0, _stream())
    _lib.check(rc, "hipie_gemm_batched_resid")
    return out


def _small_attn(fn_name, q, k, v, scale, key_mask):
    lib = _lib.load()
    B, Nq, H, hd = q.shape
    Nk = k.shape[1]
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda:
            raise RuntimeError("Not implemented on the CPU (%s)" % n)
        if t.dtype != torch.float32 or t.stride(-1) != 1 or (H > 1 and t.stride(2) != hd):
            raise RuntimeError("%s: %s must be fp32 (B, N, H, hd) with hd contiguous and heads hd apart" % (fn_name, n))
    out = torch.empty(B, Nq, H * hd, dtype=torch.float32, device=q.device)
    mp = None
    if key_mask is not None:
        key_mask = key_mask.to(torch.uint8).contiguous()
        mp = key_mask.data_ptr()
    rc = getattr(lib, fn_name)(q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, out.data_ptr(), B, H, Nq, Nk, hd, q.stride(0), q.stride(1),
                               k.stride(0), k.stride(1), v.stride(0), v.stride(1), float(scale), _stream())
    _lib.check(rc, fn_name)
    return out


@_timed("attn_f32")
def attn_f32(q, k, v, scale, key_mask=None):
    """hipie_attn_f32: exact fp32 softmax attention.  q (B,Nq,H,hd), k, v (B,Nk,H,hd) fp32 views with hd contiguous and heads hd apart
    (column blocks of one projection output are fine); hd 32 | 64; key_mask (B,Nk) bool / uint8 -> (B, Nq, H*hd) fp32."""
    return _small_attn("hipie_attn_f32", q, k, v, scale, key_mask)


@_timed("attn_rows")
def attn_f32_rows(q, k, v, scale, query_mask, split=False):
    """hipie_attn_f32_rows (exact fp32 FMA) / hipie_attn_split_rows (split=True: matrix pipe, fp32-class): attention with a mask per query row --
    query_mask (B, Nq, Nk) bool / uint8, True = may attend"""
    lib = _lib.load()
    B, Nq, H, hd = q.shape
    Nk = k.shape[1]
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1 or (H > 1 and t.stride(2) != hd):
            raise RuntimeError("attn_f32_rows: %s must be an fp32 device (B, N, H, hd) with hd contiguous and heads hd apart" % n)
    if tuple(query_mask.shape) != (B, Nq, Nk):
        raise RuntimeError("attn_f32_rows: query_mask (B, Nq, Nk), got %s" % (tuple(query_mask.shape),))
    qm = query_mask.to(torch.uint8).contiguous()
    out = torch.empty(B, Nq, H * hd, dtype=torch.float32, device=q.device)
    fn = lib.hipie_attn_split_rows if split else lib.hipie_attn_f32_rows
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), qm.data_ptr(), out.data_ptr(), B, H, Nq, Nk, hd, q.stride(0), q.stride(1),
            k.stride(0), k.stride(1), v.stride(0), v.stride(1), float(scale), _stream())
    _lib.check(rc, "hipie_attn_split_rows" if split else "hipie_attn_f32_rows")
    return out


@_timed("attn_split")
def attn_split(q, k, v, scale, key_mask=None):
    """hipie_attn_split: the same attention on the matrix pipe at fp32-class accuracy (operands split into fp16 pairs in the kernel, three
    products per logit, probabilities as an fp16 pair): what the split policy runs for BERT and the decoders' query self-attention."""
    return _small_attn("hipie_attn_split", q, k, v, scale, key_mask)


def attn_f32_ok(hd):
    return hd in (32, 64)


@_timed(lambda qkv, rel_h, rel_w, grid_hw, heads, scale: "vit_attn_global" if grid_hw[0] * grid_hw[1] > 256 else "vit_attn_window")
def vit_attn(qkv, rel_h, rel_w, grid_hw, heads, scale):
    """qkv (B, gh*gw, 3*heads*hd) 16-bit packed as (3, heads, hd); rel_h (B*heads, gh, N) f32 (key-row major),
    rel_w (B*heads, N, gw) f32 -> (B, N, heads*hd)."""
    lib = _lib.load()
    gh, gw = grid_hw
    B, N, C3 = qkv.shape
    hd = C3 // (3 * heads)
    out = torch.empty(B, N, heads * hd, dtype=qkv.dtype, device=qkv.device)
    rc = lib.hipie_vit_attn(_chk(qkv, "qkv"), _chk(rel_h, "rel_h", torch.float32), _chk(rel_w, "rel_w", torch.float32),
                            out.data_ptr(), B, gh, gw, heads, hd, float(scale), _DT[qkv.dtype], _stream())
    _lib.check(rc, "hipie_vit_attn")
    return out


def vit_attn_fused_ok(grid_hw, hd):
    """the geometry hipie_vit_attn_fused covers: a 64-wide token grid (1024-pixel images), <= 64 rows, head_dim 64 / 80."""
    return grid_hw[1] == 64 and 1 <= grid_hw[0] <= 64 and hd in (64, 80)


@_timed("vit_attn_global")
def vit_attn_fused(qkv, tab_h, tab_w, grid_hw, heads, scale):
    """global ViT attention with the decomposed rel-pos bias computed in the kernel prologue from the tables:
    qkv (B, gh*gw, 3*heads*hd) 16-bit, tab_h (2*gh-1, hd), tab_w (2*gw-1, hd) same dtype -> (B, N, heads*hd)."""
    lib = _lib.load()
    gh, gw = grid_hw
    B, N, C3 = qkv.shape
    hd = C3 // (3 * heads)
    out = torch.empty(B, N, heads * hd, dtype=qkv.dtype, device=qkv.device)
    rc = lib.hipie_vit_attn_fused(_chk(qkv, "qkv"), _chk(tab_h, "tab_h", qkv.dtype), _chk(tab_w, "tab_w", qkv.dtype),
                                  out.data_ptr(), B, gh, gw, heads, hd, float(scale), _DT[qkv.dtype], _stream())
    _lib.check(rc, "hipie_vit_attn_fused")
    return out


LOG2E = 1.4426950408889634
ATTN_FAST = 1            # include/hipie_mi355.h: HIPIE_ATTN_FAST


def vit_attn_rel_ok(grid_hw, hd):
    """the geometry hipie_vit_attn_rel covers: head_dim 64 / 80, token grids up to 96 wide (84 rows when wider than 64)."""
    gh, gw = grid_hw
    return hd in (64, 80) and 1 <= gw <= 96 and 1 <= gh <= (160 if gw <= 64 else 84)


@_timed(lambda qkv, tab_h, tab_w, grid_hw, heads, fast=False: "vit_attn_global" if grid_hw[0] * grid_hw[1] > 256 else "vit_attn_window")
def vit_attn_rel(qkv, tab_h, tab_w, grid_hw, heads, fast=False):
    """ViT attention (global or windowed) with the decomposed rel-pos bias computed in the kernel.  Operand contract of
    hipie_vit_attn_rel: qkv (B, gh*gw, 3*heads*hd) 16-bit whose q rows are PRE-SCALED by scale*log2(e); tab_h (2*gh-1, hd),
    tab_w (2*gw-1, hd) = the (re-interpolated) rel-pos tables DIVIDED by scale, same dtype -> (B, N, heads*hd)."""
    lib = _lib.load()
    gh, gw = grid_hw
    B, N, C3 = qkv.shape
    hd = C3 // (3 * heads)
    out = torch.empty(B, N, heads * hd, dtype=qkv.dtype, device=qkv.device)
    rc = lib.hipie_vit_attn_rel(_chk(qkv, "qkv"), _chk(tab_h, "tab_h", qkv.dtype), _chk(tab_w, "tab_w", qkv.dtype),
                                out.data_ptr(), B, gh, gw, heads, hd, _DT[qkv.dtype], ATTN_FAST if fast else 0, _stream())
    _lib.check(rc, "hipie_vit_attn_rel")
    return out


class _Workspace(object):
    """scratch memory of a kernel family, one buffer per (device, stream): concurrent streams never share a buffer.  A buffer that turns
    out too small is replaced by one of at least twice the size and the old one is KEPT ALIVE (a captured hipGraph may still hold its
    address; geometric growth bounds what is retained by the final size).  Single assumption left: launches on one stream are ordered."""

    def __init__(self):
        self.bufs, self.retired = {}, []

    def get(self, nbytes, device):
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            if b is not None:
                self.retired.append(b)
            b = self.bufs[key] = torch.empty(max(int(nbytes), 2 * (b.numel() if b is not None else 0)), dtype=torch.uint8, device=device)
        return b


_XA_WS = _Workspace()


@_timed("bi_xattn")
def bi_xattn(q, k, vv, vl, text_mask, clamp=50000.0, out_f32=False):
    """q, vv (B,Nv,H,hd); k, vl (B,L,H,hd) 16-bit contiguous; text_mask (B,L) -> out_v (B,Nv,H*hd), out_l (B,L,H*hd) (operand dtype,
    or fp32 with out_f32: the generic flash kernel then runs both directions)."""
    lib = _lib.load()
    B, Nv, H, hd = q.shape
    L = k.shape[1]
    text_mask = text_mask.to(torch.uint8).contiguous()
    odt = torch.float32 if out_f32 else q.dtype
    out_v = torch.empty(B, Nv, H * hd, dtype=odt, device=q.device)
    out_l = torch.empty(B, L, H * hd, dtype=odt, device=q.device)
    need = 0 if out_f32 else int(lib.hipie_bi_xattn_workspace(B, H, Nv, L, hd))          # split partials of the text -> image direction (0: none)
    ws = _XA_WS.get(need, q.device) if need > 0 else None
    rc = lib.hipie_bi_xattn_ws(_chk(q, "q"), _chk(k, "k"), _chk(vv, "vv"), _chk(vl, "vl"), _chk(text_mask, "text_mask"),
                               out_v.data_ptr(), out_l.data_ptr(), None if ws is None else ws.data_ptr(), need, B, H, Nv, L, hd,
                               float(clamp), _DT[q.dtype] | (OUT_F32 if out_f32 else 0), _stream())
    _lib.check(rc, "hipie_bi_xattn")
    return out_v, out_l


_ME_WS = _Workspace()


@_timed("mask_einsum")
def mask_einsum(mask_embed, mask_features, precision=1, out_dtype=torch.float32, row_bias=None, workspace=True):
    """einsum("bqc,bchw->bqhw"): mask_embed (B,Q,C) f32, mask_features (B,C,H,W) f32 -> (B,Q,H,W) out_dtype.
    precision 0 = exact fp32 MFMA, 1 = bf16x3 split (default; ~2^-16), 2 = plain bf16.  Precisions 1 | 2 run hipie_mask_einsum_ws (the
    embedding split once into a cached scratch buffer and staged by LDS-DMA) and take row_bias (B,Q) f32, added to every pixel of
    its query row; workspace=False: the workspace-free entry point hipie_mask_einsum (every workgroup splits the embedding)."""
    lib = _lib.load()
    B, Q, C = mask_embed.shape
    _, _, Hh, Ww = mask_features.shape
    out = torch.empty(B, Q, Hh, Ww, dtype=out_dtype, device=mask_embed.device)
    e, f = _chk(mask_embed, "mask_embed", torch.float32), _chk(mask_features, "mask_features", torch.float32)
    if int(precision) == 0 or not workspace:
        if row_bias is not None:
            raise RuntimeError("mask_einsum: row_bias needs precision 1 | 2 and the workspace form")
        rc = lib.hipie_mask_einsum(e, f, out.data_ptr(), B, Q, C, Hh * Ww, int(precision), _DT[out_dtype], _stream())
    else:
        rb = None
        if row_bias is not None:
            rbt = row_bias.float().contiguous()
            if tuple(rbt.shape) != (B, Q):
                raise ValueError("mask_einsum: row_bias must be (B, Q) = (%d, %d), got %s" % (B, Q, tuple(rbt.shape)))
            rb = _chk(rbt, "row_bias", torch.float32)
        need = int(lib.hipie_mask_einsum_workspace(B, Q, C))
        ws = _ME_WS.get(need, mask_embed.device)
        rc = lib.hipie_mask_einsum_ws(e, f, rb, out.data_ptr(), ws.data_ptr(), need, B, Q, C, Hh * Ww, int(precision), _DT[out_dtype], _stream())
    _lib.check(rc, "hipie_mask_einsum")
    return out


def _pad_last(x, n):
    if x.shape[-1] == n:
        return x.contiguous()
    out = torch.zeros(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    out[..., :x.shape[-1]] = x
    return out


@_timed("mask_einsum_bwd")
def mask_einsum_backward(mask_embed, mask_features, grad_out):
    """gradients of einsum("bqc,bchw->bqhw") (row f-4): mask_embed (B,Q,C), mask_features (B,C,H,W), grad_out (B,Q,H,W) fp32 ->
    (grad_embed (B,Q,C), grad_features (B,C,H,W)) fp32, both at fp32-class accuracy on hipie_gemm_batched (split operands):
      grad_embed[b]    = G[b] (Q x HW) . F[b]^T (HW x C)   -- K = HW split into up to 32 chunks (one problem each: Q x C is 2-5 tiles
                         per image, far fewer than the CUs), the partial products summed afterwards;
      grad_features[b] = E[b]^T (C x Q) . G[b] (Q x HW)    -- as A . W^T with A = E^T and W = G^T (both K = Q contiguous, Q padded to 32).
    The row-bias gradient of the forward's `row_bias` is grad_out.sum((2, 3)) (the caller's)."""
    lib = _lib.load()
    B, Q, C = mask_embed.shape
    _, _, Hh, Ww = mask_features.shape
    HW = Hh * Ww
    if tuple(grad_out.shape) != (B, Q, Hh, Ww) or tuple(mask_features.shape[:2]) != (B, C) or not grad_out.is_cuda:
        raise RuntimeError("mask_einsum_backward: mask_embed (B,Q,C), mask_features (B,C,H,W), grad_out (B,Q,H,W) device tensors")
    if C % 8:
        raise RuntimeError("mask_einsum_backward: C must be a multiple of 8")
    dev = grad_out.device
    G = grad_out.float().reshape(B, Q, HW)
    F_ = mask_features.float().reshape(B, C, HW)
    # ---- grad_embed: K = HW in nk chunks ----
    HWp = (HW + 31) // 32 * 32
    nk = max(n for n in range(1, 33) if (HWp // 32) % n == 0)
    Kc = HWp // nk
    gh, fh = to_hl8(_pad_last(G, HWp)), to_hl8(_pad_last(F_, HWp))              # (B, Q | C, 2 HWp)
    part = torch.empty(B, nk, Q, C, dtype=torch.float32, device=dev)
    rc = lib.hipie_gemm_batched(gh.data_ptr(), 2 * HWp, Q * 2 * HWp, 2 * Kc, fh.data_ptr(), 2 * HWp, C * 2 * HWp, 2 * Kc,
                                part.data_ptr(), C, nk * Q * C, Q * C, B, nk, Q, C, Kc, F32, 1.0, _stream())
    _lib.check(rc, "hipie_gemm_batched")
    grad_e = part.sum(1) if nk > 1 else part[:, 0]
    # ---- grad_features: K = Q ----
    Qp = (Q + 31) // 32 * 32
    N8 = (HW + 7) // 8 * 8
    eT = to_hl8(_pad_last(mask_embed.float().transpose(1, 2), Qp))              # (B, C, 2 Qp)
    gT = torch.zeros(B, N8, Qp, dtype=torch.float32, device=dev) if (N8 != HW or Qp != Q) else torch.empty(B, N8, Qp, dtype=torch.float32, device=dev)
    gT[:, :HW, :Q] = G.transpose(1, 2)
    gTh = to_hl8(gT)                                                             # (B, N8, 2 Qp)
    gf = torch.empty(B, C, N8, dtype=torch.float32, device=dev)
    rc = lib.hipie_gemm_batched(eT.data_ptr(), 2 * Qp, C * 2 * Qp, 0, gTh.data_ptr(), 2 * Qp, N8 * 2 * Qp, 0, gf.data_ptr(), N8, C * N8, 0,
                                B, 1, C, N8, Qp, F32, 1.0, _stream())
    _lib.check(rc, "hipie_gemm_batched")
    if N8 != HW:
        gf = gf[..., :HW].contiguous()
    return grad_e, gf.view(B, C, Hh, Ww)


@_timed("mask_einsum")
def mask_einsum16(mask_embed, mask_features, split=True, out_dtype=None, row_bias=None):
    """einsum("bqc,bchw->bqhw") on 16-bit features: mask_embed (B,Q,C) f32 (split here into 16-bit hi + lo parts: two MFMAs per
    product; split=False: hi only), mask_features (B,C,H,W) f16 | bf16 contiguous -> (B,Q,H,W) out_dtype (default: the feature
    dtype); row_bias (B,Q) is added to every pixel of its query row.  Q <= 320."""
    lib = _lib.load()
    B, Q, C = mask_embed.shape
    _, _, Hh, Ww = mask_features.shape
    dt = mask_features.dtype
    out_dtype = out_dtype or dt
    hi = mask_embed.to(dt).contiguous()
    lo = (mask_embed.float() - hi.float()).to(dt).contiguous() if split else None
    out = torch.empty(B, Q, Hh, Ww, dtype=out_dtype, device=mask_embed.device)
    rb = None if row_bias is None else _chk(row_bias.float().contiguous(), "row_bias", torch.float32)
    rc = lib.hipie_mask_einsum16(_chk(hi, "embed_hi"), None if lo is None else _chk(lo, "embed_lo"), _chk(mask_features, "mask_features"),
                                 rb, out.data_ptr(), B, Q, C, Hh * Ww, _DT[dt], _DT[out_dtype], _stream())
    _lib.check(rc, "hipie_mask_einsum16")
    return out


@_timed("dynamic_mask")
def dynamic_mask(mask_feats, ref_points, params, num_queries, stride=8, up=2, out_dtype=torch.float32, mlp_dtype=None):
    """mask_feats (B,8,H,W) f32; ref_points (B*Q,2) f32 pixels; params (B*Q,169) f32 -> (B*Q, up*H, up*W).
    mlp_dtype fp16 / bf16: the three layers on the matrix pipe with operands of that type (hipie_dynamic_mask16; up = 2,
    W % 4 == 0, stride % 4 == 0); "split": the same kernel on fp16 PAIRS (weights and activations hi + lo, three products each: fp32-class,
    the split policy); None / fp32: the fp32 VALU kernel."""
    lib = _lib.load()
    B, C, H, W = mask_feats.shape
    if C != 8 or params.shape[-1] != 169:
        raise RuntimeError("dynamic_mask: expects 8 feature channels and 169 parameters per instance")
    n = B * num_queries
    out = torch.empty(n, up * H, up * W, dtype=out_dtype, device=mask_feats.device)
    split = mlp_dtype == "split" and out_dtype in (torch.float32, torch.float16)
    if (split or mlp_dtype in (torch.float16, torch.bfloat16)) and up == 2 and W % 4 == 0 and stride % 4 == 0 and stride * max(H, W) <= 8192:
        rc = lib.hipie_dynamic_mask16(_chk(mask_feats, "mask_feats", torch.float32), _chk(ref_points, "ref_points", torch.float32),
                                      _chk(params, "params", torch.float32), out.data_ptr(), B, num_queries, H, W,
                                      int(stride), HL8 if split else _DT[mlp_dtype], _DT[out_dtype], _stream())
        _lib.check(rc, "hipie_dynamic_mask16")
        return out
    rc = lib.hipie_dynamic_mask(_chk(mask_feats, "mask_feats", torch.float32), _chk(ref_points, "ref_points", torch.float32),
                                _chk(params, "params", torch.float32), out.data_ptr(), B, num_queries, H, W,
                                int(stride), int(up), _DT[out_dtype], _stream())
    _lib.check(rc, "hipie_dynamic_mask")
    return out


@_timed("dynamic_mask_bwd")
def dynamic_mask_backward(mask_feats, ref_points, params, grad_out, num_queries, stride=8, up=2):
    """gradients of dynamic_mask (row f-4; hipie_dynamic_mask_backward): mask_feats (B,8,H,W), ref_points (B*Q,2), params (B*Q,169),
    grad_out (B*Q, up*H, up*W) fp32 -> (grad_feats (B,8,H,W), grad_refs (B*Q,2), grad_params (B*Q,169)) fp32."""
    lib = _lib.load()
    B, C, H, W = mask_feats.shape
    n = B * num_queries
    if C != 8 or params.shape[-1] != 169 or params.numel() != n * 169 or ref_points.numel() != n * 2:
        raise RuntimeError("dynamic_mask_backward: expects 8 feature channels, (B*Q, 169) parameters and (B*Q, 2) reference points")
    if tuple(grad_out.shape[-2:]) != (up * H, up * W) or grad_out.numel() != n * up * up * H * W:
        raise RuntimeError("dynamic_mask_backward: grad_out must be (B*Q, up*H, up*W), got %s" % (tuple(grad_out.shape),))
    dev = mask_feats.device
    gf = torch.empty(B, 8, H, W, dtype=torch.float32, device=dev)
    gr = torch.empty(n, 2, dtype=torch.float32, device=dev)
    gp = torch.empty(n, 169, dtype=torch.float32, device=dev)
    rc = lib.hipie_dynamic_mask_backward(_chk(mask_feats, "mask_feats", torch.float32), _chk(ref_points, "ref_points", torch.float32),
                                         _chk(params, "params", torch.float32), _chk(grad_out, "grad_out", torch.float32),
                                         gf.data_ptr(), gr.data_ptr(), gp.data_ptr(), B, num_queries, H, W, int(stride), int(up), _stream())
    _lib.check(rc, "hipie_dynamic_mask_backward")
    return gf, gr, gp


@_timed("vit_relpos")
def vit_relpos(qkv, tab_h, tab_w, grid_hw, heads):
    """qkv (B, N, 3*heads*hd) 16-bit; tab_h (2gh-1, hd), tab_w (2gw-1, hd) same dtype -> rel_h (B*heads, gh, N), rel_w
    (B*heads, N, gw) f32, the bias tables of hipie_vit_attn."""
    lib = _lib.load()
    gh, gw = grid_hw
    B, N, C3 = qkv.shape
    hd = C3 // (3 * heads)
    rel_h = torch.empty(B * heads, gh, N, dtype=torch.float32, device=qkv.device)
    rel_w = torch.empty(B * heads, N, gw, dtype=torch.float32, device=qkv.device)
    rc = lib.hipie_vit_relpos(_chk(qkv, "qkv"), _chk(tab_h, "tab_h", qkv.dtype), _chk(tab_w, "tab_w", qkv.dtype),
                              rel_h.data_ptr(), rel_w.data_ptr(), B, gh, gw, heads, hd, _DT[qkv.dtype], _stream())
    _lib.check(rc, "hipie_vit_relpos")
    return rel_h, rel_w


@_timed("add_layernorm")
def add_layernorm(x, delta, weight, bias, eps, norm_dtype, want_res=True, delta_row=None, out_src=None):
    """s = x + delta (delta may be None); returns (s in x.dtype or None, LayerNorm(s) in norm_dtype).  x (..., C).
    Row maps (int32, optional): ``out_src`` (out_rows,) = x row feeding each output row (-1: zeros) -- the normalised output
    then has out_rows rows; ``delta_row`` (rows of x,) = row of ``delta`` that belongs to x row r."""
    lib = _lib.load()
    C = x.shape[-1]
    rows = x.numel() // C
    res = torch.empty_like(x) if (want_res and delta is not None) else None
    if out_src is None:
        out = _alloc(tuple(x.shape), norm_dtype, x.device)
        out_rows = rows
    else:
        out_rows = int(out_src.shape[0])
        out = _alloc((out_rows, C), norm_dtype, x.device)
    args = (_chk(x, "x"), None if delta is None else _chk(delta, "delta"),
            _chk(weight, "weight", torch.float32), _chk(bias, "bias", torch.float32),
            None if res is None else res.data_ptr(), out.data_ptr(), out_rows, C, float(eps),
            _DT[x.dtype], _DT[x.dtype if delta is None else delta.dtype], _code(norm_dtype))
    if delta_row is None and out_src is None:
        rc = lib.hipie_add_layernorm(*args, _stream())
    else:
        rc = lib.hipie_add_layernorm_rows(*args, None if delta_row is None else _chk(delta_row, "delta_row", torch.int32),
                                          None if out_src is None else _chk(out_src, "out_src", torch.int32), _stream())
    _lib.check(rc, "hipie_add_layernorm")
    return (x if delta is None else res), out


@_timed("add_layernorm")
def add_layernorm_sum(x, delta, weight, bias, eps, addend):
    """n = LayerNorm(x + delta) and n + addend, both in x's dtype, one launch (the encoder's post-norm + next `src + pos`)."""
    lib = _lib.load()
    C = x.shape[-1]
    rows = x.numel() // C
    if addend.dtype != x.dtype or addend.shape != x.shape:
        raise RuntimeError("add_layernorm_sum: addend must match x")
    out, s = torch.empty_like(x), torch.empty_like(x)
    rc = lib.hipie_add_layernorm_sum(_chk(x, "x"), _chk(delta, "delta"), _chk(weight, "weight", torch.float32),
                                     _chk(bias, "bias", torch.float32), None, out.data_ptr(), _chk(addend, "addend"), s.data_ptr(),
                                     rows, C, float(eps), _DT[x.dtype], _DT[delta.dtype], _DT[x.dtype], _stream())
    _lib.check(rc, "hipie_add_layernorm_sum")
    return out, s


@_timed("add_layernorm")
def add_layernorm_dec(x, delta, weight, bias, eps, aux_dtype, want16=False, addend=None):
    """n = LayerNorm(x + delta) for the fp32 query stream of the decoders: returns (n f32, n in aux_dtype or None,
    (n + addend) in aux_dtype or None) from one launch.  x (..., C) f32; delta any of f32/f16/bf16; addend aux_dtype."""
    lib = _lib.load()
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty_like(x)
    n16 = _alloc(tuple(x.shape), aux_dtype, x.device) if want16 else None
    s16 = _alloc(tuple(x.shape), aux_dtype, x.device) if addend is not None else None
    if addend is not None and ((addend.dtype != aux_dtype or addend.shape != x.shape) if aux_dtype != "hl8" else
                               (addend.dtype != torch.float16 or addend.numel() != 2 * x.numel())):
        raise RuntimeError("add_layernorm_dec: addend must match x's shape in aux_dtype")
    rc = lib.hipie_add_layernorm_dec(_chk(x, "x", torch.float32), _chk(delta, "delta"), _chk(weight, "weight", torch.float32),
                                     _chk(bias, "bias", torch.float32), out.data_ptr(), None if n16 is None else n16.data_ptr(),
                                     None if addend is None else _chk(addend, "addend"), None if s16 is None else s16.data_ptr(),
                                     rows, C, float(eps), _DT[delta.dtype], _code(aux_dtype), _stream())
    _lib.check(rc, "hipie_add_layernorm_dec")
    return out, n16, s16


_GN_WS = {}


def group_norm_ok(x, groups):
    if not x.is_cuda or x.dim() != 4 or x.dtype not in _DT or x.shape[1] != groups * 8 or groups > 64 or 256 % groups:
        return False
    return x.is_contiguous(memory_format=torch.channels_last) or (x.is_contiguous() and (x.shape[2] * x.shape[3]) % 8 == 0)


@_timed("group_norm")
def group_norm(x, groups, weight, bias, eps, relu=False, prebias=None, out_dtype=None, out_nchw=False):
    """GroupNorm(groups, C = 8 * groups)(x + prebias[c]) (+ ReLU) on a (B,C,H,W) tensor in the memory format it arrives in
    (channels-last stays channels-last: no NHWC -> NCHW copy); fp32 statistics.  out_nchw: a channels-last input leaves as a dense
    NCHW tensor (transposed inside the kernel; H*W % 64 == 0) -- what `.contiguous()` would make of the result, without that pass."""
    lib = _lib.load()
    B, C, H, W = x.shape
    nhwc = x.is_contiguous(memory_format=torch.channels_last) and not (x.is_contiguous() and C > 1 and H * W > 1)
    if not nhwc and not x.is_contiguous():
        raise RuntimeError("group_norm: x must be NCHW- or channels-last-contiguous")
    out_dtype = out_dtype or x.dtype
    if out_nchw and nhwc and (H * W) % 64 == 0:
        nhwc = 2
        out = torch.empty(B, C, H, W, dtype=out_dtype, device=x.device)
    else:
        out = torch.empty_like(x, dtype=out_dtype)            # preserves the memory format
    key = (str(x.device), B * groups, torch.cuda.current_stream(x.device).cuda_stream)      # per stream: the two branches of the head overlap
    ws = _GN_WS.get(key)
    if ws is None:
        ws = _GN_WS[key] = torch.empty(2 * B * groups * 512, dtype=torch.float32, device=x.device)
    rc = lib.hipie_group_norm(x.data_ptr(), None if prebias is None else _chk(prebias, "prebias", torch.float32),
                              _chk(weight, "weight", torch.float32), _chk(bias, "bias", torch.float32), out.data_ptr(),
                              ws.data_ptr(), B, C, H * W, groups, int(nhwc), float(eps), 1 if relu else 0,
                              _DT[x.dtype], _DT[out_dtype], _stream())
    _lib.check(rc, "hipie_group_norm")
    return out


@_timed("add_cast")
def add_cast(a, b):
    """(a f32 + b 16-bit) rounded once to b's dtype; same shapes, numel % 4 == 0."""
    lib = _lib.load()
    if a.shape != b.shape:
        raise RuntimeError("add_cast: shapes differ")
    out = torch.empty_like(b)
    rc = lib.hipie_add_cast(_chk(a, "a", torch.float32), _chk(b, "b"), out.data_ptr(), a.numel(), _DT[b.dtype], _stream())
    _lib.check(rc, "hipie_add_cast")
    return out


@_timed("add_cast")
def add_to_hl8(a, b_hl8):
    """HL8 of (a + b): a (..., C) fp32, b_hl8 (..., 2C) fp16 HL8 of the same rows -> (..., 2C) HL8, one pass (hipie_add_cast, HL8 form)."""
    lib = _lib.load()
    if b_hl8.dtype != torch.float16 or b_hl8.numel() != 2 * a.numel() or a.shape[-1] % 8 or not a.is_cuda:
        raise RuntimeError("add_to_hl8: a (.., C) fp32 and b (.., 2C) HL8 device tensors, C %% 8 == 0")
    out = torch.empty_like(b_hl8)
    rc = lib.hipie_add_cast(_chk(a, "a", torch.float32), _chk(b_hl8, "b"), out.data_ptr(), a.numel(), HL8, _stream())
    _lib.check(rc, "hipie_add_cast")
    return out


@_timed("batched_nms")
def batched_nms(boxes, scores, classes, iou_threshold, coordinate_trick=None):
    """boxes (B,Q,4) normalised cxcywh, scores (B,Q), classes (B,Q) int64 -> keep (B,Q) int32 (query indices in
    decreasing-score order, -1 padded), count (B) int32.  torchvision.ops.batched_nms semantics per image."""
    lib = _lib.load()
    B, Q = scores.shape
    if coordinate_trick is None:
        coordinate_trick = Q * 4 <= 4000                  # torchvision/ops/boxes.py batched_nms switch
    order = scores.sort(dim=1, descending=True, stable=True)[1].to(torch.int32)
    keep = torch.empty(B, Q, dtype=torch.int32, device=boxes.device)
    count = torch.empty(B, dtype=torch.int32, device=boxes.device)
    rc = lib.hipie_batched_nms(_chk(boxes, "boxes", torch.float32), _chk(classes, "classes", torch.int64), order.data_ptr(),
                               keep.data_ptr(), count.data_ptr(), B, Q, float(iou_threshold), int(coordinate_trick), _stream())
    _lib.check(rc, "hipie_batched_nms")
    return keep, count


@_timed("mask_finalize")
def mask_finalize(masks, qidx, up, crop_hw, out_hw, threshold):
    """masks (Q,hm,wm) stride-`up` logits, qidx (n) int32 rows (or None) -> (n,out_h,out_w) uint8:
    bilinear x`up` -> sigmoid > threshold -> crop -> nearest resize to out_hw."""
    lib = _lib.load()
    Q, hm, wm = masks.shape
    n = Q if qidx is None else int(qidx.shape[0])
    out = torch.empty(n, out_hw[0], out_hw[1], dtype=torch.uint8, device=masks.device)
    rc = lib.hipie_mask_finalize(_chk(masks, "masks"), _DT[masks.dtype], None if qidx is None else _chk(qidx, "qidx", torch.int32),
                                 n, hm, wm, int(up), int(crop_hw[0]), int(crop_hw[1]), int(out_hw[0]), int(out_hw[1]),
                                 float(threshold), out.data_ptr(), _stream())
    _lib.check(rc, "hipie_mask_finalize")
    return out


SEM_PAN_CLASSES = 160          # classes per launch of hipie_sem_pan (register budget of its accumulators); more are tiled


def sem_pan_ok(n_queries, n_classes):
    return 0 < n_classes and 0 < n_queries <= 8192


@_timed("sem_pan")
def sem_pan(masks_lo, cls_all, pscore, up, crop_hw, out_hw, precision=0):
    """fused semantic + panoptic maps of one image.  masks_lo (N,hm,wm) f32 stride-`up` logits, cls_all (N,C) f32 class
    probabilities, pscore (N) f32 (score of kept queries, <= 0 otherwise) ->
    sem (C,oh,ow) f32, pan_idx (oh,ow) int32 (-1: no kept query), pan_own (oh,ow) bool, area (N) int32.
    More than SEM_PAN_CLASSES classes (ADE-847, LVIS-1203: BASELINE configs[3]/[4]) run as class tiles of one launch each;
    the panoptic outputs do not depend on the classes and are taken from the first tile."""
    lib = _lib.load()
    N, C = cls_all.shape
    _, hm, wm = masks_lo.shape
    dev = masks_lo.device
    npad = (N + 15) // 16 * 16
    ps = torch.full((npad,), -1.0, dtype=torch.float32, device=dev)
    ps[:N] = pscore
    oh, ow = int(out_hw[0]), int(out_hw[1])
    sem = torch.empty(C, oh, ow, dtype=torch.float32, device=dev)
    pan_idx = torch.empty(oh, ow, dtype=torch.int32, device=dev)
    pan_own = torch.empty(oh, ow, dtype=torch.uint8, device=dev)
    area = torch.zeros(npad, dtype=torch.int32, device=dev)
    masks_ptr = _chk(masks_lo, "masks", torch.float32)
    spare = None
    for c0 in range(0, C, SEM_PAN_CLASSES):
        cn = min(SEM_PAN_CLASSES, C - c0)
        cp = 32 if cn <= 32 else 96 if cn <= 96 else 160
        cls_t = torch.zeros(cp, npad, dtype=torch.float32, device=dev)
        cls_t[:cn, :N] = cls_all[:, c0:c0 + cn].t()
        hi = cls_t.to(torch.bfloat16)
        lo = (cls_t - hi.float()).to(torch.bfloat16) if precision == 0 else None
        if c0 == 0:
            outs = (pan_idx, pan_own, area)
        else:
            if spare is None:
                spare = (torch.empty_like(pan_idx), torch.empty_like(pan_own), torch.zeros_like(area))
            outs = spare
        rc = lib.hipie_sem_pan(masks_ptr, hi.data_ptr(), None if lo is None else lo.data_ptr(),
                               ps.data_ptr(), sem[c0:c0 + cn].data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                               N, npad, cn, hm, wm, int(up), int(crop_hw[0]), int(crop_hw[1]), oh, ow, int(precision), _stream())
        _lib.check(rc, "hipie_sem_pan")
    return sem, pan_idx, pan_own.bool(), area[:N]


_DIM_T = {}


@_timed("sine_embed")
def sine_embed(ref, num_pos_feats=128, temperature=10000, out_dtype=torch.float32):
    """get_sine_pos_embed(ref, exchange_xy=True): ref (..., 2|4) f32 (last-dim stride 1) -> (..., ncoord * num_pos_feats)."""
    import math
    lib = _lib.load()
    nc = ref.shape[-1]
    ref = ref.float()
    rs = nc
    if ref.dim() == 3 and ref.stride(-1) == 1 and ref.stride(0) == ref.shape[1] * ref.stride(1) and ref.stride(1) >= nc:
        rs = ref.stride(1)                    # a row-strided view such as ref_in[:, :, 0, :]: read in place
    elif not ref.is_contiguous():
        ref = ref.contiguous()
    key = (num_pos_feats, temperature, str(ref.device))
    dim_t = _DIM_T.get(key)
    if dim_t is None:
        d = torch.arange(num_pos_feats, dtype=torch.float32, device=ref.device)
        dim_t = (temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)).contiguous()
        _DIM_T[key] = dim_t
    n = ref.numel() // nc
    out = torch.empty(ref.shape[:-1] + (nc * num_pos_feats,), dtype=out_dtype, device=ref.device)
    if not ref.is_cuda:
        raise RuntimeError("Not implemented on the CPU (ref must be a CUDA/HIP tensor)")
    rc = lib.hipie_sine_embed(ref.data_ptr(), dim_t.data_ptr(), out.data_ptr(), n, nc, num_pos_feats, rs,
                              2 * math.pi, _DT[out_dtype], _stream())
    _lib.check(rc, "hipie_sine_embed")
    return out


def _transposed(mod_layers, dtype=None):
    """(in, out) copies of the Linear weights of a small MLP head + its biases, cached on the parameters' versions."""
    owner = mod_layers[0]
    key = tuple((l.weight.data_ptr(), l.weight._version, l.weight.dtype, l.bias._version) for l in mod_layers)
    if getattr(owner, "_wt_key", None) != key:
        owner._wt = [(l.weight.t().contiguous(), l.bias.contiguous()) for l in mod_layers]
        owner._wt_key = key
    return owner._wt


def ref_point_mlp_ok(ref, head):
    ls = getattr(head, "layers", None)
    return (ls is not None and len(ls) == 2 and ref.is_cuda and ref.shape[-1] == 4 and tuple(ls[0].weight.shape) == (256, 512)
            and tuple(ls[1].weight.shape) == (256, 256) and ls[0].weight.dtype == ls[1].weight.dtype and ls[0].weight.dtype in _DT)


@_timed("ref_point_mlp")
def ref_point_mlp(ref, head, num_pos_feats=128, temperature=10000):
    """query_pos = head(get_sine_pos_embed(ref)) in one launch: ref (B, Q, 4) f32 (row-strided view allowed), head = the
    2-layer ref_point_head (512 -> 256 -> 256) -> (B, Q, 256) in the head's weight dtype."""
    import math
    lib = _lib.load()
    ref = ref.float()
    rs = 4
    if ref.dim() == 3 and ref.stride(-1) == 1 and ref.stride(0) == ref.shape[1] * ref.stride(1) and ref.stride(1) >= 4:
        rs = ref.stride(1)
    elif not ref.is_contiguous():
        ref = ref.contiguous()
    key = (num_pos_feats, temperature, str(ref.device))
    dim_t = _DIM_T.get(key)
    if dim_t is None:
        d = torch.arange(num_pos_feats, dtype=torch.float32, device=ref.device)
        dim_t = (temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)).contiguous()
        _DIM_T[key] = dim_t
    (w1t, b1), (w2t, b2) = _transposed(list(head.layers))
    dt = w1t.dtype
    n = ref.numel() // 4
    out = torch.empty(ref.shape[:-1] + (256,), dtype=dt, device=ref.device)
    rc = lib.hipie_ref_point_mlp(ref.data_ptr(), dim_t.data_ptr(), w1t.data_ptr(), b1.data_ptr(), w2t.data_ptr(), b2.data_ptr(),
                                 out.data_ptr(), n, rs, 2 * math.pi, _DT[dt], _stream())
    _lib.check(rc, "hipie_ref_point_mlp")
    return out


def box_head_ok(x, mlp):
    ls = getattr(mlp, "layers", None)
    return (ls is not None and len(ls) == 3 and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256
            and [tuple(l.weight.shape) for l in ls] == [(256, 256), (256, 256), (4, 256)]
            and all(l.weight.dtype == torch.float32 for l in ls))


@_timed("box_head")
def box_head(x, ref, mlp, eps=1e-5):
    """sigmoid(mlp(x) + inverse_sigmoid(ref)) in one launch: x (..., 256) f32, ref (..., 4) f32, mlp = MLP(256, 256, 4, 3) f32."""
    lib = _lib.load()
    x, ref = x.contiguous(), ref.float().contiguous()
    (w1t, b1), (w2t, b2), (w3t, b3) = _transposed(list(mlp.layers))
    out = torch.empty_like(ref)
    rc = lib.hipie_box_head(_chk(x, "x", torch.float32), _chk(ref, "ref", torch.float32), w1t.data_ptr(), b1.data_ptr(), w2t.data_ptr(),
                            b2.data_ptr(), w3t.data_ptr(), b3.data_ptr(), out.data_ptr(), ref.numel() // 4, float(eps), _stream())
    _lib.check(rc, "hipie_box_head")
    return out


@_timed("box_refine")
def box_refine(delta, ref, eps=1e-5):
    """sigmoid(delta + inverse_sigmoid(ref)): delta (..., 4) f32|f16|bf16, ref (..., 4) f32 -> (..., 4) f32."""
    lib = _lib.load()
    delta, ref = delta.contiguous(), ref.float().contiguous()
    out = torch.empty_like(ref)
    rc = lib.hipie_box_refine(_chk(delta, "delta"), _chk(ref, "ref", torch.float32), out.data_ptr(), ref.numel(), float(eps),
                              _DT[delta.dtype], _stream())
    _lib.check(rc, "hipie_box_refine")
    return out


def selftest(which, a, b=None):
    lib = _lib.load()
    out = torch.empty(32 * 32 if which == 0 else 256, dtype=torch.float32, device=a.device)
    rc = lib.hipie_selftest(which, a.data_ptr(), None if b is None else b.data_ptr(), out.data_ptr(), _stream())
    _lib.check(rc, "hipie_selftest")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# hipie_gemm: the linears on the hand-written MFMA GEMM.  HL8 ("split fp16") tensors are plain fp16 torch tensors whose last
# dimension is 2K: K/8 groups of [8 hi | 8 lo] (include/hipie_mi355.h HIPIE_HL8).
def hl8_pack(x, scale=1.0):
    """(..., K) float tensor -> (..., 2K) fp16 in HL8 layout, written with torch ops (weights, once per checkpoint; tests)."""
    K = x.shape[-1]
    assert K % 8 == 0
    xs = (x.float() * scale).clamp(-65504.0, 65504.0)      # saturate like the kernels' hl_split (no inf hi / NaN lo)
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return torch.stack([hi.reshape(*x.shape[:-1], K // 8, 8), lo.reshape(*x.shape[:-1], K // 8, 8)], dim=-2).reshape(*x.shape[:-1], 2 * K).contiguous()


def hl8_unpack(x):
    """(..., 2K) fp16 HL8 -> (..., K) fp32 (hi + lo)."""
    K2 = x.shape[-1]
    g = x.reshape(*x.shape[:-1], K2 // 16, 2, 8).float()
    return (g[..., 0, :] + g[..., 1, :]).reshape(*x.shape[:-1], K2 // 2)


@_timed("to_hl8")
def to_hl8(x, scale=1.0):
    """(rows..., K) fp32 | fp16 device tensor (last dim contiguous, uniform row stride) -> (rows..., 2K) fp16 HL8 (hipie_to_hl8)."""
    lib = _lib.load()
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    out = torch.empty(*x.shape[:-1], 2 * K, dtype=torch.float16, device=x.device)
    if not x.is_cuda:
        raise RuntimeError("Not implemented on the CPU (to_hl8)")
    rc = lib.hipie_to_hl8(x2.data_ptr(), x2.stride(0), out.data_ptr(), 2 * K, x2.shape[0], K, _DT[x.dtype], float(scale), _stream())
    _lib.check(rc, "hipie_to_hl8")
    return out


@_timed("to_hl8_t")
def to_hl8_t(x, pad_to=32, scale=1.0):
    """(M, C) fp32 device tensor (last dim contiguous, any row stride) -> (C, 2 Mp) fp16 HL8 of x^T, Mp = M rounded up to `pad_to`, the padding
    columns zero (hipie_to_hl8_t): a split operand whose K runs over the rows of x."""
    lib = _lib.load()
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError("to_hl8_t: a 2-D fp32 device tensor")
    if x.stride(1) != 1:
        x = x.contiguous()
    M, C = x.shape
    Mp = -(-M // pad_to) * pad_to
    out = torch.empty(C, 2 * Mp, dtype=torch.float16, device=x.device)
    rc = lib.hipie_to_hl8_t(x.data_ptr(), x.stride(0), out.data_ptr(), 2 * Mp, M, C, Mp, float(scale), _stream())
    _lib.check(rc, "hipie_to_hl8_t")
    return out


def f16_pair(x, cols=None, scale=None):
    """fp32 (.., C) -> the two contiguous fp16 planes (hi = fp16(s x), lo = fp16(s x - hi)) of the fused training attention's operands, the
    last dimension zero-padded to `cols` (a multiple of 8); s = `scale`, a one-element DEVICE tensor (no host wait), or 1; values beyond the
    fp16 range saturate (hl_split).  One pass on the device (hipie_to_f16_pair)."""
    C = x.shape[-1]
    Cp = C if cols is None else max(cols, C)
    if x.is_cuda and x.dtype == torch.float32 and Cp % 8 == 0:
        lib = _lib.load()
        x2 = x.reshape(-1, C)
        if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < C):
            x2 = x2.contiguous()
        hi = torch.empty(*x.shape[:-1], Cp, dtype=torch.float16, device=x.device)
        lo = torch.empty_like(hi)
        sc = 0 if scale is None else scale.float().reshape(1).data_ptr()
        rc = lib.hipie_to_f16_pair(x2.data_ptr(), x2.stride(0) if x2.shape[0] > 1 else C, hi.data_ptr(), lo.data_ptr(), x2.shape[0], C, Cp, sc, _stream())
        _lib.check(rc, "hipie_to_f16_pair")
        return hi, lo
    raise RuntimeError("f16_pair: an fp32 device tensor and a column count that is a multiple of 8 (got %s %s on %s, %d columns)"
                       % (x.dtype, tuple(x.shape), x.device, Cp))


@_timed("attn_train_fwd")
def attn_train_forward(q_pair, k_pair, v_pair):
    """hipie_attn_train_forward: q', k' (BH, N, 224) and v (BH, N, 80) as fp16 pairs (f16_pair) -> (out (BH, N, 80) f32, lse (BH, N) f32)"""
    lib = _lib.load()
    qh, ql = q_pair
    kh, kl = k_pair
    vh, vl = v_pair
    BH, N, DQ = qh.shape
    if DQ != 224 or kh.shape != qh.shape or tuple(vh.shape) != (BH, N, 80) or N % 128 or not qh.is_cuda:
        raise RuntimeError("attn_train_forward: q', k' (BH, N, 224), v (BH, N, 80) device pairs with N %% 128 == 0, got %s / %s / %s"
                           % (tuple(qh.shape), tuple(kh.shape), tuple(vh.shape)))
    for t in (qh, ql, kh, kl, vh, vl):
        if t.dtype != torch.float16 or not t.is_contiguous():
            raise RuntimeError("attn_train_forward: contiguous fp16 planes")
    out = torch.empty(BH, N, 80, dtype=torch.float32, device=qh.device)
    lse = torch.empty(BH, N, dtype=torch.float32, device=qh.device)
    rc = lib.hipie_attn_train_forward(qh.data_ptr(), ql.data_ptr(), kh.data_ptr(), kl.data_ptr(), vh.data_ptr(), vl.data_ptr(), out.data_ptr(),
                                      lse.data_ptr(), BH, N, _stream())
    _lib.check(rc, "hipie_attn_train_forward")
    return out, lse


@_timed("attn_train_bwd")
def attn_train_backward(q_pair, k_pair, v96_pair, do96_pair, lse, delta):
    """hipie_attn_train_backward: the forward's q', k' pairs, v and dO as (BH, N, 96) pairs (f16_pair(.., 96); dO scaled into fp16's range by the
    caller), lse, delta = rowsum(dO * out) -> (dq' (BH, N, 224), dk (BH, N, 80), dv (BH, N, 80)) fp32, in the scale of the dO given"""
    lib = _lib.load()
    qh, ql = q_pair
    kh, kl = k_pair
    vh, vl = v96_pair
    dh, dl = do96_pair
    BH, N, _ = qh.shape
    if tuple(vh.shape) != (BH, N, 96) or tuple(dh.shape) != (BH, N, 96) or tuple(lse.shape) != (BH, N) or tuple(delta.shape) != (BH, N):
        raise RuntimeError("attn_train_backward: v / dO as (BH, N, 96) pairs, lse / delta (BH, N)")
    for t in (qh, ql, kh, kl, vh, vl, dh, dl):
        if t.dtype != torch.float16 or not t.is_contiguous() or not t.is_cuda:
            raise RuntimeError("attn_train_backward: contiguous fp16 device planes")
    lse, delta = lse.float().contiguous(), delta.float().contiguous()
    dq = torch.empty(BH, N, 224, dtype=torch.float32, device=qh.device)
    dk = torch.empty(BH, N, 80, dtype=torch.float32, device=qh.device)
    dv = torch.empty(BH, N, 80, dtype=torch.float32, device=qh.device)
    rc = lib.hipie_attn_train_backward(qh.data_ptr(), ql.data_ptr(), kh.data_ptr(), kl.data_ptr(), vh.data_ptr(), vl.data_ptr(), dh.data_ptr(),
                                       dl.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), BH, N, _stream())
    _lib.check(rc, "hipie_attn_train_backward")
    return dq, dk, dv


@_timed("fill_rows")
def fill_rows(dst, rows, src_row):
    """dst[rows[i]] = src_row for every i (hipie_fill_rows): dst (R, W) contiguous device tensor, rows int32 (n,), src_row (W,) of dst's
    dtype; row bytes a multiple of 16.  In place; returns dst."""
    lib = _lib.load()
    if dst.dim() != 2 or not dst.is_cuda or src_row.dtype != dst.dtype or src_row.numel() != dst.shape[1]:
        raise RuntimeError("fill_rows: dst (R, W) on the device, src_row (W,) of the same dtype")
    rc = lib.hipie_fill_rows(_chk(dst, "dst"), dst.stride(0) * dst.element_size(), _chk(rows, "rows", torch.int32), rows.numel(),
                             _chk(src_row, "src_row"), dst.shape[1] * dst.element_size(), _stream())
    _lib.check(rc, "hipie_fill_rows")
    return dst


ACT_NONE, ACT_GELU, ACT_RELU, ACT_QGELU = 0, 1, 2, 3


def _gemm_tag(a, w, *args, **kw):
    tag = kw.get("tag", "gemm")
    if PROFILE.shapes and tag == "gemm":
        split = kw.get("split")
        K = w.shape[1] // 2 if split else w.shape[1]
        tag = "gemm M%d N%d K%d %s->%s%s" % (a.numel() // a.shape[-1], w.shape[0], K, "f32" if a.dtype == torch.float32 else ("hl8" if split else "f16"),
                                          {F32: "f32", F16: "f16", HL8: "hl8"}.get(kw.get("out_fmt", F32), "?"), " +res" if kw.get("resid") is not None else "")
    return tag


@_timed(_gemm_tag)
def gemm(a, w, bias=None, resid=None, out_fmt=F32, act=ACT_NONE, alpha=1.0, oscale=1.0, split=None, out=None, tag="gemm", out_row=None,
         out_rows=None, a_row=None):
    """out = ((act(alpha * a . w^T + bias)) + resid) * oscale on hipie_gemm.
    a (..., K) fp16 and w (N, K) fp16 (one product), or both HL8: a (..., 2K), w (N, 2K) fp16 (three products, fp32-class);
    `split` tells which (default: inferred -- pass it when K is ambiguous).  With split=True a may also be PLAIN fp32 (..., K) rows
    (16-byte aligned): the kernel splits them after the LDS read -- no hipie_to_hl8 launch.  a may be a row-strided 2-d view.  bias (N) f32,
    resid (..., N) f32.  out_fmt F32 | F16 | HL8 -> (..., N) f32 / f16 or (..., 2N) fp16 HL8."""
    lib = _lib.load()
    a_f32 = a.dtype == torch.float32 and bool(split)
    if (a.dtype != torch.float16 and not a_f32) or w.dtype != torch.float16 or not a.is_cuda:
        raise RuntimeError("gemm: operands must be fp16 (plain or HL8) device tensors (a may be fp32 against HL8 weights)")
    if split is None:
        raise RuntimeError("gemm: say split=True (HL8 operands) or split=False (plain fp16)")
    N = w.shape[0]
    Kw = w.shape[1] // 2 if split else w.shape[1]
    lead = a.shape[:-1]
    a2 = a if a.dim() == 2 else a.reshape(-1, a.shape[-1])
    if a2.stride(-1) != 1 or a2.shape[-1] != (Kw if a_f32 else w.shape[1]):
        raise RuntimeError("gemm: a rows must be contiguous and as long as w rows (%s vs %s)" % (tuple(a.shape), tuple(w.shape)))
    if a_f32 and (a2.stride(0) % 4 or a2.data_ptr() % 16):
        raise RuntimeError("gemm: fp32 a rows must be 16-byte aligned")
    M = a2.shape[0]
    if a_row is not None:            # gather: product row m reads operand row a_row[m] (hipie_gemm_gather); M = the map's length
        if a_row.dtype != torch.int32 or not a_row.is_contiguous() or not split:
            raise RuntimeError("gemm: a_row must be a contiguous int32 vector (split operands only)")
        M = a_row.numel()
        lead = (M,)
        if out is not None and out_row is None and out.numel() // out.shape[-1] < M:
            raise RuntimeError("gemm: `out` has fewer rows (%d) than the row map (%d)" % (out.numel() // out.shape[-1], M))
        if resid is not None and resid.numel() // N < M:
            raise RuntimeError("gemm: `resid` has fewer rows (%d) than the row map (%d)" % (resid.numel() // N, M))
    if out_row is not None:
        if out is None:
            lead = (int(out_rows),)
        if out_row.dtype != torch.int32 or out_row.numel() != M or not out_row.is_contiguous():
            raise RuntimeError("gemm: out_row must be a contiguous int32 vector with one entry per row of a")
    if out is None:
        if out_fmt == F32:
            out = torch.empty(*lead, N, dtype=torch.float32, device=a.device)
        elif out_fmt == F16:
            out = torch.empty(*lead, N, dtype=torch.float16, device=a.device)
        else:
            out = torch.empty(*lead, 2 * N, dtype=torch.float16, device=a.device)
    o2 = out.reshape(-1, out.shape[-1])
    r2 = None
    if resid is not None:
        r2 = resid.reshape(-1, N)
        if r2.dtype != torch.float32 or r2.stride(-1) != 1:
            raise RuntimeError("gemm: resid must be fp32 with contiguous rows")
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        raise RuntimeError("gemm: bias must be contiguous fp32")
    if a_row is not None:
        rc = lib.hipie_gemm_gather(a2.data_ptr(), a2.stride(0), a2.shape[0], a_row.data_ptr(), _chk(w, "w"), w.shape[1],
                                   None if bias is None else bias.data_ptr(), None if r2 is None else r2.data_ptr(), 0 if r2 is None else r2.stride(0),
                                   o2.data_ptr(), o2.stride(0), None if out_row is None else out_row.data_ptr(), M, N, Kw,
                                   F32 if a_f32 else HL8, int(out_fmt), int(act), float(alpha), float(oscale), _stream())
        _lib.check(rc, "hipie_gemm_gather")
        return out
    rc = lib.hipie_gemm(a2.data_ptr(), a2.stride(0), _chk(w, "w"), w.shape[1], None if bias is None else bias.data_ptr(),
                        None if r2 is None else r2.data_ptr(), 0 if r2 is None else r2.stride(0), o2.data_ptr(), o2.stride(0),
                        None if out_row is None else out_row.data_ptr(), M, N, Kw, F32 if a_f32 else (HL8 if split else F16), int(out_fmt), int(act), float(alpha), float(oscale), _stream())
    _lib.check(rc, "hipie_gemm")
    return out


@_timed(lambda qkv, tab_h, tab_w, grid_hw, heads: "vit_attn_global" if grid_hw[0] * grid_hw[1] > 256 else "vit_attn_window")
def vit_attn_split(qkv, tab_h, tab_w, grid_hw, heads):
    """hipie_vit_attn_split: qkv (B, gh*gw, 2 * 3*heads*hd) fp16 HL8 rows (q pre-scaled by scale*log2 e), tab_h (2*gh-1, 2*hd),
    tab_w (2*gw-1, 2*hd) HL8 (tables / scale) -> (B, N, 2 * heads*hd) HL8."""
    lib = _lib.load()
    gh, gw = grid_hw
    B, N, C6 = qkv.shape
    hd = C6 // (6 * heads)
    if qkv.dtype != torch.float16 or tab_h.dtype != torch.float16 or tab_w.dtype != torch.float16:
        raise RuntimeError("vit_attn_split: HL8 (fp16) operands expected")
    out = torch.empty(B, N, 2 * heads * hd, dtype=torch.float16, device=qkv.device)
    rc = lib.hipie_vit_attn_split(_chk(qkv, "qkv"), _chk(tab_h, "tab_h"), _chk(tab_w, "tab_w"), out.data_ptr(), B, gh, gw, heads, hd, _stream())
    _lib.check(rc, "hipie_vit_attn_split")
    return out


@_timed("gemm_split_k")
def gemm_split_k(a_hl8, w_hl8, nk):
    """a . w^T for HL8 operands a (M, 2K), w (N, 2K) with the contraction cut into nk chunks (K / nk a multiple of 32): one problem per
    chunk in ONE hipie_gemm_batched launch, partial products summed -> (M, N) fp32.  For products whose M x N is a few tiles and whose K is
    long (the weight gradients of the training step)."""
    lib = _lib.load()
    M, K2 = a_hl8.shape
    N = w_hl8.shape[0]
    K = K2 // 2
    if w_hl8.shape[1] != K2 or K % (32 * nk) or a_hl8.dtype != torch.float16 or w_hl8.dtype != torch.float16 or not a_hl8.is_cuda:
        raise RuntimeError("gemm_split_k: HL8 device operands (M, 2K) / (N, 2K), K a multiple of 32 * nk")
    if N % 8:
        raise RuntimeError("gemm_split_k: N must be a multiple of 8")
    Kc = K // nk
    part = torch.empty(nk, M, N, dtype=torch.float32, device=a_hl8.device)
    rc = lib.hipie_gemm_batched(a_hl8.data_ptr(), K2, 0, 2 * Kc, w_hl8.data_ptr(), K2, 0, 2 * Kc, part.data_ptr(), N, 0, M * N, 1, nk, M, N, Kc,
                                F32, 1.0, _stream())
    _lib.check(rc, "hipie_gemm_batched")
    return part.sum(0)


@_timed("gemm_batched")
def matmul_nt_batched(a, w, alpha=1.0):
    """out[b] = alpha * a[b] . w[b]^T at fp32-class accuracy on hipie_gemm_batched: a (B, M, K), w (B, N, K) fp32 device tensors,
    K % 32 == 0 -> (B, M, N) fp32.  The operands are split to HL8 on the way in; N is padded to a multiple of 8 with zero rows.  For the
    per-image products of the heads (class logits = queries . token embeddings^T, VL_Align: deformable_detr.py:55-73)."""
    lib = _lib.load()
    B, M, K = a.shape
    N = w.shape[1]
    if w.shape[0] != B or w.shape[2] != K or K % 32 or not a.is_cuda:
        raise RuntimeError("matmul_nt_batched: a (B, M, K), w (B, N, K) device tensors with K %% 32 == 0, got %s / %s" % (tuple(a.shape), tuple(w.shape)))
    Np = (N + 7) // 8 * 8
    if Np != N:
        wp = torch.zeros(B, Np, K, dtype=torch.float32, device=w.device)
        wp[:, :N] = w
        w = wp
    ah, wh = to_hl8(a.float().contiguous()), to_hl8(w.float().contiguous())
    out = torch.empty(B, M, Np, dtype=torch.float32, device=a.device)
    rc = lib.hipie_gemm_batched(ah.data_ptr(), 2 * K, M * 2 * K, 0, wh.data_ptr(), 2 * K, Np * 2 * K, 0, out.data_ptr(), Np, M * Np, 0,
                                B, 1, M, Np, K, F32, float(alpha), _stream())
    _lib.check(rc, "hipie_gemm_batched")
    return out if Np == N else out[..., :N]


def vit_attn_split_ok(grid_hw, hd):
    """geometry hipie_vit_attn_split covers: head_dim 64 / 80, 14-wide windows, token grids up to 96 wide (64 x 64 at 1024^2, 84 x 84 at
    1344^2) and -- walked column by column -- grids wider than 96 whose height is <= 96 (64 x 128: a 1024 x 2048 image, the eval yamls'
    MAX_SIZE_TEST).  Only grids wider than 96 in BOTH directions (beyond 1536 x 1536 pixels) are not covered."""
    gh, gw = grid_hw

    def rows_ok(kh, kw):           # LDS: the per-key-row bias table of the <= 64-wide instances holds kh rows
        return (kw <= 32 and kh <= 160) or (kw <= 64 and kh <= 132) or (64 < kw <= 96 and kh <= 160)
    return hd in (64, 80) and ((gw == 14 and gh <= 96) or rows_ok(gh, gw) or (gw > 96 and gh <= 96 and rows_ok(gw, gh)))


# ---- split ("fp32-class") linears: cached HL8 copies of (derived) weights + the GEMM call ----------------------------------
def split_weight(owner, key, params, weight_fn, bias_fn=None):
    """HL8 copy of a weight (N, K) -> ((Np, 2K) fp16, bias (Np,) f32 | None, N), N padded to a multiple of 8 with zero rows.
    Cached on ``owner`` under ``key``; the cache entry carries (data_ptr, version) of ``params``, so load_state_dict / .to() /
    in-place updates invalidate it.  weight_fn / bias_fn build the (possibly folded / concatenated) fp32 tensors."""
    ver = tuple((q.data_ptr(), q._version, str(q.device)) for q in params)
    cache = owner.__dict__.setdefault("_hl8_cache", {})
    e = cache.get(key)
    if e is None or e[0] != ver:
        w = weight_fn().detach().float()
        b = None if bias_fn is None else bias_fn()
        b = None if b is None else b.detach().float()
        N = w.shape[0]
        Np = (N + 7) // 8 * 8
        if Np != N:
            w = torch.nn.functional.pad(w, (0, 0, 0, Np - N))
            b = None if b is None else torch.nn.functional.pad(b, (0, Np - N))
        e = (ver, hl8_pack(w), None if b is None else b.contiguous(), N)
        cache[key] = e
    return e[1], e[2], e[3]


def conv3x3_split_ok(x, conv):
    """3 x 3, stride 1, padding 1, dense: the shapes hipie_conv3x3_split runs well (full 256-column tiles)"""
    return (x.is_cuda and x.dim() == 4 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.padding_mode == "zeros" and conv.groups == 1 and conv.weight.dtype == torch.float32 and conv.in_channels % 32 == 0 and conv.out_channels % 256 == 0
            and x.shape[0] * (x.shape[2] + 2) * (x.shape[3] + 2) >= 16384)


_CONV_STAGE = {}
_RETIRED = []           # evicted cache entries stay alive: a captured hipGraph may still hold their addresses (as _Workspace.retired does)


@_timed("conv3x3_split")
def conv3x3_split(x, conv, act=ACT_NONE):
    """conv(x) for a 3 x 3 / stride 1 / padding 1 nn.Conv2d at the split policy's accuracy (hipie_conv3x3_split: implicit GEMM, K = 9 C).
    x (B, C, H, W) in any memory format -> (B, N, H, W) fp32 in channels-last memory.  The input is copied once onto a zero-padded pixel grid
    (with guard rows), the kernel computes on that grid and the interior is cropped."""
    lib = _lib.load()
    B, C, H, W = x.shape
    N = conv.out_channels
    Hp, Wp = H + 2, W + 2
    guard = Wp + 1
    rows = B * Hp * Wp
    # the zero-padded staging grid is cached per (geometry, device, stream): its border and guard rows stay zero, only the interior is
    # re-written per call (one full-map zero-fill pass less per convolution)
    skey = (B, C, H, W, str(x.device), torch.cuda.current_stream(x.device).cuda_stream)
    buf = _CONV_STAGE.get(skey)
    if buf is None:
        if len(_CONV_STAGE) >= 8:
            _RETIRED.append(_CONV_STAGE.pop(next(iter(_CONV_STAGE))))
        buf = _CONV_STAGE[skey] = torch.zeros(rows + 2 * guard, C, dtype=torch.float32, device=x.device)
    buf[guard:guard + rows].view(B, Hp, Wp, C)[:, 1:-1, 1:-1].copy_(x.permute(0, 2, 3, 1))
    w, b, _ = split_weight(conv, "w3x3", [conv.weight] + ([conv.bias] if conv.bias is not None else []),
                           lambda: conv.weight.permute(0, 2, 3, 1).reshape(N, 9 * C), (lambda: conv.bias) if conv.bias is not None else None)
    out = torch.empty(rows, N, dtype=torch.float32, device=x.device)
    xin = buf[guard:]
    rc = lib.hipie_conv3x3_split(xin.data_ptr(), C, w.data_ptr(), None if b is None else b.data_ptr(), out.data_ptr(), N, rows, Wp, C, N,
                                 F32, F32, int(act), _stream())
    _lib.check(rc, "hipie_conv3x3_split")
    return out.view(B, Hp, Wp, N)[:, 1:-1, 1:-1].permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)


_FFN_PERM = {}


def ffn_fused_ok(x_hl8, lin1, lin2):
    """the shape hipie_ffn_fused is built for: 256 -> 2048 -> 256 on at least a few thousand tokens (the two deformable encoders)"""
    return (x_hl8.is_cuda and lin1.weight.shape == (2048, 256) and lin2.weight.shape == (256, 2048) and lin1.bias is not None
            and lin2.bias is not None and lin1.weight.dtype == torch.float32 and x_hl8.numel() // 512 >= 4096)


@_timed("ffn_fused")
def ffn_fused(x_hl8, lin1, lin2):
    """linear2(relu(linear1(x))) in ONE launch at the split policy's accuracy (hipie_ffn_fused): x (..., 2*256) HL8 -> (..., 256) fp32.
    The hidden activations stay in registers; W2's columns are permuted once per parameter version into the order the MFMA C layout
    hands them over (rows 0-3, 8-11, 4-7, 12-15 of every 16)."""
    lib = _lib.load()
    w1, b1, _ = split_weight(lin1, "w", [lin1.weight, lin1.bias], lambda: lin1.weight, lambda: lin1.bias)
    F_ = lin2.weight.shape[1]
    perm = _FFN_PERM.get(F_)
    if perm is None:
        base = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
        perm = _FFN_PERM[F_] = (torch.arange(0, F_, 16)[:, None] + base[None, :]).reshape(-1)
    w2, b2, _ = split_weight(lin2, "w_ffn_perm", [lin2.weight, lin2.bias], lambda: lin2.weight[:, perm.to(lin2.weight.device)], lambda: lin2.bias)
    lead = x_hl8.shape[:-1]
    x2 = x_hl8.reshape(-1, x_hl8.shape[-1])
    if x2.dtype != torch.float16 or x2.stride(-1) != 1 or x2.shape[-1] != 512:
        raise RuntimeError("ffn_fused: x must be HL8 rows of 2 * 256 fp16 values")
    M = x2.shape[0]
    out = torch.empty(M, 256, dtype=torch.float32, device=x2.device)
    rc = lib.hipie_ffn_fused(x2.data_ptr(), x2.stride(0), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(), 256,
                             M, 256, F_, _stream())
    _lib.check(rc, "hipie_ffn_fused")
    return out.view(*lead, 256)


def split_ok(K):
    return K % 32 == 0


def split_linear(x, owner, key, weight, bias=None, act=ACT_NONE, out_fmt=F32, resid=None, x_hl8=False, weight_fn=None, bias_fn=None,
                 tag="gemm", params=None, out=None, out_row=None, a_row=None):
    """F.linear(x, weight, bias) at fp32-class accuracy on hipie_gemm's split operands.  x fp32 / fp16 (converted with hipie_to_hl8)
    or already HL8 (x_hl8).  weight_fn / bias_fn: derived weights (folded constants, concatenations) built once per parameter version."""
    if params is None:
        params = [weight] + ([bias] if bias is not None else [])
    w, b, N = split_weight(owner, key, params, weight_fn or (lambda: weight), bias_fn or ((lambda: bias) if bias is not None else None))
    if x_hl8:
        a = x
    elif x.dtype == torch.float32 and x.stride(-1) == 1 and x.data_ptr() % 16 == 0 and (x.dim() < 2 or x.stride(-2) % 4 == 0):
        a = x if x.dim() <= 2 or x.is_contiguous() else x.contiguous()      # fp32 rows: split inside the GEMM
    else:
        a = to_hl8(x if x.dtype in (torch.float32, torch.float16) else x.float())
    if out is not None or out_row is not None or a_row is not None:
        if N != w.shape[0]:
            raise RuntimeError("split_linear: in-place / row-mapped outputs need N % 8 == 0")
        return gemm(a, w, b, resid, out_fmt=out_fmt, act=act, split=True, tag=tag, out=out, out_row=out_row, a_row=a_row)
    out = gemm(a, w, b, resid, out_fmt=out_fmt, act=act, split=True, tag=tag)
    if N != w.shape[0]:
        out = out[..., :(2 * N if out_fmt == HL8 else N)].contiguous()
    return out


def split_linear_ln_ok(x, weight, norm_weight):
    """the fused projection + residual LayerNorm (hipie_gemm_ln) applies: 256 output features = one column tile, split-able K."""
    return (x.is_cuda and weight.dtype == torch.float32 and weight.shape[0] == 256 and norm_weight.numel() == 256 and weight.shape[1] % 32 == 0
            and not torch.is_grad_enabled())


@_timed("gemm_ln")
def split_linear_ln(x, owner, key, weight, bias, resid, norm_weight, norm_bias, eps, x_hl8=False, want_hl8=True):
    """LayerNorm(resid + F.linear(x, weight, bias)) over 256 features as ONE launch (hipie_gemm_ln): returns (n fp32, n as HL8 or None).
    x (..., K) fp32 rows or HL8 (x_hl8); resid (..., 256) fp32; the HL8 copy of `weight` is cached on `owner` under `key` (split_weight)."""
    lib = _lib.load()
    w, b, N = split_weight(owner, key, [weight] + ([bias] if bias is not None else []), lambda: weight, (lambda: bias) if bias is not None else None)
    if N != 256 or w.shape[0] != 256:
        raise RuntimeError("split_linear_ln: 256 output features expected")
    K = w.shape[1] // 2
    a_f32 = not x_hl8
    if a_f32 and (x.dtype != torch.float32 or x.stride(-1) != 1):
        raise RuntimeError("split_linear_ln: x must be fp32 rows or HL8")
    a2 = x.reshape(-1, x.shape[-1])
    if a2.shape[-1] != (K if a_f32 else 2 * K) or (a_f32 and (a2.stride(0) % 4 or a2.data_ptr() % 16)):
        raise RuntimeError("split_linear_ln: operand rows %s against weight %s" % (tuple(x.shape), tuple(weight.shape)))
    r2 = resid.reshape(-1, 256)
    M = a2.shape[0]
    if r2.dtype != torch.float32 or r2.stride(-1) != 1 or r2.shape[0] != M or not r2.is_cuda:
        raise RuntimeError("split_linear_ln: resid must be (rows, 256) fp32 on the device")
    out = torch.empty(*resid.shape[:-1], 256, dtype=torch.float32, device=x.device)
    o16 = torch.empty(*resid.shape[:-1], 512, dtype=torch.float16, device=x.device) if want_hl8 else None
    rc = lib.hipie_gemm_ln(a2.data_ptr(), a2.stride(0), _chk(w, "w"), w.shape[1], None if b is None else b.data_ptr(), r2.data_ptr(), r2.stride(0),
                           _chk(norm_weight, "norm_weight", torch.float32), _chk(norm_bias, "norm_bias", torch.float32), float(eps),
                           out.data_ptr(), 256, None if o16 is None else o16.data_ptr(), 512, M, K, F32 if a_f32 else HL8, 1.0, _stream())
    _lib.check(rc, "hipie_gemm_ln")
    return out, o16


_SHUFFLE_MAPS = {}


def _shuffle_maps(B, H, W, device):
    """row maps of ConvTranspose2d(k = 2, s = 2) as four linears: input pixel (b, h, w), tap (i, j) -> row of the (B, 2H, 2W, C) output."""
    key = (B, H, W, str(device))
    m = _SHUFFLE_MAPS.get(key)
    if m is None:
        b = torch.arange(B, device=device).view(B, 1, 1)
        h = torch.arange(H, device=device).view(1, H, 1)
        w = torch.arange(W, device=device).view(1, 1, W)
        m = [((b * 2 * H + 2 * h + i) * 2 * W + 2 * w + j).reshape(-1).to(torch.int32).contiguous() for i in range(2) for j in range(2)]
        if len(_SHUFFLE_MAPS) > 16:
            _RETIRED.extend(_SHUFFLE_MAPS.values())
            _SHUFFLE_MAPS.clear()
        _SHUFFLE_MAPS[key] = m
    return m


def convt2x2_split(rows, owner, key, weight, bias, B, H, W):
    """ConvTranspose2d(kernel 2, stride 2) of a channels-last map at fp32-class accuracy: rows (B*H*W, C_in) fp32 pixel rows, weight
    (C_in, C_out, 2, 2), bias (C_out) or None -> (B, 2H, 2W, C_out) fp32 channels-last.  One split linear per tap whose output rows go
    straight to their place in the up-sampled map (hipie_gemm's row map): the pixel shuffle costs no pass of its own."""
    Cin, Cout = weight.shape[0], weight.shape[1]
    if rows.dim() != 2 or rows.shape[0] != B * H * W or rows.shape[1] != Cin or Cout % 8:
        raise RuntimeError("convt2x2_split: rows (B*H*W, C_in), weight (C_in, C_out, 2, 2) with C_out %% 8 == 0, got %s / %s" % (tuple(rows.shape), tuple(weight.shape)))
    out = torch.empty(B, 2 * H, 2 * W, Cout, dtype=torch.float32, device=rows.device)
    o2 = out.view(-1, Cout)
    maps = _shuffle_maps(B, H, W, rows.device)
    a = rows if rows.dtype == torch.float32 and rows.is_contiguous() else rows.float().contiguous()
    for t in range(4):
        i, j = divmod(t, 2)
        split_linear(a, owner, "%s_tap%d" % (key, t), weight, bias, weight_fn=lambda i=i, j=j: weight[:, :, i, j].t().contiguous(),
                     out=o2, out_row=maps[t], tag="gemm_convt")
    return out


@_timed("topk")
def topk(x, k, want_values=False):
    """row-wise top-k of a (rows, n) fp32 device tensor (row stride may exceed n), k <= 1024 -> indices (rows, k) int64 in descending value
    order (ties: ascending index) [, values].  hipie_topk: one launch, hipGraph-replay safe (torch.topk on this stack is neither)."""
    lib = _lib.load()
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.stride(1) != 1:
        raise RuntimeError("topk: (rows, n) fp32 device tensor with contiguous rows expected")
    rows, n = x.shape
    idx = torch.empty(rows, k, dtype=torch.int64, device=x.device)
    val = torch.empty(rows, k, dtype=torch.float32, device=x.device) if want_values else None
    rc = lib.hipie_topk(x.data_ptr(), x.stride(0), rows, n, int(k), idx.data_ptr(), None if val is None else val.data_ptr(), _stream())
    _lib.check(rc, "hipie_topk")
    return (idx, val) if want_values else idx


def topk_ok(x, k):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and 0 < k <= min(1024, x.shape[1])
