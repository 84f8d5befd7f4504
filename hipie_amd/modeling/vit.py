"""ViT backbone with windowed / global attention and decomposed relative position bias.

Mirror of hipie/backbone/vit.py (ViT, Block, Attention, D2ViT) and hipie/backbone/utils.py with the reference's
parameter names, so reference checkpoints load unchanged.  The attention core runs on the hand-written HIP kernels
(hipie_vit_attn_rel); the linears are library GEMMs.  The residual stream is kept in ``precision.resid`` (fp32 in the
parity and fast policies), the GEMM / attention operands use the policy's 16-bit types.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def get_rel_pos(q_size, k_size, rel_pos):
    """hipie/backbone/utils.py:63-93."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
        r = r.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        r = rel_pos
    q_coords = torch.arange(q_size, device=rel_pos.device)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size, device=rel_pos.device)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


def resize_rel_pos(size, rel_pos):
    """the table-resize half of get_rel_pos for q_size == k_size == size: (2*size-1, hd)."""
    n = 2 * size - 1
    if rel_pos.shape[0] == n:
        return rel_pos
    r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=n, mode="linear")
    return r.reshape(-1, n).permute(1, 0)


def get_abs_pos(abs_pos, has_cls_token, hw):
    """hipie/backbone/utils.py:128-157."""
    h, w = hw
    if has_cls_token:
        abs_pos = abs_pos[:, 1:]
    size = int(math.sqrt(abs_pos.shape[1]))
    if size != h or size != w:
        new = F.interpolate(abs_pos.reshape(1, size, size, -1).permute(0, 3, 1, 2), size=(h, w), mode="bicubic",
                            align_corners=False)
        return new.permute(0, 2, 3, 1)
    return abs_pos.reshape(1, h, w, -1)


def window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    if Hp > H or Wp > W:
        x = x[:, :H, :W, :].contiguous()
    return x


_ROW_MAPS = {}
_PAD_ROWS = {}


def window_pad_rows(out_src):
    """int32 indices of the PADDING rows of the window layout (out_src < 0), cached per row map: computed once per geometry (the
    nonzero is a host round trip), not per block."""
    key = (out_src.data_ptr(), out_src.numel(), str(out_src.device))
    r = _PAD_ROWS.get(key)
    if r is None:
        r = _PAD_ROWS[key] = torch.nonzero(out_src < 0).flatten().to(torch.int32).contiguous()
    return r


class _QkvBuffers(object):
    """window-layout qkv buffers of the windowed ViT blocks (split policy).  The qkv GEMM of such a block runs over the REAL tokens only
    and scatters into the window layout; the padding rows hold the HL8 qkv bias.  Two ways to provide them:

      persistent   one buffer per (block, geometry), stored ON the block (it dies with the model) with its padding rows written once:
                   nothing to refill per step (0.6 GB per block at ViT-H bs 8, 4.9 GB for the 8 windowed blocks) -- while the buffers of
                   all LIVE blocks of the process fit the budget (HIPIE_QKV_CACHE_GB, default 8 GiB; least recently used buffers of OTHER
                   geometries are dropped first, so alternating image sizes do not accumulate);
      shared       beyond the budget (large per-GPU batches): ONE scratch buffer per geometry for all blocks, the padding rows re-written
                   by hipie_fill_rows in front of every block's qkv GEMM (99 MB per block at bs 8) -- memory stays bounded.
    """

    def __init__(self):
        import collections
        import os
        self.budget = int(float(os.environ.get("HIPIE_QKV_CACHE_GB", "8")) * (1 << 30))
        self.entries = collections.OrderedDict()          # id(block) -> (weakref(block), geometry key, bytes); the buffer is on the block
        self.shared = {}                                  # geometry key -> scratch buffer

    def _live_bytes(self):
        for k in [k for k, e in self.entries.items() if e[0]() is None]:
            del self.entries[k]
        return sum(e[2] for e in self.entries.values())

    def get(self, owner, rows, width, device, pad_rows, bias_row, bias_ver):
        import weakref
        # the key names the GEOMETRY, not only the size: two token grids with the same padded window count (50x76 and 50x72 at ws 14:
        # 4704 rows either way) have their padding rows in different places, and rows that were real tokens of the previous geometry
        # would otherwise keep its qkv instead of the bias.  pad_rows is the cached per-geometry index tensor (window_pad_rows), so
        # its address identifies the geometry for the life of the process.
        key = (rows, width, str(device), pad_rows.data_ptr(), pad_rows.numel())
        st = owner.__dict__.get("_qkv_state")
        if st is not None and st[0] == key:
            if id(owner) in self.entries:
                self.entries.move_to_end(id(owner))
            if st[2] != bias_ver:                           # new weights: re-write the padding rows
                ops.fill_rows(st[1], pad_rows, bias_row)
                owner._qkv_state = (key, st[1], bias_ver)
            return st[1]
        owner._qkv_state = None                             # another geometry: this block's old buffer goes first
        self.entries.pop(id(owner), None)
        need = rows * width * 2
        while self._live_bytes() + need > self.budget:      # then least recently used buffers of OTHER geometries
            victim = next((k for k, e in self.entries.items() if e[1] != key), None)
            if victim is None:
                break
            blk = self.entries.pop(victim)[0]()
            if blk is not None:
                blk._qkv_state = None
        if self._live_bytes() + need <= self.budget:
            buf = torch.empty(rows, width, dtype=torch.float16, device=device)
            ops.fill_rows(buf, pad_rows, bias_row)
            owner._qkv_state = (key, buf, bias_ver)
            self.entries[id(owner)] = (weakref.ref(owner), key, need)
            return buf
        buf = self.shared.get(key)
        if buf is None:
            self.shared.clear()                             # one geometry at a time in the shared mode
            buf = self.shared[key] = torch.empty(rows, width, dtype=torch.float16, device=device)
        return ops.fill_rows(buf, pad_rows, bias_row)


QKV_BUFFERS = _QkvBuffers()


def window_row_maps(B, H, W, ws, device):
    """int32 row maps between the token layout (B, H, W) and the padded window layout (B * nWin, ws, ws) of
    window_partition: out_src[window row] = token row or -1 (pad), delta_row[token row] = window row.  Cached per geometry."""
    key = (B, H, W, ws, str(device))
    m = _ROW_MAPS.get(key)
    if m is None:
        assert B * H * W < (1 << 24)
        tok = torch.arange(B * H * W, dtype=torch.float32).view(B, H, W, 1)
        win, _ = window_partition(tok + 1.0, ws)                        # pad -> 0, token r -> r + 1 (exact in fp32 < 2^24)
        out_src = (win.reshape(-1) - 1.0).to(torch.int32)
        valid = out_src >= 0
        delta_row = torch.empty(B * H * W, dtype=torch.int32)
        delta_row[out_src[valid].long()] = torch.nonzero(valid).flatten().to(torch.int32)
        m = (out_src.to(device), delta_row.to(device), win.shape[0])
        _ROW_MAPS[key] = m
    return m


class PatchEmbed(nn.Module):
    """hipie/backbone/utils.py:160-186.  kernel == stride, so the conv is a GEMM over unfolded patches (a library GEMM on
    (B*h*w, 3*p*p) x (3*p*p, E) instead of a 3-input-channel convolution)."""

    def __init__(self, kernel_size=(16, 16), stride=(16, 16), in_chans=3, embed_dim=768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride)
        self.patch = kernel_size[0]

    def forward(self, x):
        B, C, H, W = x.shape
        p = self.patch
        w = self.proj.weight
        x = x.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, H // p, W // p, C * p * p)
        if getattr(self, "split", False) and x.is_cuda and ops.split_ok(x.shape[-1]):
            return ops.split_linear(x.contiguous(), self, "proj", w, self.proj.bias, weight_fn=lambda: w.reshape(w.shape[0], -1))
        return F.linear(x.to(w.dtype), w.reshape(w.shape[0], -1), self.proj.bias)


class Attention(nn.Module):
    """hipie/backbone/vit.py:27-83."""

    def __init__(self, dim, num_heads, input_size, precision):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, dim // num_heads))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, dim // num_heads))
        self.precision = precision

    def forward(self, x):
        """x (B,H,W,C) already LayerNorm-ed, in the GEMM dtype -> (B,H,W,C) in the GEMM dtype."""
        B, H, W, C = x.shape
        nh = self.num_heads
        hd = C // nh
        if ops.vit_attn_rel_ok((H, W), hd):
            # hipie_vit_attn_rel: rel-pos bias computed in the kernel; its operand contract (q rows * scale*log2(e), tables /
            # scale) is met by folding the two constants into a copy of the qkv weights / the tables once (_folded)
            w, b, th, tw = self._folded(H, W)
            qkv16 = F.linear(x.to(w.dtype), w, b).reshape(B, H * W, 3 * C).to(self.precision.attn).contiguous()
            o = ops.vit_attn_rel(qkv16, th, tw, (H, W), nh, fast=self.precision.attn_fast)
        else:       # token grids wider than 96: bias tables through HBM (hipie_vit_relpos + hipie_vit_attn)
            qkv16 = self.qkv(x).reshape(B, H * W, 3 * C).to(self.precision.attn).contiguous()
            th, tw = self._rel_tables(H, W)
            rel_h, rel_w = ops.vit_relpos(qkv16, th, tw, (H, W), nh)
            o = ops.vit_attn(qkv16, rel_h, rel_w, (H, W), nh, self.scale)
        return self.proj(o.to(x.dtype)).view(B, H, W, C)

    def forward_split(self, y, B, H, W, resid, out_row=None, tok2win=None):
        """the split policy: y (B*H*W, 2C) HL8 (LayerNorm output) -> resid + proj(attention(qkv(y))), written IN PLACE into the fp32
        residual stream ``resid`` (rows, C).  qkv leaves its GEMM as HL8 (q rows pre-scaled, fold cached per parameter version), the
        attention forms logits and bias from both halves of both operands (hipie_vit_attn_split) and writes HL8, which the projection
        GEMM consumes; its epilogue adds the residual and -- for the windowed blocks -- stores through ``out_row`` (window row ->
        token row, -1 for the padding): window_unpartition costs no pass."""
        C = self.qkv.weight.shape[1]
        nh = self.num_heads
        c1 = self.scale * ops.LOG2E

        def wq():
            w = self.qkv.weight.detach().float().clone()
            w[:C] *= c1
            return w

        def bq():
            b = self.qkv.bias.detach().float().clone()
            b[:C] *= c1
            return b
        if getattr(self.precision, "vit_attn16", False) and ops.vit_attn_rel_ok((H, W), C // nh):
            # `mixed` policy: split linears around the single-fp16 attention core (hipie_vit_attn_rel; its output goes back to HL8 with a
            # zero lo half for the projection GEMM)
            qkv16 = ops.split_linear(y, self, "qkv", self.qkv.weight, self.qkv.bias, out_fmt=ops.F16, x_hl8=True, weight_fn=wq, bias_fn=bq,
                                     tag="gemm_qkv")
            key = (H, W, "f16", self._versions())
            if getattr(self, "_tabs16_key", None) != key:
                self._tabs16 = ((resize_rel_pos(H, self.rel_pos_h.detach().float()) / self.scale).half().contiguous(),
                                (resize_rel_pos(W, self.rel_pos_w.detach().float()) / self.scale).half().contiguous())
                self._tabs16_key = key
            o16 = ops.vit_attn_rel(qkv16.view(B, H * W, 3 * C), self._tabs16[0], self._tabs16[1], (H, W), nh, fast=True)
            o = ops.to_hl8(o16.view(B * H * W, C))
            return ops.split_linear(o, self, "proj", self.proj.weight, self.proj.bias, x_hl8=True, tag="gemm_proj",
                                    resid=resid, out=resid, out_row=out_row)
        if tok2win is not None:
            # windowed block: the qkv rows of the PADDING tokens are the bias (their LayerNorm output is zero, utils.py:29-37) and their
            # projections are discarded -- the two GEMMs run over the real tokens only (gather / scatter through the window row map) and the
            # padding rows are filled with the HL8 bias row
            rows = y.shape[0]
            w_, b_, _ = ops.split_weight(self, "qkv", [self.qkv.weight, self.qkv.bias], wq, bq)
            ver = self._versions()
            if getattr(self, "_bias_hl8_key", None) != ver:
                self._bias_hl8, self._bias_hl8_key = ops.to_hl8(b_.view(1, -1)).view(-1), ver
            # the window-layout qkv buffer with the bias in its padding rows: per block while the process-wide budget allows (padding
            # written once), else one shared scratch buffer refilled per block (QKV_BUFFERS)
            qkv = QKV_BUFFERS.get(self, rows, 6 * C, y.device, window_pad_rows(out_row), self._bias_hl8, ver)
            ops.split_linear(y, self, "qkv", self.qkv.weight, self.qkv.bias, out_fmt=ops.HL8, x_hl8=True, weight_fn=wq, bias_fn=bq,
                             tag="gemm_qkv", out=qkv, out_row=tok2win, a_row=tok2win)
        else:
            qkv = ops.split_linear(y, self, "qkv", self.qkv.weight, self.qkv.bias, out_fmt=ops.HL8, x_hl8=True, weight_fn=wq, bias_fn=bq,
                                   tag="gemm_qkv")
        key = (H, W, self._versions())
        if getattr(self, "_tabs_key", None) != key:
            self._tabs = (ops.hl8_pack(resize_rel_pos(H, self.rel_pos_h.detach().float()) / self.scale),
                          ops.hl8_pack(resize_rel_pos(W, self.rel_pos_w.detach().float()) / self.scale))
            self._tabs_key = key
        o = ops.vit_attn_split(qkv.view(B, H * W, 6 * C), self._tabs[0], self._tabs[1], (H, W), nh)
        if tok2win is not None:          # the projection reads the real tokens' rows out of the window layout and writes token order
            return ops.split_linear(o.view(B * H * W, 2 * C), self, "proj", self.proj.weight, self.proj.bias, x_hl8=True, tag="gemm_proj",
                                    resid=resid, out=resid, a_row=tok2win)
        return ops.split_linear(o.view(B * H * W, 2 * C), self, "proj", self.proj.weight, self.proj.bias, x_hl8=True, tag="gemm_proj",
                                resid=resid, out=resid, out_row=out_row)

    def _versions(self):
        ps = (self.qkv.weight, self.qkv.bias, self.rel_pos_h, self.rel_pos_w)
        return tuple((p.data_ptr(), p._version, p.dtype, str(p.device)) for p in ps)

    def _folded(self, H, W):
        """(qkv weight, bias) with the q rows multiplied by scale*log2(e) and the two rel-pos tables (re-interpolated to
        2*size-1 rows, get_rel_pos utils.py:63-86) divided by scale, rounded ONCE to the GEMM / attention dtypes.  Cached per
        token grid; the key carries the parameters' version counters, so load_state_dict / .to() invalidate it."""
        key = (H, W, self.precision.attn, self._versions())
        if getattr(self, "_fold_key", None) != key:
            C = self.qkv.weight.shape[1]
            c1 = self.scale * ops.LOG2E
            w = self.qkv.weight.detach().float().clone()
            b = self.qkv.bias.detach().float().clone()
            w[:C] *= c1
            b[:C] *= c1
            gd = self.qkv.weight.dtype
            th = (resize_rel_pos(H, self.rel_pos_h.detach().float()) / self.scale).to(self.precision.attn).contiguous()
            tw = (resize_rel_pos(W, self.rel_pos_w.detach().float()) / self.scale).to(self.precision.attn).contiguous()
            self._fold = (w.to(gd).contiguous(), b.to(gd).contiguous(), th, tw)
            self._fold_key = key
        return self._fold


    def _rel_tables(self, H, W):
        """rel_pos_h / rel_pos_w linearly re-interpolated to 2*size-1 rows when needed (get_rel_pos, utils.py:63-86), in the
        attention operand dtype, cached per token grid.  Entry [hq - hk + H - 1] is Rh[hq, hk] (:88-93)."""
        key = (H, W, self.precision.attn, self._versions())
        if getattr(self, "_rel_key", None) != key:
            self._rel_cache = (resize_rel_pos(H, self.rel_pos_h.float()).to(self.precision.attn).contiguous(),
                               resize_rel_pos(W, self.rel_pos_w.float()).to(self.precision.attn).contiguous())
            self._rel_key = key
        return self._rel_cache


class Mlp(nn.Module):
    """timm.models.layers.Mlp as used at vit.py:193-197 (exact-erf GELU)."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))

    def forward_split(self, h, resid):
        """h (rows, 2C) HL8 -> resid + mlp(h), IN PLACE in the fp32 stream ``resid`` (rows, C): fc1 with the exact-erf GELU and the HL8
        split in its epilogue, fc2 on that operand with the residual add in its epilogue."""
        mid = ops.split_linear(h, self, "fc1", self.fc1.weight, self.fc1.bias, act=ops.ACT_GELU, out_fmt=ops.HL8, x_hl8=True, tag="gemm_fc1")
        return ops.split_linear(mid, self, "fc2", self.fc2.weight, self.fc2.bias, x_hl8=True, tag="gemm_fc2", resid=resid, out=resid)


class Block(nn.Module):
    """hipie/backbone/vit.py:147-230; LayerNorm eps 1e-6."""

    def __init__(self, dim, num_heads, mlp_ratio, window_size, input_size, precision):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, input_size if window_size == 0 else (window_size, window_size), precision)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.window_size = window_size
        self.precision = precision

    def forward(self, x, delta=None):
        """x: residual stream (B,H,W,C) in the policy's activation dtype; ``delta``: the previous block's MLP output that
        has not been added yet.  Returns (x, delta'): every residual add is fused with the LayerNorm that follows it
        (hipie_add_layernorm), so the stream is read and written once per half-block."""
        gd = self.precision.gemm
        ws = self.window_size
        if self.precision.split and x.is_cuda:
            return self._forward_split(x, delta)
        if ws == 0:
            x, y = ops.add_layernorm(x, delta, self.norm1.weight, self.norm1.bias, self.norm1.eps, gd)
            y = self.attn(y)
            x, h = ops.add_layernorm(x, y.contiguous(), self.norm2.weight, self.norm2.bias, self.norm2.eps, gd)
            return x, self.mlp(h)
        # windowed block: window_partition (zero padding included) is the output row map of the first fused add+LN and
        # window_unpartition the delta row map of the second one (utils.py:16-60) -- no partition / crop copies
        B, H, W, C = x.shape
        out_src, delta_row, nwin = window_row_maps(B, H, W, ws, x.device)
        x, y = ops.add_layernorm(x, delta, self.norm1.weight, self.norm1.bias, self.norm1.eps, gd, out_src=out_src)
        y = self.attn(y.view(nwin, ws, ws, C))
        x, h = ops.add_layernorm(x, y.reshape(-1, C), self.norm2.weight, self.norm2.bias, self.norm2.eps, gd, delta_row=delta_row)
        return x, self.mlp(h)


def _block_forward_split(self, x, delta):
    """Block.forward of the split policy.  The fp32 residual stream is updated IN PLACE by the epilogues of the projection and fc2
    GEMMs (x += proj(attn), x += mlp), so each LayerNorm pass only reads the stream once and writes the next GEMM operand as HL8;
    window partition = the row map of the first LayerNorm pass, window un-partition = the store index of the projection GEMM."""
    B, H, W, C = x.shape
    ws = self.window_size
    n1, n2 = self.norm1, self.norm2
    if delta is not None:
        x = x + delta
    xs = x.view(B * H * W, C)
    if ws == 0:
        if not ops.vit_attn_split_ok((H, W), C // self.attn.num_heads):
            raise NotImplementedError("split policy: global attention on a %dx%d token grid (not covered: wider than 96 tokens in BOTH directions)" % (H, W))
        y = ops.add_layernorm(x, None, n1.weight, n1.bias, n1.eps, "hl8")[1]
        self.attn.forward_split(y.view(B * H * W, 2 * C), B, H, W, xs)
    else:
        if not ops.vit_attn_split_ok((ws, ws), C // self.attn.num_heads):
            raise NotImplementedError("split policy: window size %d" % ws)
        out_src, delta_row, nwin = window_row_maps(B, H, W, ws, x.device)
        y = ops.add_layernorm(x, None, n1.weight, n1.bias, n1.eps, "hl8", out_src=out_src)[1]
        trim = not getattr(self.attn.precision, "vit_attn16", False)
        self.attn.forward_split(y, nwin, ws, ws, xs, out_row=out_src, tok2win=delta_row if trim else None)
    h = ops.add_layernorm(x, None, n2.weight, n2.bias, n2.eps, "hl8")[1]
    self.mlp.forward_split(h.view(B * H * W, 2 * C), xs)
    return x, None


Block._forward_split = _block_forward_split


class ViT(nn.Module):
    """hipie/backbone/vit.py:233-374 (use_abs_pos, use_rel_pos, no residual conv blocks, pretrain cls token)."""

    def __init__(self, cfg, precision):
        super().__init__()
        E = cfg.vit_embed_dim
        self.patch_embed = PatchEmbed((cfg.vit_patch,) * 2, (cfg.vit_patch,) * 2, 3, E)
        n = (cfg.vit_pretrain_img_size // cfg.vit_patch) ** 2 + 1
        self.pos_embed = nn.Parameter(torch.zeros(1, n, E))
        g = cfg.vit_img_size // cfg.vit_patch
        self.blocks = nn.ModuleList([
            Block(E, cfg.vit_heads, cfg.vit_mlp_ratio, cfg.vit_window if i in cfg.vit_window_blocks else 0, (g, g), precision)
            for i in range(cfg.vit_depth)])
        self.fpn1 = nn.Sequential(nn.ConvTranspose2d(E, E // 2, kernel_size=2, stride=2))
        self.fpn2 = nn.Identity()
        self.fpn3 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.precision = precision
        self._out_feature_channels = {"res3": E // 2, "res4": E, "res5": E}
        self._out_feature_strides = {"res3": 8, "res4": 16, "res5": 32}
        nn.init.trunc_normal_(self.pos_embed, std=0.02)

    size_divisibility = 32

    def forward(self, x):
        """x (B,3,H,W) fp32 normalised image -> {"res3","res4","res5"}: logical NCHW, channels-last memory, activation dtype."""
        gd, ad, rd = self.precision.gemm, self.precision.act, self.precision.resid
        self.patch_embed.split = self.precision.split
        x = self.patch_embed(x).float()
        x = (x + self._abs_pos((x.shape[1], x.shape[2]))).to(rd)
        x, delta = x.contiguous(), None
        for blk in self.blocks:
            x, delta = blk(x, delta)
        x = (x if delta is None else x + delta.to(x.dtype)).to(ad)
        # fpn1: ConvTranspose2d(k=2, s=2) == one GEMM (E -> 4 * E/2, bias in the epilogue) + a pixel shuffle (vit.py:341-343).
        # The features leave in the activation dtype and in channels-last memory (logical NCHW): the 1x1 / 3x3 projections
        # that consume them run NHWC, so no layout or dtype copy sits between the backbone and the heads.
        B, H, W, E = x.shape
        wt = self.fpn1[0].weight                                           # (E, E/2, 2, 2)
        b4 = self.fpn1[0].bias.to(wt.dtype).repeat_interleave(4)
        if self.precision.split and x.is_cuda and ops.split_ok(E) and (E // 2) % 8 == 0:
            # one linear per tap, each writing its rows of the up-sampled map directly (ops.convt2x2_split): no shuffle pass
            res3 = ops.convt2x2_split(x.reshape(B * H * W, E), self, "fpn1", wt, self.fpn1[0].bias, B, H, W).to(ad).permute(0, 3, 1, 2)
        else:
            y = F.linear(x.to(wt.dtype), wt.reshape(E, -1).t(), b4).view(B, H, W, E // 2, 2, 2)
            res3 = y.permute(0, 1, 4, 2, 5, 3).reshape(B, 2 * H, 2 * W, E // 2).to(ad).permute(0, 3, 1, 2)
        xp = x.permute(0, 3, 1, 2)
        return {"res3": res3, "res4": xp, "res5": self.fpn3(xp)}

    def _abs_pos(self, hw):
        """bicubic-resized absolute position table, cached per token grid (weights are frozen at inference)."""
        key = (hw, self.pos_embed.data_ptr(), self.pos_embed._version, str(self.pos_embed.device))
        if getattr(self, "_abs_pos_key", None) != key:
            self._abs_pos_cache = get_abs_pos(self.pos_embed.float(), True, hw)
            self._abs_pos_key = key
        return self._abs_pos_cache

    def cast_weights(self):
        """put the GEMM/conv weights in the policy dtype (norms, pos tables stay fp32)."""
        gd = self.precision.gemm
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
                m.to(gd)
        return self


class D2ViT(ViT):
    """registered as "D2ViT" in detectron2's BACKBONE_REGISTRY when detectron2 is importable (hipie_amd/d2_registry.py)."""

    def output_shape(self):
        return {k: dict(channels=self._out_feature_channels[k], stride=self._out_feature_strides[k]) for k in ("res3", "res4", "res5")}
