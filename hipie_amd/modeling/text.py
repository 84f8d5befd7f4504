"""BERT text encoder: BertEncoder wrapper of hipie/models/deformable_detr/bert_model.py:11-153 with its
long-prompt chunking, around a BERT-base encoder with HuggingFace's parameter names (the reference calls
transformers.BertModel; the keys ``text_encoder.body.model.*`` of reference checkpoints load unchanged).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class BertSelfAttention(nn.Module):
    def __init__(self, h, heads):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)
        self.heads = heads

    def forward(self, x, ext_mask):
        B, L, C = x.shape
        hd = C // self.heads

        def sp(t):
            return t.view(B, L, self.heads, hd).permute(0, 2, 1, 3)
        s = sp(self.query(x)) @ sp(self.key(x)).transpose(-1, -2) / math.sqrt(hd) + ext_mask
        return (s.softmax(-1) @ sp(self.value(x))).permute(0, 2, 1, 3).reshape(B, L, C)


class BertSelfOutput(nn.Module):
    def __init__(self, hin, h):
        super().__init__()
        self.dense = nn.Linear(hin, h)
        self.LayerNorm = nn.LayerNorm(h, eps=1e-12)

    def forward(self, x, res):
        return self.LayerNorm(self.dense(x) + res)


class BertAttention(nn.Module):
    def __init__(self, h, heads):
        super().__init__()
        self.self = BertSelfAttention(h, heads)
        self.output = BertSelfOutput(h, h)

    def forward(self, x, ext_mask):
        return self.output(self.self(x, ext_mask), x)


class BertIntermediate(nn.Module):
    def __init__(self, h, inter):
        super().__init__()
        self.dense = nn.Linear(h, inter)

    def forward(self, x):
        return F.gelu(self.dense(x))


class BertLayer(nn.Module):
    def __init__(self, h, heads, inter):
        super().__init__()
        self.attention = BertAttention(h, heads)
        self.intermediate = BertIntermediate(h, inter)
        self.output = BertSelfOutput(inter, h)

    def forward(self, x, ext_mask):
        x = self.attention(x, ext_mask)
        return self.output(self.intermediate(x), x)


class BertEmbeddings(nn.Module):
    def __init__(self, vocab, h, max_pos):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, h)
        self.position_embeddings = nn.Embedding(max_pos, h)
        self.token_type_embeddings = nn.Embedding(2, h)
        self.LayerNorm = nn.LayerNorm(h, eps=1e-12)

    def forward(self, ids):
        L = ids.shape[1]
        x = self.word_embeddings(ids) + self.position_embeddings.weight[:L][None] + self.token_type_embeddings.weight[0][None, None]
        return self.LayerNorm(x)


class BertLayers(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(cfg.bert_hidden, cfg.bert_heads, cfg.bert_intermediate) for _ in range(cfg.bert_layers)])


class BertModel(nn.Module):
    """BERT-base encoder, last hidden state (add_pooling_layer=False).

    fp32 (the parity policy): the plain formulation below.  With a 16-bit ``compute_dtype`` (set_compute_dtype, called from
    HIPIE_IMG.finalize) a layer is 8 launches: one fused QKV GEMM, hipie_flash_attn with the padding mask as key mask
    (the additive finfo.min mask of HuggingFace's BertSelfAttention zeroes the same keys), the output GEMM,
    hipie_add_layernorm (post-norm residual, fp32 statistics), and the GELU MLP."""

    def __init__(self, cfg):
        super().__init__()
        self.embeddings = BertEmbeddings(cfg.bert_vocab, cfg.bert_hidden, cfg.bert_max_pos)
        self.encoder = BertLayers(cfg)
        self.compute_dtype = torch.float32
        self._fused = None

    def set_compute_dtype(self, dtype):
        """cast the GEMM weights (norms and embeddings stay fp32) and build the fused QKV weights; call after loading."""
        self.compute_dtype = dtype
        self._fused = None
        if dtype == torch.float32:
            return self
        fused = []
        for layer in self.encoder.layer:
            at = layer.attention.self
            fused.append((torch.cat([at.query.weight, at.key.weight, at.value.weight]).to(dtype).contiguous(),
                          torch.cat([at.query.bias, at.key.bias, at.value.bias]).to(dtype).contiguous()))
            for lin in (layer.attention.output.dense, layer.intermediate.dense, layer.output.dense):
                lin.weight.data = lin.weight.data.to(dtype)
                lin.bias.data = lin.bias.data.to(dtype)
        self._fused = fused
        self._fused_key = self._qkv_versions()
        return self

    def _qkv_versions(self):
        ps = []
        for layer in self.encoder.layer:
            at = layer.attention.self
            ps += [at.query.weight, at.key.weight, at.value.weight, at.query.bias, at.key.bias, at.value.bias]
        return tuple((q.data_ptr(), q._version) for q in ps)

    def _forward16(self, input_ids, attention_mask):
        if self._fused is None or getattr(self, "_fused_key", None) != self._qkv_versions():
            # a sub-module load_state_dict / in-place update after finalize(): rebuild the fused 16-bit QKV weights (the other dense
            # layers are cast in place by set_compute_dtype, which a load into fp16 parameters preserves)
            self.set_compute_dtype(self.compute_dtype)
        dt = self.compute_dtype
        emb = self.embeddings
        L = input_ids.shape[1]
        x = emb.word_embeddings(input_ids) + emb.position_embeddings.weight[:L][None] + emb.token_type_embeddings.weight[0][None, None]
        ln = emb.LayerNorm
        # the post-norm stream stays fp32 (the LayerNorm outputs are both the next residual and the next GEMM input; only the
        # GEMM input is rounded), the GEMMs and the attention run on 16-bit operands
        f32 = torch.float32
        x = ops.add_layernorm(x.float().contiguous(), None, ln.weight, ln.bias, ln.eps, f32)[1]
        B, L, C = x.shape
        heads = self.encoder.layer[0].attention.self.heads
        hd = C // heads
        kmask = attention_mask.to(torch.uint8).contiguous()
        x16 = x.to(dt)
        for layer, (wqkv, bqkv) in zip(self.encoder.layer, self._fused):
            qkv = F.linear(x16, wqkv, bqkv).view(B, L, 3, heads, hd)
            ctx = ops.flash_attn(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 1.0 / math.sqrt(hd), key_mask=kmask)
            so = layer.attention.output                       # the LayerNorm pass also emits the 16-bit GEMM operand
            x, x16, _ = ops.add_layernorm_dec(x, so.dense(ctx), so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.eps, dt, want16=True)
            oo = layer.output
            x, x16, _ = ops.add_layernorm_dec(x, oo.dense(F.gelu(layer.intermediate.dense(x16))), oo.LayerNorm.weight,
                                              oo.LayerNorm.bias, oo.LayerNorm.eps, dt, want16=True)
        return x

    def _forward_split(self, input_ids, attention_mask):
        """Precision.split3: every dense layer as the split-fp16 GEMM (fused QKV, fp32-class), the attention core in fp32
        (hipie_attn_split: tests/study/prec_sim.py on the full-size fixture -- single-fp16 q / k / v here move pred_masks by 4e-3), the post-norm
        stream fp32; the LayerNorm passes emit the next GEMM operand as HL8, the intermediate GEMM applies the exact-erf GELU and
        writes HL8 for the output GEMM.  8 launches per layer."""
        emb = self.embeddings
        L = input_ids.shape[1]
        x = emb.word_embeddings(input_ids) + emb.position_embeddings.weight[:L][None] + emb.token_type_embeddings.weight[0][None, None]
        ln = emb.LayerNorm
        x, xh = ops.add_layernorm(x.float().contiguous(), None, ln.weight, ln.bias, ln.eps, torch.float32)[1], None
        B, L, C = x.shape
        heads = self.encoder.layer[0].attention.self.heads
        hd = C // heads
        kmask = attention_mask.to(torch.uint8).contiguous()
        xh = ops.to_hl8(x)
        for layer in self.encoder.layer:
            at = layer.attention.self
            exact = ops.attn_f32_ok(hd)          # the attention core at fp32-class accuracy (hipie_attn_split): at the headline configuration single-fp16 q / k / v
            qkv = ops.split_linear(xh, at, "qkv", at.query.weight, None, out_fmt=ops.F32 if exact else ops.F16, x_hl8=True,   # here cost 4e-3
                                   weight_fn=lambda at=at: torch.cat([at.query.weight, at.key.weight, at.value.weight]),
                                   bias_fn=lambda at=at: torch.cat([at.query.bias, at.key.bias, at.value.bias]),
                                   params=(at.query.weight, at.key.weight, at.value.weight, at.query.bias, at.key.bias, at.value.bias)
                                   ).view(B, L, 3, heads, hd)
            if exact:
                ctx = ops.attn_split(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 1.0 / math.sqrt(hd), key_mask=kmask)
            else:
                ctx = ops.flash_attn(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 1.0 / math.sqrt(hd), key_mask=kmask, out_f32=True)
            so = layer.attention.output
            d = ops.split_linear(ctx, so, "dense", so.dense.weight, so.dense.bias)
            x, xh, _ = ops.add_layernorm_dec(x, d, so.LayerNorm.weight, so.LayerNorm.bias, so.LayerNorm.eps, "hl8", want16=True)
            oo = layer.output
            mid = ops.split_linear(xh, layer.intermediate, "dense", layer.intermediate.dense.weight, layer.intermediate.dense.bias,
                                   act=ops.ACT_GELU, out_fmt=ops.HL8, x_hl8=True)
            d = ops.split_linear(mid, oo, "dense", oo.dense.weight, oo.dense.bias, x_hl8=True)
            x, xh, _ = ops.add_layernorm_dec(x, d, oo.LayerNorm.weight, oo.LayerNorm.bias, oo.LayerNorm.eps, "hl8", want16=True)
        return x

    def forward(self, input_ids, attention_mask):
        if getattr(self, "split", False) and input_ids.is_cuda and self.compute_dtype == torch.float32:
            return self._forward_split(input_ids, attention_mask)
        if self.compute_dtype != torch.float32:
            return self._forward16(input_ids, attention_mask)
        x = self.embeddings(input_ids)
        ext = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
        for layer in self.encoder.layer:
            x = layer(x, ext)
        return x


class BertEncoder(nn.Module):
    """hipie/models/deformable_detr/bert_model.py:11-153 (PARALLEL_DET False)."""

    def __init__(self, cfg):
        super().__init__()
        self.model = BertModel(cfg)
        self.language_dim = cfg.bert_hidden

    def forward(self, x, task=None, sep=1012, compact=False):
        """x["input_ids"], x["attention_mask"] (B, L) on the host (as a tokenizer hands them over) or on the device ->
        {"masks", "hidden", "n_keys"[, "full_len"]}.  "n_keys" (when the mask is known on the host -- no device sync is spent on it):
        the number of leading token columns that hold every attended token, rounded up to 32; the masked keys behind it have
        probability exactly 0 in the image -> text softmax (fuse_helper.py:96-109: -9e15) and are not computed.
        compact=True, prompts longer than 512 tokens only: the > 512 branch of the reference leaves the hidden states behind the last
        chunk ZERO (bert_model.py:118-127), so with PAD_MAX (hipie_img.py:904-909; MAX_QUERY_LEN 4096 in the shipped eval yamls) every padding
        row of the language stream is the same row all the way to the class logits.  The returned "hidden" / "masks" then hold the
        first `live` rows only -- every real token plus at least one padding row, `live` a multiple of 32 -- and "full_len" = L: the
        consumers (fusion, VL_Align) compute `live` columns and repeat the last one (transformer.expand_tokens)."""
        ids, mask = x["input_ids"], x["attention_mask"]
        dev = self.model.embeddings.word_embeddings.weight.device
        B, L = ids.shape
        mask_h = None if mask.is_cuda else mask
        if L <= 512:
            ids, mask = _h2d(ids, dev), _h2d(mask, dev)
            out = {"masks": mask, "hidden": self.model(ids, mask)}
            if mask_h is not None:
                out["n_keys"] = _n_keys(mask_h)
            return out
        CLS, EOS = 101, 102
        # host-side string-like processing of the token rows (bert_model.py:74-110): on host copies, so that the loop costs no device
        # round trips (a device-resident prompt pays ONE copy here)
        ids_h = ids.cpu() if ids.is_cuda else ids
        mask_h = mask.cpu() if mask_h is None else mask_h
        chunks, covered = [], 0
        for b in range(B):
            inp = ids_h[b].clone()
            begin, start_src = 0, 0
            while True:
                seps = torch.where((inp == sep) | (inp == EOS))[0]
                seps = seps[seps < 510]
                if len(seps) == 0:
                    break
                last = int(seps[-1])
                first = inp[:last + 1].clone()
                first[-1] = EOS
                on = torch.where(mask_h[b][:last + 1] == 1)[0]         # the reference never advances the mask row (bert_model.py:87)
                n = len(first)
                out_mask = torch.zeros(512, dtype=ids_h.dtype)
                if start_src == 0:
                    row = torch.cat([first, torch.zeros(512 - n, dtype=ids_h.dtype)])
                    out_mask[on] = 1
                else:
                    pad = torch.zeros(512 - n - 1, dtype=ids_h.dtype)
                    pad[0] = sep
                    row = torch.cat([torch.tensor([CLS], dtype=ids_h.dtype), first, pad])
                    out_mask[on + 1] = 1
                    out_mask[0] = 1
                chunks.append((b, row, out_mask, (start_src, start_src + n, begin, begin + n)))
                start_src = 1
                inp = inp[n:]
                begin += n
                covered = max(covered, begin)
        hid = self.model(_h2d(torch.stack([c[1] for c in chunks]), dev), _h2d(torch.stack([c[2] for c in chunks]), dev))
        attended = _n_attended(mask_h)
        live = L
        if compact:
            # rows >= max(covered, attended) have zero hidden states AND a zero mask: all alike.  Keep one of them.
            live = min(L, (max(covered, attended) + 1 + 31) // 32 * 32)
        out = torch.zeros(B, live, hid.shape[-1], dtype=torch.float32, device=dev)
        for i, (b, _, _, (s0, s1, t0, t1)) in enumerate(chunks):
            out[b, t0:t1] = hid[i, s0:s1]
        res = {"masks": _h2d(mask_h[:, :live].contiguous(), dev), "hidden": out, "n_keys": min(live, (max(attended, 1) + 31) // 32 * 32)}
        if live < L:
            res["full_len"] = L
        return res


def _h2d(t, dev):
    """host tensor -> device without synchronising the stream: through a pinned staging block (torch's caching host allocator recycles
    it only after the copy has completed) and a non-blocking copy.  A plain .to(device) of pageable memory waits for everything queued
    before it -- at the head of a step that is the whole previous step, and the host falls behind on the latency-bound configurations
    (configs[0]: 17.7 instead of 14.6 ms per image)."""
    if t.is_cuda or torch.device(dev).type != "cuda":
        return t.to(dev)
    pin = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pin.copy_(t)
    return pin.to(dev, non_blocking=True)


def _n_attended(mask_h):
    """index of the last attended token + 1 over the batch (host tensor)"""
    nz = torch.nonzero(mask_h.reshape(mask_h.shape[0], -1) != 0)
    return int(nz[:, 1].max()) + 1 if nz.numel() else 0


def _n_keys(mask_h):
    L = mask_h.shape[1]
    return min(L, (max(_n_attended(mask_h), 1) + 31) // 32 * 32)
