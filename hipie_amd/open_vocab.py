"""MaskCLIP score fusion (MODEL.CLIP.ENABLED is on in 10 of the 11 shipped eval yamls).

Mirrors hipie/open_vocab/clip.py (ClipAdapter / MaskCLIP, taken by the reference from ODISE) and the fusion of
HIPIE_IMG.get_clip_logits (hipie_img.py:811-868).  The reference builds the CLIP model with `open_clip.create_model_and_transforms(
"ViT-L-14-336", pretrained="openai")` (open-clip-torch 2.0.2, neither vendored by the reference nor installable here); this module
carries the towers itself, with open_clip's parameter names, so an OpenAI / open_clip checkpoint loads with
`MaskCLIP.load_clip_state_dict`.  (The reference's MaskCLIP.state_dict() is empty -- CLIP weights are never part of a HIPIE checkpoint
-- and so is this one's.)  Without a checkpoint the towers keep their random initialisation: the code path is exercised, the scores
are meaningless; `HIPIE_IMG` says so once.

Device path.  In the reference every token may attend only to the class token and the patches (mask tokens are never attended to,
clip.py:318-321), so the (Q + 577)^2 masked attention decomposes: the 577 image tokens run a plain ViT-L/14 (hipie_flash_attn, head dim
64) that does not depend on the masks at all, and the Q mask tokens -- copies of the class token -- are extra ROWS of the same block
weights that read the image tokens' keys / values of each layer through their (Q x 577) patch masks.  Nothing of size (Q + 577)^2 is
built.  Round 6 (the full-size measurement: 176 ms of CLIP behind a 200 ms bs-8 forward) split the two for good:
  * MaskCLIP.encode_images runs the image tokens ONCE per image and keeps every layer's keys / values (113 MB per image in fp32);
    HIPIE_IMG.inference calls the fusion twice per image (instances, hipie_img.py:592-609; semantic / panoptic, :735-747) -- both read
    the same state, and the images of a batch go through the tower together (M = 577 B rows per GEMM instead of B launches of 577);
  * MaskCLIP.mask_rows runs the mask tokens of all images as one batch of rows, ragged counts padded with fully blocked rows; a mask
    token's keys and values are never used, so its in-projection is the QUERY third only.
The linears are policy-aware (PLinear: the split-fp16 GEMM under Precision.split3), LayerNorm / softmax fp32.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .modeling.transformer import PLinear, _lin

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # the Normalize of open_clip's OpenAI preprocess (clip.py:96)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

CLIP_CONFIGS = {
    # open_clip model_configs/ViT-L-14-336.json (+ quick_gelu: the OpenAI weights); hipie/config.py MODEL.CLIP.NAME default
    "ViT-L-14-336": dict(width=1024, layers=24, heads=16, patch=14, image_size=336, embed_dim=768, text_width=768, text_layers=12,
                         text_heads=12, context=77, vocab=49408, quick_gelu=True),
    "ViT-B-32": dict(width=768, layers=12, heads=12, patch=32, image_size=224, embed_dim=512, text_width=512, text_layers=12,
                     text_heads=8, context=77, vocab=49408, quick_gelu=True),
}


class _Attention(nn.Module):
    """nn.MultiheadAttention's parameters (in_proj_weight, in_proj_bias, out_proj)."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * d, d) * d ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = PLinear(d, d)


class _Mlp(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_fc = PLinear(d, 4 * d)
        self.c_proj = PLinear(4 * d, d)


class ResidualAttentionBlock(nn.Module):
    """open_clip 2.0.2 ResidualAttentionBlock (pre-LN, nn.MultiheadAttention, QuickGELU for the OpenAI weights)."""

    def __init__(self, d, heads, quick_gelu=True):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = _Attention(d)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = _Mlp(d)
        self.heads, self.quick_gelu = heads, quick_gelu

    def _mlp(self, x):
        fc, pj = self.mlp.c_fc, self.mlp.c_proj
        if fc.split and x.is_cuda and fc.weight.dtype == torch.float32 and ops.split_ok(fc.in_features) and ops.split_ok(pj.in_features):
            # the activation in the first GEMM's epilogue, its output as the HL8 operand of the second, the residual in the second's epilogue:
            # no (rows x 4 D) fp32 tensor, no elementwise passes
            h = ops.split_linear(self.ln_2(x), fc, "w", fc.weight, fc.bias, act=ops.ACT_QGELU if self.quick_gelu else ops.ACT_GELU, out_fmt=ops.HL8)
            return ops.split_linear(h, pj, "w", pj.weight, pj.bias, resid=x.contiguous(), x_hl8=True)
        h = fc(self.ln_2(x))
        h = h * torch.sigmoid(1.702 * h) if self.quick_gelu else F.gelu(h)
        return x + pj(h)

    def forward(self, x, add_mask=None):
        """plain block on (N, L, D) with an optional additive (L, L) mask (the text tower's causal mask)."""
        N, L, D = x.shape
        hd = D // self.heads
        qkv = _lin(self.attn, "in", self.ln_1(x), self.attn.in_proj_weight, self.attn.in_proj_bias)
        q, k, v = (t.reshape(N, L, self.heads, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        s = (q * hd ** -0.5) @ k.transpose(-1, -2)
        if add_mask is not None:
            s = s + add_mask
        o = (s.softmax(-1) @ v).transpose(1, 2).reshape(N, L, D)
        return self._mlp(x + self.attn.out_proj(o))

    def forward_image(self, x):
        """the image tokens alone, x (N, T, D) -> (block output, keys (N, T, H, hd), values (N, T, H, hd)) -- what the mask tokens of this
        layer attend to."""
        N, T, D = x.shape
        H, hd = self.heads, D // self.heads
        qkv = _lin(self.attn, "in", self.ln_1(x), self.attn.in_proj_weight, self.attn.in_proj_bias).view(N, T, 3, H, hd)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]                              # (N, T, H, hd) strided views
        if x.is_cuda:
            o = ops.flash_attn(q.half(), k.half(), v.half(), hd ** -0.5, out_f32=True)  # (N, T, D)
        else:
            sc = (q.transpose(1, 2) * hd ** -0.5) @ k.permute(0, 2, 3, 1)
            o = (sc.softmax(-1) @ v.transpose(1, 2)).transpose(1, 2).reshape(N, T, D)
        return self._mlp(x + self.attn.out_proj(o.to(x.dtype))), k, v

    def _q_only(self, h):
        """the query third of the in-projection (a mask token's keys / values are never read)"""
        D = h.shape[-1]
        w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        if getattr(self.attn, "split", False) and h.is_cuda and w.dtype == torch.float32 and ops.split_ok(D):
            return ops.split_linear(h, self.attn, "inq", w, b, weight_fn=lambda: w[:D], bias_fn=lambda: b[:D], params=[w, b])
        return F.linear(h.to(w.dtype), w[:D], b[:D])

    def forward_mask_rows(self, xm, k, v, blocked):
        """the mask tokens' rows: xm (N, Q, D); k, v (N, T, H, hd) of the image tokens of THIS layer; blocked (N, Q, T) bool (True: mask
        token q may not see image token t; the class token, t = 0, is always visible, so no row is fully blocked)."""
        N, Q, D = xm.shape
        H, hd = self.heads, D // self.heads
        q = self._q_only(self.ln_1(xm)).view(N, Q, H, hd)
        if xm.is_cuda and hd in (32, 64) and q.dtype == torch.float32 and k.dtype == torch.float32 and v.dtype == torch.float32:
            # one launch per layer (hipie_attn_split_rows: the mask read per query row) instead of two batched GEMMs, a masked_fill, a softmax
            # and their transposes over the (N, H, Q, T) logits
            om = ops.attn_f32_rows(q, k, v, hd ** -0.5, ~blocked, split=True)
            return self._mlp(xm + self.attn.out_proj(om))
        sc = (q.transpose(1, 2).float() * hd ** -0.5) @ k.permute(0, 2, 3, 1).float()   # (N, H, Q, T)
        sc = sc.masked_fill(blocked[:, None], float("-inf"))
        om = (sc.softmax(-1) @ v.transpose(1, 2).float()).transpose(1, 2).reshape(N, Q, D)
        return self._mlp(xm + self.attn.out_proj(om))

    def forward_masked(self, x, Q, blocked):
        """x (N, Q + T, D) = [Q mask tokens | T image tokens] -> the same layout one block later (the one-call form of the two above)."""
        xi, k, v = self.forward_image(x[:, Q:])
        return torch.cat([self.forward_mask_rows(x[:, :Q], k, v, blocked), xi], dim=1)


class _Transformer(nn.Module):
    def __init__(self, d, layers, heads, quick_gelu):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(d, heads, quick_gelu) for _ in range(layers)])


class _Visual(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, g = c["width"], c["image_size"] // c["patch"]
        self.conv1 = nn.Conv2d(3, d, c["patch"], c["patch"], bias=False)
        self.class_embedding = nn.Parameter(d ** -0.5 * torch.randn(d))
        self.positional_embedding = nn.Parameter(d ** -0.5 * torch.randn(g * g + 1, d))
        self.ln_pre = nn.LayerNorm(d)
        self.transformer = _Transformer(d, c["layers"], c["heads"], c["quick_gelu"])
        self.ln_post = nn.LayerNorm(d)
        self.proj = nn.Parameter(d ** -0.5 * torch.randn(d, c["embed_dim"]))
        self.image_size = c["image_size"]


class CLIP(nn.Module):
    """the parts of open_clip's CLIP the reference touches, with its parameter names."""

    def __init__(self, c):
        super().__init__()
        self.visual = _Visual(c)
        self.transformer = _Transformer(c["text_width"], c["text_layers"], c["text_heads"], c["quick_gelu"])
        self.token_embedding = nn.Embedding(c["vocab"], c["text_width"])
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(c["context"], c["text_width"]))
        self.ln_final = nn.LayerNorm(c["text_width"])
        self.text_projection = nn.Parameter(c["text_width"] ** -0.5 * torch.randn(c["text_width"], c["embed_dim"]))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.context_length = c["context"]

    def encode_text(self, tokens):
        """CLIP.encode_text: causal transformer, the end-of-text token's feature (highest id of the row) @ text_projection."""
        x = self.token_embedding(tokens) + self.positional_embedding
        L = x.shape[1]
        causal = torch.full((L, L), float("-inf"), device=x.device).triu_(1)
        for blk in self.transformer.resblocks:
            x = blk(x, causal)
        x = self.ln_final(x)
        return x[torch.arange(x.shape[0], device=x.device), tokens.argmax(dim=-1)] @ self.text_projection


def ensemble_logits_with_labels(logits, labels):
    """helper.py:77-106 ("max"): one gather + amax over a padded (C, S) index table instead of the per-class loop."""
    lens = [len(l) for l in labels]
    S = max(lens)
    idx = torch.zeros(len(labels), S, dtype=torch.long)
    ok = torch.zeros(len(labels), S, dtype=torch.bool)
    st = 0
    for c, n in enumerate(lens):
        idx[c, :n] = torch.arange(st, st + n)
        ok[c, :n] = True
        st += n
    assert logits.shape[-1] == st, "%d != %d" % (logits.shape[-1], st)
    idx, ok = idx.to(logits.device), ok.to(logits.device)
    return torch.where(ok, logits[..., idx], logits.new_tensor(float("-inf"))).amax(-1)


def prompt_labels_photo(labels):
    """helper.prompt_labels(labels, "photo") (helper.py:111-122)."""
    return [["a photo of a %s." % l for l in syn] for syn in labels]


def load_openseg_labels(dataset="coco_panoptic", prompt_engineered=True, roots=None):
    """get_openseg_labels (hipie/data/coco_dataset_mapper_uni.py; label files under hipie/data/datasets/openseg_labels/): the
    `id:name[,synonym...]` file of the reference installation, found relative to the working directory (launch.py chdirs to the
    repository root) or under $HIPIE_ASSETS.  Returns [{"id", "name"}] like the reference, or None when the file is absent."""
    fn = "%s_with_prompt_eng.txt" % dataset if prompt_engineered else "%s.txt" % dataset
    roots = list(roots or []) + [os.environ.get("HIPIE_ASSETS", ""), "projects/HIPIE/hipie/data/datasets/openseg_labels",
                                 "hipie/data/datasets/openseg_labels", "openseg_labels"]
    for r in roots:
        path = os.path.join(r, fn) if r else None
        if path and os.path.exists(path):
            cats = []
            for line in open(path).read().splitlines():
                i, name = line.split(":", maxsplit=1)
                if name != "invalid_class_id":
                    cats.append({"id": int(i), "name": name})
            return cats
    return None


class MaskCLIP(nn.Module):
    """hipie/open_vocab/clip.py:243-383.  `tokenize`: callable(list of str) -> (n, context) int64 token ids (open_clip.tokenize when
    open_clip is installed; the BPE vocabulary ships inside that package)."""

    def __init__(self, name="ViT-L-14-336", cfg=None, tokenize=None):
        super().__init__()
        self.cfg = dict(cfg or CLIP_CONFIGS[name])
        self.name = name
        self.clip = CLIP(self.cfg)
        self.cache_text = {}
        self.loaded = False
        if tokenize is None:
            try:
                import open_clip
                tokenize = open_clip.tokenize
            except ImportError:
                tokenize = None
        self.tokenize = tokenize
        for p in self.parameters():
            p.requires_grad = False

    # CLIP weights are not part of a HIPIE checkpoint (clip.py:124-126)
    def state_dict(self, *a, **k):
        from collections import OrderedDict
        return OrderedDict()

    def _load_from_state_dict(self, *a, **k):
        return

    def load_clip_state_dict(self, sd, strict=True):
        """open_clip / OpenAI CLIP state dict (keys `visual.conv1.weight`, `transformer.resblocks.0...`, `token_embedding.weight`, ...)."""
        own = {k: v for k, v in nn.Module.state_dict(self.clip).items()}
        missing = [k for k in own if k not in sd]
        if strict and missing:
            raise KeyError("CLIP state dict lacks %d keys, e.g. %s" % (len(missing), missing[:3]))
        with torch.no_grad():
            for k, v in own.items():
                if k in sd:
                    v.copy_(sd[k].to(v.dtype))
        self.loaded = True
        self.cache_text.clear()
        return self

    @property
    def logit_scale(self):
        return torch.clamp(self.clip.logit_scale.exp(), max=100)

    @property
    def image_size(self):
        return (self.clip.visual.image_size, self.clip.visual.image_size)

    # ---- clip.py:291-353 ------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_images(self, image):
        """image (N,3,H,W) in 0..1 -> the mask-independent state of the visual tower: {"cls0": (N,1,D) the ln_pre'd class token every mask
        token starts from, "k" / "v": per layer (N, 577, H, hd) keys / values of the image tokens}.  One pass per image, shared by every
        get_mask_embed / mask_rows call on it."""
        vis, c = self.clip.visual, self.cfg
        image = F.interpolate(image.float(), size=self.image_size, mode="bilinear", align_corners=False)
        mean = torch.tensor(CLIP_MEAN, device=image.device).view(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD, device=image.device).view(1, 3, 1, 1)
        image = (image - mean) / std                                    # clip_preprocess: Resize / CenterCrop are no-ops at this size
        N = image.shape[0]
        x = F.conv2d(image, vis.conv1.weight.float(), None, stride=c["patch"])
        x = x.reshape(N, x.shape[1], -1).permute(0, 2, 1)
        x = torch.cat([vis.class_embedding.float().expand(N, 1, -1), x], dim=1) + vis.positional_embedding.float()
        x = vis.ln_pre(x)
        state = {"cls0": x[:, 0:1].clone(), "k": [], "v": []}
        for blk in vis.transformer.resblocks:
            x, k, v = blk.forward_image(x)
            state["k"].append(k)
            state["v"].append(v)
        return state

    @staticmethod
    def state_of(state, i):
        """the state of image i of a batched state (views)"""
        return {"cls0": state["cls0"][i:i + 1], "k": [k[i:i + 1] for k in state["k"]], "v": [v[i:i + 1] for v in state["v"]]}

    @torch.no_grad()
    def blocked_patches(self, mask):
        """mask (N,Q,h,w) logits -> (N, Q, 1 + g*g) bool, True = the mask token may NOT attend to that image token (clip.py:299-321: a
        patch is visible when the sigmoid mask, resized to the CLIP input, reaches 0.5 somewhere inside it; the class token always is)."""
        c = self.cfg
        mask = F.interpolate(mask.float(), size=self.image_size, mode="bilinear", align_corners=False)
        N, Q = mask.shape[:2]
        patch_mask = F.max_pool2d(mask.sigmoid(), kernel_size=c["patch"], stride=c["patch"])
        return torch.cat([torch.zeros(N, Q, 1, dtype=torch.bool, device=mask.device), (patch_mask < 0.5).reshape(N, Q, -1)], 2)

    @torch.no_grad()
    def blocked_patches_upsampled(self, mask, scale, crop_hw):
        """blocked_patches(F.interpolate(mask, scale_factor=scale, bilinear)[..., :crop_h, :crop_w]) without the up-sampled tensor (5 GB per
        image at 1210 masks of 256 x 256 and scale 4): up-sampling, cropping and the resize to the CLIP input are LINEAR and separable, so
        the (336 x h) row operator and the (336 x w) column operator are built by pushing identity matrices through the same three torch
        operations, and the composite is two small matrix products per mask.  Same arithmetic up to fp32 summation order.
        mask (N,Q,h,w) logits at 1 / scale resolution; crop_hw = the image size the up-sampled logits are cropped to (hipie_img.py:731-747)."""
        N, Q, h, w = mask.shape
        S = self.image_size[0]
        key = (h, w, int(scale), int(crop_hw[0]), int(crop_hw[1]), str(mask.device))
        ops_ = self.__dict__.setdefault("_resize_ops", {})
        if key not in ops_:
            def operator(n, crop):
                eye = torch.eye(n, device=mask.device).view(1, 1, n, n)
                up = F.interpolate(eye, size=(n * int(scale), n), mode="bilinear", align_corners=False)[:, :, :crop]
                return F.interpolate(up, size=(S, n), mode="bilinear", align_corners=False)[0, 0]              # (S, n)
            if len(ops_) > 8:
                ops_.clear()
            ops_[key] = (operator(h, int(crop_hw[0])), operator(w, int(crop_hw[1])))
        A_h, A_w = ops_[key]
        x = torch.matmul(A_h, torch.matmul(mask.float(), A_w.t()))                                            # (N, Q, S, S)
        patch_mask = F.max_pool2d(x.sigmoid(), kernel_size=self.cfg["patch"], stride=self.cfg["patch"])
        return torch.cat([torch.zeros(N, Q, 1, dtype=torch.bool, device=mask.device), (patch_mask < 0.5).reshape(N, Q, -1)], 2)

    @torch.no_grad()
    def mask_rows(self, state, blocked):
        """state of N images (encode_images), blocked (N, Q, T) -> mask embeddings (N, Q, embed_dim)."""
        vis = self.clip.visual
        N, Q = blocked.shape[:2]
        xm = state["cls0"].expand(-1, Q, -1)
        for l, blk in enumerate(vis.transformer.resblocks):
            xm = blk.forward_mask_rows(xm, state["k"][l], state["v"][l], blocked)
        return vis.ln_post(xm) @ vis.proj.float()

    @torch.no_grad()
    def get_mask_embed(self, image, mask, state=None):
        """image (N,3,H,W) in 0..1, mask (N,Q,h,w) logits -> (N,Q,embed_dim).  state: encode_images(image) when the caller already has it."""
        if state is None:
            state = self.encode_images(image)
        return self.mask_rows(state, self.blocked_patches(mask))

    def pred_logits(self, mask_embed, text_embed, labels):
        lg = torch.einsum("bqc,nc->bqn", F.normalize(mask_embed.float(), dim=-1), F.normalize(text_embed.float(), dim=-1)) * self.logit_scale
        return ensemble_logits_with_labels(lg, labels)

    @torch.no_grad()
    def build_text_embed(self, labels):
        """labels: list (classes) of lists (synonym prompts) of str -> (n_prompts, embed_dim); cached per label set (clip.py:367-378)."""
        key = str(labels)
        if key not in self.cache_text:
            if self.tokenize is None:
                raise RuntimeError("MaskCLIP needs a tokenizer: install open_clip (its BPE vocabulary ships inside the package) or pass "
                                   "MaskCLIP(tokenize=...)")
            flat = [t for syn in labels for t in syn]
            dev = self.clip.positional_embedding.device
            out = []
            for i in range(0, len(flat), 256):
                tok = self.tokenize(flat[i:i + 256]).to(dev)[..., :self.clip.context_length]
                out.append(self.clip.encode_text(tok))
            self.cache_text[key] = torch.cat(out, 0)
        return self.cache_text[key]

    @torch.no_grad()
    def forward(self, image, mask, text_embed, labels, state=None):
        emb = self.get_mask_embed(image, mask, state)
        out = {"mask_embed": emb}
        if text_embed is not None and labels is not None:
            out["mask_pred_open_logits"] = self.pred_logits(emb, text_embed, labels)
        return out


def _fuse(lg, pred_open_prob, ov, alpha, beta, agg_mode):
    mp = lg.sigmoid() if lg.shape[-1] == 1 else lg.softmax(dim=-1)
    p = pred_open_prob.float()
    if agg_mode == "ADD":
        base = (p * (1 - alpha) + mp * alpha + 1e-9).log() * ov
        novel = (p * (1 - beta) + mp * beta + 1e-9).log() * (1 - ov)
    else:
        base = (p ** (1 - alpha) * mp ** alpha).log() * ov
        novel = (p ** (1 - beta) * mp ** beta).log() * (1 - ov)
    return base + novel


def _vocab(clip, test_label_names, train_label_names, device):
    labels = prompt_labels_photo(test_label_names)
    train = {l for syn in train_label_names for l in syn}
    ov = torch.tensor([int(not train.isdisjoint(set(syn))) for syn in test_label_names], dtype=torch.long, device=device)
    return labels, ov, clip.build_text_embed(labels).to(device)


def get_clip_logits(clip, image01, mask_logits, test_label_names, train_label_names, pred_open_prob, alpha, beta, agg_mode="MUL", state=None):
    """HIPIE_IMG.get_clip_logits (hipie_img.py:811-868) for ONE image: image01 (3,H,W) in 0..1, mask_logits (Q,h,w), test / train label
    names = lists of synonym lists, pred_open_prob (Q,C) -> fused class logits (Q,C).  state: MaskCLIP.encode_images of this image."""
    labels, ov, text_embed = _vocab(clip, test_label_names, train_label_names, pred_open_prob.device)
    lg = clip(None if image01 is None else image01[None], mask_logits[None], text_embed, labels, state=state)["mask_pred_open_logits"][0]
    return _fuse(lg, pred_open_prob, ov, alpha, beta, agg_mode)


def get_clip_logits_batched(clip, state, mask_logits, test_label_names, train_label_names, pred_open_prob, alpha, beta, agg_mode="MUL", blocked=None):
    """the same for the N images of a batched state in ONE pass of the mask tokens: mask_logits / pred_open_prob are lists of (Q_i,h_i,w_i) /
    (Q_i,C) tensors (ragged Q_i: rows are padded with mask tokens that see the class token only and dropped again) -> list of (Q_i,C).
    Every row's result is what get_clip_logits gives for its image alone (rows of a GEMM do not interact; the softmax is per row).
    blocked: the per-image patch-visibility maps when the caller has them already (blocked_patches_upsampled)."""
    dev = pred_open_prob[0].device
    labels, ov, text_embed = _vocab(clip, test_label_names, train_label_names, dev)
    if blocked is None:                                                        # per image: the (Q_i, 336, 336) resize is the big transient
        blocked = [clip.blocked_patches(m[None])[0] for m in mask_logits]
    qs = [b.shape[0] for b in blocked]
    Qm = max(qs)
    pad = torch.ones(len(qs), Qm, blocked[0].shape[-1], dtype=torch.bool, device=dev)
    pad[:, :, 0] = False
    for i, b in enumerate(blocked):
        pad[i, :qs[i]] = b
    emb = clip.mask_rows(state, pad)
    lg = clip.pred_logits(emb, text_embed, labels)
    return [_fuse(lg[i, :qs[i]], pred_open_prob[i], ov, alpha, beta, agg_mode) for i in range(len(qs))]
