"""MaskCLIP score fusion: CPU restatement (test infrastructure -- see oracle/__init__.py).

Two layers:

* THIRD-PARTY arithmetic -- **parity unpinned**.  The reference calls `open_clip` (open-clip-torch, pinned to 2.0.2 by ODISE, from which
  hipie/open_vocab/clip.py is taken; neither vendored in /root/reference nor installed here, and its weights / BPE vocabulary cannot be
  fetched).  `clip_visual_*`, `clip_resblock`, `clip_encode_text` below restate the PUBLISHED architecture of open_clip 2.0.2's
  `CLIP` / `VisualTransformer` / `Transformer` / `ResidualAttentionBlock` for the OpenAI weights ("ViT-L-14-336", pretrained="openai":
  QuickGELU, pre-LN blocks around nn.MultiheadAttention, a BOOLEAN attn_mask means "True = may not attend" -- 2.0.2 passes the mask to
  nn.MultiheadAttention unchanged).  The reference has no test or golden vector at this boundary; the call sites that anchor the
  restatement are hipie/open_vocab/clip.py:85-92 (model creation), :152-163 (text encoder), :258-289 (vision tower with mask tokens).
* The REFERENCE's own logic on top of it -- pinned by tests/golden/maskclip.npz, which is produced by running the reference's
  `MaskCLIP` / `HIPIE_IMG.get_clip_logits` / `HIPIE_IMG.inference` themselves over a stand-in `open_clip` module (tests/golden/
  ref_shim.py) that implements the same published architecture: mask-token construction and the attention mask
  (clip.py:291-337), resizing (:339-353), cosine logits + synonym ensembling (:355-365, helper.py:77-106), the base / novel class
  geometric (or arithmetic) fusion (hipie_img.py:811-868) and its two call sites (:592-609 instances, :735-747 panoptic).

State-dict keys are open_clip's (`visual.conv1.weight`, `visual.transformer.resblocks.N....`, `token_embedding.weight`, ...) under a
prefix `p`.  cfg: dict(width, layers, heads, patch, image_size, embed_dim, text_width, text_layers, text_heads, context, vocab, quick_gelu).
"""
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # open_clip.constants.OPENAI_DATASET_MEAN / _STD (the preprocess Normalize)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _ln(x, sd, p):
    return F.layer_norm(x, x.shape[-1:], sd[p + "weight"], sd[p + "bias"], 1e-5)


def clip_resblock(x, sd, p, heads, mask=None, quick_gelu=True):
    """open_clip 2.0.2 ResidualAttentionBlock, batch-first: x (N, L, D).  mask: None, bool (N*heads, L, L) / (L, L) with True = blocked,
    or a float (L, L) additive mask (the text tower's causal mask)."""
    N, L, D = x.shape
    hd = D // heads
    h = _ln(x, sd, p + "ln_1.")
    qkv = F.linear(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
    q, k, v = (t.reshape(N, L, heads, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))         # (N, heads, L, hd)
    s = (q * hd ** -0.5) @ k.transpose(-1, -2)
    if mask is not None:
        if mask.dtype == torch.bool:
            m = mask.reshape(N, heads, L, L) if mask.dim() == 3 else mask
            s = s.masked_fill(m, float("-inf"))
        else:
            s = s + mask
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(N, L, D)
    x = x + F.linear(o, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
    h = F.linear(_ln(x, sd, p + "ln_2."), sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h) if quick_gelu else F.gelu(h)
    return x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])


def clip_visual_embed(image, sd, p, cfg):
    """VisualTransformer up to ln_pre: conv1 (patch embedding, no bias), class token, positional embedding.  image (N,3,S,S), already
    CLIP-normalised -> (N, 1 + g*g, width)."""
    x = F.conv2d(image, sd[p + "visual.conv1.weight"], None, stride=cfg["patch"])
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd[p + "visual.class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + sd[p + "visual.positional_embedding"].to(x.dtype)
    return _ln(x, sd, p + "visual.ln_pre.")


def mask_clip_forward(image, attn_mask, num_mask_tokens, sd, p, cfg):
    """MaskCLIP._mask_clip_forward, hipie/open_vocab/clip.py:258-289: Q copies of the (ln_pre'd) class token are prepended as mask
    tokens, the transformer runs with the boolean attention mask, ln_post + proj of the mask tokens."""
    x = clip_visual_embed(image, sd, p, cfg)                                # (N, 1+g*g, D)
    cls = x[:, 0:1].expand(-1, num_mask_tokens, -1)
    x = torch.cat([cls, x], dim=1)                                          # [mask tokens | cls | patches]
    for i in range(cfg["layers"]):
        x = clip_resblock(x, sd, "%svisual.transformer.resblocks.%d." % (p, i), cfg["heads"], attn_mask, cfg.get("quick_gelu", True))
    x = _ln(x[:, :num_mask_tokens], sd, p + "visual.ln_post.")
    return x @ sd[p + "visual.proj"]


def clip_preprocess(image):
    """MaskCLIP's clip_preprocess (clip.py:96) on an image that already has the model's size: Resize / CenterCrop are no-ops, Normalize."""
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (image - mean) / std


def mask_token_attention_mask(mask, sd, p, cfg):
    """clip.py:299-334: mask (N, Q, S, S) logits -> bool (N*heads, T, T), T = Q + 1 + g*g.  Every token is blocked from attending TO a
    mask token; a mask token sees the class token and the patches in which its sigmoid mask reaches 0.5 somewhere."""
    N, Q = mask.shape[:2]
    patch_mask = F.max_pool2d(mask.sigmoid(), kernel_size=cfg["patch"], stride=cfg["patch"])
    blocked = (patch_mask < 0.5).reshape(N, Q, -1)
    n_img_cls = sd[p + "visual.positional_embedding"].shape[0]
    n_img = n_img_cls - 1
    T = Q + n_img_cls
    am = torch.zeros(T, T, dtype=torch.bool)
    am[:, :Q] = True
    am = am.unsqueeze(0).repeat_interleave(N, dim=0)
    am[:, :Q, -n_img:] = blocked
    heads = cfg["heads"]
    return am.unsqueeze(1).expand(-1, heads, -1, -1).reshape(N * heads, T, T)


def get_mask_embed(image, mask, sd, p, cfg):
    """MaskCLIP.get_mask_embed + encode_image_with_mask (clip.py:291-353): image (N,3,H,W) in 0..1, mask (N,Q,h,w) logits -> (N,Q,E)."""
    S = cfg["image_size"]
    image = F.interpolate(image, size=(S, S), mode="bilinear", align_corners=False)
    mask = F.interpolate(mask, size=(S, S), mode="bilinear", align_corners=False)
    am = mask_token_attention_mask(mask, sd, p, cfg)
    return mask_clip_forward(clip_preprocess(image), am, mask.shape[1], sd, p, cfg)


def clip_encode_text(tokens, sd, p, cfg):
    """CLIP.encode_text (what build_clip_text_embed calls, clip.py:29-71): token + positional embedding, causal transformer, ln_final,
    the feature at the end-of-text token (the highest id of the row) times text_projection.  tokens (n, context) int64."""
    x = sd[p + "token_embedding.weight"][tokens] + sd[p + "positional_embedding"]
    L = x.shape[1]
    causal = torch.full((L, L), float("-inf")).triu_(1)
    for i in range(cfg["text_layers"]):
        x = clip_resblock(x, sd, "%stransformer.resblocks.%d." % (p, i), cfg["text_heads"], causal, cfg.get("quick_gelu", True))
    x = _ln(x, sd, p + "ln_final.")
    return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ sd[p + "text_projection"]


def ensemble_logits_with_labels(logits, labels):
    """helper.py:77-106, ensemble_method "max": the class logit is the maximum over the class's synonym prompts."""
    lens = [len(l) for l in labels]
    out, st = [], 0
    for n in lens:
        out.append(logits[..., st:st + n].max(dim=-1).values)
        st += n
    return torch.stack(out, dim=-1)


def pred_logits(mask_embed, text_embed, labels, sd, p):
    """MaskCLIP.pred_logits (clip.py:355-365): cosine similarity * min(exp(logit_scale), 100), ensembled per class."""
    scale = torch.clamp(sd[p + "logit_scale"].exp(), max=100)
    lg = torch.einsum("bqc,nc->bqn", F.normalize(mask_embed, dim=-1), F.normalize(text_embed, dim=-1)) * scale
    return ensemble_logits_with_labels(lg, labels)


def prompt_labels_photo(labels):
    """helper.prompt_labels(labels, "photo") (helper.py:109-125): every synonym becomes "a photo of a <name>."."""
    return [["a photo of a %s." % l for l in syn] for syn in labels]


def get_clip_logits(mask_open_logits, pred_open_prob, overlapping, alpha, beta, agg_mode="MUL"):
    """HIPIE_IMG.get_clip_logits after the CLIP call (hipie_img.py:840-868): mask_open_logits (Q, C) CLIP logits of the masks,
    pred_open_prob (Q, C) the detector's class probabilities, overlapping (C) 1 where the test class shares a name with a training
    class ("base", weight alpha) else 0 ("novel", weight beta)."""
    if mask_open_logits.shape[-1] == 1:
        mp = mask_open_logits.sigmoid()
    else:
        mp = mask_open_logits.softmax(dim=-1)
    ov = overlapping.to(torch.long)
    if agg_mode == "ADD":
        base = (pred_open_prob * (1 - alpha) + mp * alpha + 1e-9).log() * ov
        novel = (pred_open_prob * (1 - beta) + mp * beta + 1e-9).log() * (1 - ov)
    else:
        base = (pred_open_prob ** (1 - alpha) * mp ** alpha).log() * ov
        novel = (pred_open_prob ** (1 - beta) * mp ** beta).log() * (1 - ov)
    return base + novel


def category_overlap(test_labels, train_labels):
    """hipie_img.py:818-830: test_labels / train_labels are lists of synonym lists; 1 where a test class shares a name with any
    training class."""
    train = {l for syn in train_labels for l in syn}
    return torch.tensor([int(not train.isdisjoint(set(syn))) for syn in test_labels], dtype=torch.long)
