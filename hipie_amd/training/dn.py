"""Contrastive de-noising queries of the detection branch (DINO-style CDN; hipie/models/ddetrs_dn.py:1176-1388 + the matching
index helpers :1352-1368 and deformable_detr.py:785-800).

Every target of the batch becomes, in each of G groups, one POSITIVE query (its box jittered by less than half its size) and one
NEGATIVE query (jittered by between half and the whole size); the queries of all images are padded to P = the largest target count, so
the de-noising part of the decoder input is (B, 2 G P, .) in the order [group 0 positives | group 0 negatives | group 1 positives | ..].
The label side of a query is the image's own text embedding (the `dynamic_label_enc` mode the HIPIE configs run), or an embedding of
its -- optionally randomised -- class id.  The attention mask keeps the groups from seeing each other and the matching queries from
seeing any de-noising query.

The reference's extra "point" queries (`dp_number`, ddetrs_dn.py:1071-1107) are not restated: their only call site (:1304-1306) passes
`self` twice, so that every argument lands in the wrong parameter; the shipped configs leave dp_number at 0."""
import torch


def _inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def cdn_queries(targets, dn_number, box_noise_scale, num_queries, label_embed, noise=None, label_noise_ratio=0.0, num_classes=None):
    """targets: per image {"labels" (n_i,), "boxes" (n_i, 4) cxcywh in [0, 1]}.
    label_embed: (B, C) tensor -- one embedding per image (dynamic label encoding) -- or a callable ids (n,) -> (n, C).
    noise: {"sign": (2 G n, 4) of +-1, "part": (2 G n, 4) in [0, 1)} and, with class-id embeddings and label_noise_ratio > 0,
           {"p": (2 G n,), "new_label": ids, at least as many as entries with p < ratio / 2}; None: drawn here, in the reference's order
           (p, new_label, sign, part).
    -> (query_label (B, 2 G P, C), query_box (B, 2 G P, 4) in logit space, attn_mask (2 G P + num_queries,) x 2 bool with True = blocked,
        meta {"single_padding": 2 P, "dn_num": G, "dp_num": 0}), or four Nones when there is nothing to de-noise."""
    if dn_number <= 0:
        return None, None, None, None
    counts = [int(t["labels"].numel()) for t in targets]
    P = max(counts) if counts else 0
    if P == 0:
        return None, None, None, None
    B, n = len(targets), sum(counts)
    G = max(1, (2 * dn_number) // (2 * P))
    labels = torch.cat([t["labels"] for t in targets])
    boxes = torch.cat([t["boxes"] for t in targets])
    dev = boxes.device
    image_of = torch.cat([torch.full((c,), i, dtype=torch.long, device=dev) for i, c in enumerate(counts)])
    slot_in_image = torch.cat([torch.arange(c, device=dev) for c in counts])
    R = 2 * G                                               # copies of the target list: even = positive, odd = negative
    rep_labels, rep_boxes, rep_image = labels.repeat(R), boxes.repeat(R, 1), image_of.repeat(R)
    noise = dict(noise or {})
    dynamic = torch.is_tensor(label_embed)
    if not dynamic and label_noise_ratio > 0:
        p = noise["p"] if "p" in noise else torch.rand(rep_labels.shape, device=dev)
        flip = torch.nonzero(p < label_noise_ratio * 0.5).flatten()
        new = noise["new_label"] if "new_label" in noise else torch.randint(0, num_classes, flip.shape, device=dev)
        rep_labels = rep_labels.clone()
        rep_labels[flip] = new[:flip.numel()].to(rep_labels.dtype)
    if box_noise_scale > 0:
        sign = noise["sign"] if "sign" in noise else torch.randint(0, 2, rep_boxes.shape, device=dev).to(rep_boxes.dtype) * 2.0 - 1.0
        part = (noise["part"] if "part" in noise else torch.rand(rep_boxes.shape, device=dev)).clone()
        negative = (torch.arange(R, device=dev) % 2 == 1).repeat_interleave(n)
        part[negative] += 1.0
        half = rep_boxes[:, 2:] / 2
        corners = torch.cat((rep_boxes[:, :2] - half, rep_boxes[:, :2] + half), 1)
        corners = (corners + part * sign * torch.cat((half, half), 1) * box_noise_scale).clamp(0.0, 1.0)
        rep_boxes = torch.cat(((corners[:, :2] + corners[:, 2:]) / 2, corners[:, 2:] - corners[:, :2]), 1)
    emb = label_embed[rep_image] if dynamic else label_embed(rep_labels.long())
    pad = R * P
    q_label = emb.new_zeros(B, pad, emb.shape[-1])
    q_box = rep_boxes.new_zeros(B, pad, 4)
    slot = torch.arange(R, device=dev).repeat_interleave(n) * P + slot_in_image.repeat(R)
    q_label[rep_image, slot] = emb
    q_box[rep_image, slot] = _inverse_sigmoid(rep_boxes)
    size = pad + num_queries
    group = torch.arange(pad, device=dev) // (2 * P)
    mask = torch.zeros(size, size, dtype=torch.bool, device=dev)
    mask[:pad, :pad] = group[:, None] != group[None, :]     # a group sees itself only ...
    mask[pad:, :pad] = True                                 # ... and the matching queries see no de-noising query
    return q_label, q_box, mask, {"single_padding": 2 * P, "dn_num": G, "dp_num": 0}


def dn_split_outputs(per_layer, meta):
    """per_layer (layers, B, 2 G P + Q, .) -> (de-noising part, matching part) along the query axis (dn_post_process, ddetrs_dn.py:1370-1388)."""
    if not meta or meta["single_padding"] <= 0:
        return None, per_layer
    pad = meta["single_padding"] * (meta["dn_num"] + meta.get("dp_num", 0))
    return per_layer[:, :, :pad], per_layer[:, :, pad:]


def dn_match_indices(targets, meta, device=None):
    """the fixed "matching" of the de-noising part: target t of an image <-> the POSITIVE query of t in every group
    (ddetrs_dn.py:1352-1368; deformable_detr.py:785-800) -> [(query idx, target idx)] per image."""
    G, stride = meta["dn_num"] + meta.get("dp_num", 0), meta["single_padding"]
    out = []
    for t in targets:
        k = int(t["labels"].numel())
        dev = device or t["labels"].device
        tgt = torch.arange(k, device=dev).repeat(G)
        qry = (torch.arange(G, device=dev) * stride).repeat_interleave(k) + tgt
        out.append((qry, tgt))
    return out


def maskdino_dn_queries(targets, dn_num, noise_scale, num_queries, label_embed, tgt=None, refpoint=None, noise=None, num_classes=None):
    """MaskDINO's own de-noising queries (DN-DETR style; maskdino/transformer_decoder/maskdino_decoder.py:202-327, training branch):
    G = dn_num // (largest target count) groups of ONE noised copy per target -- centre moved by up to half the size, size changed by up to
    the size, both times noise_scale -- padded to P slots per image in the order [group 0 | group 1 | ..]; the label side is the image's text
    embedding (label_embed a (B, C) tensor) or an embedding of the class id, flipped to a random class with probability noise_scale / 2
    (label_embed a callable).  tgt (Q, C) / refpoint (Q, 4): the matching queries, appended behind the de-noising part when given.
    noise: {"box": (G n, 4) in [0, 1)} and, with class ids, {"p": (G n,), "new_label": ids}; None: drawn here in the reference's order (p,
    new_label, box).
    -> (query_label (B, G P [+ Q], C), query_box (B, G P [+ Q], 4) logits, attn_mask (G P + num_queries)^2 with True = blocked,
        {"pad_size": G P, "scalar": G, "single_pad": P}), or four Nones when there are no targets / dn_num is too small for one group."""
    counts = [int(t["labels"].numel()) for t in targets]
    P = max(counts) if counts else 0
    G = dn_num // P if P > 0 else 0
    if G == 0:
        return None, None, None, None
    B, n = len(targets), sum(counts)
    labels = torch.cat([t["labels"] for t in targets])
    boxes = torch.cat([t["boxes"] for t in targets])
    dev = boxes.device
    image_of = torch.cat([torch.full((c,), i, dtype=torch.long, device=dev) for i, c in enumerate(counts)])
    slot_in_image = torch.cat([torch.arange(c, device=dev) for c in counts])
    rep_labels, rep_boxes, rep_image = labels.repeat(G), boxes.repeat(G, 1), image_of.repeat(G)
    noise = dict(noise or {})
    dynamic = torch.is_tensor(label_embed)
    if noise_scale > 0:
        # the reference draws p and the replacement labels whatever the label mode (and uses them only with class-id embeddings)
        p = noise["p"] if "p" in noise else torch.rand(rep_labels.shape, device=dev)
        flip = torch.nonzero(p < noise_scale * 0.5).flatten()
        if "new_label" in noise:
            new = noise["new_label"]
        else:
            new = torch.randint(0, num_classes, flip.shape, device=dev)
        rep_labels = rep_labels.clone()
        rep_labels[flip] = new[:flip.numel()].to(rep_labels.dtype)
        u = noise["box"] if "box" in noise else torch.rand(rep_boxes.shape, device=dev)
        reach = torch.cat((rep_boxes[:, 2:] / 2, rep_boxes[:, 2:]), 1)
        rep_boxes = (rep_boxes + (u * 2 - 1.0) * reach * noise_scale).clamp(0.0, 1.0)
    emb = label_embed[rep_image] if dynamic else label_embed(rep_labels.long())
    pad = G * P
    q_label = emb.new_zeros(B, pad, emb.shape[-1])
    q_box = rep_boxes.new_zeros(B, pad, 4)
    slot = torch.arange(G, device=dev).repeat_interleave(n) * P + slot_in_image.repeat(G)
    q_label[rep_image, slot] = emb
    q_box[rep_image, slot] = _inverse_sigmoid(rep_boxes)
    if refpoint is not None:
        q_label = torch.cat((q_label, tgt[None].expand(B, -1, -1)), 1)
        q_box = torch.cat((q_box, refpoint[None].expand(B, -1, -1)), 1)
    size = pad + num_queries
    group = torch.arange(pad, device=dev) // P
    mask = torch.zeros(size, size, dtype=torch.bool, device=dev)
    mask[:pad, :pad] = group[:, None] != group[None, :]
    mask[pad:, :pad] = True
    return q_label, q_box, mask, {"pad_size": pad, "scalar": G, "single_pad": P}
