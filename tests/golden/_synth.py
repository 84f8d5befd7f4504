"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

There are no HIPIE checkpoints, tokenizer vocabularies or datasets in the build
environment, so parity is checked on *seeded random* weights.  Storing
the weights in the fixtures would make them tens of MB; instead both sides (the generator
that drives the reference modules, and the tests that drive oracle/ and hipie_amd/) rebuild
the same state_dict from (key, shape, seed) with the rules below.  A fixture then only holds
the manifest {key: shape} (which doubles as a state_dict-compatibility check) and
the reference outputs.

The rules avoid the degenerate initialisations of the reference (zero rel_pos tables, zero
sampling_offsets weights, zero last bbox layer -- SURVEY 8d) so every kernel does real work.
"""
import re
import zlib

import numpy as np
import torch


def _gen(seed, name):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


_ALIASES = (
    # the reference registers some modules under several names (shared objects): the box/class heads are
    # also attached to the decoder (deformable_detr.py:272,282), MaskDINO shares one _bbox_embed across
    # layers and the decoder (maskdino_decoder.py:158-163) and decoder.norm is decoder_norm (:147-149).
    (re.compile(r"\.transformer\.decoder\.(bbox_embed|class_embed)\."), r".\1."),
    (re.compile(r"predictor\.(decoder\.)?bbox_embed\.\d+\."), "predictor._bbox_embed."),
    (re.compile(r"predictor\.decoder\.norm\."), "predictor.decoder_norm."),
)


def canonical_key(name):
    for pat, rep in _ALIASES:
        name = pat.sub(rep, name)
    return name


# ---- second weight distribution (round 4): the reference's OWN initialisation ----------------------------------------------------
# tests/golden/refinit_stats.json is a table {canonical key: statistics} measured by gen_golden.py on the reference's modules right after
# their constructors ran (their own _reset_parameters / _init_weights / default nn init; BertModel's init for the text encoder):
#   ["c", value]            constant tensor (LayerNorm 1 / 0, zero biases, layer-scale gammas ...)
#   ["n", mean, std]        bell-shaped (normal_ / trunc_normal_)
#   ["u", mean, std]        uniform (xavier_uniform_, kaiming_uniform_, default nn.Linear / nn.Conv2d)
#   ["v", [values]]         small or structured tensors stored verbatim (<= 64 elements: e.g. bbox_embed's bias [0, 0, -2, -2])
#   ["g"]                   MSDeformAttn's sampling_offsets.bias grid (ms_deform_attn.py:64-70), rebuilt from its formula
# Tensors the reference zero-initialises and that would make a kernel trivial (rel_pos_{h,w}, sampling_offsets.weight,
# attention_weights.*, the last bbox_embed layer) are drawn N(0, 0.02) as in the default distribution's spirit.  Both sides (generator and
# tests) rebuild the same values from (key, shape, seed, table); a fixture made with the table says so in meta["dist"].
_REFINIT = None
_DEGENERATE = ("rel_pos_h", "rel_pos_w", "sampling_offsets.weight", "attention_weights.weight", "attention_weights.bias")


def refinit_stats():
    global _REFINIT
    if _REFINIT is None:
        import json
        import os
        _REFINIT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "refinit_stats.json")))
    return _REFINIT


def msda_grid_bias(n_heads=8, n_levels=4, n_points=4):
    """MSDeformAttn._reset_parameters (ms_deform_attn.py:64-70): head h looks in direction 2 pi h / n_heads (normalised to the unit
    square), point i at i + 1 pixels."""
    import math
    thetas = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(n_heads, 1, 1, 2).repeat(1, n_levels, n_points, 1)
    for i in range(n_points):
        grid[:, :, i, :] *= i + 1
    return grid.reshape(-1)


def _refinit_tensor(name, shape, g, entry, dtype):
    leaf2 = ".".join(name.split(".")[-2:])
    if any(name.endswith(d) for d in _DEGENERATE) or (entry[0] == "c" and entry[1] == 0.0 and ".bbox_embed." in "." + name and name.split(".")[-2] == "2"):
        return (0.02 * torch.randn(shape, generator=g)).to(dtype)
    kind = entry[0]
    if kind == "c":
        return torch.full(shape, float(entry[1]), dtype=dtype)
    if kind == "v":
        return torch.tensor(entry[1], dtype=dtype).reshape(shape)
    if kind == "g":
        t = msda_grid_bias()
        assert t.numel() == int(np.prod(shape)), (name, shape)
        return t.reshape(shape).to(dtype)
    if kind == "n":
        return (float(entry[1]) + float(entry[2]) * torch.randn(shape, generator=g)).to(dtype)
    if kind == "u":
        half = float(entry[2]) * 3.0 ** 0.5
        return (float(entry[1]) + half * (2.0 * torch.rand(shape, generator=g) - 1.0)).to(dtype)
    raise KeyError(kind)


def synth_tensor(name, shape, seed=0, dtype=torch.float32, dist=None):
    name = canonical_key(name)
    shape = tuple(int(s) for s in shape)
    g = _gen(seed, name)
    leaf = name.split(".")[-1]
    if dist is not None and name in dist and not (len(shape) == 0 or "position_ids" in name or "token_type_ids" in name or "num_batches" in name):
        return _refinit_tensor(name, shape, g, dist[name], dtype)
    if len(shape) == 0:
        if leaf == "logit_scale":          # CLIP's learned temperature: exp(.) near the OpenAI value 1 / 0.07
            return (2.659 + 0.1 * torch.randn((), generator=g)).to(dtype)
        return torch.zeros((), dtype=torch.long) if "num_batches" in name else torch.randn((), generator=g).to(dtype) * 0.1
    if "running_var" in name:
        return (1.0 + 0.2 * torch.rand(shape, generator=g)).to(dtype)
    if "running_mean" in name:
        return (0.05 * torch.randn(shape, generator=g)).to(dtype)
    if "position_ids" in name:
        return torch.arange(shape[-1]).expand(shape).clone()
    if "token_type_ids" in name:
        return torch.zeros(shape, dtype=torch.long)
    if "sampling_offsets.bias" in name:
        return (1.5 * torch.randn(shape, generator=g)).to(dtype)
    if "sampling_offsets.weight" in name:
        return (0.5 / np.sqrt(shape[1]) * torch.randn(shape, generator=g)).to(dtype)
    if leaf in ("gamma_v", "gamma_l"):
        return (1.0 / 6 + 0.02 * torch.randn(shape, generator=g)).to(dtype)
    if leaf == "log_scale":
        return (0.1 * torch.randn(shape, generator=g)).to(dtype)
    if leaf == "bias0":
        return (-2.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
    if len(shape) == 1:
        is_norm_scale = leaf == "weight"
        if is_norm_scale:
            return (1.0 + 0.05 * torch.randn(shape, generator=g)).to(dtype)
        return (0.02 * torch.randn(shape, generator=g)).to(dtype)
    fan_in = int(np.prod(shape[1:]))
    if "bg_query_refs" in name:  # (num_bg, 4) reference boxes in (0,1)
        return (0.1 + 0.8 * torch.rand(shape, generator=g)).to(dtype)
    if "word_embeddings" in name or "position_embeddings" in name or "token_type_embeddings" in name:
        return (0.05 * torch.randn(shape, generator=g)).to(dtype)
    std = 1.0 / np.sqrt(fan_in)
    return (std * torch.randn(shape, generator=g)).to(dtype)


def synth_state_dict(manifest, seed=0, dist=None, prefix=""):
    """manifest: {key: shape} (e.g. {k: tuple(v.shape) for k, v in module.state_dict().items()}).  dist: the reference-init statistics
    table (its keys carry `prefix`: "detr." / "text_encoder.body.")."""
    d = None if dist is None else {k[len(prefix):]: v for k, v in dist.items() if k.startswith(prefix)}
    return {k: synth_tensor(k, shp, seed, dist=d) for k, shp in manifest.items()}


def manifest_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_synth(module, seed=0, dist=None, prefix=""):
    """Fill ``module`` in place with synthetic weights; returns the manifest."""
    man = manifest_of(module)
    sd = synth_state_dict(man, seed, dist, prefix)
    cur = module.state_dict()
    for k in sd:
        sd[k] = sd[k].to(cur[k].dtype)
    module.load_state_dict(sd, strict=True)
    return man


def synth_images(sizes, seed=0):
    """uint8-valued RGB images U{0..255} as float32 (3,h,w) tensors."""
    out = []
    for i, (h, w) in enumerate(sizes):
        g = _gen(seed, "image%d" % i)
        out.append(torch.randint(0, 256, (3, h, w), generator=g).float())
    return out


def synth_token_ids(batch, n_classes, max_len, seed=0, pad_to=None):
    """BERT-style ids: [CLS]=101, class tokens in [1996,29000), '.'=1012 between classes, [SEP]=102, pad 0.

    Returns ids (B,L) int64, attention mask (B,L) int64 and positive_map_label_to_token
    {1-based class: [token indices]} for the first sample (hipie_img.py:324).
    """
    rows, pmaps = [], []
    for b in range(batch):
        g = _gen(seed, "tokens%d" % b)
        ids, pmap = [101], {}
        for c in range(n_classes):
            k = int(torch.randint(1, 4, (1,), generator=g))
            toks = torch.randint(1996, 29000, (k,), generator=g).tolist()
            if len(ids) + k + 2 > max_len:
                break
            pmap[c + 1] = list(range(len(ids), len(ids) + k))
            ids += toks + [1012]
        ids.append(102)
        rows.append(ids)
        pmaps.append(pmap)
    L = pad_to or max(len(r) for r in rows)
    out = torch.zeros(batch, L, dtype=torch.long)
    mask = torch.zeros(batch, L, dtype=torch.long)
    for b, r in enumerate(rows):
        out[b, :len(r)] = torch.tensor(r)
        mask[b, :len(r)] = 1
    return out, mask, pmaps[0]


MAX_ELEMS = 1 << 16


def sub_step(numel):
    """stride used to thin out big tensors in the fixtures (odd, so it does not alias with
    power-of-two row lengths)."""
    step = max(1, numel // MAX_ELEMS)
    return step | 1


def subsample(t, step):
    return t.reshape(-1)[::step]


def synth_full_state_dict(manifest, seed_detr=71, seed_text=72, dist=None):
    """State dict of the whole HIPIE_IMG model ("detr.*" + "text_encoder.body.*" keys) with the
    same values gen_golden.py loaded into the reference's DDETRSegmUniDN / BertEncoder (which it built
    separately, so the seeds and key prefixes differ).  dist = "refinit" (or the table itself): the reference's own initialisation
    distribution (refinit_stats.json) instead of the default rules."""
    if dist == "refinit":
        dist = refinit_stats()
    dd = None if dist is None else {canonical_key(k[len("detr."):]): v for k, v in dist.items() if k.startswith("detr.")}
    dt = None if dist is None else {k[len("text_encoder.body."):]: v for k, v in dist.items() if k.startswith("text_encoder.body.")}
    out = {}
    for k, shp in manifest.items():
        if k.startswith("detr."):
            out[k] = synth_tensor(k[len("detr."):], shp, seed_detr, dist=dd)
        elif k.startswith("text_encoder.body."):
            out[k] = synth_tensor(k[len("text_encoder.body."):], shp, seed_text, dist=dt)
        else:
            raise KeyError(k)
    return out


def synth_a22(sizes, n_bg, n_fg, n_md, L, seed=0, stride=4):
    """a synthetic a22 dictionary (DDETRSegmUniDN.coco_inference's outputs) for the post-processing row: clustered boxes
    (so NMS has duplicates to remove), token logits with a few confident queries, smooth low-frequency mask logits (so
    panoptic segments have area).  sizes: [(h, w)] per image; padded canvas = max over images."""
    B = len(sizes)
    Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
    hm, wm = Hm // stride, Wm // stride
    g = _gen(seed, "a22")
    Q = n_bg + n_fg

    def rnd(*shape):
        return torch.randn(*shape, generator=g)

    def smooth(n):
        """one blob per query (logit +5 inside, -5 outside, soft edge) at a random place, plus low-frequency noise: mostly
        disjoint segments with some overlaps, so the panoptic merge both accepts and rejects segments."""
        ys = torch.linspace(0, 1, hm).view(1, 1, hm, 1)
        xs = torch.linspace(0, 1, wm).view(1, 1, 1, wm)
        cy, cx = torch.rand(B, n, 1, 1, generator=g), torch.rand(B, n, 1, 1, generator=g)
        r = 0.08 + 0.17 * torch.rand(B, n, 1, 1, generator=g)
        d = torch.sqrt((ys - cy) ** 2 + (xs - cx) ** 2)
        blob = 5.0 * torch.tanh((r - d) * 30.0)
        lo = rnd(B * n, 1, 5, 5)
        noise = torch.nn.functional.interpolate(lo, size=(hm, wm), mode="bicubic", align_corners=False).view(B, n, hm, wm)
        return (blob + 0.7 * noise + 0.2 * rnd(B, n, hm, wm)).unsqueeze(2)
    n_clu = max(3, n_fg // 4)
    centers = torch.rand(B, n_clu, 4, generator=g) * torch.tensor([0.8, 0.8, 0.4, 0.4]) + torch.tensor([0.1, 0.1, 0.1, 0.1])
    which = torch.randint(0, n_clu, (B, Q), generator=g)
    boxes = torch.gather(centers, 1, which[..., None].expand(-1, -1, 4)) + 0.02 * rnd(B, Q, 4)
    boxes[..., 2:] = boxes[..., 2:].abs().clamp_min(0.02)
    out = {
        "pred_logits": rnd(B, Q, L) * 2.5 - 1.0,
        "pred_boxes": boxes,
        "pred_boxious": rnd(B, Q, 1) * 2.0 + 0.5,
        "pred_masks": smooth(Q),
        "pred_logits_maskdino": rnd(B, n_md, L) * 2.5 - 0.5,
        "pred_boxes_maskdino": torch.rand(B, n_md, 4, generator=g),
        "pred_masks_maskdino": smooth(n_md)[:, :, 0],
    }
    return out


PROMPT_CATEGORIES = [{"name": "person"}, {"name": "traffic light"}, {"name": "hot dog", "isthing": 1},
                     {"name": "potted_plant"}, {"name": "tv (monitor)"}, {"name": "wine glass"},
                     {"name": "hair drier"}, {"name": "sky-other", "isthing": 0}, {"name": "teddy bear"},
                     {"name": "zyxwv"}]
PROMPT_VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", ".", "-", "person", "traffic", "light", "hot", "dog",
                "potted", "plant", "tv", "wine", "glass", "hair", "dr", "##ier", "sky", "other", "teddy", "bear"]


def prompt_tokenizer(tmpdir):
    """a BertTokenizerFast over a tiny deterministic vocab (the real bert-base-uncased vocab is not available offline)."""
    import os
    from transformers import BertTokenizerFast
    path = os.path.join(str(tmpdir), "vocab.txt")
    with open(path, "w") as f:
        f.write("\n".join(PROMPT_VOCAB) + "\n")
    return BertTokenizerFast(vocab=path, do_lower_case=True)        # transformers 5: `vocab` (vocab_file= is silently ignored)


def clip_tokenize(texts, context, vocab):
    """a deterministic stand-in for open_clip.tokenize (its BPE vocabulary is not available offline): one id per whitespace /
    punctuation separated word, [start] ... [end] with the END token the highest id of the vocabulary (CLIP.encode_text reads
    the feature at argmax), zero padded to `context`."""
    import re as _re
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context, dtype=torch.long)
    sot, eot = vocab - 2, vocab - 1
    for i, t in enumerate(texts):
        words = _re.findall(r"[a-z0-9]+|[^\sa-z0-9]", t.lower())
        ids = [sot] + [1 + zlib.crc32(w.encode()) % (vocab - 3) for w in words][:context - 2] + [eot]
        out[i, :len(ids)] = torch.tensor(ids)
    return out


def msda_bwd_inputs(tag, D=None):
    """seeded inputs of the backward cases (shared with tests/test_gpu_kernels.py): value, shapes, loc, attn, grad_output -- f64."""
    if tag == "recipe":                # ops/test.py:21-27, 69-85: N, M, Lq, L, P = 1, 2, 2, 2, 2 on maps (6, 4), (3, 2); D = channels
        B, M, Lq, L, P, shp, seed = 1, 2, 2, 2, 2, [(6, 4), (3, 2)], 300 + D
        scale, lo, span = 0.01, 0.0, 1.0
    else:                              # hot-path geometry (M = 8, D = 32, L = 4, P = 4) incl. out-of-range locations
        B, M, Lq, L, P, shp, seed, D = 2, 8, 50, 4, 4, [(12, 20), (6, 10), (3, 5), (2, 3)], 41, 32
        scale, lo, span = 1.0, -0.1, 1.2
    shapes = torch.as_tensor(shp, dtype=torch.long)
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(seed)
    f64 = torch.float64
    value = torch.rand(B, S, M, D, generator=g, dtype=f64) * scale
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g, dtype=f64) * span + lo
    attn = torch.rand(B, Lq, M, L, P, generator=g, dtype=f64) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    gout = torch.randn(B, Lq, M * D, generator=g, dtype=f64)
    return value, shapes, loc, attn, gout


def hash_uniform(call, shape, device="cpu"):
    """the `call`-th random tensor of a step as a counter-based hash: uniform in [0, 1) with 24 bits, bit-identical on every device (int64
    arithmetic only).  gen_train_step_golden.py feeds the REFERENCE these values in place of torch.rand / rand_like / randint_like (its
    algorithms do not care which uniform numbers they get), and the product's training step draws the same ones in the same order -- so a
    whole training step can be compared without storing megabytes of point coordinates."""
    n = 1
    for s in shape:
        n *= int(s)
    x = torch.arange(n, dtype=torch.int64, device=device) * 2654435761 + (int(call) + 1) * 40503 + 97
    x = x & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 2246822519) & 0xFFFFFFFF
    x = x ^ (x >> 13)
    x = (x * 3266489917) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return ((x & 0xFFFFFF).to(torch.float32) / 16777216.0).reshape(tuple(int(s) for s in shape))


class HashDraws:
    """the step's source of random numbers: rand(shape) / randint(low, high, shape) in call order (see hash_uniform)"""

    def __init__(self):
        self.calls = 0

    def rand(self, shape, device="cpu"):
        r = hash_uniform(self.calls, shape, device)
        self.calls += 1
        return r

    def randint(self, low, high, shape, device="cpu"):
        return (self.rand(shape, device) * (high - low)).floor().long() + low
