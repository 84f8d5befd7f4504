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


def synth_tensor(name, shape, seed=0, dtype=torch.float32):
    name = canonical_key(name)
    shape = tuple(int(s) for s in shape)
    g = _gen(seed, name)
    leaf = name.split(".")[-1]
    if len(shape) == 0:
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


def synth_state_dict(manifest, seed=0):
    """manifest: {key: shape} (e.g. {k: tuple(v.shape) for k, v in module.state_dict().items()})."""
    return {k: synth_tensor(k, shp, seed) for k, shp in manifest.items()}


def manifest_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def load_synth(module, seed=0):
    """Fill ``module`` in place with synthetic weights; returns the manifest."""
    man = manifest_of(module)
    sd = synth_state_dict(man, seed)
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


def synth_full_state_dict(manifest, seed_detr=71, seed_text=72):
    """State dict of the whole HIPIE_IMG model ("detr.*" + "text_encoder.body.*" keys) with the
    same values gen_golden.py loaded into the reference's DDETRSegmUniDN / BertEncoder (which it built
    separately, so the seeds and key prefixes differ)."""
    out = {}
    for k, shp in manifest.items():
        if k.startswith("detr."):
            out[k] = synth_tensor(k[len("detr."):], shp, seed_detr)
        elif k.startswith("text_encoder.body."):
            out[k] = synth_tensor(k[len("text_encoder.body."):], shp, seed_text)
        else:
            raise KeyError(k)
    return out
