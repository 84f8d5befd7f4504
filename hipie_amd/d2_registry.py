"""Registration in detectron2's registries -- active only where detectron2 is importable.

``register()`` adds, under the reference's names, ``HIPIE_IMG`` to META_ARCH_REGISTRY, ``D2ViT`` to BACKBONE_REGISTRY,
``MaskDINOHead`` / ``MaskDINOEncoder`` to SEM_SEG_HEADS_REGISTRY and ``MaskDINODecoder`` to the reference's
TRANSFORMER_DECODER_REGISTRY (hipie/models/maskdino/transformer_decoder/maskdino_decoder.py:22; created here when the
reference project is not installed), all constructible from the reference's yacs CfgNode, and installs the
MultiScaleDeformableAttention shim.  After that ``launch.py --eval-only`` / ``train_net.py`` build this implementation
through ``build_model(cfg)`` unchanged, and ``DetectionCheckpointer`` loads reference checkpoints (identical state_dict keys).

Order of events in train_net.py:269-271 is build -> load checkpoint -> forward: nothing weight-derived is computed at
construction (HIPIE_IMG finalises lazily on the first forward and after every load_state_dict).
Unsupported switch positions of the yaml (PARALLEL_DET, DECOUPLE_TGT off, USE_DINO off, ...) raise NotImplementedError at construction
instead of silently evaluating something else; MODEL.CLIP.ENABLED is supported (hipie_amd/open_vocab.py, round 3).
Exercised so far only against a stand-in detectron2 (tests/test_host_logic.py::test_d2_registry_*): the image has no detectron2.
"""
from .config import HipieConfig, Precision


def load_maskdino_cfg(cfg):
    """the MaskDINO sub-model's own CfgNode, as hipie/models/maskdino/build.py:8-19 builds it."""
    from detectron2.config import get_cfg
    from detectron2.projects.deeplab import add_deeplab_config
    from detectron2.projects.hipie.models.maskdino.config import add_maskdino_config
    md = get_cfg()
    add_deeplab_config(md)
    add_maskdino_config(md)
    md.merge_from_file(cfg.MODEL.MASKDINO.CONFIG_PATH)
    return md


def register(precision=None, md_cfg_loader=load_maskdino_cfg):
    from detectron2.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY   # ImportError without detectron2
    from detectron2.utils.registry import Registry
    from . import msda_shim
    from .hipie_img import HIPIE_IMG as _Impl
    from .modeling import maskdino as _md
    from .modeling.vit import D2ViT as _ViT

    try:
        from detectron2.projects.hipie.models.maskdino.transformer_decoder.maskdino_decoder import TRANSFORMER_DECODER_REGISTRY
    except ImportError:
        TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")
    msda_shim.install()
    prec = precision or Precision.split3()       # the in-tolerance policy (1e-3 at the shipped depths); Precision.fast() is opt-in

    def hcfg(cfg):
        return HipieConfig.from_yacs(cfg, md_cfg_loader(cfg))       # errors of the MaskDINO yaml propagate

    class HIPIE_IMG(_Impl):          # same class name as the reference => same registry key
        def __init__(self, cfg):
            super().__init__(hcfg(cfg), prec, device=cfg.MODEL.DEVICE)

    class D2ViT(_ViT):
        def __init__(self, cfg, input_shape=None):
            super().__init__(HipieConfig.from_yacs(cfg), prec)

    def _channels(input_shape, c):
        if input_shape is None:
            return c.backbone_channels
        return [input_shape[k].channels for k in ("res3", "res4", "res5")]

    class MaskDINOEncoder(_md.MaskDINOEncoder):
        def __init__(self, cfg, input_shape=None):
            c = hcfg(cfg)
            super().__init__(c, _channels(input_shape, c), prec)

    class MaskDINODecoder(_md.MaskDINODecoder):
        def __init__(self, cfg, in_channels=None, mask_classification=True):
            super().__init__(hcfg(cfg), prec)

    class MaskDINOHead(_md.MaskDINOHead):
        def __init__(self, cfg, input_shape=None):
            c = hcfg(cfg)
            super().__init__(c, _channels(input_shape, c), prec)

    for reg, cls in ((META_ARCH_REGISTRY, HIPIE_IMG), (BACKBONE_REGISTRY, D2ViT), (SEM_SEG_HEADS_REGISTRY, MaskDINOEncoder),
                     (SEM_SEG_HEADS_REGISTRY, MaskDINOHead), (TRANSFORMER_DECODER_REGISTRY, MaskDINODecoder)):
        if cls.__name__ in reg:
            reg._obj_map.pop(cls.__name__)
        reg.register(cls)
    return {"HIPIE_IMG": HIPIE_IMG, "D2ViT": D2ViT, "MaskDINOEncoder": MaskDINOEncoder, "MaskDINODecoder": MaskDINODecoder,
            "MaskDINOHead": MaskDINOHead, "TRANSFORMER_DECODER_REGISTRY": TRANSFORMER_DECODER_REGISTRY}
