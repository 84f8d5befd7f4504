"""Registration in detectron2's registries -- active only where detectron2 is importable.

``register()`` adds, under the reference's names, ``HIPIE_IMG`` to META_ARCH_REGISTRY and ``D2ViT`` to BACKBONE_REGISTRY,
both constructible from the reference's yacs CfgNode, and installs the MultiScaleDeformableAttention shim.  After that
``launch.py --eval-only`` / ``train_net.py`` build this implementation through ``build_model(cfg)`` unchanged, and
``DetectionCheckpointer`` loads reference checkpoints (identical state_dict keys).
"""
from .config import HipieConfig, Precision


def register(precision=None):
    from detectron2.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY   # raises ImportError without detectron2
    from . import msda_shim
    from .hipie_img import HIPIE_IMG as _Impl
    from .modeling.vit import D2ViT as _ViT

    msda_shim.install()
    prec = precision or Precision()

    def _md_cfg(cfg):
        from detectron2.config import get_cfg
        from detectron2.projects.deeplab import add_deeplab_config
        md = get_cfg()
        add_deeplab_config(md)
        try:
            from detectron2.projects.hipie.models.maskdino.config import add_maskdino_config
            add_maskdino_config(md)
            md.merge_from_file(cfg.MODEL.MASKDINO.CONFIG_PATH)
            return md
        except Exception:
            return None

    class HIPIE_IMG(_Impl):          # same class name as the reference => same registry key
        def __init__(self, cfg):
            super().__init__(HipieConfig.from_yacs(cfg, _md_cfg(cfg)), prec, device=cfg.MODEL.DEVICE)
            self.finalize()

    class D2ViT(_ViT):
        def __init__(self, cfg, input_shape):
            super().__init__(HipieConfig.from_yacs(cfg), prec)

    for reg, cls in ((META_ARCH_REGISTRY, HIPIE_IMG), (BACKBONE_REGISTRY, D2ViT)):
        if cls.__name__ in reg:
            reg._obj_map.pop(cls.__name__)
        reg.register(cls)
    return HIPIE_IMG
