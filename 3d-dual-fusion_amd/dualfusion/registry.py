"""Self-contained registry shim: the reference builds every hot-path module from
config dicts through class registries -- Det3D `build_from_cfg` popping `type`
(CP/det3d/utils/registry.py:46-76), mmcv `Registry.build`, pcdet `__all__[cfg.NAME]`.  None of
those packages exists on the GPU box, so the same mechanism is provided here, supporting both
decorator styles (`@R.register_module` bare as in Det3D and `@R.register_module()` as in mmcv),
plus late registration into the real registries when they are importable."""
import inspect


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    def __repr__(self):
        return "%s(name=%s, items=%s)" % (self.__class__.__name__, self._name, list(self._module_dict))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError("module must be a class, but got %s" % type(cls))
        name = name or cls.__name__
        if name in self._module_dict and not force:
            raise KeyError("%s is already registered in %s" % (name, self._name))
        self._module_dict[name] = cls
        return cls

    def register_module(self, cls=None, name=None, force=False):
        if cls is not None and inspect.isclass(cls):      # @R.register_module
            return self._register(cls, name, force)

        def deco(c):                                       # @R.register_module()
            return self._register(c, name, force)
        return deco

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    """CP/det3d/utils/registry.py:46-76."""
    assert isinstance(cfg, dict) and "type" in cfg
    assert isinstance(default_args, dict) or default_args is None
    args = dict(cfg)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError("%s is not in the %s registry" % (obj_type, registry.name))
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError("type must be a str or valid type, but got %s" % type(obj_type))
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    return obj_cls(**args)


# Det3D names (CP/det3d/models/registry.py) / mmdet3d names / mmcv conv-layer names
READERS = Registry("reader")
BACKBONES = Registry("backbone")
FUSION = Registry("fusion")
NECKS = Registry("neck")
HEADS = Registry("head")
MIDDLE_ENCODERS = Registry("middle_encoder")
VOXEL_ENCODERS = Registry("voxel_encoder")
FUSION_LAYERS = Registry("fusion_layer")
CONV_LAYERS = Registry("conv layer")
MM_BACKBONES = Registry("mmdet backbone")      # 2-D BEV backbone / neck of the TransFusion tree (SECOND, SECONDFPN)
MM_NECKS = Registry("mmdet neck")
MM_HEADS = Registry("mmdet head")              # TransFusionHead (mmdet3d HEADS is mmdet's HEADS registry)
MM_DETECTORS = Registry("mmdet detector")      # TransFusionDetector (mmdet3d registers detectors into mmdet's DETECTORS)
BACKBONES_3D = Registry("pcdet backbone_3d")   # pcdet uses a plain dict `__all__`; same lookup by NAME


def late_register():
    """Register our classes into the real frameworks' registries when those are importable, so the
    reference configs resolve `type=` / `NAME:` strings to the MI355X modules unchanged."""
    import importlib
    for m in ("spconv", "voxel", "backbones", "fusion", "fusion_tf", "necks", "heads", "transfusion_head", "transfusion"):
        importlib.import_module("." + m, __package__)          # every module that registers classes (some import lazily)
    done = []
    try:
        from mmcv.cnn import CONV_LAYERS as MMCV_CONV
        for k, v in CONV_LAYERS.module_dict.items():
            MMCV_CONV.register_module(name=k, module=v, force=True)
        done.append("mmcv.cnn.CONV_LAYERS")
    except Exception:
        pass
    try:
        try:
            from mmdet3d.models.builder import FUSION_LAYERS as F3, MIDDLE_ENCODERS as M3, VOXEL_ENCODERS as V3
        except ImportError:                                     # TF/mmdet3d/models/registry.py holds them in this tree
            from mmdet3d.models.registry import FUSION_LAYERS as F3, MIDDLE_ENCODERS as M3, VOXEL_ENCODERS as V3
        for src, dst in ((FUSION_LAYERS, F3), (MIDDLE_ENCODERS, M3), (VOXEL_ENCODERS, V3)):
            for k, v in src.module_dict.items():
                dst.register_module(name=k, module=v, force=True)
        done.append("mmdet3d")
    except Exception:
        pass
    try:
        import mmdet.models as _mm
        from mmdet.models import BACKBONES as MB, HEADS as MH, NECKS as MN
        MD = getattr(_mm, "DETECTORS", None)                    # (absent from some mmdet builds: the detector is optional)
        for src, dst in ((MM_BACKBONES, MB), (MM_NECKS, MN), (MM_HEADS, MH)) + (((MM_DETECTORS, MD),) if MD is not None else ()):
            for k, v in src.module_dict.items():
                dst.register_module(name=k, module=v, force=True)
        done.append("mmdet")
    except Exception:
        pass
    try:
        from det3d.models import registry as d3
        for src, dst in ((READERS, d3.READERS), (BACKBONES, d3.BACKBONES), (FUSION, d3.FUSION), (NECKS, d3.NECKS),
                         (HEADS, d3.HEADS)):
            for k, v in src.module_dict.items():
                dst._module_dict[k] = v
        done.append("det3d")
    except Exception:
        pass
    try:
        import pcdet.models.backbones_3d as p3
        for k, v in BACKBONES_3D.module_dict.items():
            p3.__all__[k] = v
        done.append("pcdet")
    except Exception:
        pass
    return done
