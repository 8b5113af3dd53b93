"""dualfusion: host-side mirror of the 3D-Dual-Fusion hot-path interface on the MI355X kernels
(libdf3d_hip.so, C ABI in include/df3d_hip.h).  See DESIGN.md / INTEGRATION.md.

Sub-modules import lazily; every op loads the HIP library on first use and raises Df3dError if it
is missing (there is no CPU fallback)."""
from ._lib import Df3dError, LIB_PATH, load as require  # noqa: F401

__all__ = ["require", "Df3dError", "LIB_PATH", "spconv", "ops", "voxel", "msda", "actr", "fusion", "backbones",
           "pipeline", "registry", "synth", "necks", "executor", "fusion_tf", "iou3d_nms", "heads", "transfusion_head"]


def __getattr__(name):
    if name in __all__:
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
