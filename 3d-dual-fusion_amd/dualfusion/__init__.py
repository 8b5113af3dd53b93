"""dualfusion: host-side mirror of the 3D-Dual-Fusion hot-path interface on the MI355X kernels
(libdf3d_hip.so, C ABI in include/df3d_hip.h).  See DESIGN.md / INTEGRATION.md.

Sub-modules import lazily; every op loads the HIP library on first use and raises Df3dError if it
is missing (there is no CPU fallback)."""
import os as _os
import sys as _sys

# HIP maps a process's streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4) and serialises the streams that share
# one.  The frame pipeline of this package runs ~8 streams per detector (frame-head worker, geometry, voxeliser, adapter side
# streams, the caller's); with 4 queues two frames in flight did not overlap at all.  Effective only when
# this import happens before the HIP runtime initialises (import dualfusion before the first CUDA call, or export it yourself).
# kernel arguments in device memory (not host-coherent memory read over PCIe at every launch): -125 us on the ~130 dependent
# launches of a detector frame; same condition -- before the HIP runtime initialises


def configure_runtime():
    """The two process-wide HIP runtime settings this package wants (INTEGRATION.md "runtime settings"), as an explicit call
    for hosts that prefer not to rely on import order.  Values the host application exported itself are left alone.  Returns
    False -- and the settings have no effect -- when the HIP runtime of this process is already initialised."""
    missing = [k for k in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG") if k not in _os.environ]
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    _torch = _sys.modules.get("torch")
    late = bool(missing and _torch is not None and _torch.cuda.is_initialized())
    if late:
        import warnings
        warnings.warn("dualfusion: the HIP runtime was initialised before `import dualfusion`; GPU_MAX_HW_QUEUES / "
                      "HIP_FORCE_DEV_KERNARG take no effect in this process (frames in flight will not overlap, ~5 % slower "
                      "steps).  Import dualfusion first or export them in the environment.", RuntimeWarning, stacklevel=2)
    return not late


configure_runtime()

from ._lib import Df3dError, LIB_PATH, load as require  # noqa: E402,F401

__all__ = ["require", "Df3dError", "LIB_PATH", "spconv", "ops", "voxel", "msda", "actr", "fusion", "backbones",
           "pipeline", "registry", "synth", "necks", "executor", "fusion_tf", "iou3d_nms", "heads", "transfusion_head", "transfusion"]


def __getattr__(name):
    if name in __all__:
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
