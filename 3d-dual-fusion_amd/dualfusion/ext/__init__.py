"""Drop-in modules under the NAMES of the reference's compiled extensions: the same function
names, argument order, caller-allocated outputs and error behaviour as the pybind modules, on libdf3d_hip.so.

  sparse_conv_ext               TF/mmdet3d/ops/spconv/src/all.cc:21-51
  voxel_layer                   TF/mmdet3d/ops/voxel/src/voxelization.cpp:7-11
  MultiScaleDeformableAttention CP/det3d/models/model_utils/ops/src/vision.cpp:13-16
  furthest_point_sample_ext     TF/mmdet3d/ops/furthest_point_sample/src/furthest_point_sample.cpp:59-66
  ball_query_ext                TF/mmdet3d/ops/ball_query/src/ball_query.cpp:45-47
  group_points_ext              TF/mmdet3d/ops/group_points/src/group_points.cpp:59-62
  gather_points_ext             TF/mmdet3d/ops/gather_points/src/gather_points.cpp:54-59
  iou3d_cuda                    TF/mmdet3d/ops/iou3d/src/iou3d.cpp (boxes_overlap_bev_gpu / boxes_iou_bev_gpu / nms_gpu / nms_normal_gpu)

`install()` puts them into `sys.modules` under the import paths the reference's Python wrappers use, so that
`from . import sparse_conv_ext` (TF/mmdet3d/ops/spconv/ops.py:17), `from .voxel_layer import hard_voxelize`
(voxelize.py:7), `import MultiScaleDeformableAttention as MSDA` (ms_deform_attn_func.py:18) ... bind to these modules
when the compiled ones are absent -- see INTEGRATION.md.  Errors surface as RuntimeError like the pybind modules'
(C++ exceptions); CPU tensors are refused: there is no CPU fallback."""
import importlib
import sys

NAMES = ("sparse_conv_ext", "voxel_layer", "MultiScaleDeformableAttention", "furthest_point_sample_ext", "ball_query_ext",
         "group_points_ext", "gather_points_ext", "iou3d_cuda")

# import paths of the reference's wrappers (module attribute of a package, or a top-level module)
ALIASES = {
    "sparse_conv_ext": ("mmdet3d.ops.spconv.sparse_conv_ext",),
    "voxel_layer": ("mmdet3d.ops.voxel.voxel_layer",),
    "MultiScaleDeformableAttention": ("MultiScaleDeformableAttention",),
    "furthest_point_sample_ext": ("mmdet3d.ops.furthest_point_sample.furthest_point_sample_ext",
                                  "det3d.ops.furthest_point_sample.furthest_point_sample_ext"),
    "ball_query_ext": ("mmdet3d.ops.ball_query.ball_query_ext", "det3d.ops.ball_query.ball_query_ext"),
    "group_points_ext": ("mmdet3d.ops.group_points.group_points_ext", "det3d.ops.group_points.group_points_ext"),
    "gather_points_ext": ("mmdet3d.ops.gather_points.gather_points_ext", "det3d.ops.gather_points.gather_points_ext"),
    "iou3d_cuda": ("mmdet3d.ops.iou3d.iou3d_cuda",),
}


def load(name):
    if name not in NAMES:
        raise KeyError(name)
    return importlib.import_module("." + name, __name__)


def install(overwrite=False):
    """Register every shim under the reference's import paths (only where no module of that name is loaded yet, unless
    `overwrite`).  Parent packages that are already imported get the attribute too, so `from . import X` resolves."""
    done = []
    for name, paths in ALIASES.items():
        mod = load(name)
        for path in paths:
            if path in sys.modules and not overwrite:
                continue
            sys.modules[path] = mod
            parent, _, leaf = path.rpartition(".")
            if parent and parent in sys.modules:
                setattr(sys.modules[parent], leaf, mod)
            done.append(path)
    return done
