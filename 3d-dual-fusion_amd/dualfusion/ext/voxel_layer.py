"""`voxel_layer` (TF/mmdet3d/ops/voxel/src/voxelization.cpp:7-11, voxelization.h:51-87): hard / dynamic voxelisation with
CALLER-ALLOCATED outputs, as `_Voxelization.forward` uses them (TF/mmdet3d/ops/voxel/voxelize.py:41-60)."""
import torch

from .. import ops as _ops
from ._common import need_cuda_contiguous, runtime_errors


@runtime_errors
def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points, max_voxels, NDim=3):
    """Fills voxels [max_voxels, max_points, C] f32, coors [max_voxels, 3] i32 (z, y, x), num_points_per_voxel [max_voxels] i32
    in first-appearance order and returns the number of voxels (a Python int: one D2H read, like the reference)."""
    for t, n in ((points, "points"), (voxels, "voxels"), (coors, "coors"), (num_points_per_voxel, "num_points_per_voxel")):
        need_cuda_contiguous(t, n)
    if NDim != 3:
        raise RuntimeError("hard_voxelize: NDim must be 3")
    if coors.dtype != torch.int32 or num_points_per_voxel.dtype != torch.int32 or voxels.dtype != torch.float32:
        raise RuntimeError("hard_voxelize: voxels f32, coors / num_points_per_voxel int32")
    cap = int(max_voxels)
    if (voxels.shape[0] < cap or coors.shape[0] < cap or num_points_per_voxel.shape[0] < cap or voxels.shape[1] != int(max_points)
            or voxels.shape[2] != points.shape[1] or coors.shape[1] != 3):
        raise RuntimeError("hard_voxelize: output buffers do not match (max_voxels, max_points, C)")
    return _ops.hard_voxelize_into(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, int(max_points), cap)


@runtime_errors
def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    """coors [P, 3] i32 <- (z, y, x) of every point, -1 rows outside the grid."""
    need_cuda_contiguous(points, "points")
    need_cuda_contiguous(coors, "coors")
    if NDim != 3 or coors.dtype != torch.int32 or tuple(coors.shape) != (points.shape[0], 3):
        raise RuntimeError("dynamic_voxelize: coors must be int32 [P, 3]")
    _ops.dynamic_voxelize(points, voxel_size, coors_range, out=coors)


def dynamic_point_to_voxel_forward(*args, **kwargs):
    raise RuntimeError("dynamic_point_to_voxel_forward (DynamicScatter) is not on the 3D-Dual-Fusion path: its configs "
                       "use hard voxelisation + mean VFE")


def dynamic_point_to_voxel_backward(*args, **kwargs):
    raise RuntimeError("dynamic_point_to_voxel_backward (DynamicScatter) is not on the 3D-Dual-Fusion path")
