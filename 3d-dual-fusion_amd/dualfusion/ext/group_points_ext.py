"""`group_points_ext` (TF/mmdet3d/ops/group_points/src/group_points.cpp:35-62), as `GroupingOperation` calls it
(group_points.py:176-206): forward gathers features [b, c, n] by idx [b, npoints, nsample] into out [b, c, npoints,
nsample]; backward scatter-adds grad_out onto grad_points [b, c, n] (zeroed by the caller)."""
import torch

from .. import _lib
from ..ops import _ptr, _stream
from ._common import need_cuda_contiguous, runtime_errors


@runtime_errors
def forward(b, c, n, npoints, nsample, points_tensor, idx_tensor, out_tensor):
    for t, nm in ((points_tensor, "points"), (idx_tensor, "idx"), (out_tensor, "out")):
        need_cuda_contiguous(t, nm)
    if tuple(points_tensor.shape) != (b, c, n) or tuple(idx_tensor.shape) != (b, npoints, nsample) or idx_tensor.dtype != torch.int32:
        raise RuntimeError("group_points forward: points [b, c, n], idx int32 [b, npoints, nsample]")
    lib = _lib.load()
    _lib.check(lib.df3d_group_points(_ptr(points_tensor), _ptr(idx_tensor), int(b), int(c), int(n), int(npoints), int(nsample),
                                     _ptr(out_tensor), _stream()), "df3d_group_points")
    return 1


@runtime_errors
def backward(b, c, n, npoints, nsample, grad_out_tensor, idx_tensor, grad_points_tensor):
    for t, nm in ((grad_out_tensor, "grad_out"), (idx_tensor, "idx"), (grad_points_tensor, "grad_points")):
        need_cuda_contiguous(t, nm)
    if tuple(grad_out_tensor.shape) != (b, c, npoints, nsample) or tuple(grad_points_tensor.shape) != (b, c, n) \
            or idx_tensor.dtype != torch.int32:
        raise RuntimeError("group_points backward: grad_out [b, c, npoints, nsample], grad_points [b, c, n]")
    lib = _lib.load()
    _lib.check(lib.df3d_group_points_grad(_ptr(grad_out_tensor), _ptr(idx_tensor), int(b), int(c), int(n), int(npoints),
                                          int(nsample), _ptr(grad_points_tensor), _stream()), "df3d_group_points_grad")
    return 1
