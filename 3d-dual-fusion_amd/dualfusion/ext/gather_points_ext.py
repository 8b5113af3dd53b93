"""`gather_points_ext` (TF/mmdet3d/ops/gather_points/src/gather_points.cpp:26-59), as `GatherPoints` calls it
(gather_points.py:29-49): out [b, c, npoints] = points [b, c, n] at idx [b, npoints]; the gradient is scatter-added onto
grad_points [b, c, n] (zeroed by the caller)."""
import torch

from .. import _lib
from ..ops import _ptr, _stream
from ._common import need_cuda_contiguous, runtime_errors


@runtime_errors
def gather_points_wrapper(b, c, n, npoints, points_tensor, idx_tensor, out_tensor):
    for t, nm in ((points_tensor, "points"), (idx_tensor, "idx"), (out_tensor, "out")):
        need_cuda_contiguous(t, nm)
    if tuple(points_tensor.shape) != (b, c, n) or tuple(idx_tensor.shape) != (b, npoints) or idx_tensor.dtype != torch.int32:
        raise RuntimeError("gather_points_wrapper: points [b, c, n], idx int32 [b, npoints]")
    lib = _lib.load()
    _lib.check(lib.df3d_gather_points(_ptr(points_tensor), _ptr(idx_tensor), int(b), int(c), int(n), int(npoints),
                                      _ptr(out_tensor), _stream()), "df3d_gather_points")
    return 1


@runtime_errors
def gather_points_grad_wrapper(b, c, n, npoints, grad_out_tensor, idx_tensor, grad_points_tensor):
    for t, nm in ((grad_out_tensor, "grad_out"), (idx_tensor, "idx"), (grad_points_tensor, "grad_points")):
        need_cuda_contiguous(t, nm)
    if tuple(grad_out_tensor.shape) != (b, c, npoints) or tuple(grad_points_tensor.shape) != (b, c, n) or idx_tensor.dtype != torch.int32:
        raise RuntimeError("gather_points_grad_wrapper: grad_out [b, c, npoints], grad_points [b, c, n]")
    lib = _lib.load()
    _lib.check(lib.df3d_gather_points_grad(_ptr(grad_out_tensor), _ptr(idx_tensor), int(b), int(c), int(n), int(npoints),
                                           _ptr(grad_points_tensor), _stream()), "df3d_gather_points_grad")
    return 1
