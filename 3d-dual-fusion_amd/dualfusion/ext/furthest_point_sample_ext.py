"""`furthest_point_sample_ext` (TF/mmdet3d/ops/furthest_point_sample/src/furthest_point_sample.cpp:32-66), as
`FurthestPointSampling.forward` calls it (furthest_point_sample.py:28-34,63-69): caller-allocated `temp` / `idx`."""
import torch

from .. import _lib
from ..ops import _ptr, _stream
from ._common import need_cuda_contiguous, runtime_errors


def _run(entry, b, n, m, points, temp, idx, shape):
    for t, nm in ((points, "points"), (temp, "temp"), (idx, "idx")):
        need_cuda_contiguous(t, nm)
    if points.dtype != torch.float32 or temp.dtype != torch.float32 or idx.dtype != torch.int32:
        raise RuntimeError("points / temp must be float32, idx int32")
    if tuple(points.shape) != shape or temp.numel() < b * n or idx.numel() < b * m:
        raise RuntimeError("points %s does not match (b, n) = (%d, %d)" % (tuple(points.shape), b, n))
    lib = _lib.load()
    _lib.check(getattr(lib, entry)(_ptr(points), int(b), int(n), int(m), _ptr(temp), _ptr(idx), _stream()), entry)
    return 1


@runtime_errors
def furthest_point_sampling_wrapper(b, n, m, points_tensor, temp_tensor, idx_tensor):
    """points [b, n, 3] -> idx [b, m] (first pick = point 0)."""
    return _run("df3d_furthest_point_sample", b, n, m, points_tensor, temp_tensor, idx_tensor, (b, n, 3))


@runtime_errors
def furthest_point_sampling_with_dist_wrapper(b, n, m, points_tensor, temp_tensor, idx_tensor):
    """points = pairwise distances [b, n, n] -> idx [b, m]."""
    return _run("df3d_furthest_point_sample_with_dist", b, n, m, points_tensor, temp_tensor, idx_tensor, (b, n, n))
