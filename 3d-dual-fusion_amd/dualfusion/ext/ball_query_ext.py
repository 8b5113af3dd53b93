"""`ball_query_ext` (TF/mmdet3d/ops/ball_query/src/ball_query.cpp:30-47), as `BallQuery.forward` calls it
(ball_query.py:33-38): caller-allocated idx [b, m, nsample] int32."""
import torch

from .. import _lib
from ..ops import _ptr, _stream
from ._common import need_cuda_contiguous, runtime_errors


@runtime_errors
def ball_query_wrapper(b, n, m, min_radius, max_radius, nsample, new_xyz_tensor, xyz_tensor, idx_tensor):
    for t, nm in ((new_xyz_tensor, "new_xyz"), (xyz_tensor, "xyz"), (idx_tensor, "idx")):
        need_cuda_contiguous(t, nm)
    if tuple(xyz_tensor.shape) != (b, n, 3) or tuple(new_xyz_tensor.shape) != (b, m, 3) or idx_tensor.dtype != torch.int32 \
            or idx_tensor.numel() < b * m * nsample:
        raise RuntimeError("ball_query_wrapper: xyz [b, n, 3], new_xyz [b, m, 3], idx int32 [b, m, nsample]")
    lib = _lib.load()
    _lib.check(lib.df3d_ball_query(_ptr(new_xyz_tensor), _ptr(xyz_tensor), int(b), int(n), int(m), float(min_radius),
                                   float(max_radius), int(nsample), _ptr(idx_tensor), _stream()), "df3d_ball_query")
    return 1
