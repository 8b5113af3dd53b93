"""`MultiScaleDeformableAttention` (CP/det3d/models/model_utils/ops/src/vision.cpp:13-16, ms_deform_attn.h:21-62), as
`MSDeformAttnFunction` calls it (ops/functions/ms_deform_attn_func.py:21-38)."""
import torch

from .. import ops as _ops
from ._common import need_cuda_contiguous, runtime_errors


def _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    for t, n in ((value, "value"), (spatial_shapes, "spatial_shapes"), (level_start_index, "level_start_index"),
                 (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight")):
        need_cuda_contiguous(t, n)
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")


@runtime_errors
def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value [N, S, M, D], sampling_loc [N, Lq, M, L, P, 2], attn_weight [N, Lq, M, L, P] -> [N, Lq, M * D].
    `im2col_step` (the reference's batch chunking) has no effect on the result; the batch must divide by it as there."""
    _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    step = min(int(value.shape[0]), int(im2col_step))
    if step <= 0 or value.shape[0] % step != 0:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (value.shape[0], step))
    return _ops.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)


@runtime_errors
def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]."""
    _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    need_cuda_contiguous(grad_output, "grad_output")
    return list(_ops.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output))
