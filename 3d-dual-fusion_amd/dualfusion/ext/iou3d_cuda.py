"""`iou3d_cuda` of the TransFusion tree (TF/mmdet3d/ops/iou3d/src/iou3d.cpp:66-90 and :92-120: boxes_overlap_bev_gpu /
boxes_iou_bev_gpu), as LiDARInstance3DBoxes.overlaps and iou3d_utils.boxes_iou_bev call them
(core/bbox/structures/base_box3d.py:419-423, ops/iou3d/iou3d_utils.py:7-24): boxes [n, 5] (x1, y1, x2, y2, angle), the result
is written into the caller's tensor."""
import torch

from .. import ops as _ops
from ._common import need_cuda_contiguous, runtime_errors


@runtime_errors
def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_overlap, "ans_overlap")):
        need_cuda_contiguous(t, n)
    if tuple(ans_overlap.shape) != (boxes_a.shape[0], boxes_b.shape[0]):
        raise RuntimeError("ans_overlap must be [N, M]")
    ans_overlap.copy_(_ops.boxes_overlap_bev_xyxyr(boxes_a, boxes_b))
    return 1


@runtime_errors
def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_iou, "ans_iou")):
        need_cuda_contiguous(t, n)
    ov = _ops.boxes_overlap_bev_xyxyr(boxes_a, boxes_b)
    sa = ((boxes_a[:, 2] - boxes_a[:, 0]) * (boxes_a[:, 3] - boxes_a[:, 1])).view(-1, 1)
    sb = ((boxes_b[:, 2] - boxes_b[:, 0]) * (boxes_b[:, 3] - boxes_b[:, 1])).view(1, -1)
    ans_iou.copy_(ov / torch.clamp(sa + sb - ov, min=1e-8))
    return 1
