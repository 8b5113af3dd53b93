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


def _greedy_keep(over, keep):
    """The reference's host loop over the suppression matrix (iou3d.cpp:127-143): box i survives unless an earlier survivor
    suppresses it; survivors' indices go into the caller's CPU `keep` tensor, the count is returned."""
    import numpy as np
    m = over.cpu().numpy()
    n = m.shape[0]
    removed = np.zeros((n,), dtype=bool)
    out = []
    for i in range(n):
        if not removed[i]:
            out.append(i)
            removed[i + 1:] |= m[i, i + 1:]
    keep[:len(out)] = torch.as_tensor(out, dtype=keep.dtype)
    return len(out)


def _check_nms_args(boxes, keep):
    need_cuda_contiguous(boxes, "boxes")
    if boxes.dim() != 2 or boxes.shape[1] != 5:
        raise RuntimeError("boxes must be [N, 5] (x1, y1, x2, y2, ry)")
    if keep.dtype != torch.long or not keep.is_contiguous() or keep.numel() < boxes.shape[0]:
        raise RuntimeError("keep must be a contiguous int64 tensor with at least N entries")


@runtime_errors
def nms_gpu(boxes, keep, nms_overlap_thresh, device_id=0):
    """iou3d.cpp:92-145: rotated NMS of score-sorted [N, 5] boxes; `keep` (CPU int64) receives the kept indices, returns
    how many.  Pair test = the kernel's `iou_bev(a, b) > thresh` (iou3d_kernel.cu:284-332): rotated overlap / max(Sa + Sb -
    overlap, EPS) on the device, the greedy pass on the host as in the reference."""
    _check_nms_args(boxes, keep)
    if boxes.shape[0] == 0:
        return 0
    b = boxes.float()
    ov = _ops.boxes_overlap_bev_xyxyr(b, b)
    s = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    iou = ov / torch.clamp(s.view(-1, 1) + s.view(1, -1) - ov, min=1e-8)
    return _greedy_keep(iou > float(nms_overlap_thresh), keep)


@runtime_errors
def nms_normal_gpu(boxes, keep, nms_overlap_thresh, device_id=0):
    """iou3d.cpp:147-201: the same with the axis-aligned `iou_normal` (iou3d_kernel.cu:335-343; the angle is ignored)."""
    _check_nms_args(boxes, keep)
    if boxes.shape[0] == 0:
        return 0
    b = boxes.float()
    left, right = torch.max(b[:, None, 0], b[None, :, 0]), torch.min(b[:, None, 2], b[None, :, 2])
    top, bottom = torch.max(b[:, None, 1], b[None, :, 1]), torch.min(b[:, None, 3], b[None, :, 3])
    inter = torch.clamp(right - left, min=0) * torch.clamp(bottom - top, min=0)
    s = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    iou = inter / torch.clamp(s.view(-1, 1) + s.view(1, -1) - inter, min=1e-8)
    return _greedy_keep(iou > float(nms_overlap_thresh), keep)
