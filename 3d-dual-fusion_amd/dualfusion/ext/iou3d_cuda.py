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


_NMS_ROWS = 2048     # rows of the suppression matrix alive at a time: [2048, N] instead of the dense [N, N]


def _greedy_keep(n, over_rows, keep):
    """The reference's host loop over the suppression matrix (iou3d.cpp:127-143): box i survives unless an earlier survivor
    suppresses it; survivors' indices go into the caller's CPU `keep` tensor, the count is returned.  The matrix is produced
    and consumed in blocks of `_NMS_ROWS` rows -- `over_rows(i0, i1)` -> bool [i1 - i0, n - i0] on the device, columns from
    box i0 on (only later boxes matter to a row) -- so device and host memory stay at rows x N where the reference keeps
    N x N / 64 bit words; same decisions in the same order."""
    import numpy as np
    removed = np.zeros((n,), dtype=bool)
    out = []
    for i0 in range(0, n, _NMS_ROWS):
        i1 = min(n, i0 + _NMS_ROWS)
        if removed[i0:i1].all():
            continue
        m = over_rows(i0, i1).cpu().numpy()
        for i in range(i0, i1):
            if not removed[i]:
                out.append(i)
                removed[i + 1:] |= m[i - i0, i + 1 - i0:]
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
    b = boxes.float().contiguous()
    s = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    thr = float(nms_overlap_thresh)

    def rows(i0, i1):
        ov = _ops.boxes_overlap_bev_xyxyr(b[i0:i1].contiguous(), b[i0:].contiguous())
        return ov / torch.clamp(s[i0:i1].view(-1, 1) + s[i0:].view(1, -1) - ov, min=1e-8) > thr
    return _greedy_keep(b.shape[0], rows, keep)


@runtime_errors
def nms_normal_gpu(boxes, keep, nms_overlap_thresh, device_id=0):
    """iou3d.cpp:147-201: the same with the axis-aligned `iou_normal` (iou3d_kernel.cu:335-343; the angle is ignored)."""
    _check_nms_args(boxes, keep)
    if boxes.shape[0] == 0:
        return 0
    b = boxes.float().contiguous()
    s = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    thr = float(nms_overlap_thresh)

    def rows(i0, i1):
        a, c = b[i0:i1], b[i0:]
        left, right = torch.max(a[:, None, 0], c[None, :, 0]), torch.min(a[:, None, 2], c[None, :, 2])
        top, bottom = torch.max(a[:, None, 1], c[None, :, 1]), torch.min(a[:, None, 3], c[None, :, 3])
        inter = torch.clamp(right - left, min=0) * torch.clamp(bottom - top, min=0)
        return inter / torch.clamp(s[i0:i1].view(-1, 1) + s[i0:].view(1, -1) - inter, min=1e-8) > thr
    return _greedy_keep(b.shape[0], rows, keep)
