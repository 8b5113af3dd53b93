"""Mirror of the reference's rotated-IoU / NMS interface on libdf3d_hip.so:
`iou3d_nms_utils` (CP/det3d/ops/iou3d_nms/iou3d_nms_utils.py:13-106) and `rotate_nms_pcdet`
(CP/det3d/core/bbox/box_torch_ops.py:248-279).  Same names, argument meaning and return values; the bit matrix is
reduced on the device, so the only host round trip left is the number of kept boxes needed to size the result."""
import numpy as np
import torch

from . import ops as _ops


def boxes_iou_bev(boxes_a, boxes_b):
    """(N, 7), (M, 7) [x, y, z, dx, dy, dz, heading] -> (N, M) rotated BEV IoU."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return _ops.boxes_bev_pairwise(boxes_a.contiguous().float(), boxes_b.contiguous().float(), iou=True)


def to_pcdet(boxes):
    boxes = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    boxes[:, -1] = -boxes[:, -1] - np.pi / 2
    return boxes


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """3-D IoU = BEV overlap x height overlap / union volume (iou3d_nms_utils.py:35-72)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    boxes_a, boxes_b = to_pcdet(boxes_a), to_pcdet(boxes_b)
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = _ops.boxes_bev_pairwise(boxes_a.contiguous().float(), boxes_b.contiguous().float(), iou=False)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _sorted_nms(boxes, scores, thresh, mode, pre_maxsize, post_max_size=None):
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]     # ties: lower index first (reference: unspecified)
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    if order.numel() == 0:
        return order
    keep, num = _ops.nms_bev(boxes[order].contiguous().float(), thresh, mode, max_keep=post_max_size or 0)
    return order[keep[:int(num)].long()].contiguous()


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    assert boxes.shape[1] == 7
    return _sorted_nms(boxes, scores, thresh, _ops.NMS_ROTATED, pre_maxsize), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    assert boxes.shape[1] == 7
    return _sorted_nms(boxes, scores, thresh, _ops.NMS_NORMAL, None), None


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """boxes (N, 7) [x, y, z, l, w, h, theta] in Det3D's frame -> selected indices (box_torch_ops.py:248-279)."""
    boxes = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    boxes[:, -1] = -boxes[:, -1] - np.pi / 2
    return _sorted_nms(boxes, scores, thresh, _ops.NMS_ROTATED, pre_maxsize, post_max_size)


def circle_nms(boxes, min_radius, post_max_size=83):
    """boxes (N, 3) [x, y, score]; centre-distance NMS (center_head.py:506-515 + circle_nms_jit.py:4-27)."""
    b7 = boxes.new_zeros((boxes.shape[0], 7))
    b7[:, :2] = boxes[:, :2]
    return _sorted_nms(b7, boxes[:, 2], min_radius, _ops.NMS_CIRCLE, None, post_max_size)
