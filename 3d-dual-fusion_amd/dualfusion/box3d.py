"""LiDAR box container and 3-D IoU of the TransFusion tree, as far as `TransFusionHead.loss` needs them
(SURVEY.md section 8f rows 3-4): `LiDARInstance3DBoxes` (TF/mmdet3d/core/bbox/structures/base_box3d.py:36-126,
280-343,352-438; lidar_box3d.py:37-43,87-90), `xywhr2xyxyr` (structures/utils.py:64-82), `BboxOverlaps3D` /
`bbox_overlaps_3d` (core/bbox/iou_calculators/iou3d_calculator.py:55-166).

The rotated BEV overlap runs on the device (`df3d_boxes_overlap_bev_xyxyr`, csrc/tfloss.hip: the reference's
`boxes_overlap_bev_gpu`, TF/mmdet3d/ops/iou3d/src/iou3d_kernel.cu:122-258); CPU tensors are refused."""
import torch

from . import ops as _ops


def xywhr2xyxyr(boxes_xywhr):
    half_w, half_h = boxes_xywhr[:, 2] / 2, boxes_xywhr[:, 3] / 2
    return torch.stack([boxes_xywhr[:, 0] - half_w, boxes_xywhr[:, 1] - half_h, boxes_xywhr[:, 0] + half_w,
                        boxes_xywhr[:, 1] + half_h, boxes_xywhr[:, 4]], 1)


class LiDARInstance3DBoxes(object):
    """Boxes (x, y, z_bottom, w, l, h, yaw[, vx, vy]) in the LiDAR frame, origin (0.5, 0.5, 0)."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device('cpu')
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        if tensor.shape[-1] == 6:
            assert box_dim == 6
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 1)), dim=-1)
            self.box_dim, self.with_yaw = box_dim + 1, False
        else:
            self.box_dim, self.with_yaw = box_dim, with_yaw
        self.tensor = tensor.clone()
        if origin != (0.5, 0.5, 0):
            self.tensor[:, :3] += self.tensor[:, 3:6] * (self.tensor.new_tensor((0.5, 0.5, 0)) - self.tensor.new_tensor(origin))

    volume = property(lambda self: self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5])
    dims = property(lambda self: self.tensor[:, 3:6])
    yaw = property(lambda self: self.tensor[:, 6])
    height = property(lambda self: self.tensor[:, 5])
    bottom_height = property(lambda self: self.tensor[:, 2])
    top_height = property(lambda self: self.tensor[:, 2] + self.tensor[:, 5])
    bottom_center = property(lambda self: self.tensor[:, :3])
    center = bottom_center
    bev = property(lambda self: self.tensor[:, [0, 1, 3, 4, 6]])
    device = property(lambda self: self.tensor.device)

    @property
    def gravity_center(self):
        g = self.tensor[:, :3].clone()
        g[:, 2] = g[:, 2] + self.tensor[:, 5] * 0.5
        return g

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        t = self.tensor[item]
        return type(self)(t.view(1, -1) if t.dim() == 1 else t, box_dim=self.box_dim, with_yaw=self.with_yaw)

    def to(self, device):
        return type(self)(self.tensor.to(device), box_dim=self.box_dim, with_yaw=self.with_yaw)

    def clone(self):
        return type(self)(self.tensor.clone(), box_dim=self.box_dim, with_yaw=self.with_yaw)

    @classmethod
    def height_overlaps(cls, boxes1, boxes2, mode='iou'):
        lowest_top = torch.min(boxes1.top_height.view(-1, 1), boxes2.top_height.view(1, -1))
        highest_bottom = torch.max(boxes1.bottom_height.view(-1, 1), boxes2.bottom_height.view(1, -1))
        return torch.clamp(lowest_top - highest_bottom, min=0)

    @classmethod
    def overlaps(cls, boxes1, boxes2, mode='iou'):
        assert isinstance(boxes1, LiDARInstance3DBoxes) and type(boxes1) == type(boxes2)
        assert mode in ['iou', 'iof']
        if len(boxes1) * len(boxes2) == 0:
            return boxes1.tensor.new(len(boxes1), len(boxes2))
        overlaps_bev = _ops.boxes_overlap_bev_xyxyr(xywhr2xyxyr(boxes1.bev), xywhr2xyxyr(boxes2.bev))
        overlaps_3d = overlaps_bev * cls.height_overlaps(boxes1, boxes2)
        volume1, volume2 = boxes1.volume.view(-1, 1), boxes2.volume.view(1, -1)
        if mode == 'iou':
            return overlaps_3d / torch.clamp(volume1 + volume2 - overlaps_3d, min=1e-8)
        return overlaps_3d / torch.clamp(volume1, min=1e-8)


def bbox_overlaps_3d(bboxes1, bboxes2, mode='iou', coordinate='lidar'):
    """[N, 7+] x [M, 7+] -> 3-D IoU [N, M] (iou3d_calculator.py:138-166)."""
    assert bboxes1.size(-1) == bboxes2.size(-1) >= 7
    if coordinate != 'lidar':
        raise NotImplementedError("only the LiDAR box type is on the 3D-Dual-Fusion path")
    b1 = LiDARInstance3DBoxes(bboxes1, box_dim=bboxes1.shape[-1])
    b2 = LiDARInstance3DBoxes(bboxes2, box_dim=bboxes2.shape[-1])
    return b1.overlaps(b1, b2, mode=mode)


class BboxOverlaps3D(object):
    def __init__(self, coordinate):
        assert coordinate in ['camera', 'lidar', 'depth']
        self.coordinate = coordinate

    def __call__(self, bboxes1, bboxes2, mode='iou'):
        return bbox_overlaps_3d(bboxes1, bboxes2, mode, self.coordinate)

    def __repr__(self):
        return self.__class__.__name__ + '(coordinate=%s' % self.coordinate
