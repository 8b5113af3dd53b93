"""Voxelisation layer and mean voxel feature encoders with the reference's class names.

`Voxelization` = TF/mmdet3d/ops/voxel/voxelize.py:61-113 (hard voxelisation; `max_voxels` is a
(train, test) pair).  `VoxelGenerator` = CP/det3d/core/input/voxel_generator.py:5-30 (numba
semantics at the cap).  VFEs: `VoxelFeatureExtractorV3` (CP/.../voxel_encoder.py:8-24),
`HardSimpleVFE` (TF/.../voxel_encoder.py:13-44), `MeanVFE` (VR/.../mean_vfe.py:6-29).
The HIP kernel fuses the mean into the voxeliser; `voxelize_mean()` exposes that fused form and
skips the padded [M, T, C] tensor entirely."""
import torch
from torch import nn

from . import ops as _ops
from .registry import READERS, VOXEL_ENCODERS


class _Voxelization(object):
    @staticmethod
    def apply(points, voxel_size, coors_range, max_points=35, max_voxels=20000):
        """voxelize.py:13-58.  max_points == -1 or max_voxels == -1 selects dynamic voxelisation
        (per-point (z, y, x) coordinates, -1 outside the grid)."""
        if max_points == -1 or max_voxels == -1:
            return _ops.dynamic_voxelize(points.contiguous().float(), voxel_size, coors_range)
        voxels, coors, num, _ = _ops.hard_voxelize(points.contiguous().float(), voxel_size, coors_range, max_points,
                                                   max_voxels, break_at_cap=True, want_voxels=True, want_mean=False)
        return voxels, coors, num


voxelization = _Voxelization.apply


class Voxelization(nn.Module):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super(Voxelization, self).__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        if isinstance(max_voxels, tuple) or isinstance(max_voxels, list):
            self.max_voxels = tuple(max_voxels)
        else:
            self.max_voxels = (max_voxels, max_voxels)
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        input_feat_shape = grid_size[:2]
        self.grid_size = grid_size
        self.pcd_shape = [*input_feat_shape, 1][::-1]

    def _cap(self):
        return self.max_voxels[0] if self.training else self.max_voxels[1]

    def forward(self, input):
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points, self._cap())

    def voxelize_mean(self, points, break_at_cap=True, batch_index=None, while_waiting=None):
        """fused voxelise + mean VFE: (mean [M,C], coors [M,3] (z,y,x) -- or [M,4] (b,z,y,x) when `batch_index`
        is given --, num [M])."""
        _, coors, num, mean = _ops.hard_voxelize(points.contiguous().float(), self.voxel_size, self.point_cloud_range,
                                                 self.max_num_points, self._cap(), break_at_cap=break_at_cap,
                                                 want_voxels=False, want_mean=True, batch_index=batch_index,
                                                 while_waiting=while_waiting)
        return mean, coors, num

    def __repr__(self):
        return "%s(voxel_size=%s, point_cloud_range=%s, max_num_points=%s, max_voxels=%s)" % (
            self.__class__.__name__, self.voxel_size, self.point_cloud_range, self.max_num_points, self.max_voxels)


class VoxelGenerator(object):
    """CP/det3d/core/input/voxel_generator.py:5-30 on the GPU (numba `continue` semantics at the cap)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = max_num_points
        self._max_voxels = max_voxels

    def generate(self, points, max_voxels=-1):
        mv = self._max_voxels if max_voxels == -1 else max_voxels
        voxels, coors, num, _ = _ops.hard_voxelize(points.contiguous().float(), self._voxel_size,
                                                   self._point_cloud_range, self._max_num_points, mv,
                                                   break_at_cap=False, want_voxels=True, want_mean=False)
        return voxels, coors, num


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super(VoxelFeatureExtractorV3, self).__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        assert self.num_input_features == features.shape[-1]
        points_mean = features[:, :, :self.num_input_features].sum(dim=1, keepdim=False) / \
            num_voxels.type_as(features).view(-1, 1)
        return points_mean.contiguous()


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    def __init__(self, num_features=4):
        super(HardSimpleVFE, self).__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors):
        points_mean = features[:, :, :self.num_features].sum(dim=1, keepdim=False) / \
            num_points.type_as(features).view(-1, 1)
        return points_mean.contiguous()


class MeanVFE(nn.Module):
    """VR/pcdet/models/backbones_3d/vfe/mean_vfe.py:6-29 (batch_dict in / out, count clamped >= 1)."""

    def __init__(self, model_cfg=None, num_point_features=4, **kwargs):
        super(MeanVFE, self).__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        voxel_features, voxel_num_points = batch_dict['voxels'], batch_dict['voxel_num_points']
        points_mean = voxel_features[:, :, :].sum(dim=1, keepdim=False)
        normalizer = torch.clamp_min(voxel_num_points.view(-1, 1), min=1.0).type_as(voxel_features)
        batch_dict['voxel_features'] = (points_mean / normalizer).contiguous()
        return batch_dict
