"""Sparse 3-D backbones of the three reference trees, with the reference's registry names,
constructor keys, forward signatures and parameter names/shapes, built on
the MI355X sparse-conv modules.

CenterPoint:  SpMiddleResNetFHD / SpMiddleResNetFHDFusion  (CP/det3d/models/backbones/scn.py:97-236)
TransFusion:  SparseEncoder / SparseEncoderFusion          (TF/mmdet3d/models/middle_encoders/sparse_encoder.py)
Voxel-RCNN:   VoxelBackBone8x / VoxelBackBone8xFusion      (VR/pcdet/models/backbones_3d/spconv_backbone.py)
"""
import numpy as np
import torch
from torch import nn

from . import spconv
from .registry import BACKBONES, BACKBONES_3D, MIDDLE_ENCODERS
from .spconv import SparseConv3d, SubMConv3d
from .spconv.modules import can_fold, fold_batchnorm


def build_norm_layer(cfg, num_features, postfix=""):
    """CP/det3d/models/utils/norm.py (the only norm the sparse backbones use is BN1d)."""
    cfg = dict(cfg)
    t = cfg.pop("type")
    if t not in ("BN1d", "BN", "naiveSyncBN1d"):
        raise KeyError("unsupported norm type for sparse backbones: %s" % t)
    cfg.pop("requires_grad", None)
    cfg.setdefault("eps", 1e-5)
    layer = nn.BatchNorm1d(num_features, **cfg)
    return "bn" + str(postfix), layer


def replace_feature(out, new_features):
    return out.replace_feature(new_features)


def conv3x3(in_planes, out_planes, stride=1, indice_key=None, bias=True):
    return spconv.SubMConv3d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=bias,
                             indice_key=indice_key)


def conv1x1(in_planes, out_planes, stride=1, indice_key=None, bias=True):
    return spconv.SubMConv3d(in_planes, out_planes, kernel_size=1, stride=stride, padding=1, bias=bias,
                             indice_key=indice_key)


def _fused_basic_block(x, conv1, bn1, conv2, bn2, downsample):
    """conv1-bn1-relu-conv2-bn2-(+identity)-relu as two fused kernels (eval mode)."""
    s1, h1 = fold_batchnorm(bn1)
    s2, h2 = fold_batchnorm(bn2)
    identity = x
    out = conv1.forward_fused(x, scale=s1, shift=h1, relu=True)
    if downsample is not None:
        identity = downsample(x)
    return conv2.forward_fused(out, scale=s2, shift=h2, relu=True, residual=identity.features.contiguous())


class SparseBasicBlock(spconv.SparseModule):
    """CP/det3d/models/backbones/scn.py:51-94 (convs carry a bias because norm_cfg is not None)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_cfg=None, downsample=None, indice_key=None):
        super(SparseBasicBlock, self).__init__()
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        bias = norm_cfg is not None
        self.conv1 = conv3x3(inplanes, planes, stride, indice_key=indice_key, bias=bias)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU()
        self.conv2 = conv3x3(planes, planes, indice_key=indice_key, bias=bias)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if not self.training and can_fold(self.bn1) and can_fold(self.bn2):
            return _fused_basic_block(x, self.conv1, self.bn1, self.conv2, self.bn2, self.downsample)
        identity = x
        out = self.conv1(x)
        out = replace_feature(out, self.bn1(out.features))
        out = replace_feature(out, self.relu(out.features))
        out = self.conv2(out)
        out = replace_feature(out, self.bn2(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = replace_feature(out, out.features + identity.features)
        out = replace_feature(out, self.relu(out.features))
        return out


@BACKBONES.register_module
class SpMiddleResNetFHD(nn.Module):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHD", **kwargs):
        super(SpMiddleResNetFHD, self).__init__()
        self.name = name
        self.dcn = None
        self.zero_init_residual = False
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        self.conv_input = spconv.SparseSequential(
            SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"),
            build_norm_layer(norm_cfg, 16)[1],
            nn.ReLU(inplace=True))
        self.conv1 = spconv.SparseSequential(
            SparseBasicBlock(16, 16, norm_cfg=norm_cfg, indice_key="res0"),
            SparseBasicBlock(16, 16, norm_cfg=norm_cfg, indice_key="res0"))
        self.conv2 = spconv.SparseSequential(
            SparseConv3d(16, 32, 3, 2, padding=1, bias=False),
            build_norm_layer(norm_cfg, 32)[1],
            nn.ReLU(inplace=True),
            SparseBasicBlock(32, 32, norm_cfg=norm_cfg, indice_key="res1"),
            SparseBasicBlock(32, 32, norm_cfg=norm_cfg, indice_key="res1"))
        self.conv3 = spconv.SparseSequential(
            SparseConv3d(32, 64, 3, 2, padding=1, bias=False),
            build_norm_layer(norm_cfg, 64)[1],
            nn.ReLU(inplace=True),
            SparseBasicBlock(64, 64, norm_cfg=norm_cfg, indice_key="res2"),
            SparseBasicBlock(64, 64, norm_cfg=norm_cfg, indice_key="res2"))
        self.conv4 = spconv.SparseSequential(
            SparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False),
            build_norm_layer(norm_cfg, 128)[1],
            nn.ReLU(inplace=True),
            SparseBasicBlock(128, 128, norm_cfg=norm_cfg, indice_key="res3"),
            SparseBasicBlock(128, 128, norm_cfg=norm_cfg, indice_key="res3"))
        self.extra_conv = spconv.SparseSequential(
            SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False),
            build_norm_layer(norm_cfg, 128)[1],
            nn.ReLU())

    def _stem(self, voxel_features, coors, batch_size, input_shape):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        coors = coors.int()
        ret = spconv.SparseConvTensor(voxel_features, coors, sparse_shape, batch_size)
        x = self.conv_input(ret)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        return x_conv1, x_conv2, x_conv3, x_conv4

    def _tail(self, x_conv1, x_conv2, x_conv3, x_conv4):
        ret = self.extra_conv(x_conv4)
        ret = ret.dense()
        N, C, D, H, W = ret.shape
        ret = ret.view(N, C * D, H, W)
        return ret, {'conv1': x_conv1, 'conv2': x_conv2, 'conv3': x_conv3, 'conv4': x_conv4}

    def forward(self, voxel_features, coors, batch_size, input_shape):
        return self._tail(*self._stem(voxel_features, coors, batch_size, input_shape))


@BACKBONES.register_module
class SpMiddleResNetFHDFusion(SpMiddleResNetFHD):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHDFusion", **kwargs):
        super(SpMiddleResNetFHDFusion, self).__init__(num_input_features, norm_cfg, name, **kwargs)

    def forward(self, voxel_features, batch_dict, coors, batch_size, input_shape, example, fuse_func=None):
        x_conv1, x_conv2, x_conv3, x_conv4 = self._stem(voxel_features, coors, batch_size, input_shape)
        if fuse_func is not None and fuse_func.fuse_mode == 'pfat':
            x_conv4 = fuse_func(batch_dict, example, encoded_voxel_list=[x_conv2, x_conv3, x_conv4],
                                layer_name='layer1_ori', fuse_mode='pfat', d_factor_list=[2, 4, 8])
        return self._tail(x_conv1, x_conv2, x_conv3, x_conv4)
