"""Sparse 3-D backbones of the three reference trees, with the reference's registry names,
constructor keys, forward signatures and parameter names/shapes, built on
the MI355X sparse-conv modules.

CenterPoint:  SpMiddleResNetFHD / SpMiddleResNetFHDFusion  (CP/det3d/models/backbones/scn.py:97-236)
TransFusion:  SparseEncoder / SparseEncoderFusion          (TF/mmdet3d/models/middle_encoders/sparse_encoder.py)
Voxel-RCNN:   VoxelBackBone8x / VoxelBackBone8xFusion      (VR/pcdet/models/backbones_3d/spconv_backbone.py)
"""
import os

import numpy as np
import torch
from torch import nn

from . import ops as _ops
from . import spconv
from .registry import BACKBONES, BACKBONES_3D, MIDDLE_ENCODERS
from .spconv import SparseConv3d, SubMConv3d
from .spconv.modules import can_fold, fold_batchnorm, wants_grad


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
    return conv2.forward_fused(out, scale=s2, shift=h2, relu=True, residual=identity)


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
        if not self.training and can_fold(self.bn1) and can_fold(self.bn2) and not wants_grad(x, self):
            return _fused_basic_block(x, self.conv1, self.bn1, self.conv2, self.bn2, self.downsample)
        identity = x
        out = self.conv1(x)
        out = replace_feature(out, _ops.batch_norm_rows(self.bn1, out.features, relu=True))
        out = self.conv2(out)
        out = replace_feature(out, _ops.batch_norm_rows(self.bn2, out.features))
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

    def _plan(self):
        """Native executor plan of the conv chain (dualfusion/executor.py) for inference; None -> module path."""
        if self.training or os.environ.get("DF3D_EXECUTOR", "1") != "1":
            return None
        plan = getattr(self, "_exec_plan", False)
        if plan is False:
            from .executor import compile_stages
            plan = compile_stages([("conv_input", self.conv_input), ("conv1", self.conv1), ("conv2", self.conv2),
                                   ("conv3", self.conv3), ("conv4", self.conv4)],
                                  geometry_stages=[("conv4", self.extra_conv)])
            object.__setattr__(self, "_exec_plan", plan)
        return plan

    def train(self, mode=True):
        object.__setattr__(self, "_exec_plan", False)
        return super(SpMiddleResNetFHD, self).train(mode)

    # the stages the camera-fusion adapter reads and their down-sampling factors (forward() of the fusion variant below)
    FUSE_STAGES = ("conv2", "conv3", "conv4")
    FUSE_D_FACTORS = [2, 4, 8]

    def prepare_geometry(self, coors, in_channels, batch_size, input_shape):
        """Every rulebook / index set of this backbone from the voxel COORDINATES alone (executor phase 1,
        `BackbonePlan.build_geometry`): blocks the calling thread for the strided layers' output counts, so it is meant for a
        helper thread that works a frame ahead (dualfusion/prefetch.py).  -> PreparedGeometry for `forward(..., prepared=)`,
        or None when the native plan is not available (training mode, DF3D_EXECUTOR=0)."""
        plan = self._plan() if coors.is_cuda and not torch.is_grad_enabled() else None
        if plan is None or coors.shape[0] == 0:
            return None
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        return plan.build_geometry(coors.int(), in_channels, batch_size, [int(v) for v in sparse_shape],
                             ready=getattr(coors, "_df3d_ready", None))

    def _stem(self, voxel_features, coors, batch_size, input_shape, prepared=None):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        coors = coors.int()
        plan = self._plan() if voxel_features.is_cuda and not torch.is_grad_enabled() else None
        if prepared is not None and plan is not None and prepared.plan is plan:
            out = plan.run_convs(prepared, voxel_features)
            return out["conv1"], out["conv2"], out["conv3"], out["conv4"]
        if plan is not None and voxel_features.shape[0] > 0:
            out = plan.run(voxel_features, coors, batch_size, [int(v) for v in sparse_shape])
            return out["conv1"], out["conv2"], out["conv3"], out["conv4"]
        ret = spconv.SparseConvTensor(voxel_features, coors, sparse_shape, batch_size)
        ready = getattr(coors, "_df3d_ready", None)
        if ready is not None and voxel_features.is_cuda and os.environ.get("DF3D_TRAIN_GEO_STREAM", "1") == "1":
            # module path (training) on resident inputs: rulebooks on a stream of their own behind the voxeliser's event
            # (spconv/conv.py `_rulebook`): their count round trips do not wait for the previous step's backward
            from .spconv.conv import GEOMETRY_STREAM_KEY
            geo = self.__dict__.get("_rulebook_stream")
            if geo is None:
                geo = self.__dict__["_rulebook_stream"] = torch.cuda.Stream(device=coors.device)
            geo.wait_event(ready)
            coors.record_stream(geo)
            ret.indice_dict[GEOMETRY_STREAM_KEY] = geo
        x = self.conv_input(ret)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        return x_conv1, x_conv2, x_conv3, x_conv4

    def _tail(self, x_conv1, x_conv2, x_conv3, x_conv4):
        ret = self.extra_conv(x_conv4)
        if getattr(self, "dense_layout", "nchw") == "rows" and ret.features.is_cuda:
            # channels-last pixel rows for the row-kernel neck (necks.RPN.forward_rows): same values as
            # dense().view(N, C*D, H, W), laid out [N*H*W, C*D]
            D, H, W = [int(v) for v in ret.spatial_shape]
            ok = getattr(self, "dense_split", None)
            if ok is not None and (ret.features.shape[1] * D) % 8 == 0 and ok():
                # the consumer (the row-kernel neck) reads split rows only: written directly, no fp32 map, no split pass
                from .necks import SplitRows
                rows = SplitRows(ret.dense_rows(split=True), ret.features.shape[1] * D)
            else:
                rows = ret.dense_rows()
            ret = (rows, (ret.batch_size, ret.features.shape[1] * D, H, W))
        else:
            ret = ret.dense()
            N, C, D, H, W = ret.shape
            ret = ret.view(N, C * D, H, W)
        return ret, {'conv1': x_conv1, 'conv2': x_conv2, 'conv3': x_conv3, 'conv4': x_conv4}

    def forward(self, voxel_features, coors, batch_size, input_shape, prepared=None):
        return self._tail(*self._stem(voxel_features, coors, batch_size, input_shape, prepared=prepared))


@BACKBONES.register_module
class SpMiddleResNetFHDFusion(SpMiddleResNetFHD):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHDFusion", **kwargs):
        super(SpMiddleResNetFHDFusion, self).__init__(num_input_features, norm_cfg, name, **kwargs)

    def forward(self, voxel_features, batch_dict, coors, batch_size, input_shape, example, fuse_func=None, prepared=None):
        x_conv1, x_conv2, x_conv3, x_conv4 = self._stem(voxel_features, coors, batch_size, input_shape, prepared=prepared)
        if fuse_func is not None and fuse_func.fuse_mode == 'pfat':
            x_conv4 = fuse_func(batch_dict, example, encoded_voxel_list=[x_conv2, x_conv3, x_conv4],
                                layer_name='layer1_ori', fuse_mode='pfat', d_factor_list=[2, 4, 8])
        return self._tail(x_conv1, x_conv2, x_conv3, x_conv4)


# =======================================================================================
# TransFusion tree (mmdet3d 0.11 fork)
# =======================================================================================
from .registry import CONV_LAYERS  # noqa: E402

for _n, _c in (("SubMConv3d", spconv.SubMConv3d), ("SparseConv3d", spconv.SparseConv3d),
               ("SubMConv2d", spconv.SubMConv2d), ("SparseConv2d", spconv.SparseConv2d),
               ("SparseInverseConv3d", spconv.SparseInverseConv3d),
               ("SparseConvTranspose3d", spconv.SparseConvTranspose3d), ("SparseConvTranspose2d", spconv.SparseConvTranspose2d)):
    if CONV_LAYERS.get(_n) is None:
        CONV_LAYERS.register_module(_c, name=_n)


def build_conv_layer(cfg, *args, **kwargs):
    """mmcv.cnn.build_conv_layer for the sparse conv types (TF/mmdet3d/ops/spconv/conv.py:207-455 registers
    them into mmcv's CONV_LAYERS)."""
    cfg = dict(cfg)
    t = cfg.pop("type")
    cls = CONV_LAYERS.get(t)
    if cls is None:
        raise KeyError("Unrecognized conv type %s" % t)
    return cls(*args, **kwargs, **cfg)


class TFSparseBasicBlock(spconv.SparseModule):
    """TF/mmdet3d/ops/sparse_block.py:67-120 over mmdet's BasicBlock: conv1/bn1/conv2/bn2 (convs
    bias=False, no indice_key -> the rulebook of every conv is rebuilt; here the occupancy directory is
    shared, so a rebuild is one neighbour-table kernel).  `norm1`/`norm2` alias bn1/bn2 like mmdet."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        conv_cfg = conv_cfg or dict(type='SubMConv3d')
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=1, dilation=1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    @property
    def norm1(self):
        return self.bn1

    @property
    def norm2(self):
        return self.bn2

    def forward(self, x):
        if not self.training and can_fold(self.bn1) and can_fold(self.bn2) and self.downsample is None \
                and not wants_grad(x, self):
            return _fused_basic_block(x, self.conv1, self.bn1, self.conv2, self.bn2, None)
        identity = x.features
        out = self.conv1(x)
        out.features = _ops.batch_norm_rows(self.bn1, out.features, relu=True)      # train(): csrc/bnrows.hip
        out = self.conv2(out)
        out.features = _ops.batch_norm_rows(self.bn2, out.features)
        if self.downsample is not None:
            identity = self.downsample(x)
        out.features = self.relu(out.features + identity)
        return out


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type='SubMConv3d', norm_cfg=None, order=('conv', 'norm', 'act')):
    """TF/mmdet3d/ops/sparse_block.py:123-185."""
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {'conv', 'norm', 'act'} == {'conv', 'norm', 'act'}
    conv_cfg = dict(type=conv_type, indice_key=indice_key)
    layers = []
    for layer in order:
        if layer == 'conv':
            if conv_type not in ['SparseInverseConv3d', 'SparseInverseConv2d', 'SparseInverseConv1d']:
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                               padding=padding, bias=False))
            else:
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, bias=False))
        elif layer == 'norm':
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == 'act':
            layers.append(nn.ReLU(inplace=True))
    return spconv.SparseSequential(*layers)


@MIDDLE_ENCODERS.register_module()
class SparseEncoder(nn.Module):
    """TF/mmdet3d/models/middle_encoders/sparse_encoder.py:11-204 (and the fusion variant :207-448).
    forward(voxel_features, coors, batch_size) -> [B, C*D, H, W]."""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type='conv_module'):
        super().__init__()
        assert block_type in ['conv_module', 'basicblock']
        self.sparse_shape = sparse_shape
        self.in_channels = in_channels
        self.order = order
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.stage_num = len(self.encoder_channels)
        self.fp16_enabled = False
        assert isinstance(order, tuple) and len(order) == 3
        assert set(order) == {'conv', 'norm', 'act'}
        if self.order[0] != 'conv':
            self.conv_input = make_sparse_convmodule(in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key='subm1', conv_type='SubMConv3d', order=('conv',))
        else:
            self.conv_input = make_sparse_convmodule(in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key='subm1', conv_type='SubMConv3d')
        encoder_out_channels = self.make_encoder_layers(make_sparse_convmodule, norm_cfg, self.base_channels,
                                                        block_type=block_type)
        self.conv_out = make_sparse_convmodule(encoder_out_channels, self.output_channels, kernel_size=(3, 1, 1),
                                               stride=(2, 1, 1), norm_cfg=norm_cfg, padding=0,
                                               indice_key='spconv_down2', conv_type='SparseConv3d')

    def make_encoder_layers(self, make_block, norm_cfg, in_channels, block_type='conv_module',
                            conv_cfg=dict(type='SubMConv3d')):
        self.encoder_layers = spconv.SparseSequential()
        for i, blocks in enumerate(self.encoder_channels):
            blocks_list = []
            for j, out_channels in enumerate(tuple(blocks)):
                padding = tuple(self.encoder_paddings[i])[j]
                if i != 0 and j == 0 and block_type == 'conv_module':
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                  padding=padding, indice_key='spconv%d' % (i + 1),
                                                  conv_type='SparseConv3d'))
                elif block_type == 'basicblock':
                    if j == len(blocks) - 1 and i != len(self.encoder_channels) - 1:
                        blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                      padding=padding, indice_key='spconv%d' % (i + 1),
                                                      conv_type='SparseConv3d'))
                    else:
                        blocks_list.append(TFSparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg,
                                                              conv_cfg=conv_cfg))
                else:
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, padding=padding,
                                                  indice_key='subm%d' % (i + 1), conv_type='SubMConv3d'))
                in_channels = out_channels
            self.encoder_layers.add_module('encoder_layer%d' % (i + 1), spconv.SparseSequential(*blocks_list))
        return out_channels

    def _dense_out(self, x):
        out = self.conv_out(x)
        if (not torch.is_grad_enabled() and out.features.is_cuda and out.features.dtype == torch.float32
                and os.environ.get("DF3D_DENSE_ROWS", "1") != "0"):
            # inference: the same map as channels-last pixel rows behind an NCHW VIEW (necks._as_nchw) -- the row-kernel neck
            # that follows reads the rows in place instead of transposing the NCHW volume (0.17 ms per step at bs = 4)
            from .necks import _as_nchw
            D, H, W = [int(v) for v in out.spatial_shape]
            return _as_nchw(out.dense_rows(), None, out.batch_size, H, W)
        spatial_features = out.dense()
        N, C, D, H, W = spatial_features.shape
        return spatial_features.view(N, C * D, H, W)

    # ---- native executor (dualfusion/executor.py): the conv chain as one call per uninterrupted segment
    def _stage_list(self):
        return [("conv_input", self.conv_input)] + list(self.encoder_layers._modules.items())

    def _runner(self, cuts=()):
        if self.training or torch.is_grad_enabled() or os.environ.get("DF3D_EXECUTOR", "1") != "1":
            return None
        key = tuple(cuts)
        cache = self.__dict__.setdefault("_exec_runners", {})
        if key not in cache:
            from .executor import build_runner
            cache[key] = build_runner(self._stage_list(), cuts=cuts, geometry_module=self.conv_out)
        return cache[key]

    def train(self, mode=True):
        self.__dict__.pop("_exec_runners", None)
        return super(SparseEncoder, self).train(mode)

    def forward(self, voxel_features, coors, batch_size):
        coors = coors.int()
        x = spconv.SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size)
        runner = self._runner() if voxel_features.is_cuda and voxel_features.shape[0] > 0 else None
        if runner is not None:
            outs = runner.run(x)
            return self._dense_out(outs[self._stage_list()[-1][0]])
        x = self.conv_input(x)
        for encoder_layer in self.encoder_layers._modules.values():
            x = encoder_layer(x)
        return self._dense_out(x)


@MIDDLE_ENCODERS.register_module()
class SparseEncoderFusion(SparseEncoder):
    """sparse_encoder.py:207-448: SparseEncoder + a fusion layer applied to the output of the stages
    listed in `fusion_pos` (voxel centres via coor2pts, :309-319)."""

    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type='conv_module',
                 fusion_layer=None, fusion_pos=None, voxel_size=None, point_cloud_range=None, ret_img_map=False):
        super().__init__(in_channels, sparse_shape, order, norm_cfg, base_channels, output_channels, encoder_channels,
                         encoder_paddings, block_type)
        self.fusion_layer = None
        self.fusion_pos = None
        self.ret_img_map = ret_img_map
        if fusion_layer is not None:
            from .registry import FUSION_LAYERS, build_from_cfg
            from . import fusion_tf  # noqa: F401  (registers 'ACTR' into FUSION_LAYERS)
            self.fusion_layer = fusion_layer if isinstance(fusion_layer, nn.Module) else \
                build_from_cfg(fusion_layer, FUSION_LAYERS)
            self.fusion_pos = fusion_pos
            self.voxel_size = voxel_size
            self.point_cloud_range = point_cloud_range

    def coor2pts(self, x, pad=0.0):
        """:309-319: voxel index (+pad) -> metric (x, y, z); ratio taken from the y dimension."""
        ratio = self.sparse_shape[1] / x.spatial_shape[1]
        scale = _ops.device_constant([float(v) for v in (list(self.voxel_size) + [1])[::-1]], torch.float32, x.indices.device)
        pts = (x.indices.to(torch.float) + pad) * scale * ratio
        pts[:, 0] = pts[:, 0] / ratio - pad
        pts[:, 1:] += _ops.device_constant([float(v) for v in self.point_cloud_range[:3][::-1]], torch.float32, pts.device)
        pts[:, 1:] = pts[:, 1:].flip(1)            # (z, y, x) -> (x, y, z); a Python index list would be a host -> device copy
        return pts            # [N, 4] (b, x, y, z); rows are batch-sorted

    # ---- frame head a frame ahead (dualfusion/prefetch.py): voxelisation + the rulebooks of the first executor segment ----
    def prefetch(self, points_list, voxel_size, point_cloud_range, max_points, max_voxels):
        """Start voxelisation (mmdet3d's hard voxelisation: `break` at the cap) + mean VFE and every rulebook of the encoder for
        these point clouds on the module's native worker thread; returns at once.  `take_head(points_list)` later hands back
        (features, coors, prepared) for `forward(features, coors, B, ..., prepared=prepared)`.  The clouds must be complete in
        device memory and be passed as the same tensors."""
        if self.training or not points_list or not points_list[0].is_cuda:
            return False
        with torch.no_grad():
            cuts = [p + 1 for p in (self.fusion_pos or [])]
            runner = self._runner(cuts)
        if runner is None:
            return False
        head = self.__dict__.get("_head_worker")
        if head is None:
            from .prefetch import FrameHead
            head = self.__dict__["_head_worker"] = FrameHead(points_list[0].device)
        if len(head._pending) >= 2:
            head.drop_all()
        cap = max_voxels if isinstance(max_voxels, int) else max_voxels[1]
        vox = dict(voxel_size=voxel_size, coors_range=point_cloud_range, max_points=max_points, max_voxels=cap, break_at_cap=True)
        head.submit(tuple(int(p.data_ptr()) for p in points_list), runner.segments[0][2], points_list, vox,
                    [int(v) for v in self.sparse_shape], None)
        return True

    def take_head(self, points_list):
        head = self.__dict__.get("_head_worker")
        prep = head.take(tuple(int(p.data_ptr()) for p in points_list)) if head is not None else None
        if prep is None or prep.geometry is None:
            return None
        prep.hand_over()
        return prep.feats, prep.coors, prep.geometry

    def close(self):
        head = self.__dict__.pop("_head_worker", None)
        if head is not None:
            head.close()

    def forward(self, voxel_features, coors, batch_size, img_feats=None, img_metas=None, points=None,
                ret_lidar_features=False, img=None, prepared=None):
        coors = coors.int()
        x0 = spconv.SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size)
        encode_features, lidar_features = [], []

        def fuse(idx, x):
            if ret_lidar_features:
                lidar_features.append(x)
            if self.fusion_pos is not None and idx in self.fusion_pos:
                c_pts = self.coor2pts(x, 0.5)
                x = x.replace_feature(self.fusion_layer(img_feats, c_pts, x.features, img_metas, img))
            encode_features.append(x)
            return x

        # the fusion layer modifies the features after encoder layer idx = stage idx + 1: cut the chain there
        cuts = [p + 1 for p in (self.fusion_pos or [])]
        runner = self._runner(cuts) if voxel_features.is_cuda and voxel_features.shape[0] > 0 else None
        if runner is not None:
            runner.run(x0, hook=lambda i, name, t: t if i == 0 else fuse(i - 1, t), prepared=prepared)
        else:
            x = self.conv_input(x0)
            for idx, encoder_layer in enumerate(self.encoder_layers._modules.values()):
                x = fuse(idx, encoder_layer(x))
        spatial_features = self._dense_out(encode_features[-1])
        if ret_lidar_features:
            return (spatial_features, encode_features[-1], img_feats)
        return spatial_features


# =======================================================================================
# Voxel-RCNN tree (OpenPCDet 0.5.2 fork)
# =======================================================================================
from functools import partial  # noqa: E402


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type="subm",
                   norm_fn=None):
    """VR/pcdet/models/backbones_3d/spconv_backbone.py:33-75."""
    if conv_type == "subm":
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == "spconv":
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class VoxelBackBone8x(nn.Module):
    """VR/pcdet/models/backbones_3d/spconv_backbone.py:135-243: batch_dict in / out."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + [1, 0, 0])]
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key="subm1"), norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key="subm1"))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2", conv_type="spconv"),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key="subm2"),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key="subm2"))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv3", conv_type="spconv"),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key="subm3"),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key="subm3"))
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key="spconv4", conv_type="spconv"),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key="subm4"),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key="subm4"))
        last_pad = model_cfg.get("last_pad", 0) if hasattr(model_cfg, "get") else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key="spconv_down2"), norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 64}

    def _runner(self):
        """Native executor for the conv chain, cut after conv1 when a subclass fuses there (dualfusion/executor.py)."""
        if self.training or torch.is_grad_enabled() or os.environ.get("DF3D_EXECUTOR", "1") != "1":
            return None
        if "_exec_runner" not in self.__dict__:
            from .executor import build_runner
            stages = [("conv_input", self.conv_input), ("conv1", self.conv1), ("conv2", self.conv2),
                      ("conv3", self.conv3), ("conv4", self.conv4)]
            cuts = [1] if type(self)._fuse1 is not VoxelBackBone8x._fuse1 else []
            self.__dict__["_exec_runner"] = build_runner(stages, cuts=cuts, geometry_module=self.conv_out)
        return self.__dict__["_exec_runner"]

    def train(self, mode=True):
        self.__dict__.pop("_exec_runner", None)
        return super(VoxelBackBone8x, self).train(mode)

    def _fuse1(self, x_conv1, batch_dict):
        return x_conv1

    def _prefetch_fuse4(self, coords, batch_dict, ready):
        return None

    def _throttle(self, depth=2):
        """Back-pressure for callers that run ahead of the GPU (side-stream work that no longer waits for the previous frame):
        at most `depth` frames in flight -- the side stream's buffers of frame k are released when frame k + 2 starts."""
        q = self.__dict__.setdefault("_inflight", [])
        if len(q) >= depth:
            q.pop(0).synchronize()

    def _frame_done(self):
        q = self.__dict__.setdefault("_inflight", [])
        q.append(torch.cuda.current_stream().record_event())

    def _fuse4(self, x_conv2, x_conv3, x_conv4, batch_dict):
        return x_conv4

    def forward(self, batch_dict):
        voxel_features, voxel_coords = batch_dict["voxel_features"], batch_dict["voxel_coords"]
        batch_size = batch_dict["batch_size"]
        x0 = spconv.SparseConvTensor(voxel_features, voxel_coords.int(), self.sparse_shape, batch_size)
        runner = self._runner() if voxel_features.is_cuda and voxel_features.shape[0] > 0 else None
        if runner is not None:
            # (measured: starting the side stream's work here, before the first segment is queued, beats starting it after
            # conv1 -- 19.1 vs 19.6 ms per step: furthest point sampling is the longest chain of the step, every
            # microsecond it starts earlier counts; without it 20.6 ms)
            # ... and it waits for the COORDINATES only: the voxeliser's own event when it ran on the voxel stream
            # (ops.hard_voxelize_clouds), else everything queued on this stream so far
            ready = getattr(voxel_coords, "_df3d_ready", None) or torch.cuda.current_stream().record_event()
            self._throttle()
            ahead = self.__dict__.get("_fuse4_ahead", [])
            hit = next((e for e in ahead if e[0] is voxel_coords), None)
            if hit is not None:
                # prepared a batch ahead (`prefetch`); the tensors of the batch before stay referenced one frame longer
                ahead.remove(hit)
                self.__dict__["_fuse4_pre_old"] = self.__dict__.pop("_fuse4_pre", None)
                self.__dict__["_fuse4_pre"] = hit[1]
            else:
                self._prefetch_fuse4(x0.indices, batch_dict, ready)
            keep = {}

            def hook(i, name, t):
                if name == "conv1":
                    t = self._fuse1(t, batch_dict)
                elif name == "conv4":
                    t = self._fuse4(keep["conv2"], keep["conv3"], t, batch_dict)
                keep[name] = t
                return t

            runner.run(x0, hook=hook, prepared=batch_dict.get("prepared"))
            x_conv1, x_conv2, x_conv3, x_conv4 = keep["conv1"], keep["conv2"], keep["conv3"], keep["conv4"]
        else:
            x = self.conv_input(x0)
            x_conv1 = self._fuse1(self.conv1(x), batch_dict)
            x_conv2 = self.conv2(x_conv1)
            x_conv3 = self.conv3(x_conv2)
            x_conv4 = self._fuse4(x_conv2, x_conv3, self.conv4(x_conv3), batch_dict)
        out = self.conv_out(x_conv4)
        if runner is not None:
            self._frame_done()
        batch_dict.update({"encoded_spconv_tensor": out, "encoded_spconv_tensor_stride": 8})
        batch_dict.update({"multi_scale_3d_features": {"x_conv1": x_conv1, "x_conv2": x_conv2, "x_conv3": x_conv3,
                                                       "x_conv4": x_conv4}})
        batch_dict.update({"multi_scale_3d_strides": {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}})
        return batch_dict


class KittiCalibration(object):
    """The three matrices of a KITTI calibration file and the devkit's `lidar_to_img` (VR/pcdet/utils/calibration_kitti.py:
    25-93): P2 [3, 4], R0 [3, 3], Tr_velo2cam [3, 4], kept in the dtype they were read in (the devkit reads float32).
    `batch_dict['calib']` of the reference holds one such object per sample; `VoxelBackBone8xFusion` accepts them (or the
    reference's own objects: anything with P2 / R0 / V2C attributes) when no composed `lidar2img` is given and then projects
    the way the reference does -- on the host, in numpy, with the devkit's three separate products and their float32
    roundings -- so that every voxel lands in the pixel the reference puts it in (the composed fp32 matrix on the device
    agrees to ~1e-4 px, which moves a voxel across a pixel boundary about once in 10^4 voxels)."""

    def __init__(self, P2, R0, V2C):
        import numpy as np
        self.P2, self.R0, self.V2C = np.asarray(P2), np.asarray(R0), np.asarray(V2C)

    @staticmethod
    def of(obj):
        if isinstance(obj, KittiCalibration):
            return obj
        if isinstance(obj, dict):
            return KittiCalibration(obj['P2'], obj['R0'], obj['Tr_velo2cam'] if 'Tr_velo2cam' in obj else obj['V2C'])
        return KittiCalibration(obj.P2, obj.R0, obj.V2C)

    def lidar_to_img(self, pts_lidar):
        """[n, 3] LiDAR points -> ([n, 2] pixels, [n] depth in the rectified camera): homogeneous points times
        (V2C^T R0^T), then times P2^T, pixel numerators over the RECTIFIED depth (not over the homogeneous coordinate)."""
        import numpy as np
        one = np.ones((pts_lidar.shape[0], 1), dtype=np.float32)
        rect = np.dot(np.hstack((pts_lidar, one)), np.dot(self.V2C.T, self.R0.T))
        rect_h = np.hstack((rect, one))
        img_h = np.dot(rect_h, self.P2.T)
        return (img_h[:, 0:2].T / rect_h[:, 2]).T, img_h[:, 2] - self.P2.T[3, 2]


class BasicGate(nn.Module):
    """VR/pcdet/models/model_utils/attention.py:88-177 -- the image gate of the Voxel-RCNN tree (`I_FUSION_METHOD: BasicGate`,
    called at spconv_backbone.py:797-800 right before ACTR): per image level s, the voxels of `x_list[s]` are projected into the
    camera, their feature rows scattered onto the level's feature map (`pts2img`, attention.py:945-967: pixels clamped to the
    border, the last row written wins, a (H + 1) x (W + 1) canvas cropped back), two 3 x 3 convolutions reduce the canvas to
    one map, and the image features are multiplied by its sigmoid.
    Kept bug for bug: the convolution stacks live in a plain Python list (`spatial_basic_list`), so they are NOT registered --
    no `state_dict` keys, never seen by an optimizer, moved to the device inside `forward` -- exactly as in the reference.
    Device formulation: the projection is the backbone's (`lidar2img` on the device instead of the numpy `Calibration`
    object); the scatter resolves "last writer" with an arg-max over row indices (deterministic; the reference's `index_put_`
    is sequential on the CPU and racy on a GPU)."""

    def __init__(self, img_channel_list, pts_channel_list, sparse_shape, voxel_size, point_cloud_range, inv_idx, pts_idx,
                 num_conv=2):
        super(BasicGate, self).__init__()
        self.voxel_size, self.point_cloud_range, self.inv_idx = voxel_size, point_cloud_range, inv_idx
        self.g_channel_list = list(pts_channel_list)
        self.sparse_shape = sparse_shape
        self.pts_idx = pts_idx
        self.spatial_basic_list = []
        for c in self.g_channel_list:
            mods = []
            for _ in range(num_conv - 1):
                mods += [nn.Conv2d(c, c, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(c, eps=1e-3, momentum=0.01), nn.ReLU()]
            mods.append(nn.Conv2d(c, 1, kernel_size=3, stride=1, padding=1))
            self.spatial_basic_list.append(nn.Sequential(*mods))
        self.sigmoid = nn.Sigmoid()

    def canvas(self, x, uv, image_hw, feat_hw, batch_size):
        """pts2img of a whole batch: uv [n, 2] float pixels (x, y) in the image -> [B, C, Hf, Wf]."""
        h, w = image_hw
        Hf, Wf = feat_hw
        norm = (uv / uv.new_tensor([float(w), float(h)])).clamp(min=0.0, max=1.0)
        iy = (norm[:, 1] * Hf).long()
        ix = (norm[:, 0] * Wf).long()
        b = x.indices[:, 0].long()
        lin = (b * (Hf + 1) + iy) * (Wf + 1) + ix
        rows = torch.arange(lin.shape[0], device=lin.device)
        win = torch.full((batch_size * (Hf + 1) * (Wf + 1),), -1, dtype=torch.long, device=lin.device)
        win.scatter_reduce_(0, lin, rows, "amax", include_self=True)
        feats = x.features
        out = feats.new_zeros((win.shape[0], feats.shape[1]))
        hit = win >= 0
        out[hit] = feats[win[hit]]
        return out.view(batch_size, Hf + 1, Wf + 1, -1)[:, :-1, :-1].permute(0, 3, 1, 2).contiguous()

    def forward(self, x_rgb, x_list, batch_dict, project=None):
        """project(x, voxel_stride) -> (xyz, uv): the backbone's projection of a sparse tensor's voxel corners (with the
        recorded augmentations undone) -- `VoxelBackBone8xFusion._project`."""
        if project is None:
            raise ValueError("BasicGate.forward needs the backbone's projection (project=)")
        B = batch_dict["batch_size"]
        image_hw = tuple(batch_dict["images"].shape[2:]) if "images" in batch_dict else tuple(batch_dict["image_hw"])
        out = []
        for s, f in enumerate(x_rgb):
            x = x_list[s]
            ratio = self.sparse_shape[1] / x.spatial_shape[1]
            _, uv = project(x, ratio)
            pts_img = self.canvas(x, uv, image_hw, tuple(f.shape[2:]), B)
            pts = self.spatial_basic_list[s].to(device=f.device)(pts_img)
            out.append(f * torch.sigmoid(pts))
        return out


class VoxelBackBone8xFusion(VoxelBackBone8x):
    """VR/pcdet/models/backbones_3d/spconv_backbone.py:436-928, camera branch reduced to the hot path:
    MVX nearest-pixel sum at stride 1 (:733-756, FUSION_POS has 1) and ACTR(v2) dual-query fusion at
    stride 8 (:760-814, FUSION_POS has 4), one camera.

    Differences from the reference call protocol (documented in DESIGN.md): the 2-D network (`semseg`,
    frozen DeepLabV3) is out of scope, so its outputs are read from `batch_dict['img_dict']`
    ({'mvx_layer1_feat2d' | 'layer1_feat2d': [B,C,h,w]}); the KITTI calibration objects, whose
    `lidar_to_img` runs in numpy on the CPU (:717-718), are replaced by `batch_dict['lidar2img']`
    [B,3,4] on the device (`lidar2img_from_kitti` composes it from P2 / R0 / Tr the way the devkit projects); a batch_dict that
    carries the reference's `calib` objects and no `lidar2img` is projected on the host exactly as the reference does
    (`KittiCalibration`: slower, pixel for pixel the reference's).  `I_FUSION_METHOD:
    BasicGate` (the image gate in front of ACTR, `BasicGate` above) is mirrored since round 4; the shipped `_ifat` yaml cannot be
    constructed by the reference itself (no `pts_idx`) -- the default [0, 2] of the PV-RCNN yaml is taken."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__(model_cfg, input_channels, grid_size, **kwargs)
        get = model_cfg.get
        self.fusion_pos = get("FUSION_POS", [1])
        self.fusion_method = get("FUSION_METHOD", "MVX")
        self.feature_levels = get("FEATURE_LEVELS", [0])
        self.register_buffer("voxel_size", torch.tensor([0.1, 0.05, 0.05]), persistent=False)          # z, y, x
        self.register_buffer("point_cloud_range", torch.tensor([-3., -40., 0., 1., 40., 70.4]), persistent=False)
        self.img_out_channel = 16 if 1 in self.fusion_pos else 64
        self.ifat = None
        method = get("I_FUSION_METHOD", False)
        if method:
            if method != "BasicGate":
                raise NotImplementedError("I_FUSION_METHOD %r: the Voxel-RCNN 3D-DF config names BasicGate" % (method,))
            cfg = get("IFAT_CFG", None) or {}
            lv0 = self.feature_levels[0]
            # (spconv_backbone.py:546-560; the shipped yaml has no `pts_idx`, which the reference's constructor requires)
            self.ifat = BasicGate(img_channel_list=cfg["img_num_channels"][lv0:lv0 + len(self.feature_levels)],
                                  pts_channel_list=cfg["pts_num_channels"], sparse_shape=self.sparse_shape,
                                  voxel_size=self.voxel_size, point_cloud_range=self.point_cloud_range, inv_idx=[2, 1, 0],
                                  pts_idx=cfg.get("pts_idx", [0, 2]))
        if "ACTR" in self.fusion_method:
            from .actr import build as build_actr
            model_name = self.fusion_method if "MVX+" not in self.fusion_method else self.fusion_method[4:]
            actr_cfg = get("ACTR_CFG", None)
            assert actr_cfg is not None
            hybrid = dict(get("HYBRID_CFG", None) or {})
            hybrid.setdefault("gate_before_ffn", True)      # this tree's layer order (VR actr_transformer.py:503-512)
            self.actr = build_actr(actr_cfg, model_name=model_name, lt_cfg=get("LT_CFG", None), hybrid_cfg=hybrid)
            self.max_num_nev = actr_cfg.get("max_num_ne_voxel", 26000)

    @staticmethod
    def lidar2img_from_kitti(P2, R0, V2C):
        """[3, 4] matrix for `batch_dict['lidar2img']` from a KITTI calibration (P2 [3,4], R0 [3,3], Tr_velo2cam [3,4]):
        rows 0 and 1 of P2 @ R0 @ Tr, row 2 = the depth row of R0 @ Tr.  The reference's `Calibration.lidar_to_img`
        divides the pixel numerators by the rectified-camera DEPTH, not by the homogeneous coordinate of the P2 product
        (VR/pcdet/utils/calibration_kitti.py:65-93), and P2[2, 3] is not zero in the devkit files."""
        import numpy as np
        R0e, Ve = np.eye(4), np.eye(4)
        R0e[:3, :3], Ve[:3, :4] = np.asarray(R0, np.float64), np.asarray(V2C, np.float64)
        M = np.asarray(P2, np.float64) @ R0e @ Ve
        M[2] = (R0e @ Ve)[2]
        return M.astype(np.float32)

    # ------------------------------------------------------------------ geometry (device-side)
    def _voxel_xyz(self, ind, voxel_stride, batch_dict):
        """voxel corner (b,z,y,x) -> LiDAR xyz of the point cloud the camera saw."""
        v3d = ind[:, 1:].float() * voxel_stride * self.voxel_size + self.point_cloud_range[:3]      # (z,y,x)
        xyz = v3d.flip(1)                            # (z, y, x) -> (x, y, z) without a host index list
        bi = ind[:, 0].long()
        # the point cloud the camera saw: undo the recorded augmentations in the reference's order (:701-714) --
        # global scale, rotation about z by -noise_rot (rotate_points_along_z, VR/pcdet/utils/common_utils.py:35-57),
        # flip about the x axis (y -> -y), flip about the y axis (x -> -x)
        if "noise_scale" in batch_dict:
            xyz = xyz / torch.as_tensor(batch_dict["noise_scale"], dtype=xyz.dtype, device=xyz.device)[bi][:, None]
        if "noise_rot" in batch_dict:
            ang = -torch.as_tensor(batch_dict["noise_rot"], dtype=xyz.dtype, device=xyz.device)[bi]
            ca, sa = torch.cos(ang), torch.sin(ang)
            xyz = torch.stack([xyz[:, 0] * ca - xyz[:, 1] * sa, xyz[:, 0] * sa + xyz[:, 1] * ca, xyz[:, 2]], 1)
        if "flip_x" in batch_dict:
            sgn = 1.0 - 2.0 * torch.as_tensor(batch_dict["flip_x"], device=xyz.device)[bi].to(xyz.dtype)
            xyz = torch.stack([xyz[:, 0], xyz[:, 1] * sgn, xyz[:, 2]], 1)
        if "flip_y" in batch_dict:
            sgn = 1.0 - 2.0 * torch.as_tensor(batch_dict["flip_y"], device=xyz.device)[bi].to(xyz.dtype)
            xyz = torch.stack([xyz[:, 0] * sgn, xyz[:, 1], xyz[:, 2]], 1)
        return xyz, bi

    @staticmethod
    def _pixels(xyz, bi, batch_dict):
        """LiDAR xyz -> image pixel (float) through lidar2img [B,3,4]; without it, through the KITTI calibration objects of
        `batch_dict['calib']` on the host, as the reference does (`KittiCalibration`)."""
        if "lidar2img" not in batch_dict:
            import numpy as np
            calibs = [KittiCalibration.of(c) for c in batch_dict["calib"]]
            host, bh = xyz.detach().cpu().numpy(), bi.cpu().numpy()           # (the reference's own host round trip, :717-718)
            uv = np.zeros((host.shape[0], 2), np.float64)
            for b, cal in enumerate(calibs):
                sel = bh == b
                if sel.any():
                    uv[sel] = cal.lidar_to_img(host[sel])[0]
            return torch.from_numpy(uv.astype(np.float32)).to(xyz.device)      # torch.Tensor(voxels_2d): fp32 from here on
        P = batch_dict["lidar2img"].float()[bi]                                                      # [N,3,4]
        # broadcast multiply-adds: einsum lowers to a batched GEMM over N tiny 3x4 matrices (2.4 ms per call at 200k voxels)
        h = P[:, :, 0] * xyz[:, 0:1] + P[:, :, 1] * xyz[:, 1:2] + P[:, :, 2] * xyz[:, 2:3] + P[:, :, 3]
        return h[:, :2] / h[:, 2:3]

    def _project(self, x, voxel_stride, batch_dict):
        """voxel corner (z,y,x) -> LiDAR xyz -> image pixel (float) through lidar2img [B,3,4]."""
        xyz, bi = self._voxel_xyz(x.indices, voxel_stride, batch_dict)
        return xyz, self._pixels(xyz, bi, batch_dict)

    @staticmethod
    def _query_slots(ind, B):
        """Rows (batch-sorted) -> (sample, slot inside the sample's padded query list, longest list).  One host read."""
        b = ind[:, 0].long()
        counts = torch.bincount(b, minlength=B)
        # host sync (the reference pads to max_num_nev then slices); the range flag of the fp16 operand format rides on it
        n_max = int(_ops.read_with_range_flag(counts.max().view(1))[0])
        starts = torch.cumsum(counts, 0) - counts
        slot = torch.arange(ind.shape[0], device=ind.device) - starts[b]       # rows are batch-sorted
        return b, slot, n_max

    # ------------------------------------------------------------------ stride-8 query geometry ahead of the convolutions
    def _prefetch_fuse4(self, coords, batch_dict, ready):
        """The coordinates of conv4's voxels follow from the input coordinates alone (three strided index sets), and so do
        the ACTRv2 queries' LiDAR positions and everything the 3-D local self-attention derives from them: furthest point
        sampling (2048 SERIAL iterations on one workgroup per sample: 4.7 ms on 8 of the 256 CUs), ball query, grouped
        coordinates, the 'unique' winners.  They are computed HERE, on a side stream that waits for the input coordinates
        only (`ready`), and run beside the backbone's convolutions; `_fuse4` waits for the event.  The reference computes them after conv4
        and again in every encoder layer (VR actr_transformer.py:482-486)."""
        # the side stream's tensors of frame k - 1 may still be read by the main stream (this chain no longer waits for it):
        # they are released one frame later, when `_throttle` has seen frame k - 1 complete
        self.__dict__["_fuse4_pre_old"] = self.__dict__.pop("_fuse4_pre", None)
        pre = self._fuse4_geometry(coords, batch_dict, ready)
        if pre is not None:
            self.__dict__["_fuse4_pre"] = pre

    # ---- frame head a batch ahead (dualfusion/prefetch.py): voxelisation + EVERY rulebook of the chain on the native worker ----
    def prefetch_frame(self, key, points_list, voxel_size, point_cloud_range, max_points, max_voxels):
        """Round 5: voxelisation (+ mean VFE) of these point clouds and every rulebook of the conv chain -- the whole chain's, on
        both sides of the fusion layers: geometry depends on the coordinates alone -- on the module's native worker thread,
        with all their count round trips; returns at once.  `take_head(key)` hands back (features, coors, prepared) for
        `forward(dict(voxel_features=, voxel_coords=, prepared=, ...))`.  The clouds must be complete in device memory."""
        if self.training or not points_list or not points_list[0].is_cuda:
            return False
        with torch.no_grad():
            runner = self._runner()
        if runner is None or runner.full is None:
            return False
        head = self.__dict__.get("_head_worker")
        if head is None:
            from .prefetch import FrameHead
            head = self.__dict__["_head_worker"] = FrameHead(points_list[0].device)
        if len(head._pending) >= 3:
            head.drop_all()
        cap = max_voxels if isinstance(max_voxels, int) else max_voxels[1]
        vox = dict(voxel_size=voxel_size, coors_range=point_cloud_range, max_points=max_points, max_voxels=cap, break_at_cap=True)
        head.submit(key, runner.full, points_list, vox, [int(v) for v in self.sparse_shape], None)
        return True

    def take_head(self, key):
        head = self.__dict__.get("_head_worker")
        prep = head.take(key) if head is not None else None
        if prep is None or prep.geometry is None:
            return None
        return prep

    def close(self):
        head = self.__dict__.pop("_head_worker", None)
        if head is not None:
            head.close()

    def prefetch(self, voxel_coords, batch_dict, prepared=None):
        """Round 5: the stride-8 query geometry of a LATER forward(batch_dict with these `voxel_coords`) started NOW -- a batch
        ahead, beside the convolutions of the batch in front of it.  Furthest point sampling is 2048 serial iterations on
        one workgroup per sample (5.2 ms on 8 of the 256 CUs); started inside its own forward it is the step's critical chain
        (FPS -> ball query -> ACTRv2 fusion: 8.9 ms per step against 3.5 ms of backbone); a batch ahead it runs under a whole
        step.  It needs the batch's voxel coordinates and calibration only -- what the reference's DataLoader workers hold
        ready before the GPU step.  Returns False when not applicable (forward() then computes it in line)."""
        if not voxel_coords.is_cuda or torch.is_grad_enabled() or self.training:
            return False
        ready = getattr(voxel_coords, "_df3d_ready", None) or torch.cuda.current_stream().record_event()
        pre = self._fuse4_geometry(voxel_coords, batch_dict, ready, prepared=prepared)
        if pre is None:
            return False
        # (a short list: the batch in front of this one has not consumed its entry yet)
        ahead = self.__dict__.setdefault("_fuse4_ahead", [])
        del ahead[:-1]
        ahead.append((voxel_coords, pre))
        return True

    def _fuse4_geometry(self, coords, batch_dict, ready, prepared=None):
        """`prepared`: the frame head's PreparedGeometry of the whole chain (`take_head`): conv4's index set is in it already."""
        previous = None
        if (4 not in self.fusion_pos or "ACTR" not in self.fusion_method or torch.is_grad_enabled()
                or os.environ.get("DF3D_VR_PREFETCH", "1") != "1" or not coords.is_cuda or coords.shape[0] == 0
                or "lidar2img" not in batch_dict):
            return None
        lts = getattr(getattr(self.actr.transformer, "encoder", None), "lidar_attns", None)
        if lts is None or len(lts) == 0:
            return None
        from .spconv.conv import SparseConvolution
        from .spconv.ops import get_conv_output_size
        # two side streams in turn: the chain of batch k + 1 (whose query count the host waits for) must not queue behind
        # the furthest point sampling of batch k
        sides = self.__dict__.get("_geo_streams")
        if sides is None:
            sides = self.__dict__["_geo_streams"] = [torch.cuda.Stream(device=coords.device) for _ in range(2)]
            self.__dict__["_geo_turn"] = 0
        self.__dict__["_geo_turn"] ^= 1
        side = sides[self.__dict__["_geo_turn"]]
        if prepared is None:
            side.wait_event(ready)
        # the previous frame's tensors stay referenced until here: their readers on the main stream were queued before `ready`
        B = batch_dict["batch_size"]
        with torch.cuda.stream(side):
            ind, shape = coords.int().contiguous(), list(self.sparse_shape)
            if prepared is not None and prepared.stages is not None and "conv4" in prepared.stages:
                # the worker built it with the rest of the chain's geometry: the side stream waits (on the device) for that
                prepared.wait()
                ind, shape = prepared.stages["conv4"].indices, list(prepared.stages["conv4"].spatial_shape)
            else:
              for stage in (self.conv_input, self.conv1, self.conv2, self.conv3, self.conv4):
                for m in stage.modules():
                    if isinstance(m, SparseConvolution) and not m.subm:
                        out_shape = get_conv_output_size(shape, m.kernel_size, m.stride, m.padding, m.dilation)
                        ind, _ = _ops.conv_out_indices(ind, B, shape, out_shape, m.kernel_size, m.stride, m.padding, m.dilation)
                        shape = list(out_shape)
            xyz, _ = self._voxel_xyz(ind, 8, batch_dict)
            b, slot, n_max = self._query_slots(ind, B)
            pts = xyz.new_zeros((B, n_max, 3))
            pts[b, slot] = xyz
            lt = lts[0]
            lt._row_plan(pts, *lt._geometry(pts))
            ev = torch.cuda.Event(enable_timing=True)       # (timing: tools/debug/vr_stall.py reads how long `_fuse4` waits for it)
            ev.record(side)
        del previous
        return dict(n=int(ind.shape[0]), shape=shape, xyz=xyz, b=b, slot=slot, n_max=n_max, pts=pts, event=ev)


    @staticmethod
    def _bilinear_taps(dst, in_size, out_size):
        """Source taps of torch's bilinear upsample (align_corners=False) for integer destination indices:
        src = max(in / out * (dst + 0.5) - 0.5, 0) in fp32, taps floor(src) and its right neighbour (if any)."""
        scale = np.float32(in_size) / np.float32(out_size)
        src = ((dst.to(torch.float32) + 0.5) * float(scale) - 0.5).clamp_(min=0)
        i0 = src.floor().long().clamp_(max=in_size - 1)
        i1 = i0 + (i0 < in_size - 1).long()
        lam = src - i0.to(torch.float32)
        return i0, i1, lam

    def _sample_int(self, fmap, b, uv, hw):
        """bilinear upsample to the image size, then nearest-integer (truncated) pixel gather (:682-731) -- evaluated at the
        gathered pixels only: the four taps and weights torch's `interpolate(mode='bilinear')` uses for that pixel
        (materialising the upsampled [B, C, 375, 1242] volume took 0.9 ms per call for ~10^5 gathered pixels)."""
        h, w = hw
        px = uv.long()                                      # torch.Tensor(voxels_2d).long(): truncation
        ok = (px[:, 1] >= 0) & (px[:, 1] < h) & (px[:, 0] >= 0) & (px[:, 0] < w)
        y0, y1, ly = self._bilinear_taps(px[:, 1].clamp(0, h - 1), fmap.shape[2], h)
        x0, x1, lx = self._bilinear_taps(px[:, 0].clamp(0, w - 1), fmap.shape[3], w)
        # channels-last copy of the (small) feature map: a tap is then one contiguous row gather
        f = fmap.permute(0, 2, 3, 1).contiguous().view(-1, fmap.shape[1])
        Hin, Win = fmap.shape[2], fmap.shape[3]
        base = b * (Hin * Win)
        tap = lambda yy, xx: f.index_select(0, base + yy * Win + xx)
        ly, lx = ly[:, None].to(fmap.dtype), lx[:, None].to(fmap.dtype)
        top = (1 - lx) * tap(y0, x0) + lx * tap(y0, x1)
        bot = (1 - lx) * tap(y1, x0) + lx * tap(y1, x1)
        feat = (1 - ly) * top + ly * bot
        return torch.where(ok[:, None], feat, torch.zeros_like(feat))

    # ------------------------------------------------------------------ the sampling as one kernel (csrc/mvx.hip)
    def _geometry_floats(self):
        hit = self.__dict__.get("_geo_floats")
        if hit is None:                                  # one device read, at the first call
            hit = self.__dict__["_geo_floats"] = (self.voxel_size.tolist(), self.point_cloud_range[:3].tolist())
        return hit

    @staticmethod
    def _aug_params(batch_dict, B, dev):
        """[B, 5] = global scale, cos(-rot), sin(-rot), flip_x sign, flip_y sign of every sample (identity where a key is
        absent), computed with the same torch operations as `_voxel_xyz`; kept in the batch_dict for the second call."""
        hit = batch_dict.get("_df3d_aug")
        if hit is not None and hit.device == dev:
            return hit
        one = torch.ones(B, dtype=torch.float32, device=dev)
        cols = [one, one, torch.zeros_like(one), one, one]
        if "noise_scale" in batch_dict:
            cols[0] = torch.as_tensor(batch_dict["noise_scale"], dtype=torch.float32, device=dev).reshape(B)
        if "noise_rot" in batch_dict:
            ang = -torch.as_tensor(batch_dict["noise_rot"], dtype=torch.float32, device=dev).reshape(B)
            cols[1], cols[2] = torch.cos(ang), torch.sin(ang)
        if "flip_x" in batch_dict:
            cols[3] = 1.0 - 2.0 * torch.as_tensor(batch_dict["flip_x"], device=dev).to(torch.float32).reshape(B)
        if "flip_y" in batch_dict:
            cols[4] = 1.0 - 2.0 * torch.as_tensor(batch_dict["flip_y"], device=dev).to(torch.float32).reshape(B)
        aug = torch.stack(cols, 1).contiguous()
        batch_dict["_df3d_aug"] = aug
        return aug

    def _sample_native(self, x, batch_dict, fmap):
        return (x.features.is_cuda and not torch.is_grad_enabled() and x.features.dtype == torch.float32
                and "lidar2img" in batch_dict               # (the sampling kernel projects through the composed matrix)
                and fmap.dtype == torch.float32 and fmap.is_contiguous() and fmap.shape[1] % 4 == 0
                and x.indices.dtype == torch.int32 and os.environ.get("DF3D_VR_SAMPLE", "1") == "1")

    def _fuse1(self, x_conv1, batch_dict):
        if 1 not in self.fusion_pos:
            return x_conv1
        img_dict = batch_dict["img_dict"]
        fmap = img_dict["mvx_layer1_feat2d"] if "mvx_layer1_feat2d" in img_dict else next(iter(img_dict.values()))
        hw = tuple(batch_dict["images"].shape[2:]) if "images" in batch_dict else tuple(batch_dict["image_hw"])
        if self._sample_native(x_conv1, batch_dict, fmap) and x_conv1.features.shape[1] == fmap.shape[1]:
            B = batch_dict["batch_size"]
            vs, r0 = self._geometry_floats()
            out, _ = _ops.voxel_image_sample(x_conv1.indices.contiguous(), B, 1.0, vs, r0,
                                             self._aug_params(batch_dict, B, fmap.device),
                                             batch_dict["lidar2img"].float().contiguous(), fmap, hw,
                                             add=x_conv1.features.contiguous())
            return x_conv1.replace_feature(out)                               # MVX, fuse_sum=True (:746-748)
        _, uv = self._project(x_conv1, 1, batch_dict)
        img_feat = self._sample_int(fmap, x_conv1.indices[:, 0].long(), uv, hw)
        return x_conv1.replace_feature(x_conv1.features + img_feat)           # MVX, fuse_sum=True (:746-748)

    def _fuse4(self, x_conv2, x_conv3, x_conv4, batch_dict):
        if 4 not in self.fusion_pos or "ACTR" not in self.fusion_method:
            return x_conv4
        img_dict = batch_dict["img_dict"]
        x_rgb = [v for k, v in img_dict.items() if k != "mvx_layer1_feat2d"]
        hw = tuple(batch_dict["images"].shape[2:]) if "images" in batch_dict else tuple(batch_dict["image_hw"])
        ind = x_conv4.indices
        B = batch_dict["batch_size"]
        feats = x_conv4.features
        pre = self.__dict__.get("_fuse4_pre")
        if pre is not None and pre["n"] == ind.shape[0] and list(pre["shape"]) == list(x_conv4.spatial_shape):
            # positions, slots and the local-attention geometry were computed beside the convolutions (`_prefetch_fuse4`:
            # the same strided index sets in the same flat-index order)
            torch.cuda.current_stream().wait_event(pre["event"])
            xyz, b, slot, n_max, pts = pre["xyz"], pre["b"], pre["slot"], pre["n_max"], pre["pts"]
            uv = None
        else:
            xyz, uv = self._project(x_conv4, 8, batch_dict)
            b, slot, n_max = self._query_slots(ind, B)
            pts = feats.new_zeros((B, n_max, 3))
            pts[b, slot] = xyz
        C = feats.shape[1]
        if self._sample_native(x_conv4, batch_dict, x_rgb[0]):
            # image features of the queries and their normalised pixels, written straight into the padded tensors
            rows = b * n_max + slot
            v_feat = feats.new_zeros((B, n_max, C))
            v_i = feats.new_zeros((B, n_max, x_rgb[0].shape[1]))
            grid = feats.new_zeros((B, n_max, 2))
            vs, r0 = self._geometry_floats()
            _ops.voxel_image_sample(ind.contiguous(), B, 8.0, vs, r0, self._aug_params(batch_dict, B, feats.device),
                                    batch_dict["lidar2img"].float().contiguous(), x_rgb[0], hw, rows=rows,
                                    out=v_i.view(B * n_max, -1), grid=grid.view(B * n_max, 2))
            v_feat.view(B * n_max, C).index_copy_(0, rows, feats)
            i_feats = x_rgb if self.ifat is None else self._gate_images(x_rgb, [x_conv2, x_conv3, x_conv4], batch_dict)
            enh = self.actr(v_feat=v_feat, v_i_feat=v_i, grid=grid, i_feats=i_feats, lidar_grid=pts)
            return x_conv4.replace_feature(enh.reshape(B * n_max, -1).index_select(0, rows) + feats)   # fuse_sum=True (:808-810)
        if uv is None:
            uv = self._pixels(xyz, b, batch_dict)
        i_feat = self._sample_int(x_rgb[0], b, uv, hw)
        v_feat = feats.new_zeros((B, n_max, C))
        v_i = feats.new_zeros((B, n_max, i_feat.shape[1]))
        grid = feats.new_zeros((B, n_max, 2))
        v_feat[b, slot] = feats
        v_i[b, slot] = i_feat
        grid[b, slot] = uv / _ops.device_constant([float(hw[1]), float(hw[0])], torch.float32, uv.device)
        i_feats = x_rgb if self.ifat is None else self._gate_images(x_rgb, [x_conv2, x_conv3, x_conv4], batch_dict)
        enh = self.actr(v_feat=v_feat, v_i_feat=v_i, grid=grid, i_feats=i_feats, lidar_grid=pts)
        return x_conv4.replace_feature(enh[b, slot] + feats)                  # fuse_sum=True (:808-810)

    def _gate_images(self, x_rgb, x_list, batch_dict):
        """spconv_backbone.py:797-800: the image features ACTR attends over pass the image gate first (the queries' own image
        features were sampled from the un-gated maps above, as in the reference)."""
        return self.ifat(x_rgb, x_list, batch_dict, project=lambda x, stride: self._project(x, stride, batch_dict))


BACKBONES_3D.register_module(VoxelBackBone8x)
BACKBONES_3D.register_module(VoxelBackBone8xFusion)
