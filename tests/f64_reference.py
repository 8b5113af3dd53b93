"""float64 torch compositions of the kernel-backed leaves of the training path, so that a detector's own Python modules
(deep-copied, `.double()`, on the CPU) become an independent float64 reference of a whole training step: the sparse
convolutions on the ORACLE's rulebooks (`oracle.get_indice_pairs`, the restated reference CPU algorithm) as
gather -> matmul -> index_add, `dense()` as an index_put, the multi-scale deformable sampling as `grid_sample`
(the reference's `ms_deform_attn_core_pytorch`, CP/det3d/models/model_utils/ops/functions/ms_deform_attn_func.py:41-61).
BatchNorm, the BEV convolutions, the decoder and the losses already take plain torch paths off the GPU.

Test infrastructure only (imports `oracle/`): the product path has no CPU fallback; `patched()` swaps the leaves for the
duration of a `with` block and only acts on HOST tensors (float64 for the reference; the same compositions in float32 are
the yardstick of what plain fp32 torch arithmetic makes of the same step)."""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from oracle import oracle as orc


def sparse_conv_f64(features, weight, bias, pairs, num, n_out):
    """features [N, Cin], weight [*k, Cin, Cout] -> [n_out, Cout]: sum over offsets of x[in] @ W[k] added at out."""
    K = int(np.prod(weight.shape[:-2]))
    w = weight.reshape(K, weight.shape[-2], weight.shape[-1])
    out = features.new_zeros((n_out, w.shape[-1]))
    for k in range(K):
        m = int(num[k])
        if m == 0:
            continue
        i = torch.from_numpy(pairs[k, 0, :m].astype(np.int64))
        o = torch.from_numpy(pairs[k, 1, :m].astype(np.int64))
        out = out.index_add(0, o, features[i] @ w[k])
    return out + bias if bias is not None else out


def msda_core_f64(value, shapes, sampling_locations, attention_weights):
    """value [N, S, M, D], shapes [(H, W)], locations [N, Lq, M, L, P, 2] in [0, 1], weights [N, Lq, M, L, P]
    -> [N, Lq, M * D]; bilinear, zero padding, align_corners=False (pixel centres at (i + 0.5) / size)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    grids = 2 * sampling_locations - 1
    out = []
    start = 0
    for lvl, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W].flatten(2).transpose(1, 2).reshape(N * M, D, H, W)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)                        # [N*M, Lq, P, 2]
        out.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
        start += H * W
    aw = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    res = (torch.stack(out, dim=-2).flatten(-2) * aw).sum(-1).view(N, M * D, Lq)
    return res.transpose(1, 2).contiguous()


@contextlib.contextmanager
def patched(projection=None):
    """Leaves -> float64 compositions while the block runs.  `projection`: (pts, cam_id, norm, pix) of the GPU run for
    `ACTRFusionLayer.project` -- the projection of voxel centres is integer-valued downstream (camera assignment,
    pixel // 4), has no learnable input, and is pinned by its own goldens; reusing it keeps an fp32-vs-fp64 rounding of
    one boundary voxel from turning into a different query set."""
    from dualfusion import fusion_tf, msda
    from dualfusion.spconv import conv as sconv
    from dualfusion.spconv import structure

    from dualfusion import backbones, ops as dops
    overlap_gpu = dops.boxes_overlap_bev_xyxyr

    def overlap(boxes_a, boxes_b, *a, **k):
        """rotated BEV overlap areas for host tensors: the oracle's restatement of the reference kernel (it only feeds the
        matching costs -- no gradient passes through it)."""
        if boxes_a.is_cuda:
            return overlap_gpu(boxes_a, boxes_b, *a, **k)
        return torch.from_numpy(orc.tf_boxes_overlap_bev(boxes_a.detach().numpy(), boxes_b.detach().numpy())).to(boxes_a.dtype)

    dops.boxes_overlap_bev_xyxyr = overlap
    saved = (sconv.SparseConvolution.forward_fused, msda.MSDeformAttnFunction.apply, structure.SparseConvTensor.dense,
             fusion_tf.ACTRFusionLayer.project, backbones.SparseEncoderFusion.coor2pts)
    books = {}

    def forward_fused(self, input, scale=None, shift=None, relu=False, residual=None):
        if input.features.is_cuda:
            return saved[0](self, input, scale, shift, relu, residual)
        assert scale is None and shift is None and residual is None and not relu and self.ndim == 3
        ind = input.indices.cpu().numpy().astype(np.int32)
        key = (ind.tobytes(), tuple(self.kernel_size), tuple(self.stride), tuple(self.padding), bool(self.subm))
        if key not in books:
            books[key] = orc.get_indice_pairs(ind, input.batch_size, [int(v) for v in input.spatial_shape],
                                              list(self.kernel_size), list(self.stride), list(self.padding),
                                              list(self.dilation), self.subm)
        outids, pairs, num, oshape = books[key]
        y = sparse_conv_f64(input.features, self.weight, self.bias, pairs, num, len(outids))
        out = structure.SparseConvTensor(y, torch.from_numpy(np.ascontiguousarray(outids)), oshape, input.batch_size)
        out.indice_dict, out.grid = input.indice_dict, input.grid
        return out

    def msda_apply(value, shapes, level_start, loc, aw, im2col_step):
        if value.is_cuda:
            return saved[1](value, shapes, level_start, loc, aw, im2col_step)
        return msda_core_f64(value, [(int(h), int(w)) for h, w in shapes.tolist()], loc, aw)

    def dense(self, channels_first=True):
        if self.features.is_cuda:
            return saved[2](self, channels_first)
        idx = self.indices.long()
        C = self.features.shape[1]
        vol = self.features.new_zeros((self.batch_size,) + tuple(int(v) for v in self.spatial_shape) + (C,))
        vol = vol.index_put(tuple(idx[:, i] for i in range(idx.shape[1])), self.features)
        nd = len(self.spatial_shape)
        return vol.permute(0, nd + 1, *range(1, nd + 1)).contiguous() if channels_first else vol

    def coor2pts(self, x, pad=0.0):
        return saved[4](self, x, pad).to(x.features.dtype)       # voxel centres in the features' precision

    def project(self, pts, img_metas):
        if projection is None or pts.is_cuda:
            return saved[3](self, pts, img_metas)
        ref_pts, cam_id, norm, pix = projection
        assert cam_id.shape[0] == pts.shape[0] == ref_pts.shape[0]
        # the GPU run's rows are in ITS rulebook's order, these in the oracle's: match them by voxel centre (computed in fp32
        # from the integer coordinates on both sides: identical values)
        key = lambda t: [r.tobytes() for r in np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))]   # noqa: E731
        where = {k: i for i, k in enumerate(key(ref_pts))}
        order = torch.tensor([where[k] for k in key(pts)], dtype=torch.long)
        return cam_id.cpu()[order], norm.cpu().to(pts.dtype)[order], pix.cpu().to(pts.dtype)[order]

    sconv.SparseConvolution.forward_fused = forward_fused
    msda.MSDeformAttnFunction.apply = staticmethod(msda_apply)
    structure.SparseConvTensor.dense = dense
    fusion_tf.ACTRFusionLayer.project = project
    backbones.SparseEncoderFusion.coor2pts = coor2pts
    try:
        yield
    finally:
        sconv.SparseConvolution.forward_fused = saved[0]
        msda.MSDeformAttnFunction.apply = saved[1]
        structure.SparseConvTensor.dense = saved[2]
        fusion_tf.ACTRFusionLayer.project = saved[3]
        backbones.SparseEncoderFusion.coor2pts = saved[4]
        dops.boxes_overlap_bev_xyxyr = overlap_gpu


@contextlib.contextmanager
def without_relu(*models):
    """Every ReLU of the training path -> identity while the block runs (module ReLUs, the ReLU fused into the BatchNorm row
    kernel, the `activation` of the transformer layers), on the GPU model and on its host copies alike.  A rectifier turns
    the last bit of a pre-activation near zero into a different gradient path: two correct 24-bit evaluations of the same
    step then differ by ~1e-3 in their gradients (tools/debug/tf_train_diff.py: a single unit of the heat-map branch with
    pre-activation -1.9e-6 / +2.7e-6 moves every upstream gradient by 7e-3).  Without rectifiers the step is smooth, every
    convolution / BatchNorm / sampling / loss kernel and its backward is still in it, and the gradients can be compared
    with float64 at the grade of the arithmetic."""
    from dualfusion import ops as dops
    ident = lambda x, *a, **k: x                                                    # noqa: E731
    bn_rows, relu_fwd = dops.batch_norm_rows, torch.nn.ReLU.forward
    acts = [(m, m.activation) for model in models for m in model.modules() if callable(getattr(m, "activation", None))]
    dops.batch_norm_rows = lambda bn, x, relu=False: bn_rows(bn, x, False)
    torch.nn.ReLU.forward = lambda self, x: x
    for m, _ in acts:
        m.activation = ident
    try:
        yield
    finally:
        dops.batch_norm_rows, torch.nn.ReLU.forward = bn_rows, relu_fwd
        for m, a in acts:
            m.activation = a


# ------------------------------------------------------------------------------------------------ a reduced TransFusion tree
SMALL_VOXEL = [0.15, 0.15, 0.2]
SMALL_RANGE = [-12.0, -12.0, -5.0, 12.0, 12.0, 3.0]              # grid 160 x 160 x 40 -> BEV 20 x 20


def small_transfusion_detector(num_proposals=24):
    """TransFusion-L + 3D-DF with the structure of TF/configs/transfusion_nusc_voxel_F.py on a 160 x 160 x 40 grid."""
    from dualfusion.backbones import SparseEncoderFusion
    from dualfusion.necks import SECOND, SECONDFPN
    from dualfusion.transfusion import TransFusionDetector
    from dualfusion.transfusion_head import TransFusionHead
    from dualfusion.voxel import HardSimpleVFE, Voxelization
    from dualfusion.workloads import TF_ACTR_CFG
    ch = ((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128))
    pad = ((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0))
    enc = SparseEncoderFusion(in_channels=5, sparse_shape=[41, 160, 160], output_channels=128, encoder_channels=ch,
                              encoder_paddings=pad, block_type='basicblock', fusion_pos=[3], voxel_size=SMALL_VOXEL,
                              point_cloud_range=SMALL_RANGE, fusion_layer=dict(type='ACTR', pfat_cfg=dict(TF_ACTR_CFG)))
    head = TransFusionHead(
        num_proposals=num_proposals, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10, num_decoder_layers=1,
        num_heads=8, initialize_by_heatmap=True, nms_kernel_size=3, ffn_channel=256, dropout=0.0,
        common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
        bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=SMALL_RANGE[:2], voxel_size=SMALL_VOXEL[:2], out_size_factor=8,
                        post_center_range=[-14.0, -14.0, -10.0, 14.0, 14.0, 10.0], score_threshold=0.0, code_size=10),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
        loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0),
        train_cfg=dict(dataset='nuScenes',
                       assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                                     cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                                     reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
                       pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[160, 160, 40], voxel_size=SMALL_VOXEL,
                       out_size_factor=8, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
                       point_cloud_range=SMALL_RANGE),
        test_cfg=dict(dataset='nuScenes', grid_size=[160, 160, 40], out_size_factor=8, pc_range=SMALL_RANGE[:2],
                      voxel_size=SMALL_VOXEL[:2], nms_type=None))
    det = TransFusionDetector(
        pts_voxel_layer=Voxelization(SMALL_VOXEL, SMALL_RANGE, 10, (20000, 20000)), pts_voxel_encoder=HardSimpleVFE(num_features=5),
        pts_middle_encoder=enc, pts_backbone=SECOND(in_channels=256, out_channels=[128, 256], layer_nums=[2, 2], layer_strides=[1, 2]),
        pts_neck=SECONDFPN(in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2], use_conv_for_no_stride=True),
        pts_bbox_head=head)
    for m in det.modules():                                       # a comparison of two arithmetic paths: no dropout noise
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return det


def small_inputs(batch, seed=0, in_hw=(128, 224), n_gt=5):
    """`batch` samples inside SMALL_RANGE: clouds (cut from the synthetic nuScenes sweeps), six camera maps per sample at
    stride 4 of a 128 x 224 input, calibration, ground-truth boxes / labels.  numpy / host values only."""
    from dualfusion import synth
    ori_hw = (900, 1600)
    cams = synth.nusc_cameras(image_hw=ori_hw, yaw_offset_deg=1.5 + seed)
    sf = [in_hw[1] / ori_hw[1], in_hw[0] / ori_hw[0]]
    metas = [dict(lidar2cam=np.stack([cams[n][0] for n in synth.NUSC_CAMS]),
                  cam_intrinsic=np.stack([cams[n][1] for n in synth.NUSC_CAMS]), ori_shape=ori_hw + (3,),
                  img_shape=in_hw + (3,), input_shape=in_hw, scale_factor=sf, flip=False) for _ in range(batch)]
    r = np.asarray(SMALL_RANGE, np.float32)
    points = []
    for b in range(batch):
        p = synth.nusc_sweep(seed=seed * 16 + b)
        keep = np.all((p[:, :3] > r[:3] + 0.01) & (p[:, :3] < r[3:] - 0.01), 1)
        points.append(np.ascontiguousarray(p[keep][:9000]))
    img = synth.camera_features(batch * 6, 256, (in_hw[0] // 4, in_hw[1] // 4), 77 + seed)
    rs = np.random.RandomState(1000 + seed)
    gts, labels = [], []
    for b in range(batch):
        n = n_gt + b
        box = np.zeros((n, 9), np.float32)
        box[:, :2] = rs.uniform(-10, 10, (n, 2))
        box[:, 2] = rs.uniform(-2.5, -1.0, n)
        box[:, 3:6] = rs.uniform(0.6, 4.5, (n, 3))
        box[:, 6] = rs.uniform(-3.1, 3.1, n)
        box[:, 7:] = rs.uniform(-2, 2, (n, 2))
        gts.append(box)
        labels.append(rs.randint(0, 10, n).astype(np.int64))
    return points, img, metas, gts, labels
