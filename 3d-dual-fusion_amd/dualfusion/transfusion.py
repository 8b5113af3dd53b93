"""`TransFusionDetector`, point-cloud branch (TF/mmdet3d/models/detectors/transfusion.py:19-260 over
mvx_two_stage.py): pts_voxel_layer -> pts_voxel_encoder (mean VFE, fused into the voxeliser) -> pts_middle_encoder
(`SparseEncoder[Fusion]`, the camera fusion layer inside) -> pts_backbone (`SECOND`) -> pts_neck (`SECONDFPN`) ->
pts_bbox_head (`TransFusionHead`), with the reference's constructor keys (config dicts resolved through the registries
of dualfusion/registry.py, or modules), `forward_train` / `simple_test` / `extract_pts_feat` / `forward_pts_train`.

The camera network is out of scope (the reference freezes it, transfusion.py:25-38): its output
feature maps are an INPUT -- `img_feats=` (a list of [B * num_cams, C, H, W] levels) wherever the reference passes `img`;
an `img_backbone` / `img_neck` given as modules is called like the reference calls them, frozen.

`training_step` is one iteration of the reference's runner for BASELINE configs[3]'s per-rank body (mmcv
`EpochBasedRunner.train` -> `OptimizerHook.after_train_iter`: `forward_train` -> `_parse_losses` -> `loss.backward()` ->
`clip_grad_norm_(max_norm=0.1, norm_type=2)` -> AdamW step, TF/configs/transfusion_nusc_voxel_F.py:302-303) with the
gradient all-reduce in few large buckets launched during backward (dualfusion/dist.py `GradBucketReducer`, what mmcv's
`MMDistributedDataParallel` does).  Every convolution of the step -- the 21 rulebook'd sparse convolutions of the encoder,
the 3x3 BEV convolutions of SECOND / SECONDFPN -- runs forward, input gradient and filter gradient on the HIP kernels
through `SparseConvFunction`; BatchNorm over rows on csrc/bnrows.hip; the fusion layer's deformable sampling and its
backward on csrc/msda.hip; Hungarian matching costs, heat-map targets and the three losses WITH their gradients on
csrc/tfloss.hip (`TransFusionHead.loss_device`)."""
import torch
from torch import nn

from . import ops as _ops
from .registry import (FUSION_LAYERS, MIDDLE_ENCODERS, MM_BACKBONES, MM_DETECTORS, MM_HEADS, MM_NECKS, VOXEL_ENCODERS,  # noqa: F401
                       build_from_cfg)


def _build(cfg, registry, **defaults):
    if cfg is None or isinstance(cfg, nn.Module):
        return cfg
    return build_from_cfg(dict(cfg), registry, defaults or None)


def parse_losses(losses):
    """mmdet `BaseDetector._parse_losses` (the reduction the runner applies to the dict `forward_train` returns): every
    entry whose key contains 'loss' is summed into the scalar that is back-propagated; tensors are averaged, lists of
    tensors summed.  -> (loss, log_vars of DEVICE scalars -- the reference's `.item()` per key is left to the caller, who
    reads them where it logs)."""
    log_vars = {}
    for name, value in losses.items():
        if torch.is_tensor(value):
            log_vars[name] = value.mean()
        elif isinstance(value, (list, tuple)):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError("%s is not a tensor or list of tensors" % name)
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, log_vars


@MM_DETECTORS.register_module()
class TransFusionDetector(nn.Module):
    def __init__(self, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None, pts_backbone=None,
                 pts_neck=None, pts_bbox_head=None, img_backbone=None, img_neck=None, freeze_img=True, train_cfg=None,
                 test_cfg=None, pretrained=None, pts_fusion_layer=None, img_roi_head=None, img_rpn_head=None, **kwargs):
        super(TransFusionDetector, self).__init__()
        from . import backbones, necks, transfusion_head, voxel  # noqa: F401   (register the classes the configs name)
        if pts_fusion_layer is not None or img_roi_head is not None or img_rpn_head is not None:
            raise NotImplementedError("pts_fusion_layer / img_roi_head / img_rpn_head: not part of the 3D-Dual-Fusion configs "
                                      "(the fusion layer lives inside pts_middle_encoder)")
        if isinstance(pts_voxel_layer, dict):
            pts_voxel_layer = voxel.Voxelization(**pts_voxel_layer)
        self.pts_voxel_layer = pts_voxel_layer
        self.pts_voxel_encoder = _build(pts_voxel_encoder, VOXEL_ENCODERS)
        self.pts_middle_encoder = _build(pts_middle_encoder, MIDDLE_ENCODERS)
        self.pts_backbone = _build(pts_backbone, MM_BACKBONES)
        self.pts_neck = _build(pts_neck, MM_NECKS)
        if isinstance(pts_bbox_head, dict):
            # mvx_two_stage.py:50-55: the detector's train_cfg.pts / test_cfg.pts become the head's
            pts_bbox_head = dict(pts_bbox_head)
            pts_bbox_head.setdefault('train_cfg', (train_cfg or {}).get('pts') if train_cfg else None)
            pts_bbox_head.setdefault('test_cfg', (test_cfg or {}).get('pts') if test_cfg else None)
        self.pts_bbox_head = _build(pts_bbox_head, MM_HEADS)
        if isinstance(img_backbone, dict) or isinstance(img_neck, dict):
            raise NotImplementedError("the camera network is an input of the hot path (SURVEY.md 2.1 #13): pass its feature maps "
                                      "as img_feats=, or hand over built modules")
        self.img_backbone, self.img_neck = img_backbone, img_neck
        self.freeze_img = freeze_img
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        # the clouds handed to forward are complete in device memory (a data loader's synchronised copies): voxelisation may
        # then run on its own stream (ops.hard_voxelize_clouds)
        self.resident_inputs = False
        if self.freeze_img:                                       # transfusion.py:25-38
            for m in (self.img_backbone, self.img_neck):
                if m is not None:
                    for p in m.parameters():
                        p.requires_grad = False

    with_pts_bbox = property(lambda self: self.pts_bbox_head is not None)
    with_pts_neck = property(lambda self: self.pts_neck is not None)
    with_img_backbone = property(lambda self: self.img_backbone is not None)
    with_img_neck = property(lambda self: self.img_neck is not None)

    # ------------------------------------------------------------------ feature extraction
    def extract_img_feat(self, img, img_metas, img_feats=None):
        """transfusion.py:40-60; `img_feats` (already extracted maps) wins."""
        if img_feats is not None:
            return list(img_feats) if isinstance(img_feats, (list, tuple)) else [img_feats]
        if self.img_backbone is None or img is None:
            return None
        input_shape = img.shape[-2:]
        for m in img_metas:
            m.update(input_shape=input_shape)
        if img.dim() == 5:
            img = img.view((-1,) + tuple(img.shape[2:]))
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.freeze_img):
            feats = self.img_backbone(img.float())
            if self.img_neck is not None:
                feats = self.img_neck(feats)
        return feats

    @torch.no_grad()
    def voxelize(self, points):
        """transfusion.py:76-108 + HardSimpleVFE fused: list of [P_b, C] clouds -> (mean features [M, C], coors [M, 4] (b, z, y, x)),
        mmdet3d's hard voxelisation (`break` at the cap)."""
        vl = self.pts_voxel_layer
        clouds = [p.contiguous().float() for p in points]
        nfeat = getattr(self.pts_voxel_encoder, "num_features", clouds[0].shape[1])
        if nfeat != clouds[0].shape[1]:
            clouds = [c[:, :nfeat].contiguous() for c in clouds]
        resident = self.resident_inputs and all(q is p for q, p in zip(clouds, points))
        return _ops.hard_voxelize_clouds(clouds, vl.voxel_size, vl.point_cloud_range, vl.max_num_points, vl._cap(),
                                         break_at_cap=True, resident_inputs=resident)

    def encode_voxels(self, voxel_features, coors, batch_size, img_feats, img_metas, img=None, prepared=None):
        """pts_middle_encoder -> pts_backbone -> pts_neck on voxel features that already exist (transfusion.py:66-74)."""
        enc = self.pts_middle_encoder
        if 'Fusion' in enc.__class__.__name__:
            kw = dict(prepared=prepared) if prepared is not None else {}
            x = enc(voxel_features, coors, batch_size, img_feats=img_feats, img_metas=img_metas, img=img, **kw)
        else:
            x = enc(voxel_features, coors, batch_size)
        x = self.pts_backbone(x)
        if self.with_pts_neck:
            x = self.pts_neck(x)
        return x

    def extract_pts_feat(self, pts, img_feats, img_metas, img=None):
        if not self.with_pts_bbox:
            return None
        voxel_features, coors = self.voxelize(pts)
        return self.encode_voxels(voxel_features, coors, len(pts), img_feats, img_metas, img)

    def extract_feat(self, points, img, img_metas, img_feats=None):
        img_feats = self.extract_img_feat(img, img_metas, img_feats)
        return img_feats, self.extract_pts_feat(points, img_feats, img_metas, img)

    # ------------------------------------------------------------------ training
    def forward_pts_train(self, pts_feats, img_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore=None):
        """transfusion.py:172-199.  On the GPU the losses come from `TransFusionHead.loss_device` (same keys / values as `loss`,
        the kernels also write the gradients); elsewhere from the plain torch formulation `loss`."""
        outs = self.pts_bbox_head(pts_feats, img_feats, img_metas)
        on_gpu = outs[0][0]['center'].is_cuda and outs[0][0]['center'].dtype == torch.float32
        fn = self.pts_bbox_head.loss_device if on_gpu else self.pts_bbox_head.loss
        return fn(gt_bboxes_3d, gt_labels_3d, outs)

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, gt_labels=None, gt_bboxes=None,
                      img=None, proposals=None, gt_bboxes_ignore=None, img_feats=None):
        """transfusion.py:110-170 (the image-branch losses belong to detectors with an image head: none in these configs)."""
        img_feats, pts_feats = self.extract_feat(points, img, img_metas, img_feats)
        losses = dict()
        if pts_feats:
            losses.update(self.forward_pts_train(pts_feats, img_feats, gt_bboxes_3d, gt_labels_3d, img_metas, gt_bboxes_ignore))
        return losses

    def forward_train_voxels(self, voxel_features, coors, batch_size, img_feats, img_metas, gt_bboxes_3d, gt_labels_3d):
        """`forward_train` behind the voxeliser (a data loader that voxelises in its workers hands these over)."""
        pts_feats = self.encode_voxels(voxel_features, coors, batch_size, img_feats, img_metas)
        return self.forward_pts_train(pts_feats, img_feats, gt_bboxes_3d, gt_labels_3d, img_metas)

    def training_step(self, points, img_feats, img_metas, gt_bboxes_3d, gt_labels_3d, reducer=None, optimizer=None,
                      grad_clip=None, voxels=None):
        """One iteration of the reference's runner on this rank's samples (module docstring): zero the gradients, forward +
        losses, backward (the reducer's buckets travel while it runs), wait for the buckets, clip, step.  -> (loss, log_vars)
        of device scalars; with `grad_clip` log_vars carries 'grad_norm' (mmcv's OptimizerHook logs it).
        reducer: `GradBucketReducer` over the trainable parameters (None: no reduction, the caller's business);
        optimizer: stepped when given; grad_clip: dict(max_norm=, norm_type=) as in `optimizer_config`;
        voxels: (voxel_features, coors) when the loader voxelised already."""
        if reducer is not None:
            reducer.zero_grad()
        elif optimizer is not None:
            optimizer.zero_grad(set_to_none=True)
        def forward_losses():
            if voxels is None:
                return self.forward_train(points=points, img_metas=img_metas, gt_bboxes_3d=gt_bboxes_3d,
                                          gt_labels_3d=gt_labels_3d, img_feats=img_feats)
            return self.forward_train_voxels(voxels[0], voxels[1], len(img_metas), img_feats, img_metas, gt_bboxes_3d,
                                             gt_labels_3d)
        with torch.enable_grad():
            # (`loss_device` reads the range flag of the fp16 operand format with its matching costs, BEFORE anything is
            # back-propagated: a forward that met a value the format cannot hold is repeated on three bf16 parts)
            losses = _ops.with_range_fallback(forward_losses)
            loss, log_vars = parse_losses(losses)
            loss.backward()
        if reducer is not None:
            reducer.finish()
        if grad_clip is not None:
            if hasattr(optimizer, "clip_grad_norm"):            # dist.BucketAdamW: over its flat gradient buffers
                log_vars['grad_norm'] = optimizer.clip_grad_norm(**grad_clip)
            else:
                params = reducer.params if reducer is not None else [p for p in self.parameters() if p.requires_grad]
                log_vars['grad_norm'] = clip_grads(params, **grad_clip)
        if optimizer is not None:
            optimizer.step()
        return loss.detach(), {k: v.detach() for k, v in log_vars.items()}

    # ------------------------------------------------------------------ inference
    def simple_test_pts(self, x, x_img, img_metas, rescale=False):
        outs = self.pts_bbox_head(x, x_img, img_metas)
        bbox_list = self.pts_bbox_head.get_bboxes(outs, img_metas, rescale=rescale)
        # bbox3d2result (mmdet3d/core/bbox/transforms.py): host copies, the evaluation's format
        return [dict(boxes_3d=b.to('cpu') if hasattr(b, 'to') else b, scores_3d=s.cpu(), labels_3d=l.cpu())
                for b, s, l in bbox_list]

    @torch.no_grad()
    def simple_test(self, points, img_metas, img=None, rescale=False, img_feats=None):
        def run():
            feats2d, pts_feats = self.extract_feat(points, img, img_metas, img_feats)
            bbox_list = [dict() for _ in range(len(img_metas))]
            if pts_feats and self.with_pts_bbox:
                for result, pts_bbox in zip(bbox_list, self.simple_test_pts(pts_feats, feats2d, img_metas, rescale=rescale)):
                    result['pts_bbox'] = pts_bbox
            return bbox_list
        # (`get_bboxes` reads the range flag of the fp16 operand format with the box counts: ops.with_range_fallback reruns a
        # frame that left the format's range on three bf16 parts)
        return _ops.with_range_fallback(run)

    def forward(self, return_loss=True, **kwargs):
        """mmdet `BaseDetector.forward`: forward_train(**kwargs) or simple_test(**kwargs)."""
        if return_loss:
            return self.forward_train(**kwargs)
        return self.simple_test(**kwargs)


def clip_grads(params, max_norm, norm_type=2):
    """mmcv `OptimizerHook.clip_grads` = torch's `clip_grad_norm_` over the parameters that hold a gradient: the total norm
    stays on the device (no host round trip) and scales the gradients in one multi-tensor launch."""
    params = [p for p in params if p.requires_grad and p.grad is not None]
    if not params:
        return torch.zeros(())
    return torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm, norm_type=norm_type, foreach=params[0].is_cuda or None)
