"""The other two module trees at their BASELINE config shapes, as bench.py workloads (`--workload tf_fusion|vr_fusion`).

  tf_fusion : BASELINE configs[2] -- TransFusion-L + 3D-DF: `SparseEncoderFusion` + ACTR fusion layer on the full
              0.075 m nuScenes grid, bs = 4 sweeps x 6 cameras (ResNet50-stride-4-shaped feature maps [24, 256, 112, 200]),
              then SECOND + SECONDFPN + TransFusionHead (LiDAR-only decoder, 200 proposals) -> boxes
              (TF/configs/transfusion_nusc_voxel_F.py:181-243 shapes).
  vr_fusion : BASELINE configs[4] -- Voxel-RCNN + 3D-DF: `VoxelBackBone8xFusion` (MVX + ACTRv2 with the 3-D local
              self-attention), KITTI 0.05 m grid, bs = 8 frames x 1 camera (small-grid / high-sparsity rulebook stress).

Inputs are synthetic and resident in HBM; every step takes the next of `--frames` distinct frames.  Voxelisation of the
batch is part of the step.  The tf_fusion step ends like the CenterPoint one: detection losses of the head
(`TransFusionHead.loss_device`: Hungarian target assignment + focal / L1 / Gaussian-focal losses) which bench.py reduces
over the ranks (`reduce_dict`) -- BASELINE configs[3]'s "RCCL all-reduce of detection losses"; `--stage boxes` ends at
the decoded boxes instead (round 2's step).  The Voxel-RCNN tree ends at its fused backbone (no head in the reference's
3D-DF addition): replicas behind bench.py's barrier."""
import numpy as np
import os

import torch

from . import ops, synth

TF_ACTR_CFG = dict(fusion_method="sum", feature_modal="hybrid",
                   hybrid_cfg=dict(attn_layer="BiGateSum1D_2", q_method="sum", q_rep_place=["weight"]),
                   num_bins=80, num_channels=[256], query_num_feat=128, num_enc_layers=2, max_num_ne_voxel=26000,
                   pos_encode_method="depth")


def _voxelize_batch(points, vs, rng, max_points, max_voxels):
    # the frames' point clouds are resident inputs (bench.py's contract): voxelise on the voxel stream
    return ops.hard_voxelize_clouds(points, vs, rng, max_points, max_voxels,
                                    resident_inputs=os.environ.get("DF3D_VOXEL_STREAM", "1") == "1")


class TransFusionWorkload(object):
    name, unit_name = "tf_fusion", "sweeps"
    hot_path_what = "the same K steps ending at the encoder's dense BEV map (voxelize + SparseEncoderFusion + ACTR; no neck / head)"

    def __init__(self, args, rank, world, dev):
        from .backbones import SparseEncoderFusion
        from .necks import SECOND, SECONDFPN
        from .transfusion_head import TransFusionHead
        self.batch = B = args.batch or 4
        self.dev = dev
        self.prefetch = bool(getattr(args, "prefetch", True)) and os.environ.get("DF3D_VOXEL_STREAM", "1") == "1"
        torch.manual_seed(0)
        ch = ((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128))
        pad = ((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0))
        self.enc = SparseEncoderFusion(in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128,
                                       encoder_channels=ch, encoder_paddings=pad, block_type='basicblock', fusion_pos=[3],
                                       voxel_size=synth.NUSC_VOXEL, point_cloud_range=synth.NUSC_RANGE,
                                       fusion_layer=dict(type='ACTR', pfat_cfg=dict(TF_ACTR_CFG))).to(dev).eval()
        self.second = SECOND(in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2]).to(dev).eval()
        self.fpn = SECONDFPN(in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2],
                             use_conv_for_no_stride=True).to(dev).eval()
        self.head = TransFusionHead(
            num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10, num_decoder_layers=1,
            num_heads=8, initialize_by_heatmap=True, nms_kernel_size=3, ffn_channel=256,
            common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
            bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075],
                            out_size_factor=8, post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                            score_threshold=0.0, code_size=10),
            # TF/configs/transfusion_nusc_voxel_L.py:212-234
            loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
            loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
            loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0),
            train_cfg=dict(dataset='nuScenes',
                           assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                                         cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                                         reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
                           pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[1440, 1440, 40],
                           voxel_size=synth.NUSC_VOXEL, out_size_factor=8,
                           code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2], point_cloud_range=synth.NUSC_RANGE),
            test_cfg=dict(dataset='nuScenes', grid_size=[1440, 1440, 40], out_size_factor=8, pc_range=[-54.0, -54.0],
                          voxel_size=[0.075, 0.075], nms_type=None)).to(dev).eval()
        from .transfusion import TransFusionDetector
        from .voxel import HardSimpleVFE, Voxelization
        # the detector the reference composes from the same config (TF/mmdet3d/models/detectors/transfusion.py): the training
        # step (`--stage train`, BASELINE configs[3]'s per-rank body) goes through it; the inference stages above call the
        # same sub-modules directly (with the frame head a batch ahead)
        self.detector = TransFusionDetector(
            pts_voxel_layer=Voxelization(synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, (120000, 160000)),
            pts_voxel_encoder=HardSimpleVFE(num_features=5), pts_middle_encoder=self.enc, pts_backbone=self.second,
            pts_neck=self.fpn, pts_bbox_head=self.head).to(dev).eval()
        self.detector.resident_inputs = os.environ.get("DF3D_VOXEL_STREAM", "1") == "1"
        ori_hw, in_hw, fh, fw = (900, 1600), (448, 800), 112, 200          # stride-4 level (the layer indexes pix // 4)
        sf = [in_hw[1] / ori_hw[1], in_hw[0] / ori_hw[0]]
        self.frames = []
        for f in range(max(1, args.frames)):
            seed = rank * 1000 + f
            cams = synth.nusc_cameras(image_hw=ori_hw, yaw_offset_deg=0.37 * (seed % 97))
            metas = [dict(lidar2cam=np.stack([cams[n][0] for n in synth.NUSC_CAMS]),
                          cam_intrinsic=np.stack([cams[n][1] for n in synth.NUSC_CAMS]), ori_shape=ori_hw + (3,),
                          img_shape=in_hw + (3,), input_shape=in_hw, scale_factor=sf, flip=False) for _ in range(B)]
            gts = [synth.nusc_gt_boxes(seed * 16 + b) for b in range(B)]    # host tensors, as the data loader hands them over
            self.frames.append(dict(
                points=[torch.from_numpy(synth.nusc_sweep(seed=seed * 16 + b)).to(dev) for b in range(B)],
                img=torch.from_numpy(synth.camera_features(B * 6, 256, (fh, fw), 1234 + seed)).to(dev), metas=metas,
                gt_boxes=[torch.from_numpy(g[0]) for g in gts], gt_labels=[torch.from_numpy(g[1]) for g in gts]))

    def describe(self):
        return ("TransFusion-L + 3D-DF (voxelize+VFE, SparseEncoderFusion + ACTR fusion layer on %d x 6 synthetic "
                "ResNet50-stride-4-shaped cam feats [256,112,200], SECOND + SECONDFPN, TransFusionHead 200 proposals, Hungarian "
                "target assignment + detection losses), 0.075 m voxel, bs=%d [BASELINE configs[2]; configs[3] per GPU]"
                % (self.batch, self.batch))

    def close(self):
        self.enc.close()

    def _train_setup(self):
        from . import dist as D
        self.detector.train()
        params = [p for p in self.detector.parameters() if p.requires_grad]
        self.reducer = D.GradBucketReducer(params)
        # TF/configs/transfusion_nusc_voxel_F.py:302-303
        if os.environ.get("DF3D_BUCKET_ADAMW", "1") == "1":
            self.optimizer = D.BucketAdamW(self.reducer, lr=1e-4, weight_decay=0.01)   # one fused launch per 16 MB bucket
        else:
            self.optimizer = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
        self.grad_clip = dict(max_norm=0.1, norm_type=2)
        self.n_params = sum(p.numel() for p in params)

    def train_step(self, i):
        """One data-parallel training iteration on this rank's `batch` sweeps (TransFusionDetector.training_step)."""
        if getattr(self, "reducer", None) is None:
            self._train_setup()
        fr = self.frames[i % len(self.frames)]
        metas = [dict(m) for m in fr["metas"]]
        loss, logs = self.detector.training_step(fr["points"], [fr["img"]], metas, fr["gt_boxes"], fr["gt_labels"],
                                                 reducer=self.reducer, optimizer=self.optimizer, grad_clip=self.grad_clip)
        return {k: v.reshape(1).float() for k, v in logs.items()}

    def step(self, i, stage):
        if stage == "train":
            return self.train_step(i)
        with torch.no_grad():
            return self._infer_step(i, stage)

    def _infer_step(self, i, stage):
        fr = self.frames[i % len(self.frames)]
        head = self.enc.take_head(fr["points"]) if self.prefetch else None
        if self.prefetch:
            # the next batch's voxelisation and rulebooks (with their count round trips) start on the encoder's worker thread
            # while this batch is queued -- what the reference's DataLoader workers do for the voxelisation
            self.enc.prefetch(self.frames[(i + 1) % len(self.frames)]["points"], synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 120000)
        metas = [dict(m) for m in fr["metas"]]                    # fresh meta dicts per iteration
        if head is not None:
            f, c, prepared = head
        else:
            f, c = _voxelize_batch(fr["points"], synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 120000)
            prepared = None
        x = self.enc(f, c, self.batch, img_feats=[fr["img"]], img_metas=metas, prepared=prepared)
        if stage == "hot_path":
            return x
        preds = self.head(self.fpn(self.second(x)))
        if stage == "boxes":
            return self.head.get_bboxes_device(preds)
        losses = self.head.loss_device(fr["gt_boxes"], fr["gt_labels"], preds)
        return {k: v.reshape(1) for k, v in losses.items()}

    def check(self, out, stage):
        if stage == "hot_path":
            assert out.shape[0] == self.batch and out.shape[-2:] == (180, 180), out.shape
        elif stage == "boxes":
            assert out[0].shape[0] == self.batch and bool(torch.isfinite(out[1]).all())
        elif stage == "train":
            assert {"loss", "loss_heatmap", "layer_-1_loss_cls", "layer_-1_loss_bbox", "grad_norm"} <= set(out), sorted(out)
            assert all(bool(torch.isfinite(v).all()) for v in out.values()), out
        else:
            assert set(out) == {"loss_heatmap", "layer_-1_loss_cls", "layer_-1_loss_bbox", "matched_ious"}, sorted(out)
            assert all(bool(torch.isfinite(v).all()) for v in out.values()), out


class VoxelRCNNWorkload(object):
    name, unit_name = "vr_fusion", "frames"
    metric = "KITTI frames/sec (0.05 m voxel, ~19k pts, 1 camera)"
    hot_path_what = "identical to the step (this tree ends at the fused sparse backbone)"

    def __init__(self, args, rank, world, dev):
        from .backbones import VoxelBackBone8xFusion
        self.batch = B = args.batch or 8
        self.dev = dev
        self.prefetch = (bool(getattr(args, "prefetch", True)) and os.environ.get("DF3D_VOXEL_STREAM", "1") == "1"
                         and os.environ.get("DF3D_VR_AHEAD", "1") == "1")
        self.head_worker = os.environ.get("DF3D_VR_HEAD", "1") == "1"
        torch.manual_seed(0)
        cfg = dict(NAME='VoxelBackBone8xFusion', USE_IMG=True, FUSION_POS=[1, 4], FUSION_METHOD='MVX+ACTRv2',
                   FEATURE_LEVELS=[0], LT_CFG=dict(npoint=2048, radius=2.0, nsample=32, num_layers=2),
                   ACTR_CFG=dict(fusion_method='sum', feature_modal='hybrid', num_bins=80, num_channels=[256],
                                 query_num_feat=64, num_enc_layers=4, max_num_ne_voxel=20000, pos_encode_method='depth'),
                   HYBRID_CFG=dict(attn_layer='BiGateSum1D_2', q_method='sum', q_rep_place=['weight']))
        self.model = VoxelBackBone8xFusion(cfg, 4, [1408, 1600, 40]).to(dev).eval()
        H, W = 384, 1280
        self.hw = (H, W)
        K = np.array([[720., 0, W / 2, 0], [0, 720., H / 2, 0], [0, 0, 1, 0]], np.float32)
        self.frames = []
        for f in range(max(1, args.frames)):
            seed = rank * 1000 + f
            rs = np.random.RandomState(seed)
            l2i = []
            for b in range(B):
                Tr = np.array([[0, -1, 0, 0.01 * rs.randn()], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], np.float32)
                l2i.append(K @ Tr)
            g = torch.Generator().manual_seed(seed)
            self.frames.append(dict(
                points=[torch.from_numpy(synth.kitti_sweep(seed=seed * 16 + b)[:, :4].copy()).to(dev) for b in range(B)],
                l2i=torch.from_numpy(np.stack(l2i)).to(dev),
                mvx=torch.randn(B, 16, H // 4, W // 4, generator=g).to(dev),
                img=torch.randn(B, 256, H // 4, W // 4, generator=g).to(dev)))

    def describe(self):
        return ("Voxel-RCNN + 3D-DF (voxelize+VFE, VoxelBackBone8xFusion: MVX point fusion + ACTRv2 with 3-D local "
                "self-attention, d_model 64, 4 encoder layers), KITTI 0.05 m voxel, bs=%d, 1 camera [BASELINE configs[4]]"
                % self.batch)

    def _dict(self, i, f, c, **extra):
        fr = self.frames[i % len(self.frames)]
        bd = dict(voxel_features=f, voxel_coords=c, batch_size=self.batch, lidar2img=fr["l2i"], image_hw=self.hw,
                  img_dict={"mvx_layer1_feat2d": fr["mvx"], "layer1_feat2d": fr["img"]})
        bd.update(extra)
        return bd

    def _batch(self, i):
        fr = self.frames[i % len(self.frames)]
        f, c = _voxelize_batch(fr["points"], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 40000)
        return self._dict(i, f, c)

    def close(self):
        self.model.close()

    @torch.no_grad()
    def step(self, i, stage):
        if not self.prefetch:
            return self.model(self._batch(i))
        # The data loader runs two batches ahead of the GPU step (round 5), as the reference's DataLoader workers do for the
        # voxelisation (VR/pcdet/datasets/dataset.py `prepare_data` -> `transform_points_to_voxels` in worker processes):
        #   batch i + 2   voxelisation + mean VFE + every rulebook of the conv chain on the backbone's native worker thread
        #                 (`prefetch_frame`: the count round trips of five index sets leave the queueing thread),
        #   batch i + 1   its head is taken and the stride-8 query geometry -- furthest point sampling (2048 serial iterations),
        #                 ball query -- starts on a side stream, beside this batch (`prefetch`),
        #   batch i       the convolutions on the prepared rulebooks, the fusion layers between them.
        heads = self.__dict__.setdefault("_heads", {})
        if self.__dict__.get("_last_step") != i - 1 and heads:
            # the caller jumped (another pass of bench.py starts over): heads prepared for other step numbers are views into
            # frame slots that later frames have recycled -- drop them all (the worker's pending jobs included)
            for bd in heads.values():
                if bd is not None and bd.get("prepared") is not None:
                    bd["prepared"].release()
            heads.clear()
            hw = self.model.__dict__.get("_head_worker")
            if hw is not None:
                hw.drop_all()
            self.model.__dict__.pop("_fuse4_ahead", None)
        self._last_step = i
        ahead = 2 if self.head_worker else 0
        for j in range(i, i + ahead + 1):
            if j not in heads and self.head_worker:
                fr = self.frames[j % len(self.frames)]
                heads[j] = None if self.model.prefetch_frame(("vr", j), fr["points"], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5,
                                                             40000) else self._batch(j)
        for j in (i, i + 1):
            if j not in heads:
                heads[j] = self._batch(j)                      # (no worker: voxelised on the voxel stream, a batch ahead)
                fresh = True
            elif heads[j] is None:
                prep = self.model.take_head(("vr", j))
                heads[j] = (self._batch(j) if prep is None else
                            self._dict(j, prep.feats, prep.coors, prepared=prep.geometry, _head=prep))
                fresh = True
            else:
                fresh = False
            if fresh:
                self.model.prefetch(heads[j]["voxel_coords"], heads[j], prepared=heads[j].get("prepared"))
        bd = heads.pop(i)
        if "_head" in bd:
            bd.pop("_head").hand_over()                        # this stream waits (on the device) for the worker's kernels
        return self.model(bd)

    def check(self, out, stage):
        assert "encoded_spconv_tensor" in out and bool(torch.isfinite(out["encoded_spconv_tensor"].features).all())


def make(args, rank, world, dev):
    return {"tf_fusion": TransFusionWorkload, "vr_fusion": VoxelRCNNWorkload}[args.workload](args, rank, world, dev)
