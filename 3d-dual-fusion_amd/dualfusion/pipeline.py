"""The hot path assembled end to end, the way the reference detectors call it
(CP/det3d/models/detectors/voxelnet.py:129-188: reader -> backbone(+fuse_func) -> dense BEV).

`CenterPointHotPath` takes raw sweeps that are already resident in HBM and produces the dense
BEV tensor the 2-D neck consumes.  The camera network is out of scope: its
output feature maps are an *input* here."""
import os

import torch
from torch import nn

from . import synth
from .backbones import SpMiddleResNetFHD, SpMiddleResNetFHDFusion
from .voxel import Voxelization


class CenterPointHotPath(nn.Module):
    """voxelize + mean VFE (fused) -> SpMiddleResNetFHD[Fusion] -> dense [B, 256, 180, 180]."""

    def __init__(self, fusion=None, voxel_size=synth.NUSC_VOXEL, pc_range=synth.NUSC_RANGE, max_points=10,
                 max_voxels=(120000, 160000), num_input_features=5, neck=None):
        super(CenterPointHotPath, self).__init__()
        self.voxel_layer = Voxelization(voxel_size, pc_range, max_points, max_voxels)
        if fusion is None:
            self.backbone = SpMiddleResNetFHD(num_input_features=num_input_features)
        else:
            self.backbone = SpMiddleResNetFHDFusion(num_input_features=num_input_features)
        self.fusion = fusion
        # set by a caller whose point clouds are complete in device memory before forward() is called (a data loader's
        # synchronised copies; bench.py): voxelisation then runs on its own stream (ops.hard_voxelize_clouds)
        self.resident_inputs = False
        self.defer_neck = False          # CenterPointDetector's launch tape runs the neck itself (with the head)
        # optional BEV neck (necks.RPN): the backbone then hands over channels-last pixel
        # rows and forward returns the neck's [B, 512, 180, 180] map instead of the dense BEV tensor
        self.neck = neck
        if neck is not None:
            self.backbone.dense_layout = "rows"
            # a neck that reads split rows only gets them straight from the scatter (necks.SplitRows)
            self.backbone.dense_split = getattr(neck, "accepts_split_rows", None)
        gs = self.voxel_layer.grid_size.tolist()
        self.grid_size_xyz = [int(gs[0]), int(gs[1]), int(gs[2])]
        object.__setattr__(self, "_ahead", None)           # dualfusion.prefetch.FrameHead, created by prefetch()

    @torch.no_grad()
    def voxelize(self, points_list, while_waiting=None):
        """list of [P_b, C] device tensors -> (features [M, C], coors [M, 4] (b,z,y,x) int32).
        CenterPoint voxelises with the numba kernel's cap semantics (point_cloud_ops.py:46-47).
        `while_waiting`: enqueued behind the first sweep's voxelizer while the host waits for its voxel count."""
        from . import ops as _ops
        vl = self.voxel_layer                       # a batch: all clouds queued, ONE round trip for their voxel counts
        clouds = [p.contiguous().float() for p in points_list]
        # `resident_inputs` promises clouds that are COMPLETE in device memory; a cloud this method had to convert (dtype /
        # layout copy queued on the caller's stream just now) is not -- the voxel stream would race with that copy
        resident = self.resident_inputs and all(q is p for q, p in zip(clouds, points_list))
        return _ops.hard_voxelize_clouds(clouds, vl.voxel_size, vl.point_cloud_range,
                                         vl.max_num_points, vl._cap(), break_at_cap=False, while_waiting=while_waiting,
                                         resident_inputs=resident)

    # ------------------------------------------------------------------ geometry of the next frame on a helper thread
    @staticmethod
    def _frame_key(points_list, batch_dict):
        # address, in-place version counter and shape of every cloud: a loader that rewrites the same buffers between
        # prefetch() and forward() changes the key, and the stale head is dropped instead of being used; the
        # frame head keeps `batch_dict` itself as the entry's owner, so its id cannot be recycled while the entry lives
        return (tuple((int(p.data_ptr()), int(p._version), tuple(p.shape)) for p in points_list),
                id(batch_dict) if batch_dict is not None else 0)

    def prefetch(self, points_list, batch_dict=None):
        """Start the head of a LATER forward(points_list, batch_dict=batch_dict) now, on the detector's native worker thread
        (dualfusion/prefetch.py, `df3d_frame_head_*`): voxelisation, the backbone's rulebooks, the fusion adapter's projection /
        query slots -- and their host round trips.  Returns at once (False: not applicable, forward() does it in line).
        The inputs must be complete in device memory (`resident_inputs`, what a data loader hands over) and must be passed to
        forward() as the same objects.  Inference on the native executor plan only."""
        if not points_list or not points_list[0].is_cuda or not self.resident_inputs or self.training:
            return False
        if any(p.dtype != torch.float32 or not p.is_contiguous() for p in points_list):
            return False
        with torch.no_grad():
            plan = self.backbone._plan()
        if plan is None:
            return False
        dev = points_list[0].device
        if self._ahead is None:
            from .prefetch import FrameHead
            object.__setattr__(self, "_ahead", FrameHead(dev))
        if len(self._ahead._pending) >= 2:                  # stale submissions (a caller that changed its frame order)
            self._ahead.drop_all()
        vl = self.voxel_layer
        vox = dict(voxel_size=vl.voxel_size, coors_range=vl.point_cloud_range, max_points=vl.max_num_points,
                   max_voxels=vl._cap(), break_at_cap=False)
        shape = [self.grid_size_xyz[2] + 1, self.grid_size_xyz[1], self.grid_size_xyz[0]]     # backbones._stem: (z + 1, y, x)
        cam = None
        fusion = self.fusion
        if fusion is not None and batch_dict is not None and hasattr(fusion, "head_request"):
            layers = [plan.exports[name] for name in self.backbone.FUSE_STAGES]
            cam = fusion.head_request(batch_dict, 'layer1_ori', layers, self.backbone.FUSE_D_FACTORS, dev)
        self._ahead.submit(self._frame_key(points_list, batch_dict), plan, points_list, vox, shape, cam, owner=batch_dict)
        return True

    def close(self):
        if self._ahead is not None:
            self._ahead.close()
            object.__setattr__(self, "_ahead", None)

    @torch.no_grad()
    def forward(self, points_list, batch_dict=None, example=None):
        prep = self._ahead.take(self._frame_key(points_list, batch_dict)) if self._ahead is not None else None
        if prep is not None:
            B = len(points_list)
            if prep.geometry is None:                       # an empty sweep: nothing was prepared beyond the voxeliser
                return self._forward_inline(points_list, batch_dict, example)
            prep.hand_over()
            if self.fusion is not None and prep.fusion is not None:
                self.fusion.use_prepared(batch_dict, 'layer1_ori', prep.fusion)
                if os.environ.get("DF3D_EARLY_IMGPROJ", "1") == "1":
                    self.fusion.prefetch_inline(batch_dict, 'layer1_ori')
            if self.fusion is None:
                bev, multi = self.backbone(prep.feats, prep.coors, B, self.grid_size_xyz, prepared=prep.geometry)
            else:
                bev, multi = self.backbone(prep.feats, batch_dict, prep.coors, B, self.grid_size_xyz, example,
                                           fuse_func=self.fusion, prepared=prep.geometry)
            if self.neck is not None and not self.defer_neck:
                rows, (nb, _, h, w) = bev
                bev = self.neck.forward_rows(rows, nb, h, w)
            return bev, multi
        return self._forward_inline(points_list, batch_dict, example)

    def _forward_inline(self, points_list, batch_dict=None, example=None):
        if (self.fusion is not None and batch_dict is not None and hasattr(self.fusion, "prefetch")
                and os.environ.get("DF3D_PREFETCH", "0") == "1"):
            # opt-in experiment: start the image-side projection (depends on the camera maps only) on a side stream.
            # Measured on MI355X it LOSES 3-9 % (the GPU is ~85 % busy already; co-running GEMMs slow the conv
            # kernels more than the filled gaps gain), so it is off by default.
            self.fusion.prefetch(batch_dict, 'layer1_ori')
        early = None
        if (self.fusion is not None and batch_dict is not None and hasattr(self.fusion, "prefetch_inline")
                and os.environ.get("DF3D_EARLY_IMGPROJ", "1") == "1" and os.environ.get("DF3D_PREFETCH", "0") != "1"):
            # the image-side projection depends on the camera maps only: queue it right behind the voxelizer, on the
            # same stream, so that the GPU has ~250 us of work while the host waits for the voxel count
            early = lambda: self.fusion.prefetch_inline(batch_dict, 'layer1_ori')   # noqa: E731
        feats, coors = self.voxelize(points_list, while_waiting=early)
        B = len(points_list)
        if self.fusion is None:
            bev, multi = self.backbone(feats, coors, B, self.grid_size_xyz)
        else:
            bev, multi = self.backbone(feats, batch_dict, coors, B, self.grid_size_xyz, example, fuse_func=self.fusion)
        if self.neck is not None and not self.defer_neck:
            rows, (nb, _, h, w) = bev
            bev = self.neck.forward_rows(rows, nb, h, w)
        return bev, multi


# nuScenes head of the 3D-DF CenterPoint config
# (CP/configs/nusc/voxelnet/nusc_centerpoint_voxelnet_0075voxel_fix_bn_z_multimodal_pfat_hybrid7_ifat.py:7-14,110-133)
NUSC_TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
              dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
              dict(num_class=2, class_names=["motorcycle", "bicycle"]),
              dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]
NUSC_COMMON_HEADS = {'reg': (2, 2), 'height': (1, 2), 'dim': (3, 2), 'rot': (2, 2), 'vel': (2, 2)}
NUSC_CODE_WEIGHTS = [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0]
NUSC_TEST_CFG = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                     nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
                              nms_post_max_size=83, nms_iou_threshold=0.2),
                     score_threshold=0.1, pc_range=[-54, -54], out_size_factor=8, voxel_size=[0.075, 0.075])


class CenterPointDetector(nn.Module):
    """The detector the way the reference composes it (`VoxelNet` / `VoxelNetFusion.forward`,
    CP/det3d/models/detectors/voxelnet.py:150-188): reader -> backbone (+ fuse_func) -> neck -> bbox_head ->
    `loss` (return_loss=True) or `predict`.  Inference-mode composition on the MI355X kernels: the hot path hands the
    neck channels-last pixel rows, the head reads the neck's rows, the losses / detections are computed on the device
    (`CenterHead.loss_device` / `predict_device`); the camera network's feature maps are an input (`batch_dict`)."""

    def __init__(self, fusion=None, neck=None, bbox_head=None, test_cfg=None, **hot_path_kwargs):
        super(CenterPointDetector, self).__init__()
        from .heads import CenterHead
        from .necks import RPN
        if neck is None:
            neck = RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256)
        if bbox_head is None:
            bbox_head = CenterHead(in_channels=512, tasks=NUSC_TASKS, dataset='nuscenes', weight=0.25,
                                   code_weights=NUSC_CODE_WEIGHTS, common_heads=dict(NUSC_COMMON_HEADS),
                                   share_conv_channel=64, dcn_head=False)
        self.hot_path = CenterPointHotPath(fusion=fusion, neck=neck, **hot_path_kwargs)
        self.bbox_head = bbox_head
        self.test_cfg = dict(NUSC_TEST_CFG) if test_cfg is None else test_cfg

    @property
    def neck(self):
        return self.hot_path.neck

    def training_step(self, points_list, example, batch_dict=None, host_copies=True):
        """One forward + backward of the detector in train() mode, the way the reference's trainer drives it
        (`model(example, return_loss=True)` -> `parse_second_losses` -> `loss.backward()`, CP/det3d/torchie/trainer/
        trainer.py:366-380): voxelisation without gradients, then backbone (sparse-conv autograd Functions, BatchNorm in
        torch; with `fusion=` the camera adapter's differentiable composition, `VoxelWithPointProjection.
        forward_autograd`, between conv4 and the dense map) -> dense BEV -> RPN neck -> CenterHead -> `loss` (the
        reference's composition) -> backward.  The caller owns gradient reduction and the optimizer.  Returns the merged
        loss dict.
        host_copies: True = the reference's logging copies (`hm_loss` / `loc_loss_elem` as CPU tensors: `.cpu()` right after the
        backward is queued -- the host then waits for the whole backward before it queues the optimizer step and the next frame);
        "async" = the same values into pinned host memory without waiting: they are valid once `rets["host_copies_ready"]`
        (an event) has completed, and the host goes on queueing -- a trainer that logs every N steps synchronises there."""
        hp = self.hot_path
        if hp.fusion is not None and batch_dict is None:
            raise ValueError("training_step: a detector with a camera-fusion adapter needs batch_dict (camera features / calibration)")
        with torch.no_grad():
            feats, coors = hp.voxelize(points_list)
        layout, hp.backbone.dense_layout = getattr(hp.backbone, "dense_layout", "nchw"), "nchw"
        try:
            with torch.enable_grad():
                if hp.fusion is None:
                    bev, _ = hp.backbone(feats, coors, len(points_list), hp.grid_size_xyz)
                else:
                    bev, _ = hp.backbone(feats, batch_dict, coors, len(points_list), hp.grid_size_xyz, example,
                                         fuse_func=hp.fusion)
                preds = self.bbox_head(self.neck(bev))
                rets = self.bbox_head.loss_rows(example) if hasattr(self.bbox_head, "loss_rows") else None
                if rets is None:
                    rets = self.bbox_head.loss(example, preds, {}, host_copies=False)
                sum(rets["loss"]).backward()
        finally:
            hp.backbone.dense_layout = layout
        from . import ops as _ops
        flag = _ops.overflow_word() if (_ops.CONV_PRECISION == "split" and points_list[0].is_cuda) else None
        if host_copies == "async":
            if flag is not None:
                # the range flag of the fp16 operand format, valid with the other logging copies (nonzero: ops.raise_if_range_flag)
                rets["range_flag"] = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                rets["range_flag"].copy_(flag, non_blocking=True)
            # (the device tensors stay available for a loss reduction over the ranks: reading them back from the pinned copies
            # would be a blocking H2D copy queued behind the whole step)
            rets["on_device"] = {key: list(rets[key]) for key in ("hm_loss", "loc_loss_elem", "loc_loss") if key in rets}
            for key in ("hm_loss", "loc_loss_elem"):
                outs = []
                for v in rets[key]:
                    h = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                    h.copy_(v.detach(), non_blocking=True)
                    outs.append(h)
                rets[key] = outs
            rets["host_copies_ready"] = torch.cuda.current_stream().record_event()
            return rets
        for key in ("hm_loss", "loc_loss_elem"):            # the reference's host copies, once the backward is queued
            rets[key] = [v.cpu() for v in rets[key]]
        if flag is not None:                                 # (the host has just waited for the step: the flag costs no wait)
            _ops.RANGE_STATS["range_checks"] += 1
            _ops.raise_if_range_flag(int(flag.cpu()[0]))      # raised BEFORE the caller's optimizer step
        return rets

    def prefetch(self, points_list, batch_dict=None):
        """`CenterPointHotPath.prefetch`: the head of a later forward() of the same inputs, on the helper thread."""
        return self.hot_path.prefetch(points_list, batch_dict)

    def close(self):
        self.hot_path.close()

    # ---- launch tape over neck + head (dualfusion/tape.py), opt-in: the predictions then live in the tape's buffers and are
    #      valid until the next forward() of this detector
    launch_tape = os.environ.get("DF3D_LAUNCH_TAPE", "0") == "1"

    def train(self, mode=True):
        tape = self.__dict__.get("_tail_tape")
        if tape is not None:
            tape.reset()
        self.__dict__.pop("_tape_tensors", None)
        return super(CenterPointDetector, self).train(mode)

    def _tail_versions(self):
        """Cheap change detector over every parameter / buffer of neck and head: the sum of the version counters (in-place
        updates -- optimizer steps, load_state_dict -- bump them; train() / eval() drop the tapes)."""
        ts = self.__dict__.get("_tape_tensors")
        if ts is None:
            mods = (self.hot_path.neck, self.bbox_head)
            ts = self.__dict__["_tape_tensors"] = [t for m in mods for t in list(m.parameters()) + list(m.buffers())]
        v = 0
        for t in ts:
            v += t._version
        return v

    def _neck_head(self, bev):
        rows, (nb, _, h, w) = bev
        return self.bbox_head(self.hot_path.neck.forward_rows(rows, nb, h, w))

    def _taped_tail(self, bev):
        from . import ops as _ops
        from .necks import SplitRows
        from .tape import TapedSection
        rows, (nb, c, h, w) = bev
        inp = rows.split if isinstance(rows, SplitRows) else rows
        tape = self.__dict__.get("_tail_tape")
        if tape is None:
            tape = self.__dict__["_tail_tape"] = TapedSection()
        key = (nb, c, h, w, tuple(inp.shape), inp.dtype, _ops.CONV_PRECISION, self._tail_versions(),
               os.environ.get("DF3D_HEAD_FINAL"), os.environ.get("DF3D_HEADFINAL"))
        return tape.run(key, lambda: self._neck_head(bev), [inp], _ops._stream())

    @torch.no_grad()
    def simple_test(self, points_list, batch_dict=None, example=None):
        """Detections on the host, the reference's `VoxelNet.forward(..., return_loss=False)` -> `bbox_head.predict`
        (CP/det3d/models/detectors/voxelnet.py:150-188): per sample {'box3d_lidar', 'scores', 'label_preds', 'metadata'}.
        The one device -> host copy of the tail (the box counts) also carries the range flag of the fp16 operand format: a
        frame with a value that format cannot hold (|activation| >= 2047 after a checkpoint with unusual BatchNorm scales,
        say) is rerun on three bf16 parts by itself (`ops.with_range_fallback`; `ops.RANGE_STATS` counts them)."""
        from . import ops as _ops

        def run():
            preds = self._predictions(points_list, batch_dict, example)
            return self.bbox_head.predict(example if example is not None else {}, preds, self.test_cfg)
        return _ops.with_range_fallback(run)

    def _predictions(self, points_list, batch_dict=None, example=None):
        hp = self.hot_path
        taped = (self.launch_tape and not self.training and hp.neck is not None and hasattr(hp.neck, "forward_rows")
                 and getattr(hp.backbone, "dense_layout", "nchw") == "rows")
        hp.defer_neck = taped
        try:
            x, _ = hp(points_list, batch_dict=batch_dict, example=example)
        finally:
            hp.defer_neck = False
        return self._taped_tail(x) if taped else self.bbox_head(x)

    @torch.no_grad()
    def forward(self, points_list, batch_dict=None, example=None, return_loss=True):
        hp = self.hot_path
        taped = (self.launch_tape and not self.training and hp.neck is not None and hasattr(hp.neck, "forward_rows")
                 and getattr(hp.backbone, "dense_layout", "nchw") == "rows")
        hp.defer_neck = taped
        try:
            x, _ = hp(points_list, batch_dict=batch_dict, example=example)
        finally:
            hp.defer_neck = False
        preds = self._taped_tail(x) if taped else self.bbox_head(x)
        if return_loss:
            return self.bbox_head.loss_device(example, preds)
        return self.bbox_head.predict_device(preds, self.test_cfg)
