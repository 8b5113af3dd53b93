#!/usr/bin/env python3
"""Timing of the other two module trees at their BASELINE config shapes (not the driver's bench line):
  tf : TransFusion-L SparseEncoderFusion + ACTR fusion layer, 0.075 m nuScenes grid, bs=4, 6 cameras (configs[2] shape, fp32)
  vr : Voxel-RCNN VoxelBackBone8xFusion (MVX + ACTRv2), KITTI 0.05 m grid, bs=8, one camera (configs[4] shape)
  neck : CenterPoint RPN BEV neck on [B, 256, 180, 180] (0.1 m nuScenes grid), row kernels vs the torch/MIOpen composition
  tfhead : TransFusionHead (LiDAR-only, 200 proposals) forward + get_bboxes on [B, 512, 180, 180], device path vs plain torch
  head : CenterPoint CenterHead (6 tasks) forward + predict on the neck's [B, 512, 180, 180] map, row kernels + device tail
         vs the torch/MIOpen forward; then sweep -> boxes end to end (LiDAR hot path + neck + head + predict)
  train : CenterPoint SpMiddleResNetFHD in train() mode, one sweep: forward + dense + loss + backward through the sparse
          conv backward kernels (training row of SURVEY section 8f; LiDAR-only, no optimizer)
usage: bench_trees.py [tf|vr|neck|head|tfhead|train] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dualfusion import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "tf"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)


def voxels(batch, sweep, vs, rng, maxp, maxv, feat_c):
    feats, coors = [], []
    for b in range(batch):
        pts = torch.from_numpy(sweep(seed=b)[:, :feat_c].copy()).to(dev)
        _, c, _, mean = ops.hard_voxelize(pts, vs, rng, maxp, maxv, want_voxels=False, batch_index=b)
        feats.append(mean)
        coors.append(c)
    return torch.cat(feats), torch.cat(coors)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("DF3D_TORCH_PROFILE") == "1":       # one profiled step: top ops by GPU time, with shapes
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            fn()
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=10,
                                                                  max_name_column_width=36, max_shapes_column_width=80))
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if which == "tf":
    from dualfusion.backbones import SparseEncoderFusion
    from make_golden import ACTR_CFG
    B = 4
    TF_CH = ((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128))
    TF_PAD = ((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0))
    enc = SparseEncoderFusion(in_channels=5, sparse_shape=[41, 1440, 1440], output_channels=128, encoder_channels=TF_CH,
                              encoder_paddings=TF_PAD, block_type='basicblock', fusion_pos=[3],
                              voxel_size=synth.NUSC_VOXEL, point_cloud_range=synth.NUSC_RANGE,
                              fusion_layer=dict(type='ACTR', pfat_cfg=dict(ACTR_CFG))).to(dev).eval()
    f, c = voxels(B, synth.nusc_sweep, synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 120000, 5)
    ori_hw, in_hw, fh, fw = (900, 1600), (448, 800), 112, 200          # stride-4 level (the layer indexes pix // 4)
    cams = synth.nusc_cameras(image_hw=ori_hw)
    sf = [in_hw[1] / ori_hw[1], in_hw[0] / ori_hw[0]]
    metas = [dict(lidar2cam=np.stack([cams[n][0] for n in synth.NUSC_CAMS]),
                  cam_intrinsic=np.stack([cams[n][1] for n in synth.NUSC_CAMS]), ori_shape=ori_hw + (3,),
                  img_shape=in_hw + (3,), input_shape=in_hw, scale_factor=sf, flip=False) for _ in range(B)]
    img = torch.randn(B * 6, 256, fh, fw, device=dev)

    def step():
        with torch.no_grad():
            return enc(f, c, B, img_feats=[img], img_metas=metas)
    ms = timeit(step)
    print("tf  SparseEncoderFusion+ACTR bs=%d (%d voxels): %.2f ms/step = %.1f sweeps/s  [DF3D_EXECUTOR=%s]" % (
        B, f.shape[0], ms, B / ms * 1e3, os.environ.get("DF3D_EXECUTOR", "1")))
    if os.environ.get("DF3D_TF_DETECTOR", "1") == "1":     # ... + SECOND + SECONDFPN + TransFusionHead -> boxes
        from dualfusion.necks import SECOND, SECONDFPN
        from dualfusion.transfusion_head import TransFusionHead
        bb = SECOND(in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2]).to(dev).eval()
        fpn = SECONDFPN(in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2],
                        use_conv_for_no_stride=True).to(dev).eval()
        head = TransFusionHead(num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
                               num_decoder_layers=1, num_heads=8, initialize_by_heatmap=True, nms_kernel_size=3,
                               ffn_channel=256, common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2),
                                                                  vel=(2, 2)),
                               bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=[-54.0, -54.0],
                                               voxel_size=[0.075, 0.075], out_size_factor=8,
                                               post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                               score_threshold=0.0, code_size=10), loss_cls=dict(use_sigmoid=True),
                               test_cfg=dict(dataset='nuScenes', grid_size=[1440, 1440, 40], out_size_factor=8,
                                             pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], nms_type=None)).to(dev).eval()

        def detect():
            with torch.no_grad():
                x = enc(f, c, B, img_feats=[img], img_metas=metas)
                x = fpn(bb(x))
                return head.get_bboxes_device(head(x))
        ms_det = timeit(detect)
        with torch.no_grad():
            x = fpn(bb(enc(f, c, B, img_feats=[img], img_metas=metas)))
            ms_neck = timeit(lambda: fpn(bb(enc(f, c, B, img_feats=[img], img_metas=metas)))) - ms
            ms_head = timeit(lambda: head.get_bboxes_device(head(x)))
        print("tf  voxel features -> boxes (encoder + ACTR + SECOND + SECONDFPN + TransFusionHead + decode) bs=%d: %.2f ms/step"
              " = %.1f sweeps/s  (neck %.2f ms, head + decode %.2f ms)" % (B, ms_det, B / ms_det * 1e3, ms_neck, ms_head))
elif which == "neck":
    from dualfusion.necks import RPN
    B = int(os.environ.get("DF3D_NECK_BATCH", "1"))
    neck = RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256).to(dev).eval()
    for m in neck.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.uniform_(0.5, 1.5)
            m.running_mean.normal_(0, 0.1)
    x = torch.randn(B, 256, 180, 180, device=dev) * (torch.rand(B, 1, 180, 180, device=dev) < 0.3)
    rows = x.permute(0, 2, 3, 1).reshape(-1, 256).contiguous()
    with torch.no_grad():
        ms_rows = timeit(lambda: neck.forward_rows(rows, B, 180, 180))
        ms_nchw = timeit(lambda: neck(x))
        ms_lib = timeit(lambda: neck.forward_reference(x))
        xcl = x.contiguous(memory_format=torch.channels_last)
        neck_cl = neck.to(memory_format=torch.channels_last)
        ms_lib_cl = timeit(lambda: neck_cl.forward_reference(xcl))
        err = float((neck(x) - neck.forward_reference(x)).abs().max())
    gf = 125.8 * B
    if B == 1:                                       # whole LiDAR path + neck, one sweep per step (bench.py's workload)
        from dualfusion import synth as _s
        from dualfusion.pipeline import CenterPointHotPath
        hp = CenterPointHotPath(neck=neck).eval().to(dev)
        hp0 = CenterPointHotPath().eval().to(dev)
        pts = [torch.from_numpy(_s.nusc_sweep(seed=0)).to(dev)]
        ms_hp0 = timeit(lambda: hp0(pts))
        ms_hp = timeit(lambda: hp(pts))
        print("LiDAR hot path %.3f ms/sweep; with the row-kernel neck %.3f ms/sweep" % (ms_hp0, ms_hp))
    print("neck RPN bs=%d: rows %.3f ms (%.0f TFLOP/s), NCHW in %.3f ms | torch/MIOpen NCHW %.3f ms, channels_last %.3f ms"
          " | max abs diff %.2e" % (B, ms_rows, gf / ms_rows, ms_nchw, ms_lib, ms_lib_cl, err))
elif which == "train":
    from dualfusion import synth as _s
    from dualfusion.pipeline import CenterPointHotPath
    hp = CenterPointHotPath().to(dev)
    pts = [torch.from_numpy(_s.nusc_sweep(seed=0)).to(dev)]
    with torch.no_grad():
        feats, coors = hp.voxelize(pts)
    hp.backbone.train()
    params = [p for p in hp.backbone.parameters() if p.requires_grad]

    def fwd():
        bev, _ = hp.backbone(feats, coors, 1, hp.grid_size_xyz)
        return bev

    def step():
        for p in params:
            p.grad = None
        fwd().square().mean().backward()
    ms_f = timeit(lambda: fwd())
    ms = timeit(step)
    hp.backbone.eval()
    with torch.no_grad():
        ms_inf = timeit(lambda: hp.backbone(feats, coors, 1, hp.grid_size_xyz))
    print("train SpMiddleResNetFHD, 1 sweep (%d voxels): forward (train mode, unfused BN) %.2f ms, forward + backward %.2f ms "
          "| inference forward %.2f ms" % (feats.shape[0], ms_f, ms, ms_inf))
    # whole CenterPoint detector: backbone -> RPN neck -> CenterHead -> loss -> backward (targets as the assigner gives them)
    from dualfusion.heads import CenterHead
    from dualfusion.necks import RPN
    TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
             dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
             dict(num_class=2, class_names=["motorcycle", "bicycle"]), dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]
    neck = RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256).to(dev).train()
    head = CenterHead(in_channels=512, tasks=TASKS, weight=0.25, code_weights=[1.0] * 8 + [0.2, 0.2],
                      common_heads={'reg': (2, 2), 'height': (1, 2), 'dim': (3, 2), 'rot': (2, 2), 'vel': (2, 2)},
                      share_conv_channel=64).to(dev).train()
    M, HW = 500, 180 * 180
    ex = dict(hm=[torch.rand(1, t["num_class"], 180, 180, device=dev) ** 8 for t in TASKS],
              ind=[torch.randint(0, HW, (1, M), device=dev) for _ in TASKS],
              mask=[(torch.rand(1, M, device=dev) < 0.1).to(torch.uint8) for _ in TASKS],
              cat=[torch.randint(0, t["num_class"], (1, M), device=dev) for t in TASKS],
              anno_box=[torch.randn(1, M, 10, device=dev) for _ in TASKS])
    hp.backbone.train()
    all_params = params + [p for m in (neck, head) for p in m.parameters()]

    def det_step():
        for p in all_params:
            p.grad = None
        rets = head.loss(ex, head(neck(fwd())), {})
        sum(rets["loss"]).backward()
    ms_det = timeit(det_step)
    print("train CenterPoint detector, 1 sweep: backbone + RPN neck + CenterHead + loss, forward + backward %.2f ms" % ms_det)
elif which == "head":
    from dualfusion.heads import CenterHead
    from dualfusion.necks import RPN
    TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
             dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
             dict(num_class=2, class_names=["motorcycle", "bicycle"]), dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]
    TEST_CFG = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                    nms=dict(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=0.2), score_threshold=0.1,
                    pc_range=[-54, -54], out_size_factor=8, voxel_size=[0.075, 0.075])
    B = int(os.environ.get("DF3D_NECK_BATCH", "1"))
    head = CenterHead(in_channels=512, tasks=TASKS, common_heads={'reg': (2, 2), 'height': (1, 2), 'dim': (3, 2),
                                                                  'rot': (2, 2), 'vel': (2, 2)}, share_conv_channel=64).to(dev).eval()
    with torch.no_grad():                      # car-sized boxes (a random-init 'dim' head gives 50 m boxes: every pair overlaps)
        for task in head.tasks:
            task.dim[3].weight.mul_(0.05)
            task.dim[3].bias.fill_(0.5)
    x = torch.randn(B, 512, 180, 180, device=dev).relu()
    with torch.no_grad():
        ms_fw = timeit(lambda: head(x))
        ms_fw_lib = timeit(lambda: head.forward_reference(x))
        preds = head(x)
        ms_pred = timeit(lambda: head.predict_device(preds, TEST_CFG))
        ms_pred_host = timeit(lambda: head.predict({}, preds, TEST_CFG))
        n_det = sum(len(d["scores"]) for d in head.predict({}, preds, TEST_CFG))
    print("head bs=%d: forward rows %.3f ms (%.0f TFLOP/s) | torch/MIOpen %.3f ms | predict (device, no sync) %.3f ms, with result "
          "lists on the host side %.3f ms (%d boxes)" % (B, ms_fw, 108.0 * B / ms_fw, ms_fw_lib, ms_pred, ms_pred_host, n_det))
    if B == 1:
        from dualfusion import synth as _s
        from dualfusion.pipeline import CenterPointHotPath
        neck = RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256).to(dev).eval()
        hp = CenterPointHotPath(neck=neck).eval().to(dev)
        pts = [torch.from_numpy(_s.nusc_sweep(seed=0)).to(dev)]

        def e2e():
            with torch.no_grad():
                bev, _ = hp(pts)
                return head.predict_device(head(bev), TEST_CFG)
        print("sweep -> boxes (LiDAR hot path + neck + head + predict): %.3f ms/sweep" % timeit(e2e))
elif which == "tfhead":
    from dualfusion.transfusion_head import TransFusionHead
    B = int(os.environ.get("DF3D_NECK_BATCH", "1"))
    head = TransFusionHead(num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
                           num_decoder_layers=1, num_heads=8, initialize_by_heatmap=True, nms_kernel_size=3, ffn_channel=256,
                           common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
                           bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075],
                                           out_size_factor=8, post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                           score_threshold=0.0, code_size=10), loss_cls=dict(use_sigmoid=True),
                           test_cfg=dict(dataset='nuScenes', grid_size=[1440, 1440, 40], out_size_factor=8,
                                         pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], nms_type=None)).to(dev).eval()
    x = torch.randn(B, 512, 180, 180, device=dev).relu()
    with torch.no_grad():
        ms_fw = timeit(lambda: head([x]))
        ms_fw_lib = timeit(lambda: head.forward_reference(x))
        preds = head([x])
        ms_box = timeit(lambda: head.get_bboxes_device(preds))
        ms_box_host = timeit(lambda: head.get_bboxes(preds))
        ref = head.forward_reference(x)
        ms_box_torch = timeit(lambda: head.bbox_coder.decode(
            ref[0]['heatmap'].sigmoid() * ref[0]['query_heatmap_score'], ref[0]['rot'], ref[0]['dim'], ref[0]['center'],
            ref[0]['height'], ref[0]['vel'], filter=True))
    print("TransFusionHead bs=%d (200 proposals): forward, device path %.3f ms | plain torch/MIOpen path %.3f ms | get_bboxes: "
          "device call %.3f ms, with per-sample lists %.3f ms, torch decode %.3f ms" % (B, ms_fw, ms_fw_lib, ms_box, ms_box_host,
                                                                                     ms_box_torch))
else:
    from dualfusion.backbones import VoxelBackBone8xFusion
    B = 8
    cfg = dict(NAME='VoxelBackBone8xFusion', USE_IMG=True, FUSION_POS=[1, 4], FUSION_METHOD='MVX+ACTRv2',
               FEATURE_LEVELS=[0], LT_CFG=dict(npoint=2048, radius=2.0, nsample=32, num_layers=2),
               ACTR_CFG=dict(fusion_method='sum', feature_modal='hybrid', num_bins=80, num_channels=[256],
                             query_num_feat=64, num_enc_layers=4, max_num_ne_voxel=20000, pos_encode_method='depth'),
               HYBRID_CFG=dict(attn_layer='BiGateSum1D_2', q_method='sum', q_rep_place=['weight']))
    mf = VoxelBackBone8xFusion(cfg, 4, [1408, 1600, 40]).to(dev).eval()
    f, c = voxels(B, synth.kitti_sweep, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 40000, 4)
    H, W = 384, 1280
    K = np.array([[720., 0, W / 2, 0], [0, 720., H / 2, 0], [0, 0, 1, 0]], np.float32)
    Tr = np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], np.float32)
    l2i = torch.from_numpy(np.stack([K @ Tr] * B)).to(dev)
    bd0 = dict(voxel_features=f, voxel_coords=c, batch_size=B, lidar2img=l2i, image_hw=(H, W),
               img_dict={"mvx_layer1_feat2d": torch.randn(B, 16, H // 4, W // 4, device=dev),
                         "layer1_feat2d": torch.randn(B, 256, H // 4, W // 4, device=dev)})

    def step():
        with torch.no_grad():
            return mf(dict(bd0))
    ms = timeit(step)
    print("vr  VoxelBackBone8xFusion (MVX+ACTRv2) bs=%d (%d voxels): %.2f ms/step = %.1f frames/s  [DF3D_EXECUTOR=%s]" % (
        B, f.shape[0], ms, B / ms * 1e3, os.environ.get("DF3D_EXECUTOR", "1")))
