"""Host-side mirror of the reference's neck / head interfaces: constructor keys,
parameter names and shapes of the checkpoints, registry resolution, torch composition on the CPU -- and that the
device-only entry points refuse CPU tensors instead of falling back.  No GPU needed."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"])]
COMMON = {'reg': (2, 2), 'height': (1, 2), 'dim': (3, 2), 'rot': (2, 2), 'vel': (2, 2)}


def test_rpn_matches_reference_checkpoint_layout():
    from dualfusion.necks import RPN
    from dualfusion.registry import NECKS, build_from_cfg
    m = build_from_cfg(dict(type="RPN", layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256],
                            us_layer_strides=[1, 2], us_num_filters=[256, 256], num_input_features=256), NECKS)
    assert isinstance(m, RPN)
    sd = m.state_dict()
    # CP/det3d/models/necks/rpn.py:60-118: blocks.<i> = [ZeroPad2d, Conv2d, BN, ReLU, (Conv2d, BN, ReLU) x n]
    assert tuple(sd["blocks.0.1.weight"].shape) == (128, 256, 3, 3) and tuple(sd["blocks.1.1.weight"].shape) == (256, 128, 3, 3)
    assert tuple(sd["blocks.0.16.weight"].shape) == (128, 128, 3, 3) and "blocks.0.17.running_mean" in sd
    assert tuple(sd["deblocks.0.0.weight"].shape) == (256, 128, 1, 1)       # stride 1 -> Conv2d
    assert tuple(sd["deblocks.1.0.weight"].shape) == (256, 256, 2, 2)       # stride 2 -> ConvTranspose2d [cin, cout, s, s]
    assert m.blocks[0][2].eps == 1e-3 and m.blocks[0][2].momentum == 0.01
    with torch.no_grad():
        y = m.eval()(torch.randn(1, 256, 12, 16))                           # CPU: the torch composition
    assert tuple(y.shape) == (1, 512, 12, 16)
    assert m.downsample_factor == 1


def test_second_and_fpn_layout():
    from dualfusion.registry import MM_BACKBONES, MM_NECKS, build_from_cfg
    bb = build_from_cfg(dict(type="SECOND", in_channels=256, out_channels=[128, 256], layer_nums=[5, 5],
                             layer_strides=[1, 2]), MM_BACKBONES)
    fpn = build_from_cfg(dict(type="SECONDFPN", in_channels=[128, 256], out_channels=[256, 256],
                              upsample_strides=[1, 2], use_conv_for_no_stride=True), MM_NECKS)
    assert tuple(bb.state_dict()["blocks.1.0.weight"].shape) == (256, 128, 3, 3) and bb.blocks[1][0].stride == (2, 2)
    assert "blocks.0.16.running_var" in bb.state_dict() and len(bb.blocks[0]) == 18
    with torch.no_grad():
        out = fpn.eval()(bb.eval()(torch.randn(2, 256, 8, 10)))
    assert isinstance(out, list) and tuple(out[0].shape) == (2, 512, 8, 10)
    with pytest.raises(KeyError):
        build_from_cfg(dict(type="SECOND", norm_cfg=dict(type="GN", num_groups=4)), MM_BACKBONES)


def test_centerhead_layout_and_cpu_forward():
    from dualfusion.heads import CenterHead
    from dualfusion.registry import HEADS, build_from_cfg
    h = build_from_cfg(dict(type="CenterHead", in_channels=512, tasks=TASKS, dataset='nuscenes', weight=0.25,
                            code_weights=[1.0] * 10, common_heads=dict(COMMON), share_conv_channel=64), HEADS)
    assert isinstance(h, CenterHead) and h.num_classes == [1, 2] and h.box_n_dim == 9
    sd = h.state_dict()
    assert tuple(sd["shared_conv.0.weight"].shape) == (64, 512, 3, 3) and "shared_conv.0.bias" in sd
    assert tuple(sd["tasks.1.hm.3.weight"].shape) == (2, 64, 3, 3) and tuple(sd["tasks.0.dim.3.weight"].shape) == (3, 64, 3, 3)
    assert float(sd["tasks.0.hm.3.bias"][0]) == pytest.approx(-2.19)
    with torch.no_grad():
        preds = h.eval()(torch.randn(1, 512, 6, 7))
    assert len(preds) == 2 and set(preds[0]) == {"reg", "height", "dim", "rot", "vel", "hm"}
    assert tuple(preds[1]["hm"].shape) == (1, 2, 6, 7)
    with pytest.raises(NotImplementedError):
        CenterHead(in_channels=512, tasks=TASKS, common_heads=dict(COMMON), dcn_head=True)
    with pytest.raises(KeyError):                                   # loss needs the assigner's targets in `example`
        h.loss({}, [dict(p) for p in preds])
    # the decode / NMS tail is device-only: CPU maps are refused, nothing falls back
    cfg = dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.1, pc_range=[-54, -54],
               out_size_factor=8, voxel_size=[0.075, 0.075],
               nms=dict(nms_pre_max_size=100, nms_post_max_size=20, nms_iou_threshold=0.2))
    with pytest.raises(Exception) as e:
        h.predict({}, preds, cfg)
    assert "CUDA" in str(e.value) or "libdf3d" in str(e.value) or "device" in str(e.value).lower()


def test_iou3d_nms_mirror_refuses_cpu():
    from dualfusion import iou3d_nms
    b = torch.zeros(4, 7)
    with pytest.raises(Exception):
        iou3d_nms.boxes_iou_bev(b, b)
    with pytest.raises(Exception):
        iou3d_nms.rotate_nms_pcdet(b, torch.zeros(4), 0.2)


def _tf_head_cfg(**over):
    cfg = dict(type="TransFusionHead", num_proposals=16, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
               num_decoder_layers=1, num_heads=8, learnable_query_pos=False, initialize_by_heatmap=True, nms_kernel_size=3,
               ffn_channel=256, dropout=0.1, bn_momentum=0.1, activation='relu',
               common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
               bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075],
                               out_size_factor=8, post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                               score_threshold=0.0, code_size=10),
               loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
               loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
               loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0),
               test_cfg=dict(dataset='nuScenes', grid_size=[96, 112, 40], out_size_factor=8, pc_range=[-54.0, -54.0],
                             voxel_size=[0.075, 0.075], nms_type=None))
    cfg.update(over)
    return cfg


def test_transfusion_head_matches_reference_checkpoint_layout():
    """TF/configs/transfusion_nusc_voxel_F.py:244-300 resolves through the registry; parameter names and shapes are the
    reference's (transfusion_head.py:594-756), the CPU path is the plain torch composition, the unsupported branches
    say so instead of silently doing something else."""
    from dualfusion.registry import HEADS, MM_HEADS, build_from_cfg
    from dualfusion.transfusion_head import TransFusionBBoxCoder, TransFusionHead
    head = build_from_cfg(_tf_head_cfg(), HEADS)
    assert isinstance(head, TransFusionHead) and MM_HEADS.get("TransFusionHead") is TransFusionHead
    sd = head.state_dict()
    assert tuple(sd["shared_conv.weight"].shape) == (128, 512, 3, 3) and "shared_conv.bias" in sd
    assert tuple(sd["heatmap_head.0.conv.weight"].shape) == (128, 128, 3, 3) and "heatmap_head.0.conv.bias" not in sd
    assert tuple(sd["heatmap_head.1.weight"].shape) == (10, 128, 3, 3) and tuple(sd["class_encoding.weight"].shape) == (128, 10, 1)
    assert tuple(sd["decoder.0.self_attn.in_proj_weight"].shape) == (384, 128)
    assert tuple(sd["decoder.0.multihead_attn.out_proj.weight"].shape) == (128, 128)
    assert tuple(sd["decoder.0.cross_posembed.position_embedding_head.0.weight"].shape) == (128, 2, 1)
    assert tuple(sd["decoder.0.linear1.weight"].shape) == (256, 128)
    assert tuple(sd["prediction_heads.0.heatmap.0.conv.weight"].shape) == (64, 128, 1) and "prediction_heads.0.vel.0.bn.running_var" in sd
    assert tuple(sd["prediction_heads.0.dim.1.weight"].shape) == (3, 64, 1)
    assert head.heatmap_head[0].bn.momentum == 0.1 and tuple(head.bev_pos.shape) == (1, 12 * 14, 2)
    assert head.bev_pos[0, 15].tolist() == [1.5, 1.5]                         # entry i*W + j = (j + .5, i + .5)
    with torch.no_grad():
        res = head.eval()([torch.randn(2, 512, 12, 14)], None, [{}])
    p = res[0][0]
    assert tuple(p["center"].shape) == (2, 2, 16) and tuple(p["heatmap"].shape) == (2, 10, 16)
    assert tuple(p["query_heatmap_score"].shape) == (2, 10, 16) and tuple(p["dense_heatmap"].shape) == (2, 10, 12, 14)
    dets = head.get_bboxes(res)
    assert len(dets) == 2 and dets[0][0].shape[1] == 9 and dets[0][2].dtype == torch.int32
    with pytest.raises(NotImplementedError):
        build_from_cfg(_tf_head_cfg(fuse_img=True, num_views=6), HEADS)
    with pytest.raises(RuntimeError):                                         # no train_cfg: no target assignment
        head.loss([torch.zeros(1, 9)], [torch.zeros(1, dtype=torch.long)], res)
    coder = TransFusionBBoxCoder(pc_range=[-54.0, -54.0], out_size_factor=8, voxel_size=[0.075, 0.075],
                                 post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], code_size=10)
    boxes = torch.tensor([[1.0, -2.0, 0.5, 4.0, 2.0, 1.5, 0.3, 0.1, -0.2]])
    t = coder.encode(boxes)                                                   # encode -> decode round trip
    d = coder.decode(torch.ones(1, 1, 1), t[:, 6:8].t()[None], t[:, 3:6].t()[None], t[:, 0:2].t()[None], t[:, 2:3].t()[None],
                     t[:, 8:10].t()[None], filter=True)[0]["bboxes"]
    assert torch.allclose(d, boxes, atol=1e-5)


def test_spconv_pool_inverse_and_dynamic_voxelize_interfaces():
    from dualfusion import Df3dError, spconv, voxel
    pool = spconv.SparseMaxPool3d(3, 2, 1)
    assert pool.kernel_size == [3, 3, 3] and pool.stride == [2, 2, 2] and pool.padding == [1, 1, 1] and not pool.subm
    up = spconv.SparseInverseConv3d(32, 16, 3, indice_key="d")
    assert up.inverse and tuple(up.weight.shape) == (3, 3, 3, 32, 16)
    with pytest.raises(Df3dError):                                            # device-only: no CPU fallback
        voxel.voxelization(torch.zeros(4, 4), [0.1, 0.1, 0.1], [0, 0, 0, 1, 1, 1], -1, 100)


def _tfl_head():
    from make_golden import TFH_KW, TFL_CODER, TFL_LOSSES, TFL_TEST_CFG, TFL_TRAIN_CFG, tfh_weight_shift
    import detgen
    from dualfusion.transfusion_head import TransFusionHead
    head = TransFusionHead(train_cfg=dict(TFL_TRAIN_CFG), test_cfg=dict(TFL_TEST_CFG), loss_iou=dict(type='VarifocalLoss'),
                           bbox_coder=dict(type='TransFusionBBoxCoder', **TFL_CODER), **TFL_LOSSES, **TFH_KW)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = tfh_weight_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return head.eval(), shapes


def test_transfusion_head_loss_vs_reference_golden(golden):
    """TransFusionHead.get_targets / loss (plain torch path, CPU) against the reference module's own outputs
    (tests/golden/transfusion_head_loss.npz: transfusion_head.py:1048-1283 + hungarian_assigner.py run by make_golden.py).
    The 3-D IoU of the matcher comes from the oracle here (the product's BboxOverlaps3D is a device kernel and refuses
    CPU tensors); everything else is the product code."""
    import detgen
    from make_golden import TFL_SHAPE
    from dualfusion import Df3dError
    from dualfusion.box3d import LiDARInstance3DBoxes
    from oracle import oracle as orc
    g = golden("transfusion_head_loss.npz")
    head, shapes = _tfl_head()
    assert sorted(shapes) == list(g["keys"])                                  # the reference's parameter names
    x = torch.from_numpy(detgen.randn("tfl_x_%d" % int(g["seed"]), TFL_SHAPE)).requires_grad_(True)
    res = head([x], None, [{}])
    p = res[0][0]
    for k in ("center", "height", "dim", "rot", "vel", "heatmap", "query_heatmap_score"):
        assert np.abs(p[k].detach().numpy() - g["pred_" + k]).max() < 2e-4, k
    assert np.array_equal(head.query_labels.numpy(), g["query_labels"])
    gt_boxes = [LiDARInstance3DBoxes(torch.from_numpy(g["gt_boxes_%d" % b]), box_dim=9) for b in range(TFL_SHAPE[0])]
    gt_labels = [torch.from_numpy(g["gt_labels_%d" % b]) for b in range(TFL_SHAPE[0])]
    with pytest.raises(Df3dError):                                            # no CPU fallback of the IoU kernel
        head.get_targets(gt_boxes, gt_labels, res[0])
    head.bbox_assigner.iou_calculator = lambda a, b: torch.from_numpy(orc.tf_bbox_overlaps_3d(a.detach().numpy(), b.detach().numpy()))
    for b in range(TFL_SHAPE[0]):                                             # cost matrix of the matcher
        one = {k: v[b:b + 1].detach() for k, v in p.items()}
        boxes = head.bbox_coder.decode(one["heatmap"].clone(), one["rot"].clone(), one["dim"].clone(), one["center"].clone(),
                                       one["height"].clone(), one["vel"].clone())[0]["bboxes"]
        cost, _ = head.bbox_assigner.cost_matrix(boxes, gt_boxes[b].tensor, gt_labels[b], one["heatmap"], head.train_cfg)
        assert np.abs(cost.numpy() - g["cost_%d" % b]).max() < 2e-4
    t = head.get_targets(gt_boxes, gt_labels, res[0])
    assert np.array_equal(t[0].numpy(), g["labels"]) and np.array_equal(t[1].numpy(), g["label_weights"])
    assert np.array_equal(t[3].numpy(), g["bbox_weights"]) and t[5] == int(g["num_pos"])
    assert np.abs(t[2].numpy() - g["bbox_targets"]).max() < 1e-5 and np.abs(t[4].numpy() - g["ious"]).max() < 1e-5
    assert abs(t[6] - float(g["matched_ious"])) < 1e-5
    assert np.array_equal(t[7].numpy(), g["heatmap"])                         # Gaussian targets: bit for bit
    losses = head.loss(gt_boxes, gt_labels, res)
    for k in ("loss_heatmap", "layer_-1_loss_cls", "layer_-1_loss_bbox", "matched_ious"):
        ref = float(g["loss_" + k])
        assert abs(losses[k].item() - ref) < 2e-5 * max(1.0, abs(ref)), (k, losses[k].item(), ref)
    sum(v for n, v in losses.items() if "loss" in n).backward()
    gx = np.array([x.grad.sum(dtype=torch.float64).item(), x.grad.abs().sum(dtype=torch.float64).item()])
    assert np.abs(gx - g["gx"]).max() < 1e-3 * g["gx"][1]
    # fp32 noise floor of this (saturated heat map) problem: the module in float64 differs from BOTH fp32 runs by 1.2e-4 /
    # 1.5e-4 at the worst element and 2.4e-6 on average (gradient scale: max 1.8e-2, mean 3.2e-3)
    d = np.abs(x.grad[:, :8].numpy() - g["gx_slice"])
    assert d.max() < 1.5e-2 * np.abs(g["gx_slice"]).max() and d.mean() < 2e-3 * np.abs(g["gx_slice"]).mean()
    gw = [head.shared_conv.weight.grad, head.heatmap_head[1].bias.grad, head.prediction_heads[0].center[1].weight.grad,
          head.decoder[0].multihead_attn.in_proj_weight.grad]
    gw = np.array([w.abs().sum(dtype=torch.float64).item() for w in gw])
    assert np.abs(gw / g["gw"] - 1).max() < 3e-3           # |.|-sums collect the rounding noise of near-zero entries
    # a frame without ground truth (the reference cannot take one): every proposal is background, no box loss
    empty = head.get_targets([torch.zeros((0, 9))] + gt_boxes[1:], [torch.zeros((0,), dtype=torch.long)] + gt_labels[1:], res[0])
    assert (empty[0][0] == head.num_classes).all() and empty[3][0].sum() == 0 and empty[7][0].sum() == 0


def test_oracle_tf_overlap_matches_cp_convention():
    """The two trees' rotated-overlap kernels differ in box format and turning sense only: TF's xyxyr boxes turned by
    -angle (iou3d_kernel.cu:106-114) cover the same area as CP's centre boxes with heading = -angle.  Ties the new
    oracle restatement to the one pinned bit for bit against the reference's CPU build (oracle/_ref, iou3d.npz)."""
    import detgen
    from oracle import oracle as orc
    a = detgen.bev_boxes("tfov_a", 150, 8.0)
    b = detgen.bev_boxes("tfov_b", 120, 8.0, special=False)

    def xyxyr(v):
        return np.stack([v[:, 0] - v[:, 3] / 2, v[:, 1] - v[:, 4] / 2, v[:, 0] + v[:, 3] / 2, v[:, 1] + v[:, 4] / 2, -v[:, 6]], 1)
    tf = orc.tf_boxes_overlap_bev(xyxyr(a), xyxyr(b))
    cp = orc.boxes_pairwise_bev(a, b, "overlap")
    assert (tf > 0).sum() > 200
    # the corner-containment margins differ (1e-5 vs 1e-2): touching configurations may gain / lose a sliver
    assert np.abs(tf - cp).max() < 5e-2 and np.median(np.abs(tf - cp)[tf > 0]) < 1e-5


def test_voxel_rcnn_basic_gate_module_contract():
    """`BasicGate` of the Voxel-RCNN tree (attention.py:88-177): constructor keys of the reference, the convolution stacks in a
    plain list (no state_dict keys -- bug for bug), `pts2img` semantics of the canvas on the CPU: clamped pixels, last row
    wins, the (H + 1) x (W + 1) canvas cropped."""
    import types
    from dualfusion.backbones import BasicGate
    gate = BasicGate(img_channel_list=[256], pts_channel_list=[8], sparse_shape=[41, 1600, 1408], voxel_size=None,
                     point_cloud_range=None, inv_idx=[2, 1, 0], pts_idx=[0])
    assert len(gate.state_dict()) == 0 and len(gate.spatial_basic_list) == 1
    mods = list(gate.spatial_basic_list[0])
    assert [type(m).__name__ for m in mods] == ["Conv2d", "BatchNorm2d", "ReLU", "Conv2d"] and mods[-1].out_channels == 1
    feats = torch.arange(5 * 8, dtype=torch.float32).view(5, 8) + 1
    ind = torch.tensor([[0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 2], [1, 0, 0, 0], [1, 0, 0, 1]], dtype=torch.int32)
    x = types.SimpleNamespace(features=feats, indices=ind, spatial_shape=[21, 800, 704])
    # image 40 x 80, feature map 4 x 8: rows 0 and 1 hit the same pixel (1, 2); row 2 lies outside (clamped to the cropped
    # border column); row 3 projects left of the image (clamped to column 0); row 4 to pixel (3, 7)
    uv = torch.tensor([[25.0, 12.0], [29.9, 19.9], [95.0, 10.0], [-7.0, 5.0], [79.9, 39.9]])
    c = gate.canvas(x, uv, (40, 80), (4, 8), 2)
    assert tuple(c.shape) == (2, 8, 4, 8)
    assert torch.equal(c[0, :, 1, 2], feats[1])                      # the later row wins
    assert float(c[0].abs().sum()) == float(feats[1].abs().sum())    # row 2 fell into the cropped border
    assert torch.equal(c[1, :, 0, 0], feats[3]) and torch.equal(c[1, :, 3, 7], feats[4])
    out = gate([torch.ones(2, 256, 4, 8)], [x], dict(batch_size=2, image_hw=(40, 80)), project=lambda t, s: (None, uv))
    assert tuple(out[0].shape) == (2, 256, 4, 8) and bool(((out[0] > 0) & (out[0] < 1)).all())


def test_bucket_adamw_equals_torch_adamw():
    """dist.BucketAdamW (one flat parameter / state buffer per gradient bucket) against torch.optim.AdamW on the same toy model:
    same parameters after three steps incl. gradient clipping; the re-seated parameters' version counters move with every step
    (the library's weight packs are keyed by them); a parameter without a gradient sees zeros."""
    import copy
    from dualfusion import dist as D
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 4), torch.nn.LayerNorm(4))
    ref = copy.deepcopy(net)
    red = D.GradBucketReducer(list(net.parameters()), bucket_mb=0.0002)          # several buckets
    assert len(red.buckets) > 1
    opt = D.BucketAdamW(red, lr=1e-2, weight_decay=0.05)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.05)
    for step in range(3):
        x = torch.randn(5, 6)
        red.zero_grad()
        ropt.zero_grad(set_to_none=True)
        net(x).square().sum().backward()
        ref(x).square().sum().backward()
        red.finish()
        n1 = opt.clip_grad_norm(max_norm=0.5)
        n0 = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=0.5)
        assert abs(float(n1) - float(n0)) <= 1e-5 * float(n0)
        before = [p._version for p in net.parameters()]
        opt.step()
        ropt.step()
        assert all(p._version > v for p, v in zip(net.parameters(), before))
        for a, b in zip(net.parameters(), ref.parameters()):
            assert float((a - b).abs().max()) <= 2e-5, step            # (lr 1e-2; the clip factor differs in its last bits)
    # the parameters ARE views of the optimizer's flat buffers
    for P, b in zip(opt.flat, red.buckets):
        for p in b["params"]:
            off = b["offsets"][id(p)]
            assert p.data_ptr() == P.data_ptr() + off * P.element_size()
