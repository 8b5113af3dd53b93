"""Host-side mirror of the reference's neck / head interfaces: constructor keys,
parameter names and shapes of the checkpoints, registry resolution, torch composition on the CPU -- and that the
device-only entry points refuse CPU tensors instead of falling back.  No GPU needed."""
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
    with pytest.raises(NotImplementedError):
        head.loss(None, None, res)
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
