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
    with pytest.raises(NotImplementedError):
        h.loss({}, preds)
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
