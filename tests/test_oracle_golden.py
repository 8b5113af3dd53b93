"""The oracle (oracle/oracle.py + oracle/df3d_oracle.c) against the golden vectors that
tests/golden/make_golden.py generated from the reference's own code, and - when the
reference build oracle/_ref is present - against that build directly on fresh inputs.
CPU only."""
import zlib

import numpy as np
import pytest

import detgen
from oracle import oracle as orc
from oracle import ref
from dualfusion import synth

from make_golden import RB_CASES, RB_SHAPE, RB_BATCH  # constants only (no reference import at module load)


def test_voxelize_reference_test_vector(golden):
    g = golden("voxelize.npz")
    v, c, n = orc.hard_voxelize(g["tg_points"], [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000)
    # literals of TF/tests/test_models/test_voxel_encoder/test_voxel_generator.py:15-22
    assert np.array_equal(c, g["tg_expected_coors"])
    assert np.array_equal(n, g["tg_expected_num"])
    assert np.array_equal(c, g["tg_ref_coors"]) and np.array_equal(n, g["tg_ref_num"])
    assert np.array_equal(v.sum(1), g["tg_ref_voxel_sum"])


@pytest.mark.parametrize("tag", ["nocap", "cap", "mp3"])
def test_voxelize_sweep(golden, tag):
    g = golden("voxelize.npz")
    maxp, maxv = [int(x) for x in g["sw_%s_params" % tag]]
    v, c, n = orc.hard_voxelize(g["sw_points"], synth.NUSC_VOXEL, synth.NUSC_RANGE, maxp, maxv)
    assert np.array_equal(c, g["sw_%s_coors" % tag])
    assert np.array_equal(n, g["sw_%s_num" % tag])
    assert np.array_equal(v.sum(1), g["sw_%s_voxel_sum" % tag])
    assert np.array_equal(v[:, 0], g["sw_%s_first" % tag])
    if tag == "cap":
        assert len(c) == maxv


def test_voxelize_numba_variant_differs_only_at_cap(golden):
    g = golden("voxelize.npz")
    a = orc.hard_voxelize(g["sw_points"], synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 20000, "cpp")
    b = orc.hard_voxelize(g["sw_points"], synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 20000, "numba")
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    a = orc.hard_voxelize(g["sw_points"], synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 3000, "cpp")
    b = orc.hard_voxelize(g["sw_points"], synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 3000, "numba")
    assert np.array_equal(a[1], b[1])          # same voxels...
    assert (b[2] >= a[2]).all() and (b[2] > a[2]).any()   # ...numba keeps filling them after the cap


@pytest.mark.parametrize("name", sorted(RB_CASES))
def test_rulebook_and_conv(golden, name):
    g = golden("rulebook_conv.npz")
    ks, st, pd, dl, subm, cin, cout = RB_CASES[name]
    ind = g["indices"]
    outids, pairs, num, oshape = orc.get_indice_pairs(ind, RB_BATCH, RB_SHAPE, ks, st, pd, dl, subm)
    # raw CPU order of the reference, bit-exact
    assert list(oshape) == list(g[name + "_oshape"])
    assert np.array_equal(outids, g[name + "_outids"])
    assert np.array_equal(num, g[name + "_num"])
    assert np.array_equal(pairs, g[name + "_pairs"])
    feats = detgen.randn("feat_" + name, (len(ind), cin))
    filt = detgen.randn("filt_" + name, (ks[0], ks[1], ks[2], cin, cout), 0.2)
    y = orc.indice_conv(feats, filt, pairs, num, len(outids), subm)
    np.testing.assert_allclose(y, g[name + "_y"], rtol=0, atol=2e-5)


def test_conv_equals_dense_conv3d(golden):
    """Independent pin of the arithmetic convention (cross-correlation, offset row-major kz,ky,kx)."""
    import torch
    g = golden("rulebook_conv.npz")
    ind = g["indices"]
    name = "subm3"
    ks, st, pd, dl, subm, cin, cout = RB_CASES[name]
    feats = detgen.randn("feat_" + name, (len(ind), cin))
    filt = detgen.randn("filt_" + name, (3, 3, 3, cin, cout), 0.2)
    d = torch.from_numpy(orc.dense(feats, ind, RB_SHAPE, RB_BATCH))
    r = torch.nn.functional.conv3d(d, torch.from_numpy(filt).permute(4, 3, 0, 1, 2), padding=1)
    r = r[ind[:, 0], :, ind[:, 1], ind[:, 2], ind[:, 3]].numpy()
    np.testing.assert_allclose(g[name + "_y"], r, atol=2e-5)


def test_msda_reference_test_vector(golden):
    g = golden("msda.npz")
    y = orc.ms_deform_attn(g["t_value"], g["t_shapes"], g["t_loc"], g["t_aw"])
    np.testing.assert_allclose(y, g["t_out"], rtol=1e-5, atol=1e-7)


def test_msda_hot_and_multilevel(golden):
    g = golden("msda.npz")
    N, M, D, Lq, L, P, H, W = [int(x) for x in g["h_dims"]]
    value = detgen.randn("msda_h_value", (N, H * W, M, D))
    loc = detgen.rand("msda_h_loc", (N, Lq, M, L, P, 2), -0.15, 1.15)
    aw = _softmax(detgen.randn("msda_h_aw", (N, Lq, M, L * P))).reshape(N, Lq, M, L, P)
    np.testing.assert_allclose(orc.ms_deform_attn(value, [(H, W)], loc, aw), g["h_out"], atol=2e-5)
    N, M, D, Lq, L, P = [int(x) for x in g["m_dims"]]
    shp = [tuple(int(v) for v in r) for r in g["m_shapes"]]
    S = sum(h * w for h, w in shp)
    value = detgen.randn("msda_m_value", (N, S, M, D))
    loc = detgen.rand("msda_m_loc", (N, Lq, M, L, P, 2), -0.1, 1.1)
    aw = _softmax(detgen.randn("msda_m_aw", (N, Lq, M, L * P))).reshape(N, Lq, M, L, P)
    np.testing.assert_allclose(orc.ms_deform_attn(value, shp, loc, aw), g["m_out"], atol=2e-5)


def _softmax(x):
    import torch
    return torch.softmax(torch.from_numpy(x), -1).numpy()


# ---- direct comparison with the compiled reference on fresh inputs (skipped where oracle/_ref is absent)
needs_ref = pytest.mark.skipif(not ref.available("sparse_conv_ext"), reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_vs_ref_build_fresh(seed):
    ind = detgen.clustered_voxels("fresh%d" % seed, 3, [7, 24, 30], n_seeds=5, walk=150)
    for ks, st, pd, subm in (([3, 3, 3], [1, 1, 1], [1, 1, 1], 1), ([3, 3, 3], [2, 2, 2], [1, 1, 1], 0),
                             ([3, 1, 1], [2, 1, 1], [0, 0, 0], 0), ([1, 3, 3], [1, 2, 2], [0, 1, 1], 0)):
        a = orc.get_indice_pairs(ind, 3, [7, 24, 30], ks, st, pd, [1, 1, 1], subm)
        b = ref.get_indice_pairs(ind, 3, [7, 24, 30], ks, st, pd, [1, 1, 1], subm)
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(x, y)
        f = detgen.randn("ff%d" % seed, (len(ind), 12))
        w = detgen.randn("fw%d" % seed, (ks[0], ks[1], ks[2], 12, 20), 0.2)
        np.testing.assert_allclose(orc.indice_conv(f, w, a[1], a[2], len(a[0]), subm),
                                   ref.indice_conv(f, w, b[1], b[2], len(b[0]), subm), atol=2e-5)


@needs_ref
def test_oracle_voxelize_vs_ref_build_fresh():
    pts = synth.nusc_sweep(seed=11)[:20000]
    a = orc.hard_voxelize(pts, synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 12000)
    b = ref.hard_voxelize(pts, synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 12000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# ---- point ops: the reference's own (GPU-only) unit-test literals, recorded by make_golden.gen_pointops()
def test_pointops_reference_test_vectors(golden):
    g = golden("pointops_tests.npz")
    assert np.array_equal(orc.furthest_point_sample(g["test_fps__xyz"], 3), g["test_fps__expected_idx"])
    # the fixture holds the test's final (dilated, min_radius 0.2) ball query; the plain one is re-derived below
    assert np.array_equal(orc.ball_query(0.2, 0.4, 5, g["test_ball_query__xyz"], g["test_ball_query__new_xyz"]),
                          g["test_ball_query__expected_idx"])
    plain = orc.ball_query(0, 0.2, 5, g["test_ball_query__xyz"], g["test_ball_query__new_xyz"])
    assert np.array_equal(plain[0, 0], [0, 0, 0, 0, 0]) and np.array_equal(plain[0, 1], [6, 6, 6, 6, 6])
    np.testing.assert_allclose(orc.group_points(g["test_grouping_points__festures"], g["test_grouping_points__idx"]),
                               g["test_grouping_points__expected_output"])
    np.testing.assert_allclose(orc.gather_points(g["test_gather_points__features"], g["test_gather_points__idx"]),
                               g["test_gather_points__expected_output"])


def test_local_transformer_oracle_vs_reference_golden(golden):
    import oracle_models as om
    from make_golden import LT_DIMS, lt_inputs
    g = golden("local_transformer.npz")
    shapes = {str(k): eval(str(s)) for k, s in zip(g["param_names"], g["param_shapes"])}
    sd = detgen.det_state_dict(shapes)
    xyz, feat = lt_inputs()
    d = LT_DIMS
    y = om.local_transformer(sd, xyz, feat, d["npoint"], d["radius"], d["nsample"], num_layers=d["num_layers"])
    np.testing.assert_allclose(y, g["out"], atol=2e-5)


def test_actr_oracle_vs_reference_golden(golden):
    import oracle_models as om
    from make_golden import actr_inputs
    g = golden("actr.npz")
    shapes = {str(k): eval(str(s)) for k, s in zip(g["param_names"], g["param_shapes"])}
    sd = detgen.det_state_dict(shapes)
    v_feat, grid, i_feat, lidar_grid, v_i_feat = actr_inputs()
    y = om.actr_forward(sd, v_feat, grid, i_feat, lidar_grid, v_i_feat)
    np.testing.assert_allclose(y, g["out"], atol=5e-5)


def test_centerpoint_fusion_oracle_vs_reference_golden(golden):
    import oracle_models as om
    from make_golden import FUS
    g = golden("fusion_cp.npz")
    shapes = {str(k): eval(str(s)) for k, s in zip(g["param_names"], g["param_shapes"])}
    sd = detgen.det_state_dict(shapes)
    sets = [g["coords2"].astype(np.int32), g["coords3"].astype(np.int32), g["coords4"].astype(np.int32)]
    feats = [detgen.randn("fus_feat%d" % i, (len(s), c)) for i, (s, c) in enumerate(zip(sets, [32, 64, 128]))]
    cams = synth.nusc_cameras(image_hw=FUS["raw_hw"], focal=FUS["focal"])
    B = FUS["batch"]
    img = {n: detgen.randn("fus_img_" + n, (B, 256) + tuple(FUS["feat_hw"])) for n in synth.NUSC_CAMS}
    calib = {n: (np.stack([cams[n][0]] * B), np.stack([cams[n][1]] * B)) for n in synth.NUSC_CAMS}
    out = om.centerpoint_fusion(sd, list(zip(sets, feats)), img, calib, FUS["img_hw"], synth.NUSC_CAMS,
                                FUS["voxel_size"], FUS["pc_range"], FUS["image_scale"], FUS["depth_thres"])
    np.testing.assert_allclose(out, g["out"], atol=1e-4)


# ---- detection tail: rotated BEV IoU / NMS
@pytest.mark.parametrize("tag,n,spread", [("dense", 192, 6.0), ("sparse", 300, 25.0)])
def test_iou_bev_oracle_vs_reference_golden(golden, tag, n, spread):
    g = golden("iou3d.npz")
    a = detgen.bev_boxes("iou_a_" + tag, n, spread)
    b = detgen.bev_boxes("iou_b_" + tag, n - 17, spread, special=False)
    assert np.array_equal(orc.boxes_pairwise_bev(a, b), g["iou_" + tag])          # bit-exact with the reference CPU path
    assert np.array_equal(orc.boxes_pairwise_bev(a, a), g["self_iou_" + tag])
    for thr in (0.2, 0.7):
        keep, _ = orc.nms_bev(a, thr, True)
        assert np.array_equal(keep, g["keep_%s_%d" % (tag, int(thr * 100))])
    ov = orc.boxes_pairwise_bev(a[:8], a[:8], mode="overlap")
    assert ov[0, 1] == 4.0 and ov[0, 2] == 0.0 and abs(ov[0, 7] - 1.0) < 1e-6     # identical / touching / nested


def test_nms_variants_oracle():
    b = detgen.bev_boxes("nmsv", 120, 5.0)
    keep_n, _ = orc.nms_bev(b, 0.3, False)
    # axis-aligned greedy NMS restated with numpy (iou3d_nms_kernel.cu:309-320)
    x1, x2 = b[:, 0] - b[:, 3] / 2, b[:, 0] + b[:, 3] / 2
    y1, y2 = b[:, 1] - b[:, 4] / 2, b[:, 1] + b[:, 4] / 2
    removed, want = np.zeros(len(b), bool), []
    for i in range(len(b)):
        if removed[i]:
            continue
        want.append(i)
        w = np.maximum(np.minimum(x2[i], x2) - np.maximum(x1[i], x1), 0)
        h = np.maximum(np.minimum(y2[i], y2) - np.maximum(y1[i], y1), 0)
        iou = w * h / np.maximum(b[i, 3] * b[i, 4] + b[:, 3] * b[:, 4] - w * h, 1e-8)
        removed[i + 1:] |= iou[i + 1:] > 0.3
    assert keep_n.tolist() == want
    keep_c, _ = orc.nms_bev(b, 4.0, "circle")
    removed, want = np.zeros(len(b), bool), []
    for i in range(len(b)):
        if removed[i]:
            continue
        want.append(i)
        d = (b[i, 0] - b[:, 0]) ** 2 + (b[i, 1] - b[:, 1]) ** 2
        removed[i + 1:] |= d[i + 1:] <= 4.0
    assert keep_c.tolist() == want
    sel, _ = orc.rotate_nms_pcdet(b, detgen.rand("nmsv_s", (120,)), 0.2, pre_maxsize=100, post_max_size=10)
    assert len(sel) == 10 and len(set(sel.tolist())) == 10


@pytest.mark.skipif(not ref.available("iou3d_nms_cuda"), reason="oracle/_ref not built")
def test_iou_bev_oracle_vs_ref_build_fresh():
    a = detgen.bev_boxes("fresh_iou_a", 257, 9.0)
    b = detgen.bev_boxes("fresh_iou_b", 131, 9.0, special=False)
    assert np.array_equal(orc.boxes_pairwise_bev(a, b), ref.boxes_iou_bev_cpu(a, b))


def _mirror_head():
    """dualfusion.heads.CenterHead (plain torch modules on the CPU) with the golden's deterministic weights."""
    import torch
    from dualfusion.heads import CenterHead
    from make_golden import HEAD_COMMON, HEAD_TASKS, head_bias_shift
    head = CenterHead(in_channels=512, tasks=HEAD_TASKS, dataset='nuscenes', weight=0.25, code_weights=[1.0] * 10,
                      common_heads=dict(HEAD_COMMON), share_conv_channel=64, dcn_head=False)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = head_bias_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return head.eval(), shapes


def test_centerhead_oracle_vs_reference_golden(golden):
    """Reference CenterHead.forward + predict (imported from /root/reference when the fixture was made): the mirror
    module has the same parameter names and the same forward values; the oracle's predict restatement returns the
    reference's detections (same boxes in the same order)."""
    import torch
    from make_golden import HEAD_SHAPE, HEAD_TEST_CFG
    g = golden("centerhead.npz")
    head, shapes = _mirror_head()
    assert sorted(shapes) == g["keys"].tolist()
    x = torch.from_numpy(detgen.randn("head_x_%d" % int(g["seed"]), HEAD_SHAPE))
    with torch.no_grad():
        preds = head(x)
    pn = [{k: v.numpy() for k, v in p.items()} for p in preds]
    for t, p in enumerate(pn):
        for name, v in p.items():
            chk = np.array([v.sum(dtype=np.float64), np.abs(v).sum(dtype=np.float64)])
            np.testing.assert_allclose(chk, g["chk_%d_%s" % (t, name)], rtol=1e-6, atol=1e-6)
    dets = orc.centerhead_predict(pn, HEAD_TEST_CFG, [1, 2, 2, 1, 2, 2])
    for i, d in enumerate(dets):
        assert np.array_equal(d["label_preds"], g["labels_%d" % i])
        np.testing.assert_allclose(d["scores"], g["scores_%d" % i], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(d["box3d_lidar"], g["boxes_%d" % i], rtol=1e-5, atol=2e-6)
    assert len(dets[0]["scores"]) > 100 and len(set(dets[0]["label_preds"].tolist())) >= 8


def _mirror_tf_head():
    """dualfusion.transfusion_head.TransFusionHead (plain torch modules on the CPU) with the golden's weights."""
    import torch
    from dualfusion.transfusion_head import TransFusionHead
    from make_golden import TFH_CODER, TFH_KW, TFH_TEST_CFG, tfh_weight_shift
    head = TransFusionHead(loss_cls=dict(use_sigmoid=True), test_cfg=dict(TFH_TEST_CFG),
                           bbox_coder=dict(type='TransFusionBBoxCoder', **TFH_CODER), **TFH_KW)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = tfh_weight_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return head.eval(), shapes, sd


def test_transfusion_head_oracle_and_mirror_vs_reference_golden(golden):
    """Reference TransFusionHead.forward + get_bboxes (imported from /root/reference when the fixture was made): the
    mirror module has the reference's parameter names, and both its torch path and the oracle restatement return the
    reference's predictions (same proposals, same values) and detections."""
    import torch
    import oracle_models as om
    from make_golden import TFH_CODER, TFH_KW, TFH_SHAPE
    g = golden("transfusion_head.npz")
    head, shapes, sd = _mirror_tf_head()
    assert sorted(shapes) == g["keys"].tolist()
    x = detgen.randn("tfh_x_%d" % int(g["seed"]), TFH_SHAPE)
    K = TFH_KW["num_proposals"]
    preds, labels = om.transfusion_head(sd, x, K)
    with torch.no_grad():
        res = head([torch.from_numpy(x)], None, [{}])
    assert np.array_equal(labels, g["query_labels"]) and np.array_equal(head.query_labels.numpy(), g["query_labels"])
    names = ["center", "height", "dim", "rot", "vel", "heatmap", "query_heatmap_score", "dense_heatmap"]
    assert sorted(res[0][0]) == sorted(names)
    for name in names:
        want = g["pred_" + name]
        for got in (preds[name], res[0][0][name].numpy()):
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), name
    dets_o = om.transfusion_get_bboxes(preds, labels, K, TFH_CODER)
    dets_m = head.get_bboxes(res)
    n = 0
    for b in range(TFH_SHAPE[0]):
        for box, score, lab in (dets_o[b], [t.numpy() for t in dets_m[b]]):
            assert np.array_equal(lab, g["labels_%d" % b])
            np.testing.assert_allclose(score, g["scores_%d" % b], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(box, g["boxes_%d" % b], rtol=1e-5, atol=2e-5)
        n += len(g["labels_%d" % b])
    assert 0 < n < TFH_SHAPE[0] * K                       # the centre-range mask removed some proposals, not all


@pytest.mark.parametrize("tag", ["hot", "multi"])
def test_msda_backward_oracle_vs_reference_autograd(golden, tag):
    """Oracle col2im restatement against the float64 autograd gradients of the reference's pure-torch core."""
    from make_golden import msda_bwd_inputs
    g = golden("msda_bwd.npz")
    value, shp, loc, aw, gout = msda_bwd_inputs(tag)
    gv, gl, ga = orc.ms_deform_attn_backward(value, shp, loc, aw, gout)
    for got, key in ((gv, "_gv"), (gl, "_gl"), (ga, "_ga")):
        want = g[tag + key]
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    assert np.abs(gl).max() > 0 and (gl == 0).any()          # some samples fall outside the map


@needs_ref
@pytest.mark.parametrize("subm", [1, 0])
def test_conv_backward_oracle_vs_ref_build(subm):
    """Oracle indiceConvBackward restatement against the reference's own compiled CPU code (oracle/_ref)."""
    ind = detgen.clustered_voxels("bwd%d" % subm, 2, [7, 20, 22], n_seeds=4, walk=120)
    ks, st, pd = ([3, 3, 3], [1, 1, 1], [1, 1, 1]) if subm else ([3, 3, 3], [2, 2, 2], [1, 1, 1])
    outids, pairs, num, _ = ref.get_indice_pairs(ind, 2, [7, 20, 22], ks, st, pd, [1, 1, 1], subm)
    f = detgen.randn("bwd_f%d" % subm, (len(ind), 12))
    w = detgen.randn("bwd_w%d" % subm, (3, 3, 3, 12, 20), 0.2)
    go = detgen.randn("bwd_g%d" % subm, (len(outids), 20))
    a = orc.indice_conv_backward(f, w, go, pairs, num, subm)
    b = ref.indice_conv_backward(f, w, go, pairs, num, subm)
    for x, y in zip(a, b):
        assert x.shape == y.shape
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=2e-4)


# ---- sparse max pool, inverse convolution, dynamic voxelisation (boundary rows of SURVEY section 8b)
def _pool_case():
    from make_golden import CONV_BWD_BATCH, CONV_BWD_SHAPE, conv_bwd_case
    ind, ks, st, pd, f, w = conv_bwd_case(0)
    outids, pairs, num, _ = orc.get_indice_pairs(ind, CONV_BWD_BATCH, CONV_BWD_SHAPE, ks, st, pd, [1, 1, 1], 0)
    return ind, f, outids, pairs, num


def test_maxpool_inverse_conv_dynamic_voxelize_oracle_vs_reference_golden(golden):
    """Outputs of the reference's compiled CPU code (oracle/_ref when the fixture was made).  The oracle's rulebook
    emits the out-voxels in the reference CPU path's order for this geometry (pinned by the rulebook fixtures), the
    fixture also stores the canonical order."""
    from make_golden import POOL_RANGE, POOL_VS, pool_points
    g = golden("pool.npz")
    ind, f, outids, pairs, num = _pool_case()
    order = np.lexsort(outids.T[::-1])
    assert np.array_equal(outids[order], g["outids"])
    y = orc.indice_maxpool(f, pairs, num, len(outids))
    assert np.array_equal(y[order], g["y"][g["order"]])
    assert (y >= 0).all() and (y == 0).any()                      # zero-initialised output: negatives never survive
    go = np.empty_like(y)
    go[order] = detgen.randn("pool_g", y.shape)[g["order"]]      # the same gradient per out-voxel
    assert np.array_equal(orc.indice_maxpool_backward(f, y, go, pairs, num), g["gin"])
    fq = np.round(f * 2) / 2
    yq = orc.indice_maxpool(fq, pairs, num, len(outids))
    assert np.array_equal(yq[order], g["yq"][g["order"]])
    ginq = orc.indice_maxpool_backward(fq, yq, go, pairs, num)
    assert np.array_equal(ginq, g["ginq"]) and (np.abs(ginq) > 0).sum() > (np.abs(g["gin"]) > 0).sum()   # ties fan out
    fo = np.empty((len(outids), 20), np.float32)
    fo[order] = detgen.randn("inv_f", (len(outids), 20))[g["order"]]
    wi = detgen.randn("inv_w", (3, 3, 3, 20, 12), 0.2)
    inv = orc.indice_conv(fo, wi, pairs, num, len(ind), 0, inverse=True)
    assert inv.shape == g["inv"].shape and np.abs(inv - g["inv"]).max() <= 2e-5 * np.abs(g["inv"]).max()
    dyn = orc.dynamic_voxelize(pool_points(), POOL_VS, POOL_RANGE)
    assert np.array_equal(dyn, g["dyn"]) and (dyn[:, 0] == -1).sum() > 100 and (dyn[:, 0] >= 0).sum() > 1000


def test_transposed_conv_oracle_vs_reference_golden(golden):
    """get_indice_pairs(transpose=True) + indice_conv of the oracle against the reference's compiled CPU code on four
    geometries (incl. output_padding): output sets and pair counts bit for bit, features <= 2e-5."""
    from make_golden import CONV_BWD_BATCH, CONV_BWD_SHAPE, CONVT_CASES, convt_case
    g = golden("conv_transpose.npz")
    ind, f = convt_case()
    for tag, ks, st, pd, op in CONVT_CASES:
        outids, pairs, num, oshape = orc.get_indice_pairs_transpose(ind, CONV_BWD_BATCH, CONV_BWD_SHAPE, ks, st, pd, [1, 1, 1], op)
        assert list(oshape) == list(g["oshape_" + tag])
        order = np.lexsort(outids.T[::-1])
        assert np.array_equal(outids[order], g["outids_" + tag]) and np.array_equal(num, g["num_" + tag])
        assert int(num.sum()) == len(ind) * int(np.prod(ks)) or tag != "k2s2"       # stride = kernel: every input writes all taps
        w = detgen.randn("convt_w_" + tag, tuple(ks) + (16, 16), 0.2)
        y = orc.indice_conv(f, w, pairs, num, len(outids), 0)
        ref_y = g["y_" + tag]
        assert np.abs(y[order] - ref_y).max() <= 2e-5 * np.abs(ref_y).max()


@needs_ref
def test_transposed_rulebook_oracle_vs_ref_build_fresh():
    ind = detgen.clustered_voxels("convt_fresh", 2, [6, 14, 11], n_seeds=5, walk=90)
    for ks, st, pd, op in (([3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 0, 1]), ([2, 3, 2], [2, 1, 2], [0, 1, 0], [0, 0, 0]),
                           ([3, 3, 3], [1, 1, 1], [1, 1, 1], [0, 0, 0])):
        o = orc.get_indice_pairs_transpose(ind, 2, [6, 14, 11], ks, st, pd, [1, 1, 1], op)
        r = ref.get_indice_pairs_transpose(ind, 2, [6, 14, 11], ks, st, pd, [1, 1, 1], op)
        n = len(o[0])
        assert np.array_equal(o[0], r[0][:n]) and np.array_equal(o[2], r[2]) and list(o[3]) == list(r[3])
        for k in range(len(o[2])):
            assert np.array_equal(o[1][k, :, :o[2][k]], r[1][k, :, :r[2][k]])


@needs_ref
def test_maxpool_dynamic_voxelize_oracle_vs_ref_build_fresh():
    ind = detgen.clustered_voxels("pool_fresh", 2, [9, 18, 16], n_seeds=5, walk=150)
    outids, pairs, num, _ = ref.get_indice_pairs(ind, 2, [9, 18, 16], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], 0)
    f = np.round(detgen.randn("pool_fresh_f", (len(ind), 8)) * 4) / 4
    y = ref.indice_maxpool(f, pairs, num, len(outids))
    assert np.array_equal(orc.indice_maxpool(f, pairs, num, len(outids)), y)
    go = detgen.randn("pool_fresh_g", y.shape)
    assert np.array_equal(orc.indice_maxpool_backward(f, y, go, pairs, num), ref.indice_maxpool_backward(f, y, go, pairs, num))
    pts = detgen.rand("dyn_fresh", (3000, 5), -60.0, 60.0)
    rng, vs = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], [0.075, 0.075, 0.2]
    assert np.array_equal(orc.dynamic_voxelize(pts, vs, rng), ref.dynamic_voxelize(pts, vs, rng))


def test_centerhead_loss_vs_reference_golden(golden):
    """CenterHead.loss (focal heat-map loss + code-weighted L1 box loss) and its gradient against the reference's own
    `loss` on the same weights, input and assigner outputs (center_head.py:250-298, losses/centernet_loss.py)."""
    import torch
    from dualfusion.heads import CenterHead
    from make_golden import HEAD_COMMON, HEAD_SHAPE, HEAD_TASKS, head_bias_shift, head_loss_example
    g = golden("centerhead_loss.npz")
    head = CenterHead(in_channels=512, tasks=HEAD_TASKS, dataset='nuscenes', weight=0.25,
                      code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0], common_heads=dict(HEAD_COMMON),
                      share_conv_channel=64, dcn_head=False)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    head.load_state_dict({k: torch.from_numpy(v) for k, v in head_bias_shift(detgen.det_state_dict(shapes)).items()})
    head.eval()
    x = torch.from_numpy(detgen.randn("head_loss_x", HEAD_SHAPE)).requires_grad_(True)
    ex = {k: [torch.from_numpy(a) for a in v] for k, v in head_loss_example().items()}
    rets = head.loss(ex, head(x), {})
    sum(rets["loss"]).backward()
    for key in ("loss", "hm_loss", "loc_loss", "num_positive"):
        np.testing.assert_allclose([v.item() for v in rets[key]], g[key], rtol=2e-5)
    np.testing.assert_allclose(np.stack([v.numpy() for v in rets["loc_loss_elem"]]), g["loc_loss_elem"], rtol=2e-5, atol=1e-6)
    gx = [x.grad.sum(dtype=torch.float64).item(), x.grad.abs().sum(dtype=torch.float64).item()]
    np.testing.assert_allclose(gx, g["gx"], rtol=1e-4)
    gw = [head.shared_conv[0].weight.grad.abs().sum(dtype=torch.float64).item(),
          head.tasks[1].hm[3].bias.grad.abs().sum(dtype=torch.float64).item()]
    np.testing.assert_allclose(gw, g["gw"], rtol=1e-4)


def test_transfusion_projection_vs_reference_golden(golden):
    """ACTRFusionLayer.project (pure device-side torch, runs on the CPU as well) against the reference's own
    get_2d_coor_multi / projection walk through nuScenes records (golden tf_fusion.npz, tests/golden/make_golden.py
    gen_tf_fusion): camera assignment (last visible camera wins, unseen -> camera 0 at (0, 0)) exact, image coordinates
    to 5e-3 px -- the reference projects in float64 through five frames, the layer in fp32 through the composed
    lidar2cam.  Second case: a 3-D augmentation flow (flip, flip, rotate, scale, translate) that must be undone first."""
    import torch
    from dualfusion.fusion_tf import ACTRFusionLayer
    from make_golden import ACTR_CFG, TFF, tff_inputs, tff_metas
    g = golden("tf_fusion.npz")
    layer = ACTRFusionLayer(pfat_cfg=dict(ACTR_CFG)).eval()
    pts, _, _ = tff_inputs()
    cat = torch.cat([torch.cat([torch.full((len(p), 1), float(b)), torch.from_numpy(p)], 1) for b, p in enumerate(pts)])
    for tag, aug in (("plain", False), ("aug", True)):
        metas, _, _, _ = tff_metas(aug)
        ours = []
        for b, m in enumerate(metas):
            mm = {k: v for k, v in m.items() if k not in ("sample_idx", "filename")}
            mm["lidar2cam"], mm["cam_intrinsic"] = g[tag + "_lidar2cam"][b], g[tag + "_intrinsic"][b]
            ours.append(mm)
        cam_id, norm, pix = layer.project(cat, ours)
        want, want_o = g[tag + "_coor_2d"], g[tag + "_coor_2d_o"]
        assert np.array_equal(cam_id.numpy(), want[:, 0].astype(np.int64)), tag
        assert len(set(want[:, 0].astype(int))) == 6 and (want_o[:, 1:] == 0).all(1).sum() > 5      # all cameras, some unseen
        np.testing.assert_allclose(pix.numpy(), want_o[:, 1:], atol=5e-3, rtol=0)
        np.testing.assert_allclose(norm.numpy(), want[:, 1:], atol=5e-5, rtol=0)


# ---- round 6: the fusion adapters of the other two trees as oracle compositions (the full-size parity tests of configs[2] / [4]
#      use them instead of the device layer)
@pytest.mark.parametrize("tag,aug", [("plain", False), ("aug", True)])
def test_transfusion_fusion_oracle_vs_reference_golden(golden, tag, aug):
    """tests/oracle_models.transfusion_fusion (projection through the composed nuScenes chain in float64, last visible camera
    wins, zero-padded per-camera lists, image feature at pixel // 4, ACTR, additive write-back) against the output of the
    reference's own point_fusion.ACTR.forward (golden tf_fusion.npz; TF/mmdet3d/models/fusion_layers/point_fusion.py:342-643)."""
    import oracle_models as om
    from make_golden import ACTR_CFG, tff_inputs, tff_metas    # noqa: F401
    g = golden("tf_fusion.npz")
    from dualfusion.fusion_tf import ACTRFusionLayer
    shapes = {k: tuple(v.shape) for k, v in ACTRFusionLayer(pfat_cfg=dict(ACTR_CFG)).state_dict().items()}
    assert sorted(shapes) == list(g["param_names"])
    sd = detgen.det_state_dict(shapes)
    pts, feats, img = tff_inputs()
    metas, _, _, _ = tff_metas(aug)
    ours = []
    for b, m in enumerate(metas):
        mm = {k: v for k, v in m.items() if k not in ("sample_idx", "filename")}
        mm["lidar2cam"], mm["cam_intrinsic"] = g[tag + "_lidar2cam"][b], g[tag + "_intrinsic"][b]
        ours.append(mm)
    c2 = np.concatenate([om.transfusion_project(p, m)[0] for p, m in zip(pts, ours)])
    c2o = np.concatenate([om.transfusion_project(p, m)[1] for p, m in zip(pts, ours)])
    assert np.array_equal(c2[:, 0], g[tag + "_coor_2d"][:, 0])                       # camera assignment: exact
    # (the reference walks five frames with its points held in a float32 array -- the global frame's ~1 km coordinates cost it
    # ~1e-4 m --, the composition goes through the composed matrix in float64: a few 1e-3 px)
    np.testing.assert_allclose(c2o[:, 1:], g[tag + "_coor_2d_o"][:, 1:], atol=5e-3, rtol=0)
    assert np.array_equal(c2o[:, 1:].astype(np.int64) // 4, g[tag + "_coor_2d_o"][:, 1:].astype(np.int64) // 4)   # same feature pixels
    out = om.transfusion_fusion(sd, pts, feats, img, ours)
    want = g[tag + "_fused"]
    assert np.abs(out - want).max() <= 2e-4 * np.abs(want).max(), np.abs(out - want).max()


@pytest.mark.parametrize("tag,with_aug", [("plain", False), ("aug", True)])
def test_voxel_rcnn_fusion_oracle_vs_reference_golden(golden, tag, with_aug):
    """tests/oracle_models.voxel_rcnn_mvx / voxel_rcnn_actr_fusion against the reference's own VoxelBackBone8xFusion.
    point_fusion (golden vr_fusion.npz; VR/pcdet/models/backbones_3d/spconv_backbone.py:650-827): the MVX nearest-pixel sum at
    stride 1 and the ACTRv2 dual-query fusion at stride 8 (LocalTransformer per layer, gate before the feed-forward blocks)."""
    import oracle_models as om
    from make_golden import VRF, vrf_inputs
    g = golden("vr_fusion.npz")
    from dualfusion import actr as actr_mod
    model = actr_mod.build(dict(VRF["actr"]), model_name="ACTRv2", lt_cfg=dict(VRF["lt"]),
                           hybrid_cfg=dict(VRF["hybrid"], gate_before_ffn=True))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert sorted(shapes) == list(g["param_names"])
    sd = detgen.det_state_dict(shapes)
    ind1, f1, ind4, f4, mvx, img, aug = vrf_inputs()
    aug = aug if with_aug else None
    l2i = g["lidar2img"]
    y1 = om.voxel_rcnn_mvx(ind1, f1, mvx, l2i, VRF["hw"], 1, aug)
    want1 = g[tag + "_mvx"]
    assert np.abs(y1 - want1).max() <= 1e-5 * max(1.0, np.abs(want1).max()), np.abs(y1 - want1).max()
    y4 = om.voxel_rcnn_actr_fusion(sd, ind4, f4, img, l2i, VRF["hw"], VRF["lt"], 8, aug, num_layers=VRF["actr"]["num_enc_layers"])
    want4 = g[tag + "_actr"]
    assert np.abs(y4 - want4).max() <= 2e-4 * np.abs(want4).max(), np.abs(y4 - want4).max()
