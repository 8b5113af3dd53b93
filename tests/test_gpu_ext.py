"""Boundary, GPU part: the extension shims (`dualfusion/ext/*`, the reference's pybind names) driven
with the argument shapes the reference's Python wrappers use -- caller-allocated outputs for `hard_voxelize`
(TF/mmdet3d/ops/voxel/voxelize.py:46-57), the rulebook / conv triple of TF/mmdet3d/ops/spconv/ops.py:46-126, int64
`spatial_shapes` + `im2col_step` of ms_deform_attn_func.py:21-38, pre-allocated `temp` / `idx` / `out` of the point ops --
against the same golden vectors of the reference's compiled code that pin the kernels themselves."""
import numpy as np
import pytest

import detgen
from make_golden import RB_BATCH, RB_CASES, RB_SHAPE
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def test_voxel_layer_with_caller_allocated_outputs(golden):
    from dualfusion import synth
    from dualfusion.ext import voxel_layer
    g = golden("voxelize.npz")
    for tag in ("nocap", "cap", "mp3"):
        maxp, maxv = [int(x) for x in g["sw_%s_params" % tag]]
        points = T(g["sw_points"])
        voxels = points.new_zeros(size=(maxv, maxp, points.size(1)))                  # voxelize.py:46-52
        coors = points.new_zeros(size=(maxv, 3), dtype=torch.int)
        num = points.new_zeros(size=(maxv,), dtype=torch.int)
        n = voxel_layer.hard_voxelize(points, voxels, coors, num, synth.NUSC_VOXEL, synth.NUSC_RANGE, maxp, maxv, 3)
        assert isinstance(n, int) and n == len(g["sw_%s_num" % tag])
        assert np.array_equal(coors[:n].cpu().numpy(), g["sw_%s_coors" % tag])
        assert np.array_equal(num[:n].cpu().numpy(), g["sw_%s_num" % tag])
        assert np.array_equal(voxels[:n, 0].cpu().numpy(), g["sw_%s_first" % tag])
    # the reference's own unit-test vector (tests/test_voxel_generator.py literals)
    points = T(g["tg_points"])
    voxels, coors, num = points.new_zeros((20000, 1000, points.size(1))), points.new_zeros((20000, 3), dtype=torch.int), points.new_zeros((20000,), dtype=torch.int)
    n = voxel_layer.hard_voxelize(points, voxels, coors, num, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000)
    assert np.array_equal(coors[:n].cpu().numpy(), g["tg_expected_coors"]) and np.array_equal(num[:n].cpu().numpy(), g["tg_expected_num"])
    # dynamic voxelisation fills the caller's coors (voxelize.py:41-44)
    from make_golden import POOL_RANGE, POOL_VS, pool_points
    pts = T(pool_points())
    dyn = pts.new_zeros(size=(pts.size(0), 3), dtype=torch.int)
    voxel_layer.dynamic_voxelize(pts, dyn, POOL_VS, POOL_RANGE, 3)
    assert np.array_equal(dyn.cpu().numpy(), golden("pool.npz")["dyn"])
    with pytest.raises(RuntimeError):
        voxel_layer.hard_voxelize(points, voxels[:10], coors, num, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000)
    with pytest.raises(RuntimeError):
        voxel_layer.hard_voxelize(points.t(), voxels, coors, num, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000)


@pytest.mark.parametrize("name", sorted(RB_CASES))
def test_sparse_conv_ext_rulebook_conv_and_backward(golden, name):
    """get_indice_pairs_3d -> indice_conv_fp32 (-> indice_conv_backward_fp32) exactly as ops.get_indice_pairs / indice_conv
    of the reference's Python layer call them; golden = the reference's compiled CPU extension."""
    from dualfusion.ext import sparse_conv_ext as ext
    g = golden("rulebook_conv.npz")
    ks, st, pd, dl, subm, cin, cout = RB_CASES[name]
    ind = T(g["indices"])
    out_shape = list(RB_SHAPE) if subm else orc.get_conv_output_size(RB_SHAPE, ks, st, pd, dl)
    outids, pairs, num = ext.get_indice_pairs_3d(ind, RB_BATCH, out_shape, list(RB_SHAPE), ks, st, pd, dl, [0, 0, 0], int(subm), 0)
    assert pairs.dtype == torch.int32 and tuple(pairs.shape) == (int(np.prod(ks)), 2, len(g["indices"])) and num.dtype == torch.int32
    ref_out, ref_lists = orc.canonical_rulebook(g[name + "_outids"], g[name + "_pairs"], g[name + "_num"])
    my_out, my_lists = orc.canonical_rulebook(outids.cpu().numpy(), pairs.cpu().numpy(), num.cpu().numpy())
    assert np.array_equal(my_out, ref_out) and np.array_equal(num.cpu().numpy(), g[name + "_num"])
    assert all(np.array_equal(a, b) for a, b in zip(my_lists, ref_lists))
    feats = T(detgen.randn("feat_" + name, (len(g["indices"]), cin)))
    filt = T(detgen.randn("filt_" + name, (ks[0], ks[1], ks[2], cin, cout), 0.2))
    y = ext.indice_conv_fp32(feats, filt, pairs, num, outids.shape[0], 0, int(subm)).cpu().numpy()
    if not subm:
        pos = {tuple(r): i for i, r in enumerate(outids.cpu().numpy())}
        y = y[np.array([pos[tuple(r)] for r in g[name + "_outids"]])]
    np.testing.assert_allclose(y, g[name + "_y"], rtol=1e-3, atol=1e-4)
    # pre-grid variant and fused bias
    o2, p2, n2 = ext.get_indice_pairs_grid_3d(ind, torch.empty(0, device=DEV), RB_BATCH, out_shape, list(RB_SHAPE), ks, st, pd, dl,
                                              [0, 0, 0], int(subm), 0)
    assert torch.equal(p2, pairs) and torch.equal(n2, num) and torch.equal(o2, outids)
    bias = T(detgen.randn("bias_" + name, (cout,)))
    yb = ext.fused_indice_conv_fp32(feats, filt, bias, pairs, num, outids.shape[0], 0, int(subm))
    y0 = ext.indice_conv_fp32(feats, filt, pairs, num, outids.shape[0], 0, int(subm))
    assert torch.allclose(yb, y0 + bias, atol=1e-5)
    # backward against the oracle's indiceConvBackward (pinned to the reference's compiled code by conv_bwd.npz)
    go = detgen.randn("go_" + name, (outids.shape[0], cout))
    gi, gw = ext.indice_conv_backward_fp32(feats, filt, T(go), pairs, num, 0, int(subm))
    ogi, ogw = orc.indice_conv_backward(feats.cpu().numpy(), filt.cpu().numpy(), go, pairs.cpu().numpy(), num.cpu().numpy(), subm)
    assert np.abs(gi.cpu().numpy() - ogi).max() <= 1e-3 * max(1.0, np.abs(ogi).max())
    assert tuple(gw.shape) == tuple(filt.shape) and np.abs(gw.cpu().numpy() - ogw).max() <= 1e-3 * max(1.0, np.abs(ogw).max())
    with pytest.raises(RuntimeError):
        ext.get_indice_pairs_3d(ind, RB_BATCH, [s + 1 for s in out_shape], list(RB_SHAPE), ks, st, pd, dl, [0, 0, 0], int(subm), 0)
    with pytest.raises(RuntimeError):
        ext.indice_conv_fp32(feats.t().contiguous().t(), filt, pairs, num, outids.shape[0], 0, int(subm))      # not contiguous


def test_sparse_conv_ext_2d_and_maxpool(golden):
    from dualfusion.ext import sparse_conv_ext as ext
    rs = np.random.RandomState(2)
    H, W = 20, 24
    flat = rs.choice(2 * H * W, 300, replace=False)
    ind2 = np.stack([flat // (H * W), (flat % (H * W)) // W, flat % W], 1).astype(np.int32)
    outids, pairs, num = ext.get_indice_pairs_2d(T(ind2), 2, [H, W], [H, W], [3, 3], [1, 1], [1, 1], [1, 1], [0, 0], 1, 0)
    assert outids.shape[1] == 3 and tuple(pairs.shape) == (9, 2, 300)
    f = T(rs.standard_normal((300, 8)).astype(np.float32))
    w = T(rs.standard_normal((3, 3, 8, 12)).astype(np.float32) * 0.2)
    y = ext.indice_conv_fp32(f, w, pairs, num, 300, 0, 1).cpu()
    dense = torch.zeros(2, 8, H, W)
    dense[ind2[:, 0], :, ind2[:, 1], ind2[:, 2]] = f.cpu()
    want = torch.nn.functional.conv2d(dense, w.cpu().permute(3, 2, 0, 1), padding=1)[ind2[:, 0], :, ind2[:, 1], ind2[:, 2]]
    assert torch.allclose(y, want, atol=2e-4)
    # max pooling entries vs the reference's compiled CPU code (pool.npz)
    from make_golden import CONV_BWD_BATCH, CONV_BWD_SHAPE, conv_bwd_case
    g = golden("pool.npz")
    ind, ks, st, pd, f, _ = conv_bwd_case(0)
    oshape = orc.get_conv_output_size(CONV_BWD_SHAPE, ks, st, pd, [1, 1, 1])
    outids, pairs, num = ext.get_indice_pairs_3d(T(ind), CONV_BWD_BATCH, oshape, list(CONV_BWD_SHAPE), ks, st, pd, [1, 1, 1], [0, 0, 0], 0, 0)
    y = ext.indice_maxpool_fp32(T(f), pairs, num, outids.shape[0])
    assert np.array_equal(outids.cpu().numpy(), g["outids"]) and np.array_equal(y.cpu().numpy(), g["y"][g["order"]])
    go = detgen.randn("pool_g", g["y"].shape)
    gin = ext.indice_maxpool_backward_fp32(T(f), y, T(go[g["order"]]), pairs, num)
    assert np.array_equal(gin.cpu().numpy(), g["gin"])


def test_msda_entries_forward_and_backward(golden):
    from dualfusion.ext import MultiScaleDeformableAttention as MSDA
    g = golden("msda.npz")
    shapes = torch.as_tensor(np.asarray(g["t_shapes"]), dtype=torch.long, device=DEV)       # int64, as the reference passes them
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    y = MSDA.ms_deform_attn_forward(T(g["t_value"]), shapes, lsi, T(g["t_loc"]), T(g["t_aw"]), 2)   # ops/test.py: im2col_step = 2
    np.testing.assert_allclose(y.cpu().numpy(), g["t_out"], rtol=1e-4, atol=1e-7)
    from make_golden import msda_bwd_inputs
    gb = golden("msda_bwd.npz")
    for tag in ("hot", "multi"):
        value, shp, loc, aw, gout = msda_bwd_inputs(tag)
        shapes = torch.as_tensor(shp, dtype=torch.long, device=DEV)
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        out = MSDA.ms_deform_attn_backward(T(value), shapes, lsi, T(loc), T(aw), T(gout), 64)
        assert isinstance(out, list) and len(out) == 3
        for got, key in zip(out, ("_gv", "_gl", "_ga")):
            want = gb[tag + key]
            assert np.abs(got.cpu().numpy() - want).max() / max(1.0, np.abs(want).max()) <= 1e-4
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(T(g["t_value"]), shapes.int(), lsi, T(g["t_loc"]), T(g["t_aw"]), 2)      # int32 shapes
    # the reference's autograd Function on top of the shim module, as ms_deform_attn_func.py binds it
    from dualfusion import ext
    import sys
    had = sys.modules.get("MultiScaleDeformableAttention")
    try:
        sys.modules.pop("MultiScaleDeformableAttention", None)
        ext.install()
        import MultiScaleDeformableAttention as M2
        assert M2.ms_deform_attn_forward is MSDA.ms_deform_attn_forward
    finally:
        if had is not None:
            sys.modules["MultiScaleDeformableAttention"] = had


def test_point_op_entries(golden):
    from dualfusion.ext import ball_query_ext, furthest_point_sample_ext, gather_points_ext, group_points_ext
    g = golden("pointops_tests.npz")
    # FurthestPointSampling.forward (furthest_point_sample.py:28-34)
    xyz = T(g["test_fps__xyz"])
    B, N = xyz.shape[:2]
    out = torch.cuda.IntTensor(B, 3)
    temp = torch.cuda.FloatTensor(B, N).fill_(1e10)
    furthest_point_sample_ext.furthest_point_sampling_wrapper(B, N, 3, xyz, temp, out)
    assert np.array_equal(out.cpu().numpy(), g["test_fps__expected_idx"])
    # the with_dist form on the squared-distance matrix picks the same points (furthest_point_sample.py:63-69)
    big = T(detgen.rand("fpsd_xyz", (2, 700, 3), -5, 5))
    d = (big[:, :, None] - big[:, None]).pow(2).sum(-1).contiguous()
    o1, o2 = torch.cuda.IntTensor(2, 64), torch.cuda.IntTensor(2, 64)
    furthest_point_sample_ext.furthest_point_sampling_wrapper(2, 700, 64, big, torch.cuda.FloatTensor(2, 700).fill_(1e10), o1)
    furthest_point_sample_ext.furthest_point_sampling_with_dist_wrapper(2, 700, 64, d, torch.cuda.FloatTensor(2, 700).fill_(1e10), o2)
    assert torch.equal(o1, o2)
    # BallQuery.forward (ball_query.py:33-38)
    bx, nx = T(g["test_ball_query__xyz"]), T(g["test_ball_query__new_xyz"])
    B, N, m = bx.shape[0], bx.shape[1], nx.shape[1]
    idx = torch.cuda.IntTensor(B, m, 5).zero_()
    ball_query_ext.ball_query_wrapper(B, N, m, 0.2, 0.4, 5, nx, bx, idx)
    assert np.array_equal(idx.cpu().numpy(), g["test_ball_query__expected_idx"])
    # GroupingOperation forward / backward (group_points.py:176-206)
    feat, gidx = T(g["test_grouping_points__festures"]), T(g["test_grouping_points__idx"], torch.int32)
    B, C, N = feat.shape
    _, npnt, ns = gidx.shape
    out = torch.cuda.FloatTensor(B, C, npnt, ns)
    group_points_ext.forward(B, C, N, npnt, ns, feat, gidx, out)
    np.testing.assert_allclose(out.cpu().numpy(), g["test_grouping_points__expected_output"])
    go = torch.randn(B, C, npnt, ns, device=DEV)
    gp = torch.cuda.FloatTensor(B, C, N).zero_()
    group_points_ext.backward(B, C, N, npnt, ns, go.contiguous(), gidx, gp)
    want = torch.zeros(B, C, N, device=DEV).scatter_add_(2, gidx.long().view(B, 1, -1).expand(B, C, -1), go.view(B, C, -1))
    assert torch.allclose(gp, want, atol=1e-5)
    # GatherPoints forward / backward (gather_points.py:29-49)
    feat, aidx = T(g["test_gather_points__features"]), T(g["test_gather_points__idx"], torch.int32)
    B, C, N = feat.shape
    npnt = aidx.shape[1]
    out = torch.cuda.FloatTensor(B, C, npnt)
    gather_points_ext.gather_points_wrapper(B, C, N, npnt, feat, aidx, out)
    np.testing.assert_allclose(out.cpu().numpy(), g["test_gather_points__expected_output"])
    go = torch.randn(B, C, npnt, device=DEV)
    gp = torch.cuda.FloatTensor(B, C, N).zero_()
    gather_points_ext.gather_points_grad_wrapper(B, C, N, npnt, go, aidx, gp)
    want = torch.zeros(B, C, N, device=DEV).scatter_add_(2, aidx.long().view(B, 1, -1).expand(B, C, -1), go)
    assert torch.allclose(gp, want, atol=1e-5)


def test_iou3d_cuda_entries_fill_the_callers_tensor():
    from dualfusion.ext import iou3d_cuda
    a = detgen.bev_boxes("ioue_a", 40, 6.0)
    b = detgen.bev_boxes("ioue_b", 30, 6.0, special=False)
    xy = lambda v: np.stack([v[:, 0] - v[:, 3] / 2, v[:, 1] - v[:, 4] / 2, v[:, 0] + v[:, 3] / 2, v[:, 1] + v[:, 4] / 2, v[:, 6]], 1).astype(np.float32)  # noqa: E731
    ov = torch.zeros(40, 30, device=DEV)
    iou3d_cuda.boxes_overlap_bev_gpu(T(xy(a)), T(xy(b)), ov)
    want = orc.tf_boxes_overlap_bev(xy(a), xy(b))
    assert np.abs(ov.cpu().numpy() - want).max() < 2e-4
    iou = torch.zeros(40, 30, device=DEV)
    iou3d_cuda.boxes_iou_bev_gpu(T(xy(a)), T(xy(b)), iou)
    sa, sb = (a[:, 3] * a[:, 4])[:, None], (b[:, 3] * b[:, 4])[None]
    assert np.abs(iou.cpu().numpy() - want / np.maximum(sa + sb - want, 1e-8)).max() < 2e-4


def test_iou3d_cuda_nms_entries_vs_oracle_and_reference_kernel():
    """`nms_gpu` / `nms_normal_gpu` under the reference's calling protocol (iou3d_utils.py:27-75): score-sorted [N, 5] boxes,
    a CPU int64 `keep`, the number kept returned -- against the greedy pass over the oracle's overlap matrix and, where it is
    built, the reference's own kernels on the same GPU (oracle/_ref/iou3d_cuda.so)."""
    from dualfusion.ext import iou3d_cuda
    from oracle import ref
    v = detgen.bev_boxes("ioue_nms", 300, 7.0)
    xy = np.stack([v[:, 0] - v[:, 3] / 2, v[:, 1] - v[:, 4] / 2, v[:, 0] + v[:, 3] / 2, v[:, 1] + v[:, 4] / 2, v[:, 6]], 1).astype(np.float32)
    ov = orc.tf_boxes_overlap_bev(xy, xy)
    s = (xy[:, 2] - xy[:, 0]) * (xy[:, 3] - xy[:, 1])
    iou = ov / np.maximum(s[:, None] + s[None] - ov, 1e-8)
    l, r = np.maximum(xy[:, None, 0], xy[None, :, 0]), np.minimum(xy[:, None, 2], xy[None, :, 2])
    t, b = np.maximum(xy[:, None, 1], xy[None, :, 1]), np.minimum(xy[:, None, 3], xy[None, :, 3])
    inter = np.maximum(r - l, 0) * np.maximum(b - t, 0)
    iou_n = inter / np.maximum(s[:, None] + s[None] - inter, 1e-8)

    def greedy(m, th):
        rem, out = np.zeros(len(m), bool), []
        for i in range(len(m)):
            if not rem[i]:
                out.append(i)
                rem[i + 1:] |= m[i, i + 1:] > th
        return out
    for name, mat, lo, hi in (("nms_gpu", iou, 0.15, 0.25), ("nms_normal_gpu", iou_n, 0.25, 0.35)):
        # a pair within rounding of the threshold may flip between float implementations: the threshold is put into the
        # middle of the widest gap between the IoUs of this box set inside [lo, hi]
        vals = np.sort(np.concatenate([[lo], mat[np.triu_indices(len(mat), 1)], [hi]]))
        vals = vals[(vals >= lo) & (vals <= hi)]
        g = int(np.argmax(np.diff(vals)))
        th = float((vals[g] + vals[g + 1]) / 2)
        assert np.abs(mat[np.triu_indices(len(mat), 1)] - th).min() > 1e-4
        keep = torch.zeros(len(xy), dtype=torch.long)
        n = getattr(iou3d_cuda, name)(T(xy), keep, th, 0)
        want = greedy(mat, th)
        assert n == len(want) and keep[:n].tolist() == want and 5 < n < len(xy)
        if ref.available("iou3d_cuda"):
            keep_r = torch.zeros(len(xy), dtype=torch.long)
            n_r = getattr(ref.load("iou3d_cuda"), name)(T(xy), keep_r, th, 0)
            assert n_r == n and keep_r[:n_r].tolist() == want
    assert iou3d_cuda.nms_gpu(torch.zeros((0, 5), device=DEV), torch.zeros(0, dtype=torch.long), 0.2, 0) == 0
    with pytest.raises(RuntimeError):
        iou3d_cuda.nms_gpu(T(xy), torch.zeros(len(xy), dtype=torch.int32), 0.2, 0)
