"""CenterPoint sparse backbone (SpMiddleResNetFHD) on the MI355X modules vs the oracle composition,
same state_dict, same synthetic sweep.  fp32, tolerance 1e-3 (north_star)."""
import os

import numpy as np
import pytest

import detgen
import oracle_models as om
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _model_and_sd(dev):
    from dualfusion.backbones import SpMiddleResNetFHD
    model = SpMiddleResNetFHD(num_input_features=5).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = detgen.det_state_dict(shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.to(dev), sd


def test_parameter_names_match_reference_checkpoint_layout():
    from dualfusion.backbones import SpMiddleResNetFHD
    sd = SpMiddleResNetFHD(num_input_features=5).state_dict()
    # SURVEY.md Appendix B
    assert tuple(sd["conv_input.0.weight"].shape) == (3, 3, 3, 5, 16)
    assert tuple(sd["conv2.0.weight"].shape) == (3, 3, 3, 16, 32)
    assert tuple(sd["conv4.0.weight"].shape) == (3, 3, 3, 64, 128)
    assert tuple(sd["extra_conv.0.weight"].shape) == (3, 1, 1, 128, 128)
    assert "conv1.0.conv1.bias" in sd and "conv_input.0.bias" not in sd
    assert "conv3.4.bn2.running_var" in sd


@pytest.mark.parametrize("batch", [1, 2])
def test_centerpoint_backbone_vs_oracle(batch):
    from dualfusion import ops, synth
    dev = torch.device("cuda:0")
    model, sd = _model_and_sd(dev)
    # a reduced grid keeps the oracle's dense rulebook grids small: 16 m x 16 m around the sensor
    rng = [-9.6, -9.6, -5.0, 9.6, 9.6, 3.0]
    feats, coors = [], []
    o_feats, o_coors = [], []
    for b in range(batch):
        pts = synth.nusc_sweep(seed=20 + b)
        v, c, n, mean = ops.hard_voxelize(torch.from_numpy(pts).to(dev), synth.NUSC_VOXEL, rng, 10, 120000)
        ov, oc, on = orc.hard_voxelize(pts, synth.NUSC_VOXEL, rng, 10, 120000)
        assert np.array_equal(c.cpu().numpy(), oc)
        feats.append(mean)
        coors.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32, device=dev), c], 1))
        o_feats.append(orc.mean_vfe(ov, on))
        o_coors.append(np.concatenate([np.full((len(oc), 1), b, np.int32), oc], 1))
    grid_xyz = [256, 256, 40]
    with torch.no_grad():
        dense, ms = model(torch.cat(feats), torch.cat(coors), batch, grid_xyz)
    o_dense, o_ms = om.centerpoint_backbone(sd, np.concatenate(o_feats), np.concatenate(o_coors), batch, grid_xyz)
    assert tuple(dense.shape) == o_dense.shape == (batch, 256, 32, 32)
    for name in ("conv1", "conv2", "conv3", "conv4"):
        mi, mf = om.sort_rows(ms[name].indices.cpu().numpy(), ms[name].features.cpu().numpy())
        oi, of = om.sort_rows(o_ms[name].indices, o_ms[name].features)
        assert np.array_equal(mi, oi), name                   # voxel sets bit-exact
        scale = np.abs(of).max()
        assert np.abs(mf - of).max() <= 1e-3 * max(scale, 1.0), (name, np.abs(mf - of).max(), scale)
    np.testing.assert_allclose(dense.cpu().numpy(), o_dense, rtol=1e-3, atol=1e-3 * max(np.abs(o_dense).max(), 1.0))


# ------------------------------------------------------------------------- ACTR (dual-query fusion encoder)
def test_actr_parameter_layout_matches_reference(golden):
    from dualfusion import actr
    from make_golden import ACTR_CFG
    g = golden("actr.npz")
    m = actr.build(dict(ACTR_CFG), model_name="ACTR")
    sd = m.state_dict()
    assert sorted(sd) == list(g["param_names"])
    for k, s in zip(g["param_names"], g["param_shapes"]):
        assert str(tuple(sd[str(k)].shape)) == str(s), k
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"]) == 1212484


def test_actr_forward_vs_reference_golden(golden):
    """Same weights (detgen), same inputs -> the reference ACTR.forward output stored by make_golden.py."""
    from dualfusion import actr
    from make_golden import ACTR_CFG, actr_inputs
    dev = torch.device("cuda:0")
    g = golden("actr.npz")
    m = actr.build(dict(ACTR_CFG), model_name="ACTR").eval()
    sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev)
    v_feat, grid, i_feat, lidar_grid, v_i_feat = [torch.from_numpy(a).to(dev) for a in actr_inputs()]
    with torch.no_grad():
        y = m(v_feat=v_feat, grid=grid, i_feats=[i_feat], lidar_grid=lidar_grid, v_i_feat=v_i_feat)
    ref = g["out"]
    err = np.abs(y.cpu().numpy() - ref).max()
    assert err <= 1e-3 * max(1.0, np.abs(ref).max()), err


# ------------------------------------------------------------------------- CenterPoint fusion adapter (a7-a9)
def test_centerpoint_fusion_adapter_vs_reference_golden(golden):
    """VoxelWithPointProjection (projection, image-side gate, per-camera query assembly, ACTR, write-back)
    against the output of the REFERENCE module run by make_golden.py on the same inputs and weights."""
    from dualfusion import fusion as fz, spconv, synth
    from make_golden import ACTR_CFG, FUS, FUS_IFAT, FUS_LT
    dev = torch.device("cuda:0")
    g = golden("fusion_cp.npz")
    sets = [g["coords2"].astype(np.int32), g["coords3"].astype(np.int32), g["coords4"].astype(np.int32)]
    feats = [detgen.randn("fus_feat%d" % i, (len(s), c)) for i, (s, c) in enumerate(zip(sets, [32, 64, 128]))]
    mod = fz.VoxelWithPointProjection(fuse_mode='pfat', interpolate=False, voxel_size=FUS["voxel_size"],
                                      pc_range=FUS["pc_range"], image_list=synth.NUSC_CAMS,
                                      image_scale=FUS["image_scale"], depth_thres=FUS["depth_thres"],
                                      pfat_cfg=dict(ACTR_CFG), lt_cfg=dict(FUS_LT), ifat_cfg=dict(FUS_IFAT),
                                      model_name='ACTR').eval()
    sd_shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    assert sorted(sd_shapes) == list(g["param_names"])          # fusion.pfat.* / fusion.ifat.* checkpoint layout
    for k, s in zip(g["param_names"], g["param_shapes"]):
        assert str(sd_shapes[str(k)]) == str(s), k
    sd = detgen.det_state_dict(sd_shapes)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    mod = mod.to(dev)
    B = FUS["batch"]
    cams = synth.nusc_cameras(image_hw=FUS["raw_hw"], focal=FUS["focal"])
    H, W = FUS["img_hw"]
    batch_dict = {'image_shape': {}, 'img_feat': {'layer1_ori_feat2d': {}}, 'calib': {}}
    for n in synth.NUSC_CAMS:
        key = n.lower()
        batch_dict['image_shape'][key] = torch.tensor([[H, W, 3]] * B)
        batch_dict['img_feat']['layer1_ori_feat2d'][key] = torch.from_numpy(
            detgen.randn("fus_img_" + n, (B, 256) + tuple(FUS["feat_hw"]))).to(dev)
        T, K = cams[n]
        batch_dict['calib']['lidar2cam_' + key.lstrip('cam_')] = torch.from_numpy(np.stack([T] * B)).to(dev)
        batch_dict['calib']['cam_intrinsic_' + key.lstrip('cam_')] = torch.from_numpy(np.stack([K] * B)).to(dev)
    shapes = [[21, 128, 128], [11, 64, 64], [5, 32, 32]]
    xs = [spconv.SparseConvTensor(torch.from_numpy(f).to(dev), torch.from_numpy(i).to(dev), shp, B)
          for f, i, shp in zip(feats, sets, shapes)]
    # visible-voxel counts per (sample, camera): integer parity of the projection + masks
    inp = mod._gather_inputs(batch_dict, 'layer1_ori', dev)
    grid, mask, pinv = mod._project(xs[2], 8, inp)
    bcol = torch.from_numpy(sets[2][:, 0]).to(dev)
    counts = np.array([[int((mask[c].bool() & (bcol == b)).sum()) for c in range(6)] for b in range(B)])
    assert np.array_equal(counts, g["counts"]), (counts, g["counts"])
    out = mod(batch_dict, {}, encoded_voxel_list=xs, layer_name='layer1_ori', fuse_mode='pfat',
              d_factor_list=[2, 4, 8])
    ref = g["out"]
    err = np.abs(out.features.cpu().numpy() - ref).max()
    assert err <= 1e-3 * max(1.0, np.abs(ref).max()), err
    # second case: 3-D augmentation records to undo before the projection (batch_dict['aug_matrix_inv'],
    # point_to_image_projection.py:121-128) -- golden from the same reference module
    from make_golden import fusion_aug_inv
    ga = golden("fusion_cp_aug.npz")
    batch_dict = dict(batch_dict, aug_matrix_inv=fusion_aug_inv())
    xs = [spconv.SparseConvTensor(torch.from_numpy(f).to(dev), torch.from_numpy(i).to(dev), shp, B)
          for f, i, shp in zip(feats, sets, shapes)]
    inp = mod._gather_inputs(batch_dict, 'layer1_ori', dev)
    grid, mask, pinv = mod._project(xs[2], 8, inp)
    counts = np.array([[int((mask[c].bool() & (bcol == b)).sum()) for c in range(6)] for b in range(B)])
    # an augmented lattice has no exact ties, but a corner may sit within 1 ulp of a pixel boundary: allow one voxel
    assert np.abs(counts - ga["counts"]).sum() <= 1, (counts, ga["counts"])
    out = mod(batch_dict, {}, encoded_voxel_list=xs, layer_name='layer1_ori', fuse_mode='pfat', d_factor_list=[2, 4, 8])
    ref = ga["out"]
    row_err = np.abs(out.features.cpu().numpy() - ref).max(1)
    assert (row_err > 1e-3 * max(1.0, np.abs(ref).max())).sum() <= (0 if np.array_equal(counts, ga["counts"]) else 40), row_err.max()
    assert np.abs(ref - g["out"]).max() > 1.0               # the augmentation records do change the result


def test_centerpoint_fusion_adapter_training_path(golden):
    """The differentiable formulation of the adapter (forward_autograd: same native integer work, torch compositions +
    the ACTR module path for every floating-point stage) reproduces the REFERENCE module's output on the golden inputs,
    and gradients reach the voxel features of every scale it reads and every fusion parameter."""
    from dualfusion import fusion as fz, spconv, synth
    from make_golden import ACTR_CFG, FUS, FUS_IFAT, FUS_LT
    dev = torch.device("cuda:0")
    g = golden("fusion_cp.npz")
    sets = [g["coords2"].astype(np.int32), g["coords3"].astype(np.int32), g["coords4"].astype(np.int32)]
    feats = [detgen.randn("fus_feat%d" % i, (len(s), c)) for i, (s, c) in enumerate(zip(sets, [32, 64, 128]))]
    mod = fz.VoxelWithPointProjection(fuse_mode='pfat', interpolate=False, voxel_size=FUS["voxel_size"],
                                      pc_range=FUS["pc_range"], image_list=synth.NUSC_CAMS,
                                      image_scale=FUS["image_scale"], depth_thres=FUS["depth_thres"],
                                      pfat_cfg=dict(ACTR_CFG), lt_cfg=dict(FUS_LT), ifat_cfg=dict(FUS_IFAT),
                                      model_name='ACTR')
    sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in mod.state_dict().items()})
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    mod = mod.to(dev).train()
    for m in mod.modules():                                     # the golden was taken without dropout
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    B = FUS["batch"]
    cams = synth.nusc_cameras(image_hw=FUS["raw_hw"], focal=FUS["focal"])
    H, W = FUS["img_hw"]
    batch_dict = {'image_shape': {}, 'img_feat': {'layer1_ori_feat2d': {}}, 'calib': {}}
    for n in synth.NUSC_CAMS:
        key = n.lower()
        batch_dict['image_shape'][key] = torch.tensor([[H, W, 3]] * B)
        batch_dict['img_feat']['layer1_ori_feat2d'][key] = torch.from_numpy(
            detgen.randn("fus_img_" + n, (B, 256) + tuple(FUS["feat_hw"]))).to(dev)
        T, K = cams[n]
        batch_dict['calib']['lidar2cam_' + key.lstrip('cam_')] = torch.from_numpy(np.stack([T] * B)).to(dev)
        batch_dict['calib']['cam_intrinsic_' + key.lstrip('cam_')] = torch.from_numpy(np.stack([K] * B)).to(dev)
    shapes = [[21, 128, 128], [11, 64, 64], [5, 32, 32]]
    leaves = [torch.from_numpy(f).to(dev).requires_grad_(True) for f in feats]
    xs = [spconv.SparseConvTensor(f, torch.from_numpy(i).to(dev), shp, B) for f, i, shp in zip(leaves, sets, shapes)]
    out = mod(batch_dict, {}, encoded_voxel_list=xs, layer_name='layer1_ori', fuse_mode='pfat', d_factor_list=[2, 4, 8])
    assert out.features.requires_grad                           # forward() took the differentiable composition
    ref = g["out"]
    err = np.abs(out.features.detach().cpu().numpy() - ref).max()
    assert err <= 1e-3 * max(1.0, np.abs(ref).max()), err
    wgt = torch.from_numpy(detgen.randn("fus_train_w", tuple(out.features.shape))).to(dev)
    (out.features * wgt).sum().backward()
    for i in (0, 2):                                            # the gate reads scales 0 and 2, the queries scale 2
        gr = leaves[i].grad
        assert gr is not None and torch.isfinite(gr).all() and float(gr.abs().sum()) > 0, i
    assert leaves[1].grad is None                               # scale 1 is not an input of this configuration
    missing = [k for k, p_ in mod.named_parameters()
               if p_.requires_grad and (p_.grad is None or not torch.isfinite(p_.grad).all() or float(p_.grad.abs().sum()) == 0)]
    # parameters the 3D-DF configuration never reaches, in the reference as here: the level embedding (one level, no
    # positional term), the image-query half of the LAST layer's gate (nobody reads its qi output), the gate's reduction
    # conv of a scale that is not in voxel_idx
    nlay = len(mod.pfat.transformer.encoder.layers)
    unused = ["pfat.transformer.level_embed", "pfat.transformer.encoder.layers.%d.fusion_layer.a_conv1d." % (nlay - 1)] + \
             ["ifat.reduced_dim.%d." % i for i in range(mod.ifat.voxel_idx[-1]) if i not in mod.ifat.voxel_idx]
    assert all(any(k.startswith(u) for u in unused) for k in missing), missing
    # round 4: the default formulation factors the per-pixel gate out of input_proj (att * (W img) + b, no gated copy of the
    # camera maps); the composition over the gated maps (the reference's order, DF3D_TRAIN_GATED=1) gives the same output and
    # the same gradients for every voxel feature and every parameter
    got = {"out": out.features.detach().clone(), "leaf0": leaves[0].grad.clone(), "leaf2": leaves[2].grad.clone()}
    got.update({k: p_.grad.clone() for k, p_ in mod.named_parameters() if p_.grad is not None})
    mod.zero_grad(set_to_none=True)
    leaves2 = [torch.from_numpy(f).to(dev).requires_grad_(True) for f in feats]
    xs2 = [spconv.SparseConvTensor(f, torch.from_numpy(i).to(dev), shp, B) for f, i, shp in zip(leaves2, sets, shapes)]
    os.environ["DF3D_TRAIN_GATED"] = "1"
    try:
        out2 = mod(batch_dict, {}, encoded_voxel_list=xs2, layer_name='layer1_ori', fuse_mode='pfat', d_factor_list=[2, 4, 8])
        (out2.features * wgt).sum().backward()
    finally:
        os.environ.pop("DF3D_TRAIN_GATED", None)
    want = {"out": out2.features.detach(), "leaf0": leaves2[0].grad, "leaf2": leaves2[2].grad}
    want.update({k: p_.grad for k, p_ in mod.named_parameters() if p_.grad is not None})
    assert set(want) == set(got)
    for k in want:
        # (two fp32 evaluation orders of the same network: where a rectifier of the layers' feed-forward blocks sees a
        # pre-activation within rounding of zero the two take different branches for that one unit -- a few rows then differ by
        # ~1e-3 of scale while everything else agrees to 1e-6; round 6 saw it when the input projection's product changed its
        # summation order.  Bounded in the L2 sense, loosely entry-wise.)
        scale = float(want[k].abs().max())
        diff = (got[k] - want[k]).double()
        # (one flipped hidden unit of a feed-forward block on one row = one row of that weight's gradient off by that row's
        # share: 1.7 % of the largest entry measured, 1e-6 everywhere else)
        assert float(diff.abs().max()) <= 5e-2 * max(scale, 1e-6), (k, float(diff.abs().max()), scale)
        assert float(diff.norm()) <= 2e-3 * max(float(want[k].double().norm()), 1e-6), (k, float(diff.norm()), float(want[k].norm()))
    # d(loss)/d(voxel feature) of a row no camera sees is exactly the upstream weight (identity write-back)
    inp = mod._gather_inputs(batch_dict, 'layer1_ori', dev)
    _, mask, _ = mod._project(xs[2], 8, inp)
    unseen = (mask.sum(0) == 0)
    win_free = unseen.clone()                                   # ... unless it also owns a pixel of the gate canvas (it cannot: unseen)
    assert int(win_free.sum()) > 0
    np.testing.assert_allclose(leaves[2].grad[win_free].cpu().numpy(), wgt[win_free].cpu().numpy(), rtol=0, atol=1e-6)


# ------------------------------------------------------------------------- LocalTransformer (a13) / point ops
def test_pointops_reference_test_vectors_on_gpu(golden):
    from dualfusion import ops
    dev = torch.device("cuda:0")
    g = golden("pointops_tests.npz")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    assert np.array_equal(ops.furthest_point_sample(T(g["test_fps__xyz"]), 3).cpu().numpy(), g["test_fps__expected_idx"])
    bq = ops.ball_query(0.2, 0.4, 5, T(g["test_ball_query__xyz"]), T(g["test_ball_query__new_xyz"]))
    assert np.array_equal(bq.cpu().numpy(), g["test_ball_query__expected_idx"])
    gp = ops.group_points(T(g["test_grouping_points__festures"]), T(g["test_grouping_points__idx"]))
    np.testing.assert_allclose(gp.cpu().numpy(), g["test_grouping_points__expected_output"])
    ga = ops.gather_points(T(g["test_gather_points__features"]), T(g["test_gather_points__idx"]))
    np.testing.assert_allclose(ga.cpu().numpy(), g["test_gather_points__expected_output"])


def test_local_transformer_vs_reference_golden(golden):
    from dualfusion.pointformer import LocalTransformer
    from make_golden import LT_DIMS, lt_inputs
    dev = torch.device("cuda:0")
    g = golden("local_transformer.npz")
    d = LT_DIMS
    m = LocalTransformer(d["npoint"], d["radius"], d["nsample"], d["C"], d["C"], num_layers=d["num_layers"],
                         attn_feat_agg_method="unique", feat_agg_method="replace").eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sorted(shapes) == list(g["param_names"])
    sd = detgen.det_state_dict(shapes)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev)
    xyz, feat = lt_inputs()
    with torch.no_grad():
        y = m(torch.from_numpy(xyz).to(dev), torch.from_numpy(feat.copy()).to(dev))
    np.testing.assert_allclose(y.cpu().numpy(), g["out"], rtol=1e-3, atol=1e-4)
    # the encoder hands the LocalTransformer a permuted VIEW of its [B, N, C] query rows: that takes the row-layout
    # path (no transposes, in-place 'replace' update) and must give the reference result too
    rows = torch.from_numpy(np.ascontiguousarray(feat.transpose(0, 2, 1))).to(dev)            # [B, N, C]
    with torch.no_grad():
        y2 = m(torch.from_numpy(xyz).to(dev), rows.permute(0, 2, 1))
    assert y2.data_ptr() == rows.data_ptr()                                                    # updated in place
    np.testing.assert_allclose(y2.cpu().numpy(), g["out"], rtol=1e-3, atol=1e-4)


def test_actrv2_builds_and_runs_with_local_transformer():
    """ACTRv2 (CP `...pfatv2.py`, VR config 5): LocalTransformer before every deformable layer; parameter
    names transformer.encoder.lidar_attns.i.* (SURVEY.md Appendix B)."""
    from dualfusion import actr
    from make_golden import ACTR_CFG
    dev = torch.device("cuda:0")
    lt = dict(npoint=128, radius=2.0, nsample=16, num_layers=2, attn_feat_agg_method='unique', feat_agg_method='replace')
    m = actr.build(dict(ACTR_CFG), model_name="ACTRv2", lt_cfg=lt).eval().to(dev)
    names = set(m.state_dict())
    assert "transformer.encoder.lidar_attns.1.chunk.layers.0.self_attn.in_proj_weight" in names
    assert "transformer.encoder.lidar_attns.0.pe.0.conv.weight" in names
    N, Q = 2, 400
    gen = torch.Generator().manual_seed(0)
    v = torch.randn(N, Q, 128, generator=gen).to(dev)
    grid = torch.rand(N, Q, 2, generator=gen).to(dev)
    img = torch.randn(N, 256, 20, 30, generator=gen).to(dev)
    lid = (torch.rand(N, Q, 3, generator=gen) * 40 - 20).to(dev)
    vi = torch.randn(N, Q, 256, generator=gen).to(dev)
    with torch.no_grad():
        y = m(v_feat=v, grid=grid, i_feats=[img], lidar_grid=lid, v_i_feat=vi)
    assert tuple(y.shape) == (N, Q, 128) and bool(torch.isfinite(y).all())


def test_centerpoint_hot_path_end_to_end_vs_oracle():
    """BASELINE configs[1] module tree end to end on a reduced grid: voxelize + mean VFE ->
    SpMiddleResNetFHDFusion -> VoxelWithPointProjection (projection, image gate, ACTR, write-back) -> dense
    BEV, against the oracle composition with the same weights and inputs."""
    from dualfusion import fusion as fz, synth
    from dualfusion.pipeline import CenterPointHotPath
    from make_golden import FUS
    dev = torch.device("cuda:0")
    rng = FUS["pc_range"]
    fus = fz.build_centerpoint_fusion(pc_range=rng, image_scale=FUS["image_scale"])
    fus.depth_thres = FUS["depth_thres"]
    model = CenterPointHotPath(fusion=fus, pc_range=rng).eval()
    sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)
    B = 2
    pts_np = [synth.nusc_sweep(seed=90 + b) for b in range(B)]
    cams = synth.nusc_cameras(image_hw=FUS["raw_hw"], focal=FUS["focal"], yaw_offset_deg=1.37)   # no lattice ties
    H, W = FUS["img_hw"]
    img = {n: detgen.randn("e2e_img_" + n, (B, 256) + tuple(FUS["feat_hw"])) for n in synth.NUSC_CAMS}
    bd = {'image_shape': {}, 'img_feat': {'layer1_ori_feat2d': {}}, 'calib': {}}
    for n in synth.NUSC_CAMS:
        key = n.lower()
        bd['image_shape'][key] = torch.tensor([[H, W, 3]] * B)
        bd['img_feat']['layer1_ori_feat2d'][key] = torch.from_numpy(img[n]).to(dev)
        bd['calib']['lidar2cam_' + key.lstrip('cam_')] = torch.from_numpy(np.stack([cams[n][0]] * B)).to(dev)
        bd['calib']['cam_intrinsic_' + key.lstrip('cam_')] = torch.from_numpy(np.stack([cams[n][1]] * B)).to(dev)
    dense, ms = model([torch.from_numpy(p).to(dev) for p in pts_np], batch_dict=bd, example={})
    # ---- oracle
    feats, coors = [], []
    for b, p in enumerate(pts_np):
        ov, oc, on = orc.hard_voxelize(p, synth.NUSC_VOXEL, rng, 10, 160000, "numba")
        feats.append(orc.mean_vfe(ov, on))
        coors.append(np.concatenate([np.full((len(oc), 1), b, np.int32), oc], 1))
    sd_b = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    sd_f = {k[len("fusion."):]: v for k, v in sd.items() if k.startswith("fusion.")}
    calib = {n: (np.stack([cams[n][0]] * B), np.stack([cams[n][1]] * B)) for n in synth.NUSC_CAMS}

    def srt(c):
        i, f = om.sort_rows(c.indices, c.features)
        return np.ascontiguousarray(i), np.ascontiguousarray(f)

    dbg = {}

    def fuse(c2, c3, c4):
        lv = [srt(c) for c in (c2, c3, c4)]        # the adapter needs batch-sorted rows (spconv's GPU order)
        out = om.centerpoint_fusion(sd_f, lv, img, calib, FUS["img_hw"], synth.NUSC_CAMS, synth.NUSC_VOXEL, rng,
                                    FUS["image_scale"], FUS["depth_thres"], debug=dbg)
        c4.indices, c4.features = lv[2][0], out
        c4.rulebooks = {}
        return c4
    o_dense, o_ms = om.centerpoint_backbone(sd_b, np.concatenate(feats), np.concatenate(coors), B, [256, 256, 40],
                                            fuse=fuse)
    mi, mf = om.sort_rows(ms["conv4"].indices.cpu().numpy(), ms["conv4"].features.cpu().numpy())
    oi, of = om.sort_rows(o_ms["conv4"].indices, o_ms["conv4"].features)
    assert np.array_equal(mi, oi)
    # Voxel -> pixel projection truncates three times (.long()); the oracle projects with torch-CPU matmuls,
    # the kernel with an explicit k-ordered FMA chain (bit-identical on the reference golden,
    # test_centerpoint_fusion_adapter_vs_reference_golden).  A voxel that projects exactly onto a pixel
    # boundary can fall on either side; the rig is turned by 1.37 deg so the voxel lattice has no such ties,
    # and the visible sets must then agree exactly.
    tol = 1e-3 * max(1.0, np.abs(of).max())
    row_err = np.abs(mf - of).max(1)
    inp = fus._gather_inputs(bd, 'layer1_ori', dev)
    _, m_, _ = fus._project(ms["conv4"], 8, inp)
    ind4 = ms["conv4"].indices.cpu().numpy()
    order = np.lexsort((ind4[:, 3], ind4[:, 2], ind4[:, 1], ind4[:, 0]))
    mm = m_.cpu().numpy()[:, order].astype(bool)
    for ci in range(6):
        for b in range(B):
            assert np.array_equal(dbg["per"][(b, ci)]["mask"].numpy(), mm[ci][mi[:, 0] == b]), (ci, b)
    assert row_err.max() <= tol, (row_err.max(), int((row_err > tol).sum()))
    d_err = np.abs(dense.cpu().numpy() - o_dense)
    assert (d_err > 1e-3 * max(1.0, np.abs(o_dense).max())).mean() < 0.01


def test_native_executor_matches_module_path():
    """df3d_backbone_run (one native call for the conv chain) launches the same kernels as the per-module path:
    features and indices of every exported stage are bit-identical, at nuScenes size and on a tiny input."""
    import os
    from dualfusion import synth
    from dualfusion.pipeline import CenterPointHotPath
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = CenterPointHotPath().eval().to(dev)
    for seed, take in ((0, None), (1, 300)):
        pts = torch.from_numpy(synth.nusc_sweep(seed=seed)).to(dev)
        if take:
            pts = pts[:take].contiguous()
        with torch.no_grad():
            feats, coors = model.voxelize([pts])
            assert model.backbone._plan() is not None
            fast = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
            os.environ["DF3D_EXECUTOR"] = "0"
            try:
                slow = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
            finally:
                os.environ["DF3D_EXECUTOR"] = "1"
            for a, b in zip(fast, slow):
                assert torch.equal(a.indices, b.indices) and a.spatial_shape == b.spatial_shape
                assert torch.equal(a.features, b.features)
            # the tail (strided conv on conv4 + dense) runs through the module path on the executor's tensors
            y_fast, _ = model.backbone._tail(*fast)
            y_slow, _ = model.backbone._tail(*slow)
            assert torch.equal(y_fast, y_slow)


def test_prepared_chain_geometry_runs_range_by_range():
    """Round 5, `df3d_backbone_convs_range`: the geometry of a WHOLE conv chain built ahead (`BackbonePlan.build_geometry` on
    the uncut plan of a `SegmentedRunner`), the convolutions run range by range around hooks -- (1) with identity hooks every
    stage is bit-identical to the one-call run of the uncut chain; (2) with a hook that REPLACES a stage's rows (what a
    fusion layer does) everything behind the cut equals the per-segment path fed the same replaced rows; (3) the SparseConv
    geometry the tail needs (conv_out's rulebook) is attached to the last stage either way."""
    from dualfusion import ops, spconv, synth
    from dualfusion.backbones import VoxelBackBone8x
    from dualfusion.executor import build_runner
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B = 2
    m = VoxelBackBone8x(dict(NAME='VoxelBackBone8x'), 4, [1408, 1600, 40])
    import detgen                                           # lively deterministic weights / BatchNorm statistics
    sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    clouds = [torch.from_numpy(synth.kitti_sweep(seed=70 + b)[:, :4].copy()).to(dev) for b in range(B)]
    f, c = ops.hard_voxelize_clouds(clouds, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 40000)
    stages = [("conv_input", m.conv_input), ("conv1", m.conv1), ("conv2", m.conv2), ("conv3", m.conv3), ("conv4", m.conv4)]
    with torch.no_grad():
        whole = build_runner(stages, cuts=[], geometry_module=m.conv_out)
        cut = build_runner(stages, cuts=[1, 3], geometry_module=m.conv_out)
        assert whole is not None and cut is not None and len(cut.segments) == 3 and cut.full is not None
        assert cut.ranges[0][0] == 0 and cut.ranges[-1][1] == len(cut.full.specs)
        assert all(a[1] == b[0] for a, b in zip(cut.ranges, cut.ranges[1:]))
        x0 = spconv.SparseConvTensor(f, c, m.sparse_shape, B)
        want = whole.run(x0)
        geo = cut.full.build_geometry(c, f.shape[1], B, m.sparse_shape)
        got = cut.run(x0, prepared=geo)
        assert geo.handle is None                              # the last range released the handle
        for name, _ in stages:
            assert torch.equal(got[name].indices, want[name].indices), name
            assert torch.equal(got[name].features, want[name].features), name
        assert torch.equal(m.conv_out(got["conv4"]).features, m.conv_out(want["conv4"]).features)

        # a hook that replaces rows at both cuts
        def hook(i, name, t):
            if name == "conv1":
                return t.replace_feature(t.features * 0.5 + 0.25)
            if name == "conv3":
                return t.replace_feature(torch.relu(t.features - 0.1))
            return t
        ref = cut.run(x0, hook=hook)                            # per-segment plans (geometry built inside each call)
        geo = cut.full.build_geometry(c, f.shape[1], B, m.sparse_shape)
        got = cut.run(x0, hook=hook, prepared=geo)
        for name, _ in stages:
            assert torch.equal(got[name].indices, ref[name].indices), name
            assert torch.equal(got[name].features, ref[name].features), name
        assert float(want["conv4"].features.abs().max()) > 0 and not torch.equal(got["conv4"].features, want["conv4"].features)
        assert torch.equal(m.conv_out(got["conv4"]).features, m.conv_out(ref["conv4"]).features)
    torch.cuda.synchronize()


def test_native_executor_table_follows_every_parameter():
    """an in-place edit of ANY tensor the folded table is built from -- conv bias, BatchNorm bias /
    running_mean alone, not only filters / running_var / BatchNorm weight -- must rebuild the table: after each edit the
    executor's output equals the per-module path's (which reads the parameters directly)."""
    import os
    from dualfusion import synth
    from dualfusion.pipeline import CenterPointHotPath
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    model = CenterPointHotPath().eval().to(dev)
    pts = torch.from_numpy(synth.nusc_sweep(seed=2)[:20000].copy()).to(dev)
    bb = model.backbone
    bn0 = next(m for m in bb.modules() if isinstance(m, torch.nn.BatchNorm1d))
    block_conv = bb.conv1[0].conv1                                    # a BasicBlock conv: it has a bias (scn.py:68-73)
    assert block_conv.bias is not None

    def both():
        with torch.no_grad():
            feats, coors = model.voxelize([pts])
            fast = bb._stem(feats, coors, 1, model.grid_size_xyz)
            os.environ["DF3D_EXECUTOR"] = "0"
            try:
                slow = bb._stem(feats, coors, 1, model.grid_size_xyz)
            finally:
                os.environ["DF3D_EXECUTOR"] = "1"
        return fast[-1].features.clone(), slow[-1].features.clone()
    f0, s0 = both()
    assert torch.equal(f0, s0)
    # (edits under no_grad bump the tensors' version counters, as optimizers and load_state_dict do; writes through
    # `.data` bypass the counters by design and are not tracked by any of the caches)
    for edit in (lambda: bn0.bias.add_(0.37), lambda: bn0.running_mean.add_(0.21), lambda: block_conv.bias.add_(0.4)):
        with torch.no_grad():
            edit()
        f1, s1 = both()
        assert torch.equal(f1, s1), "stale executor table"
        assert not torch.equal(f1, f0)
        f0 = f1


# ------------------------------------------------------------------------- sparse conv backward
@pytest.mark.parametrize("subm", [1, 0])
def test_conv_backward_golden_and_autograd(golden, subm):
    """df3d_sparse_conv_grad_filters + the forward kernel on the inverse table against indice_conv_backward_fp32 of the
    reference's compiled CPU code (golden) -- through the autograd Function of the spconv mirror (loss.backward())."""
    from dualfusion import spconv
    from make_golden import CONV_BWD_BATCH, CONV_BWD_SHAPE, conv_bwd_case
    dev = torch.device("cuda:0")
    g = golden("conv_bwd.npz")
    ind, ks, st, pd, f, w = conv_bwd_case(subm)
    cls = spconv.SubMConv3d if subm else spconv.SparseConv3d
    conv = cls(12, 20, 3, stride=st[0], padding=pd[0], bias=True).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
    feats = torch.from_numpy(f).to(dev).requires_grad_(True)
    x = spconv.SparseConvTensor(feats, torch.from_numpy(ind).to(dev), CONV_BWD_SHAPE, CONV_BWD_BATCH)
    y = conv(x)
    oi = y.indices.cpu().numpy()
    assert np.array_equal(oi[np.lexsort(oi.T[::-1])], g["outids_%d" % subm])
    # the golden out-grad rows are in the reference's row order: map them onto ours by coordinates
    ref_ids = g["outids_%d" % subm]
    go_ref_sorted = detgen.randn("bwd_g%d" % subm, (len(oi), 20))[g["order_%d" % subm]]
    key = lambda a: (((a[:, 0] * 64 + a[:, 1]) * 64 + a[:, 2]) * 64 + a[:, 3])
    pos = np.searchsorted(key(ref_ids), key(oi))
    go = torch.from_numpy(go_ref_sorted[pos]).to(dev)
    (y.features * go).sum().backward()
    gi, gw = feats.grad.cpu().numpy(), conv.weight.grad.cpu().numpy()
    assert np.abs(gi - g["gi_%d" % subm]).max() <= 1e-4 * max(1.0, np.abs(g["gi_%d" % subm]).max())
    assert np.abs(gw - g["gw_%d" % subm]).max() <= 1e-4 * max(1.0, np.abs(g["gw_%d" % subm]).max())
    np.testing.assert_allclose(conv.bias.grad.cpu().numpy(), go_ref_sorted.sum(0), rtol=1e-4, atol=1e-4)


def test_conv_backward_inverse_table_and_full_size():
    """SubM: the mirrored forward table IS the inverse table; nuScenes-size conv4 layer: directional derivatives
    <grad_in, dX> and <grad_W, dW> equal the change of <conv(x), g> (the op is bilinear: exact up to rounding)."""
    from dualfusion import ops, synth
    from dualfusion.pipeline import CenterPointHotPath
    import os
    dev = torch.device("cuda:0")
    os.environ["DF3D_EXECUTOR"] = "0"
    try:
        model = CenterPointHotPath().eval().to(dev)
        pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
        with torch.no_grad():
            feats, coors = model.voxelize(pts)
            xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
    finally:
        os.environ["DF3D_EXECUTOR"] = "1"
    x4 = xs[3]
    blk = model.backbone.conv4[3]
    rb = x4.find_indice_pair(blk.conv1.indice_key)
    n = x4.features.shape[0]
    assert torch.equal(rb.nbr.flip(0), ops.invert_neighbors(rb.nbr, n))
    # strided layer (conv3 -> conv4 downsample): inverse table round trip
    down = model.backbone.conv4[0]
    rbd = xs[2].find_indice_pair(down.indice_key) if down.indice_key else None
    if rbd is not None:
        inv = ops.invert_neighbors(rbd.nbr, xs[2].features.shape[0])
        k, o = torch.nonzero(rbd.nbr >= 0, as_tuple=True)
        assert torch.equal(inv[k, rbd.nbr[k, o].long()], o.int())
    gen = torch.Generator().manual_seed(9)
    f = x4.features.contiguous()
    w = blk.conv1.weight.detach().view(27, 128, 128).contiguous()
    g = torch.randn(n, 128, generator=gen).to(dev)
    gi, gw = ops.sparse_conv_backward(f, w, g, rb.nbr, True)
    dx = torch.randn(n, 128, generator=gen).to(dev)
    dw = (torch.randn(27, 128, 128, generator=gen) * 0.05).to(dev)
    lhs = float((gi.double() * dx.double()).sum())
    rhs = float((ops.sparse_conv_fused(dx, w, rb.nbr, n).double() * g.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))
    lhs = float((gw.double() * dw.double()).sum())
    rhs = float((ops.sparse_conv_fused(f, dw, rb.nbr, n).double() * g.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))


def test_training_step_gradients_vs_dense_torch():
    """A training step through the spconv mirror (SubM conv -> BN(train) -> ReLU -> strided conv -> BN -> ReLU -> dense ->
    loss.backward()) against the same network written with dense torch ops in float64 on the CPU (conv3d on the
    densified tensor, masked to the active sites; BatchNorm over the active rows)."""
    from dualfusion import spconv
    F = torch.nn.functional
    dev = torch.device("cuda:0")
    shape, B, C0 = [6, 10, 12], 2, 8
    ind = detgen.clustered_voxels("train", B, shape, n_seeds=3, walk=60)
    f = detgen.randn("train_f", (len(ind), C0))
    w1 = detgen.randn("train_w1", (3, 3, 3, C0, 16), 0.3)
    w2 = detgen.randn("train_w2", (3, 3, 3, 16, 32), 0.2)
    seq = spconv.SparseSequential(spconv.SubMConv3d(C0, 16, 3, bias=False, indice_key="a"), torch.nn.BatchNorm1d(16),
                                  torch.nn.ReLU(), spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False),
                                  torch.nn.BatchNorm1d(32), torch.nn.ReLU()).to(dev).train()
    with torch.no_grad():
        seq[0].weight.copy_(torch.from_numpy(w1))
        seq[3].weight.copy_(torch.from_numpy(w2))
    feats = torch.from_numpy(f).to(dev).requires_grad_(True)
    out = seq(spconv.SparseConvTensor(feats, torch.from_numpy(ind).to(dev), shape, B)).dense()
    G = torch.from_numpy(detgen.randn("train_G", tuple(out.shape)))
    (out * G.to(dev)).sum().backward()

    # ---- dense float64 reference
    idx = torch.from_numpy(ind).long()
    fr = torch.from_numpy(f).double().requires_grad_(True)
    W1 = torch.from_numpy(w1).double().requires_grad_(True)
    W2 = torch.from_numpy(w2).double().requires_grad_(True)
    g1, b1 = torch.ones(16, dtype=torch.float64, requires_grad=True), torch.zeros(16, dtype=torch.float64, requires_grad=True)
    def densify(rows, where, dims, C):
        cl = torch.zeros(B, *dims, C, dtype=torch.float64).index_put((where[:, 0], where[:, 1], where[:, 2], where[:, 3]), rows)
        return cl.permute(0, 4, 1, 2, 3)
    X = densify(fr, idx, shape, C0)
    m0 = torch.zeros(B, 1, *shape, dtype=torch.float64)
    m0[idx[:, 0], 0, idx[:, 1], idx[:, 2], idx[:, 3]] = 1
    y1 = F.conv3d(X, W1.permute(4, 3, 0, 1, 2), padding=1)
    r1 = y1[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
    r1 = torch.relu(F.batch_norm(r1, None, None, g1, b1, training=True, eps=1e-5))
    Y1 = densify(r1, idx, shape, 16)
    m1 = F.conv3d(m0, torch.ones(1, 1, 3, 3, 3, dtype=torch.float64), stride=2, padding=1) > 0
    oid = torch.nonzero(m1[:, 0])
    y2 = F.conv3d(Y1, W2.permute(4, 3, 0, 1, 2), stride=2, padding=1)
    r2 = y2[oid[:, 0], :, oid[:, 1], oid[:, 2], oid[:, 3]]
    r2 = torch.relu(F.batch_norm(r2, None, None, torch.ones(32, dtype=torch.float64), torch.zeros(32, dtype=torch.float64),
                                 training=True, eps=1e-5))
    ref = densify(r2, oid, list(y2.shape[2:]), 32)
    assert tuple(ref.shape) == tuple(out.shape)
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-4
    (ref * G.double()).sum().backward()
    for got, want, name in ((feats.grad, fr.grad, "features"), (seq[0].weight.grad, W1.grad, "w1"),
                            (seq[3].weight.grad, W2.grad, "w2"), (seq[1].weight.grad, g1.grad, "bn1.weight"),
                            (seq[1].bias.grad, b1.grad, "bn1.bias")):
        err = float((got.cpu().double() - want).abs().max() / max(1.0, float(want.abs().max())))
        assert err <= 2e-4, (name, err)


@pytest.mark.parametrize("L,G", [(32, 70), (16, 257), (5, 3)])
def test_local_transformer_layer_on_row_kernels_vs_float64(L, G):
    """TransformerEncoderLayerPreNorm at the ACTRv2 sizes (64 channels, 4 heads of 16, FFN 128): the row-kernel path
    (`_forward_rows`: conv kernels over an identity table for the four linears, `df3d_group_attention`, fused
    add + LayerNorm) against the module's torch composition in float64 (pointformer.py:10-44)."""
    import copy
    from dualfusion import ops
    from dualfusion.pointformer import TransformerEncoderLayerPreNorm
    if ops.CONV_PRECISION != "split":
        pytest.skip("row path of the layer runs on the split-precision kernels")
    dev = torch.device("cuda:0")
    m = TransformerEncoderLayerPreNorm(d_model=64, nhead=4, dim_feedforward=128, dropout=0.0).eval()
    sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = torch.from_numpy(detgen.randn("ltl_x_%d_%d" % (L, G), (L, G, 64)))
    with torch.no_grad():
        ref = copy.deepcopy(m).double()(x.double())
        md = m.to(dev)
        assert md._rows_fit(x.to(dev), None, None)
        y = md(x.to(dev))
    err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, err
    # the attention core alone against torch's
    qkv = torch.from_numpy(detgen.randn("ltl_qkv_%d_%d" % (L, G), (L * G, 192)))
    o = ops.group_attention(qkv.to(dev), L, G, 4).cpu()
    q, k, v = [t.view(L, G, 4, 16).permute(1, 2, 0, 3).double() for t in qkv.split(64, 1)]
    want = torch.softmax(q @ k.transpose(-1, -2) / 4.0, -1) @ v                          # [G, H, L, 16]
    want = want.permute(2, 0, 1, 3).reshape(L * G, 64)
    assert float((o.double() - want).abs().max()) < 1e-5


@pytest.mark.parametrize("G", [1, 70, 2100, 5000])
def test_local_transformer_layer_as_one_kernel_vs_float64_and_row_kernels(G):
    """df3d_lt_layer (csrc/ltlayer.hip: a wave owns a group of 32 tokens from load to store, every product on the matrix
    cores in one register layout) against the module's torch composition in float64 (pointformer.py:10-44) and against
    the eight-launch row-kernel path; group counts below / above one pass of the persistent waves, input untouched."""
    import copy
    import os
    from dualfusion import ops
    from dualfusion.pointformer import TransformerEncoderLayerPreNorm
    if ops.CONV_PRECISION != "split":
        pytest.skip("row path of the layer runs on the split-precision kernels")
    dev = torch.device("cuda:0")
    m = TransformerEncoderLayerPreNorm(d_model=64, nhead=4, dim_feedforward=128, dropout=0.0).eval()
    sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = torch.from_numpy(detgen.randn("ltf_x_%d" % G, (32, G, 64))) * 1.5 + 0.3
    with torch.no_grad():
        ref = copy.deepcopy(m).double()(x.double())
        md = m.to(dev)
        xd = x.to(dev)
        keep = xd.clone()
        assert md._fused_fit(xd)
        y = md(xd)
        assert torch.equal(xd, keep)
        old = os.environ.get("DF3D_LT_FUSED")
        os.environ["DF3D_LT_FUSED"] = "0"
        try:
            rows = md(xd)
        finally:
            if old is None:
                del os.environ["DF3D_LT_FUSED"]
            else:
                os.environ["DF3D_LT_FUSED"] = old
    scale = float(ref.abs().max())
    # round 5: fp16 hi + lo operands (22 significand bits) -- the fused layer sits where torch's own fp32 composition does
    # (2e-7 .. 3e-7 of scale measured; rounds 3-4, bf16 pairs: 1e-5)
    assert float((y.cpu().double() - ref).abs().max()) < 2e-6 * scale
    assert float((y - rows).abs().max()) < 2e-6 * scale
    # a parameter update re-packs the fragments
    with torch.no_grad():
        md.linear2.bias.add_(1.0)
        y2 = md(xd)
    assert float((y2 - y - 1.0).abs().max()) < 4e-6 * scale


def test_local_transformer_chunk_as_fused_launches():
    """LocalTransformer at the ACTRv2 size of the Voxel-RCNN tree (64 channels, nsample 32, two layers) on the row-layout
    path: gather + positional MLP inside the first layer's load and the 'unique' / 'replace' write-back inside the last
    layer's store (df3d_lt_layer_gather / df3d_lt_layer_scatter: two launches for the module) against the module's own
    channel-first composition (grouping, positional convs, encoder layers, scatter: pointformer.py:349-380) in float64 on
    the CPU geometry, and against the unfused row path; points outside every group keep their features bit for bit."""
    import copy
    import os
    from dualfusion import pointformer
    from dualfusion.pointformer import LocalTransformer
    dev = torch.device("cuda:0")
    B, N, C = 2, 1500, 64
    m = LocalTransformer(96, 2.5, 32, C, C, num_layers=2, attn_feat_agg_method="unique", feat_agg_method="replace").eval()
    sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g = torch.Generator().manual_seed(9)
    xyz = torch.rand(B, N, 3, generator=g) * torch.tensor([40.0, 40.0, 3.0])
    rows = torch.randn(B, N, C, generator=g)
    md = copy.deepcopy(m).to(dev)
    old = os.environ.get("DF3D_LT_GATHER")
    with torch.no_grad():
        xd = xyz.to(dev)
        r1 = rows.clone().to(dev)
        y1 = md(xd, r1.permute(0, 2, 1))
        assert y1.data_ptr() == r1.data_ptr() and md._fused_chunk(r1.reshape(B * N, C), 32, B * 96) is not None
        os.environ["DF3D_LT_GATHER"] = "0"
        try:
            pointformer._GEO.__dict__.clear()
            r0 = rows.clone().to(dev)
            y0 = md(xd.clone(), r0.permute(0, 2, 1))
        finally:
            if old is None:
                os.environ.pop("DF3D_LT_GATHER", None)
            else:
                os.environ["DF3D_LT_GATHER"] = old
        # float64 module composition on the same geometry (group indices from the device ops)
        group_idx, group_xyz = md._geometry(xd)
        gi = group_idx.cpu().long()
        feats = rows.double().permute(0, 2, 1)                                          # [B, C, N]
        m64 = copy.deepcopy(m).double()
        grouped = torch.gather(feats[:, :, None, :].expand(B, C, 96, N), 3, gi[:, None].expand(B, C, 96, 32))
        x = grouped + m64.pe(group_xyz.cpu().double())
        x = x.permute(0, 2, 1, 3).reshape(-1, C, 32).permute(2, 0, 1)
        y = m64.chunk(x).permute(1, 2, 0).reshape(B, 96, C, 32).transpose(1, 2)         # [B, C, np, ns]
        want = feats.clone().contiguous()
        m64.scatter(want, y, group_idx.cpu())
        want = want.permute(0, 2, 1)
    scale = float(want.abs().max())
    assert float((y1.cpu().double() - want).abs().max()) < 4e-6 * scale
    assert float((y1 - y0).abs().max()) < 4e-6 * scale
    untouched = torch.ones(B, N, dtype=torch.bool)
    untouched.scatter_(1, gi.reshape(B, -1), False)
    assert int(untouched.sum()) > 0 and torch.equal(y1.cpu()[untouched], rows[untouched])
