"""Detection tail, part 2: CenterHead.predict on the MI355X (one device call) against the
golden detections of the reference's own CenterHead.predict, against the oracle on other configurations (no vel,
circular NMS, empty maps), and -- at the nuScenes map size, where last-ulp score / IoU ties are unavoidable --
through size-independent properties of the result (sortedness, mask, greedy-NMS validity)."""
import numpy as np
import pytest

import detgen
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"


def _head():
    from test_oracle_golden import _mirror_head
    return _mirror_head()[0]


def test_predict_equals_reference_golden(golden):
    from make_golden import HEAD_SHAPE, HEAD_TEST_CFG
    g = golden("centerhead.npz")
    head = _head().to(DEV)
    x = torch.from_numpy(detgen.randn("head_x_%d" % int(g["seed"]), HEAD_SHAPE)).to(DEV)
    with torch.no_grad():
        preds = head.forward_reference(x)          # library convolutions: this test isolates the decode / NMS tail
        dets = head.predict({}, preds, HEAD_TEST_CFG)
    assert len(dets) == 2
    for i, d in enumerate(dets):
        assert d["label_preds"].dtype == torch.int64
        assert d["label_preds"].cpu().numpy().tolist() == g["labels_%d" % i].tolist()
        np.testing.assert_allclose(d["scores"].cpu().numpy(), g["scores_%d" % i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(d["box3d_lidar"].cpu().numpy(), g["boxes_%d" % i], rtol=1e-4, atol=1e-4)


def _random_preds(name, B, H, W, ncls=(1, 2, 2), vel=True, hm_bias=-1.5):
    preds = []
    for t, nc in enumerate(ncls):
        d = dict(hm=detgen.randn("%s_hm%d" % (name, t), (B, nc, H, W), 1.5) + np.float32(hm_bias),
                 reg=detgen.rand("%s_reg%d" % (name, t), (B, 2, H, W)),
                 height=detgen.randn("%s_h%d" % (name, t), (B, 1, H, W), 0.8),
                 dim=detgen.randn("%s_d%d" % (name, t), (B, 3, H, W), 0.4) + np.float32(0.5),
                 rot=detgen.randn("%s_r%d" % (name, t), (B, 2, H, W)))
        if vel:
            d["vel"] = detgen.randn("%s_v%d" % (name, t), (B, 2, H, W))
        preds.append(d)
    return preds


def _cfg(pre=60, post=20, thr=0.2, score=0.1, z=(-0.6, 0.7)):
    return dict(post_center_limit_range=[-61.2, -61.2, z[0], 61.2, 61.2, z[1]],
                nms=dict(nms_pre_max_size=pre, nms_post_max_size=post, nms_iou_threshold=thr), score_threshold=score,
                pc_range=[-54, -54], out_size_factor=8, voxel_size=[0.075, 0.075])


def _tie_free(name, cfg, ncls, vel, B=2, H=10, W=12):
    for k in range(200):
        preds = _random_preds("%s_%d" % (name, k), B, H, W, ncls, vel)
        st = {}
        want = orc.centerhead_predict(preds, cfg, list(ncls), margin_out=st)
        if st["score_gap"] > 2e-6 and st["thr_gap"] > 1e-5 and st["range_gap"] > 1e-4 and st["iou_close"] == 0:
            return preds, want
    raise AssertionError("no tie-free input")


@pytest.mark.parametrize("vel,circle", [(True, False), (False, False), (True, True)])
def test_predict_vs_oracle_variants(vel, circle):
    from dualfusion.heads import CenterHead
    ncls = (1, 2, 3)
    cfg = _cfg()
    if circle:
        cfg.update(circular_nms=True, min_radius=[0.6, 0.6, 0.6])
    preds, want = _tie_free("var%d%d" % (vel, circle), cfg, ncls, vel)
    head = CenterHead.__new__(CenterHead)
    torch.nn.Module.__init__(head)
    head.num_classes = list(ncls)
    got = head.predict({"metadata": ["a", "b"]}, [{k: torch.from_numpy(v).to(DEV) for k, v in p.items()} for p in preds], cfg)
    for b, (g, w) in enumerate(zip(got, want)):
        assert g["metadata"] == "ab"[b]
        assert g["label_preds"].cpu().numpy().tolist() == w["label_preds"].tolist()
        assert g["box3d_lidar"].shape[1] == (9 if vel else 7)
        np.testing.assert_allclose(g["scores"].cpu().numpy(), w["scores"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g["box3d_lidar"].cpu().numpy(), w["box3d_lidar"], rtol=1e-4, atol=1e-4)


def test_predict_empty_and_strided_maps():
    from dualfusion import ops
    B, H, W = 2, 9, 11
    p = _random_preds("empty", B, H, W, (2,), True, hm_bias=-9.0)[0]           # nothing passes the score threshold
    rows = {k: torch.from_numpy(v).to(DEV).permute(0, 2, 3, 1).reshape(B * H * W, -1) for k, v in p.items()}
    out = ops.centerhead_predict([dict(rows, label_base=3)], B, H, W, 8, [0.075, 0.075], [-54, -54], None, 0.1,
                                 ops.NMS_ROTATED, 0.2, 50, 10)
    assert out[3].cpu().tolist() == [0, 0] and bool((out[2] == -1).all())
    # all heads as column slices of ONE row buffer (the layout the row kernels produce): same result as separate maps
    p = _random_preds("strided", B, H, W, (2,), True)[0]
    rows = {k: torch.from_numpy(v).to(DEV).permute(0, 2, 3, 1).reshape(B * H * W, -1).contiguous() for k, v in p.items()}
    order = ["reg", "height", "dim", "rot", "vel", "hm"]
    buf = torch.cat([rows[k] for k in order] + [torch.zeros(B * H * W, 4, device=DEV)], 1)
    views, c = {}, 0
    for k in order:
        views[k] = buf[:, c:c + rows[k].shape[1]]
        c += rows[k].shape[1]
    a = ops.centerhead_predict([dict(rows, label_base=0)], B, H, W, 8, [0.075, 0.075], [-54, -54], None, 0.1,
                               ops.NMS_ROTATED, 0.2, 50, 10)
    b = ops.centerhead_predict([dict(views, label_base=0)], B, H, W, 8, [0.075, 0.075], [-54, -54], None, 0.1,
                               ops.NMS_ROTATED, 0.2, 50, 10)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_predict_properties_at_nuscenes_size():
    """[B=2, 180x180, 6 tasks], pre_max 1000 / post_max 83 (the 3D-DF nuScenes test_cfg): scores sorted, masks
    honoured, labels in the task's range, and the kept set is a valid greedy NMS of the candidate list."""
    from dualfusion import ops
    B, H, W = 2, 180, 180
    ncls = (1, 2, 2, 1, 2, 2)
    preds = _random_preds("full", B, H, W, ncls, True, hm_bias=-2.19)
    cfg = _cfg(pre=1000, post=83, z=(-1.0, 1.0))
    tasks, base = [], 0
    for t, p in enumerate(preds):
        d = {k: torch.from_numpy(v).to(DEV).permute(0, 2, 3, 1).reshape(B * H * W, -1) for k, v in p.items()}
        d["label_base"] = base
        base += ncls[t]
        tasks.append(d)
    boxes, scores, labels, counts = ops.centerhead_predict(tasks, B, H, W, 8, [0.075, 0.075], [-54, -54],
                                                           cfg["post_center_limit_range"], 0.1, ops.NMS_ROTATED, 0.2, 1000, 83)
    torch.cuda.synchronize()
    counts = counts.cpu().numpy()
    assert counts.shape == (12,) and counts.min() > 10 and counts.max() <= 83
    base = np.concatenate([[0], np.cumsum(ncls)])
    for s in range(12):
        n, t = counts[s], s // B
        sc = scores[s, :n].cpu().numpy()
        bx = boxes[s, :n].cpu().numpy()
        lb = labels[s, :n].cpu().numpy()
        assert (np.diff(sc) <= 0).all() and sc.min() > 0.1
        assert ((lb >= base[t]) & (lb < base[t + 1])).all()
        assert (bx[:, 2] >= -1.0).all() and (bx[:, 2] <= 1.0).all()
        assert bool((labels[s, n:] == -1).all())
        # kept boxes do not suppress each other (in pcdet's frame, as rotate_nms_pcdet evaluates them)
        nb = bx[:, [0, 1, 2, 4, 3, 5, 8]].copy()
        nb[:, 6] = -nb[:, 6] - np.float32(np.pi / 2)
        iou = orc.boxes_pairwise_bev(nb, nb)
        np.fill_diagonal(iou, 0)
        assert iou.max() <= 0.2 + 1e-4
    # the best-scoring candidate of every segment is always kept: compare with the oracle's top score
    want = orc.centerhead_predict(preds[:1], dict(cfg, nms=dict(nms_pre_max_size=1000, nms_post_max_size=83,
                                                               nms_iou_threshold=0.2)), [1])
    for b in range(B):
        np.testing.assert_allclose(scores[b, 0].item(), want[b]["scores"][0], rtol=1e-5)
        np.testing.assert_allclose(boxes[b, 0].cpu().numpy(), want[b]["box3d_lidar"][0], rtol=1e-4, atol=1e-4)
        # and the whole kept list agrees wherever no tie interferes: at least 90 % identical rows
        k = min(counts[b], len(want[b]["scores"]))
        same = np.isclose(boxes[b, :k].cpu().numpy(), want[b]["box3d_lidar"][:k], rtol=1e-4, atol=1e-4).all(1)
        assert same.mean() > 0.9


def test_forward_rows_vs_torch_cpu_and_end_to_end():
    """CenterHead.forward on the row kernels (3 launches) against the torch fp32 CPU composition of the same module,
    then neck -> head -> predict chained on the device."""
    from make_golden import HEAD_TEST_CFG
    head = _head()
    x = torch.from_numpy(detgen.randn("head_fw_x", (2, 512, 20, 24)))
    with torch.no_grad():
        ref = head.forward_reference(x)
        hd = head.to(DEV)
        got = hd(x.to(DEV))
    assert len(got) == 6
    for t in range(6):
        assert set(got[t]) == set(ref[t])
        for k in ref[t]:
            assert tuple(got[t][k].shape) == tuple(ref[t][k].shape)
            err = float((got[t][k].cpu() - ref[t][k]).abs().max() / ref[t][k].abs().max())
            assert err < 1e-5, (t, k, err)                                   # fp32-grade against torch CPU fp32 (bar 1e-3)
    from dualfusion import ops
    if ops.CONV_PRECISION == "split":                       # other precisions take the library composition
        assert getattr(got[0]["hm"], "_df3d_rows", None) is not None
    dets = hd.predict({}, got, HEAD_TEST_CFG)
    dets_ref = hd.predict({}, [{k: v.to(DEV) for k, v in p.items()} for p in ref], HEAD_TEST_CFG)
    for a, b in zip(dets, dets_ref):
        k = min(len(a["scores"]), len(b["scores"]))
        assert k > 50
        same = np.isclose(a["box3d_lidar"][:k].cpu().numpy(), b["box3d_lidar"][:k].cpu().numpy(), rtol=1e-3, atol=1e-3).all(1)
        assert same.mean() > 0.9


def test_forward_rows_fp32_mode_vs_torch_cpu():
    """DF3D_CONV_PRECISION=fp32: the head's three conv depths on the exact-fp32 MFMA kernel (shared conv, first convs of
    two branches per launch, final convs on the column slices in place) against the torch fp32 CPU composition."""
    from dualfusion import ops
    head = _head()
    x = torch.from_numpy(detgen.randn("head_fw_x", (2, 512, 20, 24)))
    old = ops.CONV_PRECISION
    try:
        ops.CONV_PRECISION = "fp32"
        with torch.no_grad():
            ref = head.forward_reference(x)
            hd = head.to(DEV)
            got = hd(x.to(DEV))
            again = hd.forward_rows_fp32(x.to(DEV))
    finally:
        ops.CONV_PRECISION = old
    for t in range(6):
        assert set(got[t]) == set(ref[t])
        for k in ref[t]:
            assert tuple(got[t][k].shape) == tuple(ref[t][k].shape)
            err = float((got[t][k].cpu() - ref[t][k]).abs().max() / ref[t][k].abs().max())
            assert err < 1e-4, (t, k, err)
            assert torch.equal(got[t][k], again[t][k])
    assert getattr(got[0]["hm"], "_df3d_rows", None) is not None


def test_loss_on_device_equals_cpu():
    """CenterHead forward (training path: autograd through the torch modules) + loss + backward on the MI355X against the
    same computation on the CPU (which tests/test_oracle_golden.py pins to the reference's own loss)."""
    from dualfusion.heads import CenterHead
    from make_golden import HEAD_COMMON, HEAD_SHAPE, HEAD_TASKS, head_bias_shift, head_loss_example

    def run(dev):
        head = CenterHead(in_channels=512, tasks=HEAD_TASKS, dataset='nuscenes', weight=0.25,
                          code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0], common_heads=dict(HEAD_COMMON),
                          share_conv_channel=64, dcn_head=False)
        shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
        head.load_state_dict({k: torch.from_numpy(v) for k, v in head_bias_shift(detgen.det_state_dict(shapes)).items()})
        head = head.to(dev).eval()
        x = torch.from_numpy(detgen.randn("head_loss_x", HEAD_SHAPE)).to(dev).requires_grad_(True)
        ex = {k: [torch.from_numpy(a).to(dev) for a in v] for k, v in head_loss_example().items()}
        rets = head.loss(ex, head(x), {})
        sum(rets["loss"]).backward()
        return [v.item() for v in rets["loss"]], x.grad.cpu().numpy(), head.shared_conv[0].weight.grad.cpu().numpy()
    lg, xg, wg = run(DEV)
    lc, xc, wc = run("cpu")
    np.testing.assert_allclose(lg, lc, rtol=1e-4)
    assert np.abs(xg - xc).max() <= 1e-3 * np.abs(xc).max() and np.abs(wg - wc).max() <= 1e-3 * np.abs(wc).max()


def test_training_forward_backward_on_row_kernels_vs_float64():
    """CenterHead in train mode through `forward_rows_train` (shared conv + the 36 branches' first convs batched on the
    sparse-conv kernels, one BatchNorm over all branches, library final convs): every head map, the input gradient, all
    parameter gradients and the BatchNorm running statistics against the module's torch composition in float64."""
    from dualfusion import ops
    from test_gpu_neck import _train_case
    if ops.CONV_PRECISION != "split":
        pytest.skip("the training path batches the branches on the split-precision kernels")
    head = _head()
    x = torch.from_numpy(detgen.randn("head_train_x", (2, 512, 12, 16)))

    def run(mod, t):
        outs = mod(t) if t.is_cuda else mod.forward_reference(t)
        return torch.cat([outs[i][k] for i in range(len(outs)) for k in sorted(outs[i])], dim=1)
    # 36 branches x 384 pixels: ONE ReLU mask flipped by forward rounding moves a branch's BatchNorm / filter gradient by a
    # per cent (which branch changes with the input; the library composition on the same device is at 4e-7) -- the bound on
    # a single parameter is loose, the median over all parameters is not
    _train_case(head, run, x, grad_l2=4e-3, flip_l2=3e-2)
    # and the batched path is the one that ran
    hd = _head().to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    assert getattr(hd, "_train_tables", None) is None
    hd(xd)
    assert hd.__dict__.get("_train_tables")


@pytest.mark.parametrize("B,H,W", [(2, 11, 21), (1, 30, 44), (1, 14, 14), (1, 3, 2)])
def test_final_conv_forward_vs_torch_float64(B, H, W):
    """df3d_head_final_conv (round 3: taps as matrix-core columns + shifted sum; filters split in the kernel or packed once)
    against F.conv2d in float64 on the values the split rows carry: map sizes that cut the 14 x 14 tiles, 1..4 maps."""
    from dualfusion import ops
    F = torch.nn.functional
    ks = [2, 1, 3, 4, 2]
    G = len(ks)
    gen = torch.Generator().manual_seed(17)
    acts = torch.randn((B * H * W, G * 64 + 8), generator=gen)            # rows wider than the branches use
    ws = [torch.randn((k, 64, 3, 3), generator=gen) * 0.1 for k in ks]
    bs = [torch.randn((k,), generator=gen) for k in ks]
    cols, c0 = [], 0
    for k in ks:
        cols.append((c0, k))
        c0 += k
    width = (c0 + 7) // 8 * 8
    vol = acts[:, :G * 64].double().view(B, H, W, G, 64).permute(3, 0, 4, 1, 2)
    ref = torch.cat([F.conv2d(vol[g], ws[g].double(), bs[g].double(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, -1)
                     for g in range(G)], 1)
    w4, b4 = torch.zeros((G, 9, 64, 4)), torch.zeros((G, 4))
    for g, k in enumerate(ks):
        w4[g, :, :, :k] = ws[g].permute(2, 3, 1, 0).reshape(9, 64, k)
        b4[g, :k] = bs[g]
    w4d, b4d = w4.to(DEV), b4.to(DEV)
    cd = torch.tensor(cols, dtype=torch.int32, device=DEV)
    split = ops.split_rows(acts.to(DEV))
    out = ops.head_final_conv(split, B, H, W, w4d, b4d, cd, width)
    out_pk = ops.head_final_conv(split, B, H, W, w4d, b4d, cd, width, packed=ops.head_final_pack(w4d))
    assert torch.equal(out[:, :c0], out_pk[:, :c0])                      # the same operands, whoever split the filters
    err = float((out[:, :c0].cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 4e-6, err                                               # fp16 hi + lo operands: hi*hi + lo*hi + hi*lo


def test_final_conv_backward_kernels_vs_torch_float64():
    """df3d_head_final_conv_backward (data gradient, filter gradient) and the bias gather of `_HeadFinalFunction` against
    F.conv2d autograd in float64, branch by branch: ragged map sizes (tiles cut by the border), 1 / 2 / 3 maps per branch."""
    from dualfusion.heads import _HeadFinalFunction
    F = torch.nn.functional
    B, H, W = 2, 11, 21
    ks = [2, 1, 3, 2, 2, 1, 3]
    G = len(ks)
    gen = torch.Generator().manual_seed(5)
    acts = torch.randn((B * H * W, G * 64), generator=gen)
    ws = [torch.randn((k, 64, 3, 3), generator=gen) * 0.1 for k in ks]
    bs = [torch.randn((k,), generator=gen) for k in ks]
    cols, c0 = [], 0
    for k in ks:
        cols.append((c0, k))
        c0 += k
    width = (c0 + 7) // 8 * 8
    go = torch.randn((B * H * W, width), generator=gen)
    # float64 reference
    a64 = acts.double().requires_grad_(True)
    w64 = [w.double().requires_grad_(True) for w in ws]
    b64 = [b.double().requires_grad_(True) for b in bs]
    vol = a64.view(B, H, W, G, 64).permute(3, 0, 4, 1, 2)
    outs = [F.conv2d(vol[g], w64[g], b64[g], padding=1) for g in range(G)]
    ref = torch.cat([o.permute(0, 2, 3, 1).reshape(B * H * W, -1) for o in outs], 1)
    (ref * go[:, :c0].double()).sum().backward()
    # device
    ad = acts.to(DEV).requires_grad_(True)
    w4 = torch.zeros((G, 9, 64, 4))
    b4 = torch.zeros((G, 4))
    for g, k in enumerate(ks):
        w4[g, :, :, :k] = ws[g].permute(2, 3, 1, 0).reshape(9, 64, k)
        b4[g, :k] = bs[g]
    w4d, b4d = w4.to(DEV).requires_grad_(True), b4.to(DEV).requires_grad_(True)
    out = _HeadFinalFunction.apply(ad, w4d, b4d, torch.tensor(cols, dtype=torch.int32, device=DEV), width, B, H, W)
    (out * go.to(DEV)).sum().backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / float(b.abs().max()))
    assert rel(out[:, :c0], ref.detach()) < 1e-5
    assert rel(ad.grad, a64.grad) < 1e-5
    for g, k in enumerate(ks):
        assert rel(w4d.grad[g, :, :, :k], w64[g].grad.permute(2, 3, 1, 0).reshape(9, 64, k)) < 1e-5, g
        assert float(w4d.grad[g, :, :, k:].abs().max()) == 0.0 if k < 4 else True
        assert rel(b4d.grad[g, :k], b64[g].grad) < 1e-5, g


def test_branch_conv_function_vs_torch_float64():
    """`_BranchConvFunction` (grouped forward, grouped input gradient + sum, one filter-gradient launch) against G separate
    F.conv2d in float64."""
    from dualfusion import ops
    from dualfusion.heads import _BranchConvFunction
    if ops.CONV_PRECISION != "split":
        pytest.skip("split-precision kernels only")
    F = torch.nn.functional
    B, H, W, G = 2, 9, 14, 6
    gen = torch.Generator().manual_seed(6)
    rows = torch.randn((B * H * W, 64), generator=gen)
    w = torch.randn((G, 64, 64, 3, 3), generator=gen) * 0.05            # [G, cout, cin, 3, 3]
    b = torch.randn((G * 64,), generator=gen)
    go = torch.randn((B * H * W, G * 64), generator=gen)
    r64, w64, b64 = rows.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    x = r64.view(B, H, W, 64).permute(0, 3, 1, 2)
    ref = torch.cat([F.conv2d(x, w64[g], b64[g * 64:(g + 1) * 64], padding=1).permute(0, 2, 3, 1).reshape(-1, 64)
                     for g in range(G)], 1)
    (ref * go.double()).sum().backward()
    nbr = ops.conv2d_neighbors(B, H, W, 3, 3, 1, 1, False, DEV)[0]
    rd, bd = rows.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    out = _BranchConvFunction.apply(rd, wd.permute(0, 3, 4, 2, 1).reshape(G, 9, 64, 64), bd, nbr, nbr.flip(0).contiguous())
    (out * go.to(DEV)).sum().backward()
    rel = lambda a, c: float((a.detach().cpu().double() - c).abs().max() / float(c.abs().max()))
    assert rel(out, ref.detach()) < 1e-4
    assert rel(rd.grad, r64.grad) < 1e-4
    assert rel(wd.grad, w64.grad) < 1e-4
    assert rel(bd.grad, b64.grad) < 1e-4


def test_loss_rows_values_and_gradients_equal_the_autograd_loss():
    """`CenterHead.loss_rows` (losses of all tasks + the gradient of every head map from the two launches of csrc/loss.hip)
    against `CenterHead.loss` (the reference's torch composition through autograd) on the same packed maps: values 1e-5,
    gradient of the packed buffer 1e-4 of its scale; also with a task that has no positives."""
    from dualfusion import ops
    from make_golden import HEAD_SHAPE, head_loss_example
    if ops.CONV_PRECISION != "split":
        pytest.skip("the packed training maps come from the split-precision row path")
    head = _head().to(DEV).train()
    x = torch.from_numpy(detgen.randn("head_loss_x", HEAD_SHAPE)).to(DEV)
    ex = {k: [torch.from_numpy(a).to(DEV) for a in v] for k, v in head_loss_example().items()}
    ex["mask"][2] = torch.zeros_like(ex["mask"][2])                       # a task without positives
    preds = head(x)
    out, cols, geom = head.__dict__["_packed_train"]
    # reference: the torch loss on a leaf copy of the packed buffer
    leaf = out.detach().clone().requires_grad_(True)
    B, H, W = geom
    maps = leaf.view(B, H, W, -1)
    pr = [{k: (maps[..., c0:c0 + n].permute(0, 3, 1, 2).clone() if k == "hm" else maps[..., c0:c0 + n].permute(0, 3, 1, 2))
           for k, (c0, n) in t.items()} for t in cols]
    ref = head.loss(ex, pr, {})
    sum(ref["loss"]).backward()
    leaf2 = out.detach().clone().requires_grad_(True)
    head.__dict__["_packed_train"] = (leaf2, cols, geom)
    got = head.loss_rows(ex)
    assert got is not None
    sum(got["loss"]).backward()
    for k in ("loss", "hm_loss", "loc_loss"):
        np.testing.assert_allclose([float(v.detach()) for v in got[k]], [float(v.detach()) for v in ref[k]], rtol=2e-5, atol=1e-6)
    for a, b in zip(got["loc_loss_elem"], ref["loc_loss_elem"]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=1e-7)
    g, r = leaf2.grad, leaf.grad
    assert float((g - r).abs().max()) <= 1e-4 * float(r.abs().max())
    # per-task coefficients reach the gradient
    head.__dict__["_packed_train"] = (out.detach().clone().requires_grad_(True), cols, geom)
    leaf3 = head.__dict__["_packed_train"][0]
    got = head.loss_rows(ex)
    (2.0 * got["loss"][0] + 0.0 * sum(got["loss"][1:])).backward()
    c_hm = cols[0]["hm"]
    torch.testing.assert_close(leaf3.grad[:, c_hm[0]:c_hm[0] + c_hm[1]], 2.0 * r[:, c_hm[0]:c_hm[0] + c_hm[1]], rtol=1e-3, atol=1e-7)
    c1 = cols[1]["hm"]
    assert float(leaf3.grad[:, c1[0]:c1[0] + c1[1]].abs().max()) == 0.0
