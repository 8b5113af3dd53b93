"""Detection tail, part 3: TransFusionHead (LiDAR-only branch) on the MI355X -- the
row-kernel convolutions, `df3d_heatmap_proposals` and `df3d_transfusion_decode` through the C ABI -- against the
golden outputs of the reference's own TransFusionHead.forward / get_bboxes, against the oracle restatement on other
inputs, and at the nuScenes map size (180 x 180, 200 proposals) against the same module's plain-torch path plus
size-independent properties (local maxima, sortedness, range mask)."""
import numpy as np
import pytest

import detgen
import oracle_models as om

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"
NAMES = ["center", "height", "dim", "rot", "vel", "heatmap", "query_heatmap_score", "dense_heatmap"]


def _golden_head():
    from test_oracle_golden import _mirror_tf_head
    head, _, sd = _mirror_tf_head()
    return head.to(DEV), sd


def _close(got, want, tol, what):
    assert got.shape == want.shape, what
    err = np.abs(got - want).max()
    assert err <= tol * max(1.0, np.abs(want).max()), "%s: max err %g" % (what, err)


def test_forward_and_boxes_equal_reference_golden(golden):
    from make_golden import TFH_KW, TFH_SHAPE
    from dualfusion import ops
    g = golden("transfusion_head.npz")
    head, _ = _golden_head()
    x = torch.from_numpy(detgen.randn("tfh_x_%d" % int(g["seed"]), TFH_SHAPE)).to(DEV)
    kt = ops.KernelTimer()
    kt.start()
    with torch.no_grad():
        res = head([x], None, [{}])
        dets = head.get_bboxes(res)
    torch.cuda.synchronize()
    recs = sorted((r["cin"], r["cout"]) for r in kt.stop())
    if ops.CONV_PRECISION == "split":                       # (DF3D_CONV_PRECISION=fp32 / bf16 select the module's torch path)
        assert recs == [(128, 32), (128, 128), (512, 128)]   # row kernels, not torch convs
    assert head.query_labels.cpu().numpy().tolist() == g["query_labels"].tolist()
    for name in NAMES:
        _close(res[0][0][name].cpu().numpy(), g["pred_" + name], 1e-3, name)
    assert len(dets) == TFH_SHAPE[0]
    for b, (box, score, lab) in enumerate(dets):
        assert lab.dtype == torch.int32 and lab.cpu().numpy().tolist() == g["labels_%d" % b].tolist()
        np.testing.assert_allclose(score.cpu().numpy(), g["scores_%d" % b], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(box.cpu().numpy(), g["boxes_%d" % b], rtol=1e-3, atol=1e-3)
    assert sum(len(d[1]) for d in dets) < TFH_SHAPE[0] * TFH_KW["num_proposals"]


def _np_proposals(heat, K, exempt, pad=1):
    """numpy restatement of transfusion_head.py:845-878 on logits [B, C, H, W] (float32 sigmoid by torch-CPU)."""
    s = torch.from_numpy(heat).sigmoid().numpy()
    B, C, H, W = s.shape
    keep = np.zeros_like(s, dtype=bool)
    inner = np.ones((B, C, H - 2 * pad, W - 2 * pad), bool)
    for dy in range(-pad, pad + 1):
        for dx in range(-pad, pad + 1):
            inner &= s[:, :, pad:H - pad, pad:W - pad] >= s[:, :, pad + dy:H - pad + dy, pad + dx:W - pad + dx]
    keep[:, :, pad:H - pad, pad:W - pad] = inner
    for c in exempt:
        keep[:, c] = True
    sup = np.where(keep, s, np.float32(0)).reshape(B, C, H * W)
    top = np.argsort(-sup.reshape(B, -1), axis=1, kind="stable")[:, :K]
    return top // (H * W), top % (H * W), sup


@pytest.mark.parametrize("shape,K,exempt,ks", [((2, 10, 20, 22), 24, (8, 9), 3), ((1, 3, 33, 17), 50, (1, 2), 3),
                                                 ((3, 5, 16, 16), 300, (), 5), ((1, 10, 180, 180), 200, (8, 9), 3)])
def test_heatmap_proposals_vs_numpy(shape, K, exempt, ks):
    from dualfusion import ops
    B, C, H, W = shape
    heat = detgen.randn("prop_heat_%d_%d" % (H, K), shape, 2.0)
    feat = detgen.randn("prop_feat_%d_%d" % (H, K), (B * H * W, 16))
    cw, cb = detgen.randn("prop_cw", (16, C)), detgen.randn("prop_cb", (16,))
    cls, pix, sup = _np_proposals(heat, K, exempt, ks // 2)
    rows = torch.from_numpy(np.ascontiguousarray(heat.transpose(0, 2, 3, 1).reshape(B * H * W, C)))
    wide = torch.zeros((B * H * W, 16), dtype=torch.float32)
    wide[:, :C] = rows                                                     # a row stride larger than the class count
    tc, tp, qs, qp, qf = ops.heatmap_proposals(wide.to(DEV)[:, :C], B, C, H, W, ks, exempt, K, torch.from_numpy(feat).to(DEV),
                                               torch.from_numpy(cw).to(DEV), torch.from_numpy(cb).to(DEV))
    tc, tp = tc.cpu().numpy(), tp.cpu().numpy()
    got = np.take_along_axis(sup.reshape(B, -1), tc * (H * W) + tp, 1)
    want = np.take_along_axis(sup.reshape(B, -1), cls * (H * W) + pix, 1)
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)               # same scores in the same (descending) order
    same = (tc == cls) & (tp == pix)
    assert same.mean() > 0.99 and (np.diff(got, axis=1) <= 0).all()        # sigmoid is last-ulp different at most
    qs = qs.cpu().numpy()
    for b in range(B):
        np.testing.assert_allclose(qs[b], sup[b][:, tp[b]], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(qp[b].cpu().numpy(), np.stack([tp[b] % W + 0.5, tp[b] // W + 0.5], 1).astype(np.float32))
        wantf = feat.reshape(B, H * W, 16)[b][tp[b]] + cw[:, tc[b]].T + cb
        np.testing.assert_allclose(qf[b].cpu().numpy(), wantf, rtol=1e-6, atol=1e-6)
    if ks == 3 and not exempt:
        ys, xs = tp // W, tp % W
        assert ((ys > 0) & (ys < H - 1) & (xs > 0) & (xs < W - 1))[got > 0].all()   # border pixels never survive


def test_heatmap_proposals_rejects_bad_arguments():
    from dualfusion import Df3dError, ops
    heat = torch.zeros((400, 10), device=DEV)
    with pytest.raises(Df3dError):
        ops.heatmap_proposals(heat, 1, 10, 20, 20, 1, (), 8)               # the reference's slice is empty for kernel 1
    with pytest.raises(Df3dError):
        ops.heatmap_proposals(heat, 1, 10, 20, 20, 3, (), 4001)
    with pytest.raises(ValueError):
        ops.heatmap_proposals(heat.cpu(), 1, 10, 20, 20, 3, (), 8)


@pytest.mark.parametrize("thr,vel", [(0.0, True), (0.05, True), (0.02, False)])
def test_decode_vs_oracle(thr, vel):
    from dualfusion import ops
    B, K, C = 3, 333, 10
    coder = dict(pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], out_size_factor=8,
                 post_center_range=[-40.0, -45.0, -1.5, 45.0, 40.0, 1.5], score_threshold=thr)
    preds = dict(heatmap=detgen.randn("dec_hm", (B, C, K), 2.0), center=detgen.rand("dec_c", (B, 2, K), 0, 180),
                 height=detgen.randn("dec_h", (B, 1, K)), dim=detgen.randn("dec_d", (B, 3, K), 0.5),
                 rot=detgen.randn("dec_r", (B, 2, K)), query_heatmap_score=detgen.rand("dec_q", (B, C, K)))
    if vel:
        preds["vel"] = detgen.randn("dec_v", (B, 2, K))
    labels = (detgen.rand("dec_l", (B, K)) * C).astype(np.int64) % C
    preds["query_heatmap_score"][0, :, :7] = 0.0                            # zero scores: label falls back to class 0
    want = om.transfusion_get_bboxes(preds, labels, K, coder)
    rows = {k: torch.from_numpy(np.ascontiguousarray(v.transpose(0, 2, 1).reshape(B * K, -1))).to(DEV)
            for k, v in preds.items() if k != "query_heatmap_score"}
    boxes, scores, labs, counts = ops.transfusion_decode(rows, torch.from_numpy(preds["query_heatmap_score"]).to(DEV),
                                                         torch.from_numpy(labels.astype(np.int32)).to(DEV), B, K, C, 8,
                                                         coder["voxel_size"], coder["pc_range"],
                                                         coder["post_center_range"], thr)
    counts = counts.cpu().numpy()
    n_all = 0
    for b in range(B):
        wb, ws, wl = want[b]
        assert counts[b] == len(ws)
        assert labs[b, :counts[b]].cpu().numpy().tolist() == wl.tolist()
        np.testing.assert_allclose(scores[b, :counts[b]].cpu().numpy(), ws, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(boxes[b, :counts[b]].cpu().numpy(), wb, rtol=1e-5, atol=1e-5)
        n_all += counts[b]
    assert 0 < n_all < B * K and boxes.shape[-1] == (9 if vel else 7)


def _nusc_head():
    from dualfusion.transfusion_head import TransFusionHead
    from make_golden import tfh_weight_shift
    head = TransFusionHead(num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
                           num_decoder_layers=1, num_heads=8, learnable_query_pos=False, initialize_by_heatmap=True,
                           nms_kernel_size=3, ffn_channel=256, dropout=0.1, bn_momentum=0.1, activation='relu',
                           common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
                           bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075],
                                           out_size_factor=8, post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                           score_threshold=0.0, code_size=10),
                           loss_cls=dict(type='FocalLoss', use_sigmoid=True),
                           test_cfg=dict(dataset='nuScenes', grid_size=[1440, 1440, 40], out_size_factor=8,
                                         pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], nms_type=None))
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = tfh_weight_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return head.eval().to(DEV)


def test_nuscenes_size_against_torch_path_and_properties():
    """[2, 512, 180, 180] -> 200 proposals (transfusion_nusc_voxel_F.py:244-300): device path vs the plain-torch path
    of the same module on the same GPU (library convolutions, full argsort), and properties of the result.  Two of the
    ~94 000 candidate scores of a sample can be closer than the fp32 noise of the two convolution implementations, so
    proposals are matched by (class, pixel) rather than by rank before their predictions are compared."""
    head = _nusc_head()
    B, K, HW = 2, 200, 180 * 180
    x = torch.from_numpy(detgen.randn("tfh_full", (B, 512, 180, 180))).to(DEV)
    with torch.no_grad():
        got = head([x])[0][0]
        lab_got = head.query_labels.clone()
        dets = head.get_bboxes(([got],))
        want = head.forward_reference(x)[0]
        lab_want = head.query_labels.clone()
    _close(got["dense_heatmap"].cpu().numpy(), want["dense_heatmap"].cpu().numpy(), 1e-3, "dense_heatmap")
    # each path's proposals are the top-k of its own dense heat map (numpy restatement of the selection)
    keys = []
    for res, lab in ((got, lab_got), (want, lab_want)):
        cls, pix, _ = _np_proposals(res["dense_heatmap"].cpu().numpy(), K, (8, 9))
        assert (cls == lab.cpu().numpy()).mean() >= 0.99
        keys.append(cls * HW + pix)
    n_checked = 0
    for b in range(B):
        common, ig, iw = np.intersect1d(keys[0][b], keys[1][b], return_indices=True)
        assert len(common) >= K - 2                                        # at most a swap at the cut
        if len(common) < K:
            continue                                                       # another query set: self-attention differs
        for name in NAMES[:-1]:
            g_, w_ = got[name][b].cpu().numpy()[:, ig], want[name][b].cpu().numpy()[:, iw]
            _close(g_, w_, 2e-3, name)
        n_checked += 1
    assert n_checked >= 1
    # the proposals' own-class scores are positive local maxima of the dense map, in descending order
    own = got["query_heatmap_score"].gather(1, lab_got[:, None, :]).squeeze(1)
    assert (own > 0).all() and (own[:, 1:] <= own[:, :-1]).all()
    for b, (box, score, lab) in enumerate(dets):
        assert box.shape[1] == 9 and len(score) == len(lab) == len(box) <= 200
        assert (box[:, :2].abs() <= 61.2).all() and (box[:, 3:6] > 0).all() and (score >= 0).all() and (score <= 1).all()


def test_training_mode_uses_autograd_path():
    head, _ = _golden_head()
    from make_golden import TFH_SHAPE
    head.train()
    x = torch.from_numpy(detgen.randn("tfh_train", TFH_SHAPE)).to(DEV).requires_grad_(True)
    res = head([x])[0][0]
    (res["heatmap"].sum() + res["center"].sum() + res["dense_heatmap"].sum()).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    assert head.shared_conv.weight.grad is not None and head.decoder[0].multihead_attn.in_proj_weight.grad.abs().sum() > 0


@pytest.mark.parametrize("B,nq,nk,heads", [(1, 200, 32400, 8), (2, 24, 440, 8), (3, 300, 1000, 4), (1, 1, 17, 2),
                                           (2, 16, 16, 8)])
def test_cross_attention_vs_float64(B, nq, nk, heads):
    """df3d_cross_attention (keys split over workgroups + log-sum-exp merge) against softmax(q k^T / sqrt(d)) v in
    float64; k and v are column slices of one [B*nk, 2E] projection buffer, as the head passes them."""
    from dualfusion import ops
    E = heads * 16
    q = detgen.randn("xa_q_%d_%d" % (nq, nk), (B * nq, E))
    kv = detgen.randn("xa_kv_%d_%d" % (nq, nk), (B * nk, 2 * E))
    kv[:, :E] *= 1.5                                                       # scores of a few units: a peaked softmax
    qd, kvd = torch.from_numpy(q).to(DEV), torch.from_numpy(kv).to(DEV)
    out = ops.cross_attention(qd, kvd[:, :E], kvd[:, E:], B, heads, 0.25).cpu().numpy()
    q64 = q.astype(np.float64).reshape(B, nq, heads, 16).transpose(0, 2, 1, 3)
    k64 = kv[:, :E].astype(np.float64).reshape(B, nk, heads, 16).transpose(0, 2, 1, 3)
    v64 = kv[:, E:].astype(np.float64).reshape(B, nk, heads, 16).transpose(0, 2, 1, 3)
    s = np.einsum("bhqd,bhkd->bhqk", q64, k64) * 0.25
    p = np.exp(s - s.max(-1, keepdims=True))
    want = np.einsum("bhqk,bhkd->bhqd", p / p.sum(-1, keepdims=True), v64).transpose(0, 2, 1, 3).reshape(B * nq, E)
    assert np.abs(out - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


def test_cross_attention_rejects_bad_arguments():
    from dualfusion import Df3dError, ops
    q, kv = torch.zeros((8, 96), device=DEV), torch.zeros((64, 96), device=DEV)
    with pytest.raises(ValueError):
        ops.cross_attention(q, kv, kv, 1, 8)                               # 96 columns are not 8 heads x 16
    with pytest.raises(ValueError):
        ops.cross_attention(q.cpu(), kv.cpu(), kv.cpu(), 1, 6)
    flat = torch.zeros((8 * 80 + 1,), device=DEV)
    with pytest.raises(Df3dError):
        ops.cross_attention(flat[1:].view(8, 80), kv[:, :80].contiguous(), kv[:, :80].contiguous(), 1, 5)   # q not 16-byte aligned


@pytest.mark.parametrize("layers,aux", [(2, True), (3, False)])
def test_multi_layer_decoder_device_path_vs_torch_path(layers, aux):
    """num_decoder_layers > 1 (the reference's default is 3): query positions move to the predicted centres between
    layers, auxiliary=True concatenates the per-layer predictions along the proposal axis (transfusion_head.py:
    886-905,1016-1028).  Device path vs the same module's plain-torch path on a small map."""
    from dualfusion.transfusion_head import TransFusionHead
    from make_golden import TFH_CODER, TFH_KW, TFH_SHAPE, TFH_TEST_CFG, tfh_weight_shift
    kw = dict(TFH_KW, num_decoder_layers=layers, auxiliary=aux)
    head = TransFusionHead(loss_cls=dict(use_sigmoid=True), test_cfg=dict(TFH_TEST_CFG),
                           bbox_coder=dict(type='TransFusionBBoxCoder', **TFH_CODER), **kw)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = tfh_weight_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    head = head.eval().to(DEV)
    x = torch.from_numpy(detgen.randn("tfh_x_0", TFH_SHAPE)).to(DEV)         # the golden's tie-free input
    with torch.no_grad():
        got = head([x])[0][0]
        lab = head.query_labels.clone()
        want = head.forward_reference(x)[0]
    K = TFH_KW["num_proposals"]
    assert torch.equal(lab, head.query_labels)
    assert got["heatmap"].shape[-1] == (K * layers if aux else K)
    assert sorted(got) == sorted(want)
    for name in got:
        _close(got[name].cpu().numpy(), want[name].cpu().numpy(), 1e-3, name)
    if not aux:
        # like the reference, only the FIRST layer's dict carries 'query_heatmap_score' (transfusion_head.py:1013-1020),
        # so get_bboxes needs auxiliary=True (or one layer) -- same KeyError as the reference otherwise
        assert "query_heatmap_score" not in got
        return
    with torch.no_grad():
        boxes = head.get_bboxes(([got],))
        boxes_ref = head.get_bboxes(([{k: v.clone() for k, v in want.items()}],))
    for (b0, s0, l0), (b1, s1, l1) in zip(boxes, boxes_ref):
        assert l0.cpu().numpy().tolist() == l1.cpu().numpy().tolist()
        np.testing.assert_allclose(s0.cpu().numpy(), s1.cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(b0.cpu().numpy(), b1.cpu().numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("B,nq,nk,H", [(2, 200, 3000, 8), (1, 37, 1001, 2), (3, 256, 517, 4)])
def test_cross_attention_training_kernels(B, nq, nk, H):
    """df3d_cross_attention_train / _backward (the decoder layer's cross-attention of a training step: multi_head_attention_forward,
    transfusion_head.py:478-495) against the float64 composition softmax(scale q k^T) -> dropout -> . v and its autograd
    gradients: ragged query / key tiles, keys over several chunks; with p = 0.1 the composition uses the SAME mask (the hash over
    ((b h) q) key, read off a relu_dropout_ call on ones with that seed) and the mask keeps ~90 %."""
    from dualfusion import ops
    dev = torch.device("cuda:0")
    E = H * 16
    gen = torch.Generator().manual_seed(B * 1000 + nq)
    q, k, v = (torch.randn((B, n, E), generator=gen) for n in (nq, nk, nk))
    q = q * 1.5
    go = torch.randn((B, nq, E), generator=gen)
    for p in (0.0, 0.1):
        seed = 4242
        qa, ka, va = (t.to(dev).requires_grad_(True) for t in (q, k, v))
        out = ops.cross_attention_train(qa, ka, va, H, 0.25, p, seed=seed)
        out.backward(go.to(dev))
        mask = torch.ones((B, H, nq, nk), dtype=torch.float64)
        if p:
            kept = ops.relu_dropout_(torch.ones(B * H * nq * nk, device=dev), p, seed=seed).cpu().view(B, H, nq, nk)
            assert abs(float((kept != 0).double().mean()) - (1 - p)) < 0.01
            mask = kept.double()
        qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
        heads = lambda t: t.view(B, t.shape[1], H, 16).transpose(1, 2)
        pr = torch.softmax(heads(qr) @ heads(kr).transpose(-1, -2) * 0.25, -1) * mask
        ref = (pr @ heads(vr)).transpose(1, 2).reshape(B, nq, E)
        ref.backward(go.double())
        for a, b, name in ((out, ref, "out"), (qa.grad, qr.grad, "dq"), (ka.grad, kr.grad, "dk"), (va.grad, vr.grad, "dv")):
            err = float((a.detach().cpu().double() - b.detach()).abs().max()) / float(b.detach().abs().max())
            assert err <= 2e-5, (name, p, err)
    # the inference entry gives the same output as the training entry without dropout
    plain = ops.cross_attention(q.to(dev).view(B * nq, E), k.to(dev).view(B * nk, E), v.to(dev).view(B * nk, E), B, H, scale=0.25)
    with torch.no_grad():
        again = ops.cross_attention_train(q.to(dev), k.to(dev), v.to(dev), H, 0.25, 0.0)
    assert torch.equal(plain.view(B, nq, E), again)
