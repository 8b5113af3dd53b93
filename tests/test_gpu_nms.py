"""Detection tail, part 1: rotated BEV IoU / NMS on the MI355X, through the C ABI,
against (i) the oracle, (ii) the golden vectors made by the reference's CPU path, (iii) the reference's OWN GPU
kernels (oracle/_ref/iou3d_nms_cuda.so, the hipified build of CP/det3d/ops/iou3d_nms/src) run on the same device.
IoU values: <= 1e-5 absolute (last-ulp differences of cosf / sinf / atan2f between libm and the device library);
keep indices: identical -- the tests assert that no evaluated IoU of the oracle lies within 1e-4 of the threshold,
so a flipped decision is a bug, not rounding."""
import numpy as np
import pytest

import detgen
from oracle import oracle as orc
from oracle import ref

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def clear_boxes(name, n, spread, thr, mode=True):
    """Deterministic boxes whose greedy NMS evaluates no IoU within 1e-4 of the threshold (first such seed)."""
    for k in range(20):
        b = detgen.bev_boxes("%s_%d" % (name, k), n, spread)
        want, close = orc.nms_bev(b, thr, mode, margin=1e-4)
        if close == 0:
            return b, want
    raise AssertionError("no tie-free input found")


@pytest.mark.parametrize("tag,n,spread", [("dense", 192, 6.0), ("sparse", 300, 25.0)])
def test_pairwise_iou_golden_and_oracle(golden, tag, n, spread):
    from dualfusion import iou3d_nms, ops
    g = golden("iou3d.npz")
    a = detgen.bev_boxes("iou_a_" + tag, n, spread)
    b = detgen.bev_boxes("iou_b_" + tag, n - 17, spread, special=False)
    got = iou3d_nms.boxes_iou_bev(T(a), T(b)).cpu().numpy()
    assert np.abs(got - g["iou_" + tag]).max() <= 1e-5
    assert ((got > 0) == (g["iou_" + tag] > 0)).mean() > 0.999
    ov = ops.boxes_bev_pairwise(T(a), T(b), iou=False).cpu().numpy()
    assert np.abs(ov - orc.boxes_pairwise_bev(a, b, mode="overlap")).max() <= 1e-4
    assert ops.boxes_bev_pairwise(T(a[:0]), T(b)).shape == (0, n - 17)


@pytest.mark.parametrize("n,spread,thr", [(192, 6.0, 0.2), (300, 25.0, 0.7), (1000, 30.0, 0.2), (1, 1.0, 0.5), (64, 3.0, 0.1),
                                          (65, 3.0, 0.1), (4096, 150.0, 0.2)])
def test_nms_rotated_vs_oracle(n, spread, thr):
    from dualfusion import ops
    b, want = clear_boxes("nms_%d" % n, n, spread, thr)
    keep, num = ops.nms_bev(T(b), thr, ops.NMS_ROTATED)
    assert keep[:int(num)].cpu().numpy().tolist() == want.tolist()
    keep2, num2 = ops.nms_bev(T(b), thr, ops.NMS_ROTATED, max_keep=7)
    assert keep2[:int(num2)].cpu().numpy().tolist() == want[:7].tolist()


def test_nms_golden_keep_lists(golden):
    from dualfusion import ops
    g = golden("iou3d.npz")
    for tag, n, spread in (("dense", 192, 6.0), ("sparse", 300, 25.0)):
        a = detgen.bev_boxes("iou_a_" + tag, n, spread)
        for thr in (0.2, 0.7):
            if (np.abs(g["self_iou_" + tag] - thr) < 1e-4).any():
                continue
            keep, num = ops.nms_bev(T(a), thr, ops.NMS_ROTATED)
            assert keep[:int(num)].cpu().numpy().tolist() == g["keep_%s_%d" % (tag, int(thr * 100))].tolist()


def test_nms_batched_lists_counts_and_modes():
    from dualfusion import ops
    cap, S = 700, 5
    counts = np.array([700, 0, 1, 129, 640], np.int32)
    for mode, omode, thr in ((ops.NMS_ROTATED, True, 0.2), (ops.NMS_NORMAL, False, 0.3), (ops.NMS_CIRCLE, "circle", 2.0)):
        boxes = np.stack([clear_boxes("nmsb_%d" % s, cap, 12.0 + s, thr, omode)[0] for s in range(S)])
        keep, num = ops.nms_bev(T(boxes), thr, mode, counts=T(counts), max_keep=83)
        keep, num = keep.cpu().numpy(), num.cpu().numpy()
        for s in range(S):
            want, close = orc.nms_bev(boxes[s, :counts[s]], thr, omode, margin=1e-5)
            assert num[s] == min(len(want), 83)
            assert keep[s, :num[s]].tolist() == want[:83].tolist()


def test_mirrored_interface_rotate_nms_pcdet():
    from dualfusion import iou3d_nms
    s = detgen.rand("pcdet_s", (900,))
    for k in range(20):
        b = detgen.bev_boxes("pcdet_%d" % k, 900, 20.0)
        want, close = orc.rotate_nms_pcdet(b, s, 0.2, pre_maxsize=500, post_max_size=83, margin=1e-4)
        want2, close2 = orc.nms_bev(b[np.argsort(-s, kind="stable")[:500]], 0.2, True, margin=1e-4)
        if close == 0 and close2 == 0:
            break
    assert close == 0 and close2 == 0
    bt = T(b)
    sel = iou3d_nms.rotate_nms_pcdet(bt, T(s), 0.2, pre_maxsize=500, post_max_size=83)
    assert sel.dtype == torch.int64 and sel.cpu().numpy().tolist() == want.tolist()
    assert torch.equal(bt, T(b))                                    # like the reference, the caller's boxes are untouched
    sel2, _ = iou3d_nms.nms_gpu(T(b), T(s), 0.2, pre_maxsize=500)
    assert sel2.cpu().numpy().tolist() == np.argsort(-s, kind="stable")[:500][want2].tolist()
    iou3 = iou3d_nms.boxes_iou3d_gpu(T(b[:50]), T(b[:50])).cpu().numpy()
    assert np.allclose(np.diag(iou3), 1.0, atol=1e-5)
    assert iou3d_nms.rotate_nms_pcdet(T(b[:0]), T(s[:0]), 0.2).numel() == 0


@pytest.mark.skipif(not ref.available("iou3d_nms_cuda"), reason="oracle/_ref not built")
def test_against_reference_gpu_kernels_on_this_device():
    """The reference's own kernels (hipified by torch's build) on the MI355X, beside ours."""
    from dualfusion import ops
    m = ref.load("iou3d_nms_cuda")
    a = detgen.bev_boxes("refgpu_a", 1000, 30.0)
    at = T(a)
    iou_ref = torch.zeros(1000, 1000, device=DEV)
    m.boxes_iou_bev_gpu(at, at, iou_ref)
    iou = ops.boxes_bev_pairwise(at, at)
    assert float((iou - iou_ref).abs().max()) <= 1e-5
    ov_ref = torch.zeros(1000, 1000, device=DEV)
    m.boxes_overlap_bev_gpu(at, at, ov_ref)
    assert float((ops.boxes_bev_pairwise(at, at, iou=False) - ov_ref).abs().max()) <= 1e-4
    for thr, normal in ((0.2, False), (0.5, False), (0.3, True)):
        keep_ref = torch.zeros(1000, dtype=torch.long)
        n_ref = (m.nms_normal_gpu if normal else m.nms_gpu)(at, keep_ref, thr)
        keep, num = ops.nms_bev(at, thr, ops.NMS_NORMAL if normal else ops.NMS_ROTATED)
        if normal or not bool(((iou_ref - thr).abs() < 1e-4).any()):
            assert int(num) == n_ref and keep[:n_ref].cpu().tolist() == keep_ref[:n_ref].tolist()


# ---- df3d_topk_keys: the selection step of both heads, against a full numpy sort
def _score_keys(rs, S, n, kind):
    """[segment | 0x3F800000 - score bits | index] keys like the heads build them."""
    if kind == "uniform":
        score = rs.uniform(0.0, 1.0, size=(S, n)).astype(np.float32)
    elif kind == "peaked":                                  # sigmoid of wide logits: many scores next to 1.0
        score = (1.0 / (1.0 + np.exp(-rs.normal(4.0, 3.0, size=(S, n))))).astype(np.float32)
    elif kind == "saturated":                               # thousands of exactly equal top scores: ties by index
        score = np.where(rs.uniform(size=(S, n)) < 0.5, np.float32(1.0), rs.uniform(0, 1, size=(S, n)).astype(np.float32))
    elif kind == "clustered":                               # a near-constant heat map: thousands of scores within 1e-6
        score = (0.1006 + rs.uniform(0, 1e-6, size=(S, n))).astype(np.float32)
    elif kind == "zeros":                                   # fewer positive scores than k
        score = np.where(rs.uniform(size=(S, n)) < 0.002, rs.uniform(0, 1, size=(S, n)), 0.0).astype(np.float32)
    inv = (np.uint64(0x3F800000) - score.view(np.uint32).astype(np.uint64))
    idx = np.broadcast_to(rs.permutation(n).astype(np.uint64), (S, n))
    keys = (np.arange(S, dtype=np.uint64)[:, None] << np.uint64(56)) | (inv << np.uint64(24)) | idx
    if kind == "uniform":
        keys = np.where(rs.uniform(size=(S, n)) < 0.3, np.uint64(0xFFFFFFFFFFFFFFFF), keys)      # masked entries
    return keys


@pytest.mark.parametrize("S,n,k,kind", [(3, 32400, 1000, "uniform"), (2, 324000, 200, "peaked"), (2, 50000, 500, "saturated"),
                                        (2, 40000, 300, "zeros"), (6, 32400, 1000, "clustered"), (5, 168, 60, "uniform"), (1, 100, 4096, "uniform"),
                                        (2, 20000, 4096, "saturated"), (1, 7, 1, "peaked")])
def test_topk_keys_equals_full_sort(S, n, k, kind):
    from dualfusion import ops
    rs = np.random.RandomState(S * 1000 + k)
    keys = _score_keys(rs, S, n, kind)
    out, cnt = ops.topk_keys(torch.from_numpy(keys.view(np.int64)).to(DEV), k)
    want = np.sort(keys, axis=1)[:, :k]
    if n < k:
        want = np.concatenate([want, np.full((S, k - n), 0xFFFFFFFFFFFFFFFF, np.uint64)], 1)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    assert cnt.cpu().numpy().tolist() == (want != np.uint64(0xFFFFFFFFFFFFFFFF)).sum(1).tolist()


def test_topk_keys_rejects_bad_arguments():
    from dualfusion import Df3dError, ops
    keys = torch.zeros((2, 100), dtype=torch.int64, device=DEV)
    with pytest.raises(Df3dError):
        ops.topk_keys(keys, 4097)
    with pytest.raises(Df3dError):
        ops.topk_keys(keys.cpu(), 10)
