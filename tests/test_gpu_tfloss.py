"""TransFusionHead target assignment + losses on the MI355X (csrc/tfloss.hip through
the C ABI): rotated overlap of xyxyr boxes vs the oracle restatement AND the reference's own GPU kernel
(oracle/_ref/iou3d_cuda.so, built from TF/mmdet3d/ops/iou3d/src), matching costs / Gaussian targets / losses vs the
golden outputs of the reference's `TransFusionHead.loss` (tests/golden/transfusion_head_loss.npz), the device path vs the
module's plain-torch path (values and gradients), nuScenes-size properties, and RCCL at world size 1."""
import os
import subprocess
import sys

import numpy as np
import pytest

import detgen

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _xyxyr(v):
    return np.stack([v[:, 0] - v[:, 3] / 2, v[:, 1] - v[:, 4] / 2, v[:, 0] + v[:, 3] / 2, v[:, 1] + v[:, 4] / 2, v[:, 6]], 1).astype(np.float32)


def test_overlap_xyxyr_vs_oracle_and_reference_kernel():
    from dualfusion import ops
    from oracle import oracle as orc, ref
    a = _xyxyr(detgen.bev_boxes("tfov_a", 257, 9.0))
    b = _xyxyr(detgen.bev_boxes("tfov_b", 190, 9.0, special=False))
    got = ops.boxes_overlap_bev_xyxyr(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
    want = orc.tf_boxes_overlap_bev(a, b)
    assert (want > 0).sum() > 300
    # float polygon clipping, operation for operation: differences come from the last ulps of cosf / sinf / atan2f only
    assert np.abs(got - want).max() < 2e-4 and np.median(np.abs(got - want)[want > 0]) < 2e-6
    if ref.available("iou3d_cuda"):                          # the reference's own kernel on the same GPU
        m = ref.load("iou3d_cuda")
        out = torch.zeros((len(a), len(b)), device=DEV)
        m.boxes_overlap_bev_gpu(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), out)
        torch.cuda.synchronize()
        r = out.cpu().numpy()
        assert np.abs(got - r).max() < 2e-4 and np.abs(want - r).max() < 2e-4      # pins the oracle restatement too
    assert ops.boxes_overlap_bev_xyxyr(torch.zeros((0, 5), device=DEV), torch.from_numpy(b).to(DEV)).shape == (0, len(b))
    with pytest.raises(Exception):
        ops.boxes_overlap_bev_xyxyr(torch.from_numpy(a), torch.from_numpy(b))        # CPU tensors are refused


def _head():
    from test_host_modules import _tfl_head
    head, _ = _tfl_head()
    return head.to(DEV)


def _gt(g, B):
    from dualfusion.box3d import LiDARInstance3DBoxes
    return ([LiDARInstance3DBoxes(torch.from_numpy(g["gt_boxes_%d" % b]), box_dim=9) for b in range(B)],
            [torch.from_numpy(g["gt_labels_%d" % b]) for b in range(B)])


def test_targets_and_losses_vs_reference_golden(golden):
    """Both product paths on the device -- `loss` (torch composition over the IoU kernel) and `loss_device` (tfloss.hip) --
    against the reference module's outputs."""
    from make_golden import TFL_SHAPE
    from dualfusion import ops
    g = golden("transfusion_head_loss.npz")
    B = TFL_SHAPE[0]
    head = _head()
    x = torch.from_numpy(detgen.randn("tfl_x_%d" % int(g["seed"]), TFL_SHAPE)).to(DEV)
    gt_boxes, gt_labels = _gt(g, B)
    with torch.enable_grad():                               # the plain-torch (autograd) forward: same summation order class
        xg = x.clone().requires_grad_(True)
        res = head([xg], None, [{}])
    p = res[0][0]
    for k in ("center", "height", "dim", "rot", "vel", "heatmap"):
        assert np.abs(p[k].detach().cpu().numpy() - g["pred_" + k]).max() < 1e-3, k
    # matching costs, both formulations
    gt, lab, off, counts = head._pack_gt(gt_boxes, gt_labels, torch.device(DEV))
    rows = torch.cat([p[k] for k in ("center", "height", "dim", "rot", "vel", "heatmap")], 1).detach().permute(0, 2, 1).contiguous()
    cost, iou, boxes = ops.tf_match_cost(rows, 10, 10, gt, lab, off, max(counts), **head._match_cfg())
    for b in range(B):
        assert np.abs(cost[b, :, :counts[b]].cpu().numpy() - g["cost_%d" % b]).max() < 1e-3
        tb = head.bbox_coder.decode(*[p[k][b:b + 1].detach().clone() for k in ("heatmap", "rot", "dim", "center", "height", "vel")])[0]["bboxes"]
        assert np.abs(boxes[b].cpu().numpy() - tb[:, :7].cpu().numpy()).max() < 1e-4
        c2, i2 = head.bbox_assigner.cost_matrix(tb, gt_boxes[b].tensor.to(DEV), gt_labels[b].to(DEV), p["heatmap"][b:b + 1].detach(), head.train_cfg)
        assert np.abs(c2.cpu().numpy() - g["cost_%d" % b]).max() < 1e-3
        assert np.abs(i2.cpu().numpy() - iou[b, :, :counts[b]].cpu().numpy()).max() < 1e-4
    # targets (torch path on device tensors)
    t = head.get_targets(gt_boxes, gt_labels, res[0])
    assert np.array_equal(t[0].cpu().numpy(), g["labels"]) and np.array_equal(t[1].cpu().numpy(), g["label_weights"])
    assert np.array_equal(t[3].cpu().numpy(), g["bbox_weights"]) and t[5] == int(g["num_pos"])
    assert np.abs(t[2].cpu().numpy() - g["bbox_targets"]).max() < 1e-5 and np.abs(t[4].cpu().numpy() - g["ious"]).max() < 1e-4
    assert np.array_equal(t[7].cpu().numpy(), g["heatmap"])
    # Gaussian targets by the splat kernel: bit for bit
    cfg = head.train_cfg
    hk = ops.draw_heatmap_gaussian(gt, lab, off, B, 10, 20, 20, cfg["voxel_size"], cfg["out_size_factor"], cfg["point_cloud_range"],
                                   cfg["gaussian_overlap"], cfg["min_radius"])
    assert np.array_equal(hk.cpu().numpy(), g["heatmap"])
    # losses: device kernels
    with torch.enable_grad():
        dl = head.loss_device(gt_boxes, gt_labels, res)
        sum(v for n, v in dl.items() if "loss" in n).backward()
    gx_dev = xg.grad.clone()
    for k in ("loss_heatmap", "layer_-1_loss_cls", "layer_-1_loss_bbox", "matched_ious"):
        want = float(g["loss_" + k])
        assert abs(dl[k].item() - want) < 1e-4 * max(1.0, abs(want)), (k, dl[k].item(), want)
    assert int(head._last_num_pos.item()) == int(g["num_pos"])
    # losses: torch composition (mutates dense_heatmap like the reference, so it runs second, on a fresh forward)
    with torch.enable_grad():
        xg2 = x.clone().requires_grad_(True)
        res2 = head([xg2], None, [{}])
        tl = head.loss(gt_boxes, gt_labels, res2)
        sum(v for n, v in tl.items() if "loss" in n).backward()
    for k in dl:
        assert abs(dl[k].item() - tl[k].item()) < 1e-4 * max(1.0, abs(tl[k].item())), k
    # gradients: the two device formulations agree with each other and with the reference's (fp32 noise floor of this
    # saturated problem: 1.5e-4 at the worst element, see tests/test_host_modules.py)
    d = (gx_dev - xg2.grad).abs()
    scale = xg2.grad.abs().max().item()
    assert d.max().item() < 1.5e-2 * scale and d.mean().item() < 2e-3 * xg2.grad.abs().mean().item()
    dref = np.abs(gx_dev[:, :8].cpu().numpy() - g["gx_slice"])
    assert dref.max() < 1.5e-2 * np.abs(g["gx_slice"]).max() and dref.mean() < 2e-3 * np.abs(g["gx_slice"]).mean()
    gsum = np.array([gx_dev.sum(dtype=torch.float64).item(), gx_dev.abs().sum(dtype=torch.float64).item()])
    assert np.abs(gsum - g["gx"]).max() < 1e-3 * g["gx"][1]


@pytest.mark.parametrize("layout", ["nchw", "rows"])
def test_gaussian_focal_loss_vs_torch(layout):
    from dualfusion import ops, tf_losses
    B, C, H, W = 3, 10, 37, 41
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.normal(0, 4, (B, C, H, W)).astype(np.float32)).to(DEV)
    x[0, 0, :4, :4] = torch.tensor([-30.0, 30.0, -9.3, 9.3], device=DEV)[None]          # clamped / unclamped borders
    t = torch.from_numpy((rs.uniform(0, 1, (B, C, H, W)) ** 8).astype(np.float32)).to(DEV)
    t[t > 0.8] = 1.0
    logits = x if layout == "nchw" else x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    out, grad = ops.gaussian_focal_loss(logits, t, 2.0, 4.0, 0.7, want_grad=True)
    xr = x.double().requires_grad_(True)
    ref = tf_losses.GaussianFocalLoss(loss_weight=0.7)(torch.clamp(xr.sigmoid(), 1e-4, 1 - 1e-4), t.double(),
                                                       avg_factor=max(t.eq(1).sum().item(), 1))
    ref.backward()
    assert int(out[1].item()) == int(t.eq(1).sum().item())
    assert abs(out[0].item() - ref.item()) < 2e-5 * abs(ref.item())
    g = (grad * out[2]).cpu().double()
    # float64 clamps at exactly the same logits except within rounding of the clamp points
    far = ((xr.detach().cpu().abs() - 9.2102).abs() > 1e-3)
    assert ((g - xr.grad.cpu()).abs() * far).max().item() < 1e-5 * xr.grad.abs().max().item() + 1e-7
    with pytest.raises(Exception):
        ops.gaussian_focal_loss(x.cpu(), t.cpu())


def test_full_size_properties_and_empty_samples():
    """nuScenes map (180 x 180, 200 proposals, bs = 4): device losses vs the torch composition on random predictions; a
    sample without ground truth; Hungarian properties (one proposal per box, no proposal twice)."""
    from dualfusion import synth
    head = _nusc_head()
    B, K, C = 4, 200, 10
    rs = np.random.RandomState(3)
    p = {"center": torch.from_numpy(rs.uniform(5, 175, (B, 2, K)).astype(np.float32)),
         "height": torch.from_numpy(rs.uniform(-2, 0, (B, 1, K)).astype(np.float32)),
         "dim": torch.from_numpy(rs.normal(0.7, 0.5, (B, 3, K)).astype(np.float32)),
         "rot": torch.from_numpy(rs.normal(0, 1, (B, 2, K)).astype(np.float32)),
         "vel": torch.from_numpy(rs.normal(0, 1, (B, 2, K)).astype(np.float32)),
         "heatmap": torch.from_numpy(rs.normal(-2, 1.5, (B, C, K)).astype(np.float32)),
         "dense_heatmap": torch.from_numpy(rs.normal(-3, 1.5, (B, C, 180, 180)).astype(np.float32))}
    p = {k: v.to(DEV).requires_grad_(True) for k, v in p.items()}
    gts = [synth.nusc_gt_boxes(40 + b) for b in range(B)]
    gts[2] = (np.zeros((0, 9), np.float32), np.zeros((0,), np.int64))             # a frame without objects
    for b in (0, 1):                                                               # some boxes right on proposals: IoU > 0
        n = 6
        ctr = p["center"][b, :, :n].detach().cpu().numpy().T * 0.6 - 54.0
        gts[b][0][:n, :2] = ctr + rs.normal(0, 0.2, (n, 2))
        gts[b][0][:n, 3:6] = np.exp(p["dim"][b, :, :n].detach().cpu().numpy().T) * 1.07   # not exactly equal: no |d| = 0 ties
        gts[b][0][:n, 2] = p["height"][b, 0, :n].detach().cpu().numpy() - gts[b][0][:n, 5] / 2
    gt_boxes = [torch.from_numpy(g[0]) for g in gts]
    gt_labels = [torch.from_numpy(g[1]) for g in gts]
    preds = ([p],)
    dl = head.loss_device(gt_boxes, gt_labels, preds)
    sum(v for n, v in dl.items() if "loss" in n).backward()
    grads_dev = {k: v.grad.clone() for k, v in p.items()}
    for v in p.values():
        v.grad = None
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    tl = head.loss(gt_boxes, gt_labels, ([{k: v * 1 for k, v in leaf.items()}],))    # loss() works in place on dense_heatmap
    p2 = leaf
    sum(v for n, v in tl.items() if "loss" in n).backward()
    for k in dl:
        assert abs(dl[k].item() - tl[k].item()) < 2e-4 * max(1.0, abs(tl[k].item())), (k, dl[k].item(), tl[k].item())
    assert dl["matched_ious"].item() > 0.01
    for k in p:
        d = (grads_dev[k] - p2[k].grad).abs().max().item()
        assert d < 1e-4 * p2[k].grad.abs().max().item() + 1e-8, (k, d)
    # matching properties from the targets of the torch path
    t = head.get_targets(gt_boxes, gt_labels, [{k: v.detach() for k, v in p.items()}])
    labels, bw = t[0], t[3]
    for b in range(B):
        npos = int((labels[b] < C).sum().item())
        assert npos == min(len(gts[b][1]), K) and int(bw[b, :, 0].sum().item()) == npos
    assert int(head._last_num_pos.item()) == sum(min(len(g[1]), K) for g in gts)
    assert t[7][2].abs().sum().item() == 0 and (t[7].amax((2, 3)) <= 1).all()
    ones = int(t[7].eq(1).sum().item())
    assert 0 < ones <= sum(len(g[1]) for g in gts)


def test_batch_without_any_ground_truth():
    """every sample of the batch empty -- `loss_device` must accept it (the torch `loss` path does) and give
    the same all-negative losses and gradients."""
    head = _nusc_head()
    B, K, C = 2, 200, 10
    rs = np.random.RandomState(5)
    p = {"center": rs.uniform(5, 175, (B, 2, K)), "height": rs.uniform(-2, 0, (B, 1, K)), "dim": rs.normal(0.7, 0.5, (B, 3, K)),
         "rot": rs.normal(0, 1, (B, 2, K)), "vel": rs.normal(0, 1, (B, 2, K)), "heatmap": rs.normal(-2, 1.5, (B, C, K)),
         "dense_heatmap": rs.normal(-3, 1.5, (B, C, 180, 180))}
    p = {k: torch.from_numpy(v.astype(np.float32)).to(DEV).requires_grad_(True) for k, v in p.items()}
    gt_boxes = [torch.zeros((0, 9)) for _ in range(B)]
    gt_labels = [torch.zeros((0,), dtype=torch.int64) for _ in range(B)]
    dl = head.loss_device(gt_boxes, gt_labels, ([p],))
    sum(v for n, v in dl.items() if "loss" in n).backward()
    grads_dev = {k: (v.grad.clone() if v.grad is not None else None) for k, v in p.items()}
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    tl = head.loss(gt_boxes, gt_labels, ([{k: v * 1 for k, v in leaf.items()}],))
    sum(v for n, v in tl.items() if "loss" in n).backward()
    for k in dl:
        assert abs(dl[k].item() - tl[k].item()) < 2e-4 * max(1.0, abs(tl[k].item())), (k, dl[k].item(), tl[k].item())
    assert dl["layer_-1_loss_bbox"].item() == 0.0 and int(head._last_num_pos.item()) == 0
    for k in ("heatmap", "dense_heatmap"):
        d = (grads_dev[k] - leaf[k].grad).abs().max().item()
        assert d < 1e-4 * leaf[k].grad.abs().max().item() + 1e-8, (k, d)


def _nusc_head():
    from dualfusion import synth
    from dualfusion.transfusion_head import TransFusionHead
    torch.manual_seed(0)
    return TransFusionHead(
        num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10, num_decoder_layers=1,
        num_heads=8, initialize_by_heatmap=True, nms_kernel_size=3, ffn_channel=256,
        common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
        bbox_coder=dict(type='TransFusionBBoxCoder', pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], out_size_factor=8,
                        post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.0, code_size=10),
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
        loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0),
        train_cfg=dict(dataset='nuScenes',
                       assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                                     cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                                     reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
                       pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[1440, 1440, 40], voxel_size=synth.NUSC_VOXEL,
                       out_size_factor=8, code_weights=[1.0] * 8 + [0.2, 0.2], point_cloud_range=synth.NUSC_RANGE),
        test_cfg=dict(dataset='nuScenes', grid_size=[1440, 1440, 40], out_size_factor=8, pc_range=[-54.0, -54.0],
                      voxel_size=[0.075, 0.075], nms_type=None)).to(DEV).eval()


def test_splat_kernel_vs_host_loop_at_nuscenes_size():
    from dualfusion import synth
    head = _nusc_head()
    gts = [synth.nusc_gt_boxes(7 + b) for b in range(3)]
    gts[1][0][:4, :2] = [[-53.9, -53.9], [53.9, 53.9], [-53.9, 53.9], [0.0, 53.95]]     # Gaussians clipped at the borders
    gts[1][0][4, 3:5] = [30.0, 40.0]                                                      # a large radius
    gt = [torch.from_numpy(g[0]).to(DEV) for g in gts]
    lab = [torch.from_numpy(g[1]).to(DEV) for g in gts]
    host = head.heatmap_targets(gt, lab, torch.device(DEV))
    from dualfusion import ops
    g, l, off, _ = head._pack_gt([t.cpu() for t in gt], [t.cpu() for t in lab], torch.device(DEV))
    cfg = head.train_cfg
    dev = ops.draw_heatmap_gaussian(g, l, off, 3, 10, 180, 180, cfg["voxel_size"], cfg["out_size_factor"], cfg["point_cloud_range"],
                                    cfg["gaussian_overlap"], cfg["min_radius"])
    assert torch.equal(host, dev) and host.eq(1).sum().item() > 30


def test_rccl_world_size_one_reduce_and_buckets():
    """RCCL comes up on the device: init_process_group('nccl') at world size 1, reduce_dict of device losses, the bucketed
    gradient reducer (hooks, all_reduce on the bucket storage, finish)."""
    code = r'''
import os, sys, torch
sys.path[:0] = [%r, %r]
from dualfusion import dist as D
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
rank, local, world = D.init_from_env("nccl")
assert torch.distributed.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", 0)
out = D.reduce_dict({"loss_heatmap": torch.tensor([2.5], device=dev), "layer_-1_loss_cls": torch.tensor([1.0, 3.0], device=dev)})
assert out["loss_heatmap"].item() == 2.5 and out["layer_-1_loss_cls"].tolist() == [1.0, 3.0]
lin = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 8)).to(dev)
red = D.GradBucketReducer(list(lin.parameters()), bucket_mb=0.05)
assert len(red.buckets) >= 2
red.zero_grad()
x = torch.randn(32, 64, device=dev)
lin(x).square().mean().backward()
red.finish()
ref = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 8)).to(dev)
ref.load_state_dict(lin.state_dict())
ref(x).square().mean().backward()
for a, b in zip(lin.parameters(), ref.parameters()):
    assert torch.allclose(a.grad, b.grad, atol=1e-6), (a.grad - b.grad).abs().max()
assert abs(D.max_over_ranks(1.25, dev) - 1.25) < 1e-6
D.barrier(dev)
torch.distributed.destroy_process_group()
print("RCCL_OK")
''' % (ROOT, os.path.join(ROOT, "3d-dual-fusion_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_tf_fusion_reduces_real_losses_over_rccl():
    """bench.py --workload tf_fusion: the step ends in the head's detection losses, reduced through reduce_dict on an
    RCCL process group (launched as the driver launches N > 1: torch.distributed.run, here with one rank)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "tf_fusion", "--steps", "2",
           "--warmup", "1", "--frames", "2", "--no-extra-passes", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["collective_backend"] == "nccl" and res["world_size"] == 1 and res["n_gpus"] == 1
    rl = res["reduced_losses"]
    assert set(rl) == {"loss_heatmap", "layer_-1_loss_cls", "layer_-1_loss_bbox", "matched_ious"}
    assert all(np.isfinite(v).all() for v in rl.values()) and rl["loss_heatmap"][0] > 0 and rl["layer_-1_loss_bbox"][0] > 0
    assert "bs=4" in res["config"]["workload"] and res["value"] > 0
