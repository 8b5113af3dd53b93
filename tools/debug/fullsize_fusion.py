#!/usr/bin/env python3
"""Where does the full-size fusion output deviate?  Product (folded / fused) vs product (module composition) vs oracle port
on the SAME stride-8 inputs captured from one full-grid sweep."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch

import oracle_models as om
from dualfusion import synth
from dualfusion.fusion import CP_DEPTH_THRES, build_centerpoint_fusion, synthetic_camera_inputs
from dualfusion.pipeline import CenterPointHotPath

DEV = "cuda:0"
yaw = float(sys.argv[1]) if len(sys.argv) > 1 else 7.3
randomize = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
torch.manual_seed(0)
model = CenterPointHotPath(fusion=build_centerpoint_fusion()).eval().to(DEV)
if randomize:
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.weight.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
bd, ex = synthetic_camera_inputs(1, DEV, seed=77, yaw_offset_deg=yaw)
pts = torch.from_numpy(synth.nusc_sweep(seed=5)).to(DEV)
cap = {}
real = model.fusion


def spy(batch_dict, example, encoded_voxel_list=None, **kw):
    cap["levels"] = [(x.indices.clone(), x.features.clone()) for x in encoded_voxel_list]
    cap["shapes"] = [list(x.spatial_shape) for x in encoded_voxel_list]
    cap["kw"] = kw
    out = real(batch_dict, example, encoded_voxel_list=encoded_voxel_list, **kw)
    cap["out_a"] = out.features.clone()
    return out


spy.fuse_mode = 'pfat'
with torch.no_grad():
    model.fusion = None
    hp_fusion = real
    # run through the backbone with the spy as fuse_func
    feats, coors = model.voxelize([pts])
    bev, multi = model.backbone(feats, bd, coors, 1, model.grid_size_xyz, ex, fuse_func=spy)
print("levels:", [tuple(f.shape) for _, f in cap["levels"]], "kw:", {k: v for k, v in cap["kw"].items() if k != "img_conv_func"})
out_a = cap["out_a"].cpu().numpy()

from dualfusion.spconv.structure import SparseConvTensor


def run_variant(setup):
    xs = [SparseConvTensor(f.clone(), i.clone(), cap["shapes"][k], 1) for k, (i, f) in enumerate(cap["levels"])]
    undo = setup()
    try:
        with torch.no_grad():
            out = real(bd, ex, encoded_voxel_list=xs, **cap["kw"])
    finally:
        undo()
    return out.features.cpu().numpy()


def v_default():
    return lambda: None


def v_nofold():
    old = real.pfat.can_fold
    real.pfat.can_fold = lambda: False
    os.environ["DF3D_IMGPROJ"] = "0"

    def undo():
        real.pfat.can_fold = old
        os.environ["DF3D_IMGPROJ"] = "1"
    return undo


def v_noimgproj():
    os.environ["DF3D_IMGPROJ"] = "0"

    def undo():
        os.environ["DF3D_IMGPROJ"] = "1"
    return undo


out_a2 = run_variant(v_default)
out_b = run_variant(v_nofold)
out_b2 = run_variant(v_noimgproj)
sd_all = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
sd_f = {k[len("fusion."):]: v for k, v in sd_all.items() if k.startswith("fusion.")}
if not sd_f:
    sd_f = {k: v.detach().cpu().numpy() for k, v in real.state_dict().items()}
img = {n: bd['img_feat']['layer1_ori_feat2d'][n.lower()].cpu().numpy() for n in synth.NUSC_CAMS}
calib = {n: (bd['calib']['lidar2cam_' + n.lower().lstrip('cam_')].cpu().numpy(),
             bd['calib']['cam_intrinsic_' + n.lower().lstrip('cam_')].cpu().numpy()) for n in synth.NUSC_CAMS}
hw = tuple(int(v) for v in bd['image_shape']['cam_front'][0][:2])
dbg = {}
out_c = om.centerpoint_fusion(sd_f, [(i.cpu().numpy(), f.cpu().numpy()) for i, f in cap["levels"]], img, calib, hw,
                              synth.NUSC_CAMS, synth.NUSC_VOXEL, synth.NUSC_RANGE, 2.0 / 3.0, CP_DEPTH_THRES, debug=dbg)
base = cap["levels"][-1][1].cpu().numpy()


def cmp(name, x, y):
    d = np.abs(x - y)
    s = np.abs(y - base).max() + 1e-30
    rows = d.max(1)
    print("%-38s max abs %.3e | rel to |delta|max %.3e | rel to |out|max %.3e | rows>1e-3*outmax: %d / %d" % (
        name, d.max(), d.max() / s, d.max() / np.abs(y).max(), int((rows > 1e-3 * np.abs(y).max()).sum()), len(rows)))


print("|out|max %.3f, |out - in|max (the fusion's contribution) %.3f" % (np.abs(out_c).max(), np.abs(out_c - base).max()))
cmp("product(in backbone) vs product(replay)", out_a, out_a2)
cmp("product fused vs oracle port", out_a, out_c)
cmp("product no-fold vs oracle port", out_b, out_c)
cmp("product fold, torch imgproj vs oracle", out_b2, out_c)
cmp("product fused vs product no-fold", out_a, out_b)
print("debug keys:", list(dbg.keys())[:20])
