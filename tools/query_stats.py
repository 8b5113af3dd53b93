#!/usr/bin/env python3
"""Per-camera query counts of the CenterPoint adapter at nuScenes size (how much of [6, max_ne] is padding)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import synth  # noqa: E402
from dualfusion.fusion import VoxelWithPointProjection, build_centerpoint_fusion, synthetic_camera_inputs  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

dev = torch.device("cuda:0")
fus = build_centerpoint_fusion()
model = CenterPointHotPath(fusion=fus).eval().to(dev)
orig = VoxelWithPointProjection._query_slots


def spy(self, ind, mask, B):
    pos, max_ne = orig(self, ind, mask, B)
    c = mask.to(torch.int64).sum(1).tolist()
    print("n_voxels %d  per-camera visible %s  sum %d  max_ne %d  padded rows %d" % (
        ind.shape[0], c, sum(c), max_ne, max_ne * mask.shape[0]))
    return pos, max_ne


VoxelWithPointProjection._query_slots = spy
for seed in range(3):
    pts = [torch.from_numpy(synth.nusc_sweep(seed=seed)).to(dev)]
    bd, ex = synthetic_camera_inputs(1, dev)
    with torch.no_grad():
        model(pts, batch_dict=bd, example=ex)
