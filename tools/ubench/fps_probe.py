#!/usr/bin/env python3
"""Furthest point sampling at the Voxel-RCNN + 3D-DF shape (B = 8, ~24k points, 2048 samples): time per iteration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch
from dualfusion import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, N, m in ((8, 24000, 2048), (1, 24000, 2048), (8, 17000, 2048), (8, 12000, 2048)):
    xyz = (torch.rand(B, N, 3, device=dev) * 40 - 20).contiguous()
    ops.furthest_point_sample(xyz, m)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        ops.furthest_point_sample(xyz, m)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print("B %d N %6d m %d: %.2f ms = %.2f us per iteration" % (B, N, m, ms, ms * 1e3 / m))
