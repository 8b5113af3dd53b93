"""Fused LocalTransformer layer (df3d_lt_layer) against the row-kernel chain at the Voxel-RCNN size: 32 x 16384 x 64."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-dual-fusion_amd"))
from dualfusion.pointformer import TransformerEncoderLayerPreNorm  # noqa: E402

dev = torch.device("cuda:0")
m = TransformerEncoderLayerPreNorm(d_model=64, nhead=4, dim_feedforward=128, dropout=0.0).eval().to(dev)
x = torch.randn(32, 16384, 64, device=dev)


def t(reps=20):
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for mode in ("1", "0", "1"):
    os.environ["DF3D_LT_FUSED"] = mode
    us = t()
    print("DF3D_LT_FUSED=%s: %.1f us per layer (rows in + out %.0f MB -> %.2f TB/s)" % (mode, us, 2 * x.numel() * 4 / 1e6, 2 * x.numel() * 4 / us / 1e6))

from dualfusion import ops  # noqa: E402
hit = m.__dict__["_lt_packed"]
xg = x.permute(1, 0, 2).contiguous()
with torch.no_grad():
    for _ in range(3):
        yg = ops.lt_layer(xg, hit[1], hit[2], 4, 128, 1e-5, 1e-5, group_major=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        yg = ops.lt_layer(xg, hit[1], hit[2], 4, 128, 1e-5, 1e-5, group_major=True)
    torch.cuda.synchronize()
    print("group-major rows: %.1f us per layer" % ((time.perf_counter() - t0) / 20 * 1e6))
    os.environ["DF3D_LT_FUSED"] = "1"
    print("max |group-major - token-major| =", float((yg.permute(1, 0, 2) - m(x)).abs().max()))
