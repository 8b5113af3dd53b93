#!/usr/bin/env python3
"""df3d_cross_attention at the TransFusionHead size (200 queries x 32 400 keys x 8 heads) for several key-chunk counts
(DF3D_XATTN_WORKGROUPS is read once per process: run one process per setting).  usage: xattn_probe.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-dual-fusion_amd"))
from dualfusion import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda:0"
q = torch.randn(B * 200, 128, device=dev)
kv = torch.randn(B * 32400, 256, device=dev)
for _ in range(5):
    ops.cross_attention(q, kv[:, :128], kv[:, 128:], B, 8, 0.25)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    ops.cross_attention(q, kv[:, :128], kv[:, 128:], B, 8, 0.25)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 50 * 1e3
print("cross_attention bs=%d workgroups=%s: %.1f us (%.1f TFLOP/s fp32)" % (
    B, os.environ.get("DF3D_XATTN_WORKGROUPS", "default"), ms * 1e3, B * 2 * 2 * 200 * 32400 * 128 / ms / 1e9))
