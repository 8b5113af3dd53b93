"""Exact-fp32 detection head (CenterHead.forward_rows_fp32) timing on an MI355X: python tools/ubench/fp32_head.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-dual-fusion_amd"))
from dualfusion import ops  # noqa: E402

dev = torch.device("cuda:0")
n, K = 180 * 180, 9
nbr = ops.conv2d_neighbors(1, 180, 180, 3, 3, 1, 1, False, dev)[0]


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


x = torch.randn(n, 64, device=dev)
for G, cout in ((36, 64), (18, 128), (9, 256)):
    w = torch.randn(G, K, 64, cout, device=dev) * 0.05
    us = t(lambda: ops.sparse_conv_grouped(x, w, nbr, n, relu=True))
    print("mids  G=%2d cout=%3d: %7.1f us  %.1f TF" % (G, cout, us, 2.0 * n * K * 64 * 64 * 36 / us / 1e6))
w1 = torch.randn(K, 64, 128, device=dev) * 0.05
us = t(lambda: ops.sparse_conv_fused(x, w1, nbr, n, relu=True))
print("mid   single 64->128 (default dispatch): %7.1f us x 18 = %.1f" % (us, us * 18))
mid = torch.randn(n, 2304, device=dev)
for cout in (16, 32):
    w = torch.randn(36, K, 64, cout, device=dev) * 0.05
    us = t(lambda: ops.sparse_conv_grouped(mid, w, nbr, n, group_in=64))
    print("finals G=36 cout=%d: %7.1f us  %.1f TF (useful columns %d)" % (cout, us, 2.0 * n * K * 64 * cout * 36 / us / 1e6, cout))
