"""Does a long-running small kernel on a side stream slow the main stream's kernels?  (MI355X)
python tools/ubench/overlap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-dual-fusion_amd"))
from dualfusion import ops  # noqa: E402

dev = torch.device("cuda:0")
n, K = 180 * 180, 9
nbr = ops.conv2d_neighbors(1, 180, 180, 3, 3, 1, 1, False, dev)[0]
x = torch.randn(n, 64, device=dev)
w = torch.randn(K, 64, 64, device=dev) * 0.05
a = torch.randn(4096, 4096, device=dev)
xyz = torch.rand(8, 24000, 3, device=dev) * 50
side = torch.cuda.Stream()


def main_work(reps=60):
    for _ in range(reps):
        ops.sparse_conv_fused(x, w, nbr, n, relu=True)


def timed(side_fn):
    main_work(5)
    torch.cuda.synchronize()
    if side_fn is not None:
        with torch.cuda.stream(side):
            side_fn()
    t0 = time.perf_counter()
    main_work()
    torch.cuda.current_stream().synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3


for name, fn in (("alone", None),
                 ("fps 2048 of 24k x 8", lambda: ops.furthest_point_sample(xyz, 2048)),
                 ("sleep kernel (1 workgroup, ~5 ms)", lambda: torch.cuda._sleep(10_000_000)),
                 ("fps 512 x 4 launches-equivalent", lambda: [ops.furthest_point_sample(xyz, 512) for _ in range(4)]),
                 ("alone", None)):
    for _ in range(2):
        m, tot = timed(fn)
    print("%-40s main stream %.2f ms, all streams %.2f ms" % (name, m, tot))
