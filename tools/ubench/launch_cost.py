#!/usr/bin/env python3
"""Fixed cost of a dependent launch on one stream: N back-to-back launches of a kernel that moves (almost) nothing."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402
from dualfusion import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(64, 32, device=dev)
y = torch.zeros(1024, device=dev)
for name, fn in (("df3d_split_rows on 64 rows", lambda: ops.split_rows(x)), ("torch add_ on 1024 floats", lambda: y.add_(1.0)),
                 ("hipMemsetAsync 4 KB (zero_)", lambda: y.zero_())):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 2000
    a.record()
    for _ in range(N):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("%-32s %.2f us per launch (HIP_FORCE_DEV_KERNARG=%s)" % (name, a.elapsed_time(b) * 1e3 / N, os.environ.get("HIP_FORCE_DEV_KERNARG")))

# GPU-side cost of a dependent kernel when the host is out of the way: the same launches replayed from a hipGraph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y.add_(1.0)
    with torch.cuda.graph(g, stream=s):
        for _ in range(1000):
            y.add_(1.0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    g.replay()
b.record()
torch.cuda.synchronize()
print("hipGraph of 1000 dependent add_ kernels: %.2f us per kernel" % (a.elapsed_time(b) * 1e3 / 5000))
