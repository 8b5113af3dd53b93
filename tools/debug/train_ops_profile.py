#!/usr/bin/env python3
"""torch.profiler over training steps of configs[1] (bench.py's workload): device time by operator and, for the copy / fill /
add operators, by the Python frames that called them -- which layout copies and gradient accumulations are worth removing."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

sys.argv = [sys.argv[0], "--stage", "train", "--workload", "cp_fusion", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.make_workload(args, 0, 1, dev)
for i in range(3):
    wl.step(i, "train")
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(N):
        wl.step(3 + i, "train")
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
by = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = getattr(ev, "self_cuda_time_total", 0.0)
    if dt <= 0 or ev.name not in ("aten::copy_", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::mul", "aten::clone",
                                  "aten::contiguous", "aten::sum", "aten::cat", "aten::index", "aten::index_put_", "aten::mm",
                                  "aten::addmm", "aten::bmm", "aten::matmul"):
        continue
    frames = [f for f in (ev.stack or []) if "dualfusion" in f or "bench.py" in f][:2]
    key = (ev.name, str(ev.input_shapes)[:70], " <- ".join(f.split("/")[-1][:60] for f in frames))
    by[key][0] += dt
    by[key][1] += 1
rows = sorted(by.items(), key=lambda kv: -kv[1][0])[:40]
for (name, shapes, where), (t, n) in rows:
    print("%8.1f us/step %5.1f calls  %-14s %-70s %s" % (t / N, n / N, name, shapes, where))
if hasattr(wl, "close"):
    wl.close()
