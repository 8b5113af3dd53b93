#!/usr/bin/env python3
"""torch.profiler over steps of any bench workload: device time by operator / kernel (top 30) and the aten operators by shape.
usage: step_ops_profile.py <workload> [stage] [conv-precision]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

wlname = sys.argv[1] if len(sys.argv) > 1 else "vr_fusion"
stage = sys.argv[2] if len(sys.argv) > 2 else "detect"
prec = sys.argv[3] if len(sys.argv) > 3 else "split"
sys.argv = [sys.argv[0], "--stage", stage, "--workload", wlname, "--no-cpu-baseline", "--conv-precision", prec, "--frames", "4", "--inflight", "1"]
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.make_workload(args, 0, 1, dev)
for i in range(4):
    wl.step(i, stage)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(N):
        wl.step(4 + i, stage)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=70))
by = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = getattr(ev, "self_cuda_time_total", 0.0)
    if dt <= 0 or not ev.name.startswith("aten::"):
        continue
    key = (ev.name, str(ev.input_shapes)[:100])
    by[key][0] += dt
    by[key][1] += 1
for (name, shapes), (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us/step %5.1f calls  %-26s %s" % (t / N, n / N, name, shapes))
if hasattr(wl, "close"):
    wl.close()
