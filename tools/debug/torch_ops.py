#!/usr/bin/env python3
"""Which torch (ATen) ops run in one detector step, with the Python line that calls them (torch.profiler, stacks):
usage torch_ops.py [workload] [stage]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "3d-dual-fusion_amd"))
import torch
import bench
wl_name = sys.argv[1] if len(sys.argv) > 1 else "cp_fusion"
stage = sys.argv[2] if len(sys.argv) > 2 else "detect"
sys.argv = [sys.argv[0], "--workload", wl_name, "--stage", stage]
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.make_workload(args, 0, 1, dev)
for i in range(6):
    wl.step(i, stage)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(N):
        wl.step(i, stage)
    torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    where = "?"
    for fr in ev.stack:
        if "dualfusion" in fr or "bench.py" in fr:
            where = fr.split("3d-dual-fusion_amd/")[-1]
            break
    k = (ev.name, where)
    agg[k][0] += 1
    agg[k][1] += ev.device_time_total
tot = sum(v[1] for v in agg.values())
print("aten ops with device time: %.1f us per step, %d calls per step" % (tot / N, sum(v[0] for v in agg.values()) / N))
for (name, where), (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:60]:
    print("%7.1f us %5.1f x  %-28s %s" % (t / N, n / N, name, where[:110]))
