#!/usr/bin/env python3
"""torch.profiler view of one TransFusion training step: device time by (op, input shapes) and by phase."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
from dualfusion import ops, workloads
ops.CONV_PRECISION = os.environ.get("PREC", "bf16")
dev = torch.device("cuda:0")
WL = os.environ.get("WL", "tf")
if WL == "tf":
    wl = workloads.TransFusionWorkload(types.SimpleNamespace(batch=4, frames=2, prefetch=False), 0, 1, dev)
else:
    import argparse, bench
    ops.CONV_PRECISION = os.environ.get("PREC", "split")
    sys.argv = [sys.argv[0], "--workload", "cp_fusion", "--stage", "train", "--frames", "2"]
    wl = bench.make_workload(bench.parse(), 0, 1, dev)
for i in range(4):
    wl.step(i, "train")
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    wl.step(4, "train")
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("device time of the step: %.2f ms" % (tot / 1e3))
for e in rows[:int(os.environ.get("TOP", "70"))]:
    print("%9.1f us x%-3d %-46s %s" % (e.self_device_time_total, e.count, e.key[:46], str(e.input_shapes)[:110]))
