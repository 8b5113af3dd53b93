#!/usr/bin/env python3
"""torch.profiler view of one TransFusion / Voxel-RCNN INFERENCE step: which torch ops (with shapes) are left beside the library's own kernels."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
wlname = os.environ.get("WL", "tf_fusion")
sys.argv = [sys.argv[0], "--workload", wlname, "--frames", "2"]
args = bench.parse()
dev = torch.device("cuda:0")
from dualfusion import ops
ops.CONV_PRECISION = args.conv_precision or ("bf16" if wlname == "tf_fusion" else "split")
wl = bench.make_workload(args, 0, 1, dev)
stage = args.stage
for i in range(4):
    wl.step(i, stage)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    wl.step(4, stage)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total)
print("device time: %.2f ms" % (sum(e.self_device_time_total for e in rows) / 2e3))
for e in rows[:int(os.environ.get("TOP", "200"))]:
    if e.key.startswith("aten::") and e.self_device_time_total > 5:
        print("%9.1f us x%-3d %-40s %s" % (e.self_device_time_total, e.count, e.key[:40], str(e.input_shapes)[:120]))
