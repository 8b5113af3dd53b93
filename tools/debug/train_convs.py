#!/usr/bin/env python3
"""Which convolutions a cp_fusion training step sends through torch / MIOpen (shapes), to find the slow ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import argparse, collections
import torch
import bench
args = argparse.Namespace(batch=1, frames=2, workload="cp_fusion", backend="df3d")
dev = torch.device("cuda:0")
wl = bench.CenterPointWorkload(args, 0, 1, dev)
for i in range(2):
    wl.step(i, "train")
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    wl.step(2, "train")
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if "conv" in e.key.lower()]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print("%-40s %10.1f us x%d  %s" % (e.key, e.device_time_total, e.count, str(e.input_shapes)[:150]))
import time
for i in range(3, 9):
    torch.cuda.synchronize(); t0 = time.time()
    wl.step(i, "train")
    torch.cuda.synchronize(); print("step %d: %.1f ms" % (i, (time.time() - t0) * 1e3))
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
for e in rows[:14]:
    print("%-60s %10.1f us x%d" % (e.key[:60], e.device_time_total, e.count))
print("cpu-side top:")
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
for e in rows[:10]:
    print("%-60s %10.1f us x%d" % (e.key[:60], e.self_cpu_time_total, e.count))
