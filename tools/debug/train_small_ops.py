#!/usr/bin/env python3
"""Where the ~1200 fills / copies of a training step come from: torch.profiler, CPU-side operator events grouped by the first
Python frames inside this repository."""
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    wl.step(3, "train")
    torch.cuda.synchronize()
want = ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::clone", "aten::contiguous",
        "aten::new_zeros", "aten::empty_like", "aten::add_", "aten::cat")
by = collections.Counter()
tot = collections.Counter()
for ev in prof.events():
    if ev.name not in want:
        continue
    frames = [f for f in (ev.stack or []) if ("dualfusion" in f or "bench.py" in f)]
    where = " <- ".join(f.split("/")[-1].split(",")[0][:48] + ":" + f.split("(")[-1][:0] for f in frames[:2]) if frames else "(autograd engine / no python frame)"
    if frames:
        where = " <- ".join(f.split("/")[-1][:70] for f in frames[:2])
    by[(ev.name, where)] += 1
    tot[ev.name] += 1
print(dict(tot))
for (name, where), n in by.most_common(45):
    print("%4d  %-16s %s" % (n, name, where))
if hasattr(wl, "close"):
    wl.close()
