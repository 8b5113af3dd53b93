#!/usr/bin/env python3
"""aten::copy_ / contiguous / clone of one training step by shape and stride pattern (torch.profiler, device time)."""
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
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(N):
        wl.step(3 + i, "train")
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = getattr(ev, "self_cuda_time_total", 0.0)
    if dt <= 0 or ev.name not in ("aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::sum", "aten::cat", "aten::index",
                                  "aten::index_put_", "aten::fill_", "aten::zero_", "aten::mul_", "aten::div", "aten::where",
                                  "aten::sigmoid", "aten::relu", "aten::threshold_backward", "aten::native_layer_norm",
                                  "aten::native_layer_norm_backward", "aten::native_dropout", "aten::native_dropout_backward"):
        continue
    key = (ev.name, str(ev.input_shapes)[:90])
    by[key][0] += dt
    by[key][1] += 1
tot = collections.Counter()
for (name, _), (t, n) in by.items():
    tot[name] += t / N
print({k: round(v) for k, v in tot.most_common()})
for (name, shapes), (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:45]:
    print("%8.1f us/step %5.1f calls  %-28s %s" % (t / N, n / N, name, shapes))
if hasattr(wl, "close"):
    wl.close()
