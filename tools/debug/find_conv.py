import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/3d-dual-fusion_amd"]
import torch, bench
sys.argv = [sys.argv[0], "--workload", "tf_fusion", "--stage", "detect"]
args = bench.parse()
from dualfusion import ops
ops.CONV_PRECISION = "bf16"
dev = torch.device("cuda:0")
wl = bench.make_workload(args, 0, 1, dev)
for i in range(4):
    wl.step(i, "detect")
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    wl.step(4, "detect")
    torch.cuda.synchronize()
for ev in prof.events():
    if "conv" in ev.name.lower() and ev.name.startswith("aten::"):
        print(ev.name, ev.input_shapes, round(ev.device_time_total), [f for f in ev.stack if "dualfusion" in f or "bench" in f][:3])
