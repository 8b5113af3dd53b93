"""Host-side profile (cProfile) of bench.py's detector step (cp_fusion / cp_lidar / tf_fusion): where the Python thread spends
its time, which calls block on the GPU.  usage: cp_cpu_profile.py [workload] [stage]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

import bench  # noqa: E402


class A(object):
    workload, frames, batch, inflight = sys.argv[1] if len(sys.argv) > 1 else "cp_fusion", 8, 0, 1


stage = sys.argv[2] if len(sys.argv) > 2 else "detect"
if A.workload == "tf_fusion":
    from dualfusion import ops
    ops.CONV_PRECISION = "bf16"
wl = bench.make_workload(A(), 0, 1, torch.device("cuda:0"))
for k in range(12):
    wl.step(k, stage)
torch.cuda.synchronize()
N = 24
t0 = time.perf_counter()
for k in range(N):
    wl.step(k, stage)
torch.cuda.synchronize()
print("plain: %.3f ms/step" % ((time.perf_counter() - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for k in range(N):
    wl.step(k, stage)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats("dualfusion|bench", 40)
