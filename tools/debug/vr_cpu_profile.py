"""Host-side profile (cProfile) of the Voxel-RCNN + 3D-DF step: where the Python thread spends its time."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import workloads  # noqa: E402


class A(object):
    workload, frames, batch, inflight = sys.argv[1] if len(sys.argv) > 1 else "vr_fusion", 8, 0, 1


wl = workloads.make(A(), 0, 1, torch.device("cuda:0"))
for k in range(10):
    wl.step(k, "detect")
torch.cuda.synchronize()
N = 16
t0 = time.perf_counter()
for k in range(N):
    wl.step(k, "detect")
torch.cuda.synchronize()
print("plain: %.3f ms/step" % ((time.perf_counter() - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for k in range(N):
    wl.step(k, "detect")
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumtime").print_stats("dualfusion", 50)
