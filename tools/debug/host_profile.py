"""cProfile of bench.py's detector step on the host (which Python functions the ~2 ms of enqueue time per frame go to).
usage: host_profile.py [workload] [steps] [sort: tottime|cumtime]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

import bench  # noqa: E402


class A(object):
    workload, frames, batch, inflight, prefetch = sys.argv[1] if len(sys.argv) > 1 else "cp_fusion", 8, 0, 1, True


steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
key = sys.argv[3] if len(sys.argv) > 3 else "tottime"
wl = bench.make_workload(A(), 0, 1, torch.device("cuda:0"))
for k in range(12):
    wl.step(k, "detect")
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(12, 12 + steps):
    wl.step(k, "detect")
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(key)
print("per step = totals / %d" % steps)
st.print_stats(70)
wl.close()
