import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/3d-dual-fusion_amd"]
import torch, bench
class A: workload, frames, batch, inflight, prefetch = "cp_fusion", 8, 0, 1, False
wl = bench.make_workload(A(), 0, 1, torch.device("cuda:0"))
f = wl.model.hot_path.fusion
orig = f._query_slots
def spy(ind, mask, B):
    pos, max_ne, counts = orig(ind, mask, B)
    c = counts.cpu().tolist()
    print("counts", c, "max", max_ne, "valid", sum(c), "padded", len(c) * max_ne, "fill %.2f" % (sum(c) / (len(c) * max_ne)))
    return pos, max_ne, counts
f._query_slots = spy
for k in range(8):
    wl.step(k, "detect")
torch.cuda.synchronize()
