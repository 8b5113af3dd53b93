"""How long does the Voxel-RCNN step wait for the side stream (FPS + ball query of the stride-8 queries)?
Per step: elapsed time from the moment the main stream reaches `_fuse4` to the side stream's completion event (> 0 = stall)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import backbones, workloads  # noqa: E402


class A(object):
    workload, frames, batch, inflight = "vr_fusion", 8, 0, 1


wl = workloads.make(A(), 0, 1, torch.device("cuda:0"))
orig = backbones.VoxelBackBone8xFusion._fuse4
marks = []


def patched(self, x2, x3, x4, bd):
    pre = self.__dict__.get("_fuse4_pre")
    if pre is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((e, pre["event"]))
    return orig(self, x2, x3, x4, bd)


backbones.VoxelBackBone8xFusion._fuse4 = patched
for k in range(12):
    wl.step(k, "detect")
torch.cuda.synchronize()
print("main reaches _fuse4 -> side stream done, ms per step:", ["%.2f" % a.elapsed_time(b) for a, b in marks[4:]])
