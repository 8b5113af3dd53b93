#!/usr/bin/env python3
"""Host-side profile of the configs[1] step: where the Python thread spends its time (cProfile), and how much
of the step it spends blocked on the GPU (`.item()` / synchronize)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import synth  # noqa: E402
from dualfusion.fusion import build_centerpoint_fusion, synthetic_camera_inputs  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

dev = torch.device("cuda:0")
fus = build_centerpoint_fusion()
model = CenterPointHotPath(fusion=fus).eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
bd, ex = synthetic_camera_inputs(1, dev)
for _ in range(5):
    model(pts, batch_dict=bd, example=ex)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    model(pts, batch_dict=bd, example=ex)
torch.cuda.synchronize()
print("plain: %.3f ms/step" % ((time.perf_counter() - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    model(pts, batch_dict=bd, example=ex)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
st.sort_stats("cumtime").print_stats("dualfusion|pipeline", 45)
