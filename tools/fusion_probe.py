#!/usr/bin/env python3
"""Stage timing of the CenterPoint fusion adapter at nuScenes size (HIP events around torch modules)."""
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import synth  # noqa: E402
from dualfusion.fusion import build_centerpoint_fusion, synthetic_camera_inputs  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
fus = build_centerpoint_fusion()
model = CenterPointHotPath(fusion=fus).eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
bd, ex = synthetic_camera_inputs(1, dev)
times = OrderedDict()


def hook(name, mod):
    def pre(m, a):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        m._t0 = e

    def post(m, a, o):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        times.setdefault(name, []).append((m._t0, e))
    mod.register_forward_pre_hook(pre)
    mod.register_forward_hook(post)


for n, m in fus.named_modules():
    if n and (len(list(m.children())) == 0 or n in ("pfat", "ifat", "pfat.transformer.encoder.layers.0",
                                                    "pfat.transformer.encoder.layers.0.self_attn")):
        hook(n, m)
hook("fusion(total)", fus)
with torch.no_grad():
    for _ in range(3):
        model(pts, batch_dict=bd, example=ex)
    times.clear()
    for _ in range(5):
        model(pts, batch_dict=bd, example=ex)
torch.cuda.synchronize()
x4n = None
for n, ev in times.items():
    ms = sum(a.elapsed_time(b) for a, b in ev) / 5
    print("%-70s %8.1f us/step  (%d calls)" % (n, ms * 1e3, len(ev) // 5))
