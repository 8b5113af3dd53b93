#!/usr/bin/env python3
"""Run-to-run spread of a backbone filter gradient over three queued cp_fusion training steps, with the geometry stream on / off
(tests/test_gpu_fullsize.py::test_training_steps_with_geometry_on_its_own_stream): is a difference between the two modes larger
than the difference between two runs of ONE mode (float atomics + rectifier flips)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch
from dualfusion import synth
from dualfusion.fusion import build_centerpoint_fusion, synthetic_camera_inputs
from dualfusion.pipeline import NUSC_TASKS, CenterPointDetector
DEV = torch.device("cuda:0")
torch.manual_seed(0)
det = CenterPointDetector(fusion=build_centerpoint_fusion()).to(DEV).train()
det.hot_path.resident_inputs = True
det.hot_path.fusion.resident_inputs = True
frames = []
for k in range(3):
    pts = [torch.from_numpy(synth.nusc_sweep(seed=21 + k)).to(DEV)]
    bd, ex = synthetic_camera_inputs(1, DEV, seed=5 + k)
    tg = synth.centerhead_targets(1, [t["num_class"] for t in NUSC_TASKS], seed=8 + k)
    frames.append((pts, bd, dict(ex, **{k_: [torch.from_numpy(a).to(DEV) for a in v] for k_, v in tg.items()})))
torch.cuda.synchronize()
w = det.hot_path.backbone.conv3[0].weight
state = {k: v.detach().clone() for k, v in det.state_dict().items()}
res = []
for mode in ("1", "1", "0", "0"):
    os.environ["DF3D_TRAIN_GEO_STREAM"] = mode
    det.load_state_dict(state)
    torch.manual_seed(7)
    torch.cuda.manual_seed_all(7)
    outs = []
    for pts, bd, ex in frames:
        det.zero_grad(set_to_none=True)
        rets = det.training_step(pts, dict(ex), batch_dict=dict(bd), host_copies="async")
        outs.append((torch.stack([v.detach().reshape(()) for v in rets["loss"]]), w.grad.detach().clone()))
    torch.cuda.synchronize()
    res.append((mode, [(l.cpu(), g.cpu()) for l, g in outs]))
for a in range(4):
    for b in range(a + 1, 4):
        d = [float((ga - gb).abs().max() / gb.abs().max()) for (_, ga), (_, gb) in zip(res[a][1], res[b][1])]
        dl = [float((la - lb).abs().max() / lb.abs().max()) for (la, _), (lb, _) in zip(res[a][1], res[b][1])]
        print("modes %s vs %s: grad rel diff per step %s; loss rel diff %s" % (res[a][0], res[b][0], ["%.1e" % x for x in d], ["%.1e" % x for x in dl]))
