#!/usr/bin/env python3
"""Host-side phase times of the LiDAR training step (no syncs between phases): how long the host needs to QUEUE each phase vs
the step's wall time."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "3d-dual-fusion_amd"))
import torch
import bench
sys.argv = [sys.argv[0], "--workload", "cp_lidar", "--stage", "train"]
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.make_workload(args, 0, 1, dev)
wl._train_setup()
model = wl.model
hp = model.hot_path
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
N = 15
for it in range(5 + N):
    if it == 5:
        torch.cuda.synchronize(); T.clear(); w0 = time.perf_counter()
    fr = wl.frames[it % len(wl.frames)]
    bd, example = wl.fresh_inputs(fr)
    t = time.perf_counter()
    wl.reducer.zero_grad(); t = tick("zero_grad", t)
    with torch.no_grad():
        feats, coors = hp.voxelize(fr["points"]); t = tick("voxelize", t)
    hp.backbone.dense_layout = "nchw"
    with torch.enable_grad():
        bev, _ = hp.backbone(feats, coors, 1, hp.grid_size_xyz); t = tick("backbone fwd", t)
        x = model.neck(bev); t = tick("neck fwd", t)
        preds = model.bbox_head(x); t = tick("head fwd", t)
        rets = model.bbox_head.loss_rows(example)
        if rets is None:
            rets = model.bbox_head.loss(example, preds, {}, host_copies=False)
        t = tick("loss fwd", t)
        sum(rets["loss"]).backward(); t = tick("backward", t)
    hp.backbone.dense_layout = "rows"
    wl.reducer.finish(); t = tick("reducer", t)
    wl.optimizer.step(); t = tick("optimizer", t)
torch.cuda.synchronize()
wall = (time.perf_counter() - w0) / N
print("wall %.2f ms/step; host queue time per phase (ms):" % (wall * 1e3))
for k, v in T.items():
    print("  %-14s %.2f" % (k, v / N * 1e3))
print("  sum            %.2f" % (sum(T.values()) / N * 1e3))
