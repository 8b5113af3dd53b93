#!/usr/bin/env python3
"""The conv4 filter gradient alone, N launches (for counter passes).  usage: wgrad_run.py [launches] [channels stage 1-4]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "3d-dual-fusion_amd"))
from dualfusion import ops, synth
from dualfusion.pipeline import CenterPointHotPath

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 6
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
os.environ["DF3D_EXECUTOR"] = "0"
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
blocks = [model.backbone.conv1[0], model.backbone.conv2[3], model.backbone.conv3[3], model.backbone.conv4[3]]
x = xs[stage - 1]
nbr = x.find_indice_pair(blocks[stage - 1].conv1.indice_key).nbr
g = torch.randn((x.features.shape[0], x.features.shape[1]), device=dev)
for _ in range(n_launch):
    ops.sparse_conv_grad_filters(x.features, g, nbr)
torch.cuda.synchronize()
