#!/usr/bin/env python3
"""The conv4 128 -> 128 K = 27 layer of one synthetic sweep, launched a few times (target of rocprofv3 --pmc passes:
tools/ubench/conv4_pmc.sh).  usage: conv4_run.py [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import numpy as np
import torch
from dualfusion import ops, synth
from dualfusion.pipeline import CenterPointHotPath

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
os.environ["DF3D_EXECUTOR"] = "0"
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
x = xs[3]
rb = x.find_indice_pair(model.backbone.conv4[3].conv1.indice_key)
C, n = x.features.shape[1], x.features.shape[0]
w = torch.randn(27, C, C, device=dev) * (0.7 / np.sqrt(27 * C))
fs = ops.split_rows(torch.randn(n, C, device=dev))
packed = ops.conv_pack_weights(w)
bias = torch.randn(C, device=dev)
resid = torch.randn(n, C, device=dev)
for _ in range(n_launch):
    ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, bias=bias, residual=resid, relu=True)
torch.cuda.synchronize()
print("rows", n, "pairs", int((rb.nbr >= 0).sum()))
