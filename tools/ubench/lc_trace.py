#!/usr/bin/env python3
"""Where the waves of the loader / consumer conv kernel spend their time (library built with -DDF3D_OS_TRACE):
matrix waves: work vs barrier; loader waves: issue vs landing wait vs barrier."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import numpy as np
import torch
from dualfusion import _lib, ops, synth
from dualfusion.pipeline import CenterPointHotPath

dev = torch.device("cuda:0")
torch.manual_seed(0)
os.environ["DF3D_EXECUTOR"] = "0"
os.environ["DF3D_OS_LC"] = "1"
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.df3d_debug_set_os_trace.argtypes = [ctypes.c_void_p]
x = xs[3]
rb = x.find_indice_pair(model.backbone.conv4[3].conv1.indice_key)
C, n = x.features.shape[1], x.features.shape[0]
fs = ops.split_rows(torch.randn(n, C, device=dev))
packed = ops.conv_pack_weights(torch.randn(27, C, C, device=dev) * 0.05)
nwg = (n + 127) // 128
tr = torch.zeros((nwg, 16, 8), dtype=torch.int64, device=dev)
for rep in range(6):
    if rep == 5:
        lib.df3d_debug_set_os_trace(ctypes.c_void_p(tr.data_ptr()))
    ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, relu=True)
torch.cuda.synchronize()
lib.df3d_debug_set_os_trace(None)
t = tr.cpu().numpy().astype(np.float64)
steps = t[:, 0, 5].mean()
print("conv4: %d workgroups, %.1f steps per tile; ticks: prologue %.0f  loop %.0f  epilogue %.0f" % (
    nwg, steps, (t[:, :4, 0] - t[:, :4, 6]).mean(), (t[:, :4, 4] - t[:, :4, 0]).mean(), (t[:, :4, 7] - t[:, :4, 4]).mean()))
m = t[:, 0:4, :]
print("matrix waves: per step  work %.0f  barrier %.0f" % (m[:, :, 1].mean() / steps, m[:, :, 3].mean() / steps))
l = t[:, 4:12, :]
print("loader waves: per step  issue %.0f  landing wait %.0f  barrier %.0f" % (l[:, :, 1].mean() / steps, l[:, :, 2].mean() / steps,
                                                                               l[:, :, 3].mean() / steps))
